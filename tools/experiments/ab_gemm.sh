#!/bin/bash
# A/B of two library builds (cartnet_amd/libcartnet_hip_{A,B}.so, see tools/experiments/ab_bench.sh) on the layer GEMM variants
# (tools/experiments/exp_w128.py, sustained) and on the training step.
for v in A B A B; do
  cp cartnet_amd/libcartnet_hip_$v.so cartnet_amd/libcartnet_hip.so
  echo "== $v"; python tools/experiments/exp_w128.py 2>/dev/null | grep -v CARTNET
done
bash tools/experiments/ab_bench.sh
