#!/bin/bash
# A/B comparison of two builds of libcartnet_hip.so on ONE GPU box (boxes of the pool differ by +-2 %, more than most
# kernel changes are worth): build variant A, copy cartnet_amd/libcartnet_hip.so to cartnet_amd/libcartnet_hip_A.so,
# build variant B likewise (*.so files travel with the gpurun snapshot), then on the box:  bash tools/experiments/ab_bench.sh
# Prints ms/step and graphs/s of the fp32 and bf16x3 passes for A B A B.
for v in A B A B; do
  cp cartnet_amd/libcartnet_hip_$v.so cartnet_amd/libcartnet_hip.so
  python bench.py --no-subconfigs --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-recipe-pass --sustain-seconds 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', d['ms_per_step'], d['value'], d['bf16x3']['ms_per_step'], d['bf16x3']['value'])"
done
