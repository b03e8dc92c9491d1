#!/bin/bash
# Same-box A/B of two library builds (CARTNET_LIB): tools/ab_lib.sh A.so B.so [forms-filter]
# prints the selected GEMM forms alone (tools/bench_gemm_forms.py) and the training step, A B A B.
A=$1; B=$2; F=${3:-"dW,calibration"}
for v in $A $B $A $B; do
  echo "== $v"
  CARTNET_LIB=$PWD/$v timeout -k 10 300 python tools/bench_gemm_forms.py 0 "$F" 2>/dev/null
done
for v in $A $B $A $B; do
  echo "== $v"
  CARTNET_LIB=$PWD/$v timeout -k 10 300 python bench.py --no-subconfigs --steps 30 --warmup 10 --no-x3-pass --no-recipe-pass --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], d['value'], 'cal', d['calibration']['avg_launch_us'], 'dominant', r['kernel'][:40], r['avg_launch_us'], r['frac'], 'isolated', r.get('isolated_avg_launch_us'))"
done
