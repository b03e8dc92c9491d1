#!/bin/bash
# Same-box A/B of library builds on the fp32 training step (BASELINE configs[1]):  tools/ab_step.sh A.so B.so [...]   (A B A B)
for v in "$@" "$@"; do
  echo "== $v"
  CARTNET_LIB=$PWD/$v timeout -k 10 300 python bench.py --no-subconfigs --steps 30 --warmup 10 --no-x3-pass --no-recipe-pass --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], d['value'], 'cal', d['calibration']['avg_launch_us'], 'dominant', r['kernel'][:46], r['avg_launch_us'], r['frac'], 'isolated', r.get('isolated_avg_launch_us'), 'whole', r.get('whole_step_frac'), 'isoGEMM', r.get('isolated_gemm_ms_per_step'))"
done
