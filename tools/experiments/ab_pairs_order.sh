#!/bin/bash
# Same-box A/B of library variants on the bf16x3 and bf16 + bf16-storage steps at full size: tools/experiments/ab_pairs_order.sh A.so B.so ...
for v in "$@" "$@"; do
  echo "== $v"
  for args in "--precision 1" "--precision 2 --half-storage"; do
  CARTNET_LIB=$PWD/$v timeout -k 10 200 python bench.py $args --no-subconfigs --no-telemetry --steps 30 --warmup 10 --no-x3-pass --no-recipe-pass --no-cpu-baseline --no-calibration --no-kernel-timer --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', d['ms_per_step'])"
  done
done
