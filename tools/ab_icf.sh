#!/bin/bash
# Same-box A/B of library builds on the iComformer step (BASELINE configs[4]):  tools/ab_icf.sh A.so B.so [...]   (A B A B)
for v in "$@" "$@"; do
  echo "== $v"
  CARTNET_LIB=$PWD/$v timeout -k 10 200 python bench.py --model icomformer --no-telemetry --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timer --no-x3-pass --no-recipe-pass --no-calibration --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
