#!/bin/bash
# Same-box A B C ... x 3 rounds of the fp32 step over cache-policy variants of the library (60 timed steps each):
#   bash tools/experiments/ab_cache_policy.sh NAME1 NAME2 ...      (product first)
for round in 1 2 3; do
  for v in "" "$@"; do
    lib=cartnet_amd/libcartnet_hip${v:+_$v}.so
    CARTNET_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-subconfigs --steps 60 --warmup 10 --no-x3-pass --no-recipe-pass --no-cpu-baseline --sustain-seconds 0 --no-cold --preroll-steps 100 --no-calibration 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-product}', d['ms_per_step'])"
  done
done
