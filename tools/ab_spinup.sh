#!/bin/bash
# Same box: the default command (K = 10, W = 3) with 0 / 2 / 5 spin-up steps in front of the timed region.
for i in 1 2; do
for sp in 0 2 5; do
  timeout -k 10 200 python bench.py --spinup $sp --no-subconfigs --no-x3-pass --no-recipe-pass --no-cpu-baseline --no-calibration --sustain-seconds 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spinup $sp:', d['ms_per_step'], 'sustained', d['sustained']['ms_per_step_min_window'], d['sustained']['ms_per_step_max_window'])"
done; done
