#!/bin/bash
# Same box: the default command (K = 10, W = 3) with different numbers of spin-up steps in front of the timed region; the
# first process on a fresh box is the interesting one (cold card).
for sp in 2 40 2 40 0; do
  timeout -k 10 200 python bench.py --spinup $sp --no-subconfigs --no-x3-pass --no-recipe-pass --no-cpu-baseline --no-calibration --sustain-seconds 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['telemetry']; print('spinup $sp:', d['ms_per_step'], 'sustained', d['sustained']['ms_per_step_min_window'], d['sustained']['ms_per_step_max_window'], 'sclk before', t['before_timed'].get('sclk_mhz'), 'during', (t.get('during_timed') or {}).get('sclk_mhz'))"
done
