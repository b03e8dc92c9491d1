#!/bin/bash
# What the timers inside bench.py's timed region cost: the same command with and without the GEMM launch timer / the
# telemetry sampler, twice each (same box).
for i in 1 2; do
for args in "" "--no-kernel-timer" "--no-telemetry"; do
  timeout -k 10 200 python bench.py $args --no-subconfigs --steps 20 --warmup 5 --no-x3-pass --no-recipe-pass --no-cpu-baseline --no-calibration --sustain-seconds 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('[$args]', d['ms_per_step'], 'sustained', d['sustained']['ms_per_step_min_window'], d['sustained']['ms_per_step_max_window'], 'roofline', r.get('launches'), r.get('avg_launch_us'), r.get('frac'), r.get('share_of_step'), r.get('sampled_every'))"
done; done
