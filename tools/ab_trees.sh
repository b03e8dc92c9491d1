#!/bin/bash
# Same-box comparison of this tree's training step with ANOTHER TREE's (a copy of an earlier commit with its own built
# library under _prev/ -- for changes that alter the ABI, where a variant library cannot be loaded by this tree's binding):
#   bash tools/ab_trees.sh [_prev] [-- extra bench flags]        two rounds, A B A B
OTHER=${1:-_prev}; [ $# -gt 0 ] && shift; [ "$1" == "--" ] && shift
for round in 1 2; do
  for t in . "$OTHER"; do
    (cd $t && timeout -k 10 300 python bench.py --no-subconfigs --steps 30 --warmup 10 --no-x3-pass --no-recipe-pass \
      --no-cpu-baseline --sustain-seconds 0 --no-cold --preroll-steps 100 --no-calibration "$@" 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$t', d['ms_per_step'], d['value'], 'dominant', r['avg_launch_us'], r['frac'])")
  done
done
