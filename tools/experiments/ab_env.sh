#!/bin/bash
# Same-binary A/B on ONE GPU box: the training step with and without an environment switch (boxes of the pool differ by
# +-2 %, more than most kernel changes are worth).  usage: bash tools/experiments/ab_env.sh VAR [VALUE]   ->  A = VAR unset, B = VAR=VALUE (default 1)
# (The C library reads no environment variable since round 4: this serves Python-side switches such as CARTNET_FUSED_LOSS;
#  library variants are compared with tools/build_variant.sh + tools/ab_lib.sh.)
VAR=${1:-CARTNET_FUSED_LOSS}
VAL=${2:-1}
for v in A B A B; do
  if [ $v = B ]; then export $VAR=$VAL; else unset $VAR; fi
  python bench.py --no-subconfigs --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-recipe-pass --sustain-seconds 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v ($VAR=' + ('$VAL' if '$v' == 'B' else 'unset') + ')', d['ms_per_step'], d['value'], d['bf16x3']['ms_per_step'], d['bf16x3']['value'])"
done
