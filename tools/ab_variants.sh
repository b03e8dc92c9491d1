#!/bin/bash
# Same-box comparison of the product library with any number of variants (tools/build_variant.sh NAME ...), two rounds:
#   bash tools/ab_variants.sh NAME1 NAME2 ... [-- extra bench flags]
NAMES=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do NAMES+=("$1"); shift; done; [ "$1" == "--" ] && shift
for round in 1 2; do
  for v in "" "${NAMES[@]}"; do
    lib=cartnet_amd/libcartnet_hip${v:+_$v}.so
    CARTNET_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-subconfigs --steps 30 --warmup 10 \
      --no-x3-pass --no-recipe-pass --no-cpu-baseline --sustain-seconds 0 --no-cold --preroll-steps 100 --no-calibration "$@" 2>gpurun_out/ab_err_${v:-product}.log |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('${v:-product}', d['ms_per_step'], d['value'], 'dominant', r['avg_launch_us'], r['frac'], 'isolated gemm ms', r.get('isolated_gemm_ms_per_step'))" || tail -3 gpurun_out/ab_err_${v:-product}.log
  done
done
