#!/bin/bash
# Per-kernel rocprofv3 stats of the fp32 training step for two builds of the library (see tools/experiments/ab_bench.sh for A / B).
# -> gpurun_out/abprof_{A,B}/.../*kernel_stats.csv
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in A B; do
  cp $ROOT/cartnet_amd/libcartnet_hip_$v.so $ROOT/cartnet_amd/libcartnet_hip.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/abprof_$v -- python3 $ROOT/bench.py --preroll-steps 0 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-x3-pass --no-recipe-pass --sustain-seconds 0 > $ROOT/gpurun_out/abprof_$v.json 2> $ROOT/gpurun_out/abprof_$v.err || exit 1
  find $ROOT/gpurun_out/abprof_$v -name "*kernel_trace.csv" -delete
done
