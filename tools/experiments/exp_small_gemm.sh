#!/bin/bash
# Kernel-trace durations of tools/experiments/exp_small_gemm.py's cases (CARTNET_LIB selects the build).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/tl_small
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/tl_small -- python3 $ROOT/tools/experiments/exp_small_gemm.py > $ROOT/gpurun_out/tl_small.log 2>&1
cd $ROOT
f=$(find gpurun_out/tl_small -name "*kernel_trace.csv")
python tools/experiments/exp_small_gemm.py $f
rm -rf gpurun_out/tl_small
