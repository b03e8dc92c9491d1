#!/bin/bash
# Kernel timeline of one training step at configs[2] shapes (tools/bench_jarvis.py, precision 2 is its last pass).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/tl_jarvis -- python3 $ROOT/tools/bench_jarvis.py > $ROOT/gpurun_out/tl_jarvis.log 2>&1
cd $ROOT
f=$(find gpurun_out/tl_jarvis -name "*kernel_trace.csv")
python tools/timeline.py $f 3 --list
rm -f $f
