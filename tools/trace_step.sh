#!/bin/bash
# Kernel trace of a few fp32 training steps + the per-launch Gantt listing of one of them (tools/timeline.py --list).
# Usage (GPU box): bash tools/trace_step.sh <tag> [extra bench.py args]   -> gpurun_out/<tag>_timeline.txt
TAG=${1:-tl}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --preroll-steps 0 --no-telemetry --no-subconfigs --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-x3-pass --no-recipe-pass --no-calibration --sustain-seconds 0 "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
F=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/timeline.py" "$F" 3 --list > "$ROOT/gpurun_out/${TAG}_timeline.txt" 2>&1
rm -f "$F"
tail -3 "$ROOT/gpurun_out/${TAG}_timeline.txt"
