#!/bin/bash
# rocprofv3 kernel stats of the iComformer training step (bench.py --model icomformer), fp32 pass only.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_icomformer
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --model icomformer --preroll-steps 0 --no-telemetry --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-x3-pass --no-recipe-pass --no-calibration --sustain-seconds 0 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "exit $?"
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
