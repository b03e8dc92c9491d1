#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats of the default bench command + two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md) for the HBM traffic per kernel, and one
# more (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) for the matrix-pipe utilisation per kernel.
# Usage: tools/collect_profiles.sh <tag>   -> gpurun_out/prof_<tag>/{stats,fetch,write}/...
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err" &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > "$OUT/fetch.json" 2> "$OUT/fetch.err" &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > "$OUT/write.json" 2> "$OUT/write.err" &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > "$OUT/mfma.json" 2> "$OUT/mfma.err"
echo "exit $?"
find "$OUT" -name "*.csv" | head -20
