#!/bin/bash
# Runs on the GPU box (gpurun): the default bench line un-profiled, rocprofv3 kernel stats of the same command, and
# separate PMC passes over fp32-only steps (--no-x3-pass): FETCH_SIZE and WRITE_SIZE (they do not fit one pass,
# MI355X_MICROARCH.md) for the HBM traffic per kernel and per step, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE for the
# matrix-pipe utilisation per kernel.  PMC passes never carry a trace domain.
# Usage: tools/collect_profiles.sh <tag>   -> gpurun_out/prof_<tag>/{bench_default.json,stats,fetch,write,mfma}/...
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# which build these passes measure: the digest of the kernel sources goes into traffic.json (tools/publish_profiles.py), and
# bench.py prints `counted_is_stale` when the sources it runs on differ from it
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench._csrc_digest())" > "$OUT/csrc_digest.txt"
PMC_ARGS="--preroll-steps 0 --no-telemetry --no-subconfigs --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-x3-pass --no-recipe-pass --no-calibration --sustain-seconds 0"
python3 "$ROOT/bench.py" --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" &&
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --preroll-steps 0 --steps 10 --warmup 3 --no-telemetry --no-subconfigs --no-cpu-baseline --no-recipe-pass --sustain-seconds 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err" &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" $PMC_ARGS > "$OUT/fetch.json" 2> "$OUT/fetch.err" &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" $PMC_ARGS > "$OUT/write.json" 2> "$OUT/write.err" &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_x3" -- python3 "$ROOT/bench.py" $PMC_ARGS --precision 1 > "$OUT/fetch_x3.json" 2> "$OUT/fetch_x3.err" &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_x3" -- python3 "$ROOT/bench.py" $PMC_ARGS --precision 1 > "$OUT/write_x3.json" 2> "$OUT/write_x3.err" &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python3 "$ROOT/bench.py" --preroll-steps 0 --no-telemetry --no-subconfigs --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-recipe-pass --no-calibration --sustain-seconds 0 > "$OUT/mfma.json" 2> "$OUT/mfma.err"
echo "exit $?"
# keep what is published small: stats + counter CSVs only
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
find "$OUT" -name "*.csv" | head -20
