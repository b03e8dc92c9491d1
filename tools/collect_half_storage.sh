#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and kernel stats of the precision-2 step with and without bf16
# storage.  -> gpurun_out/prof_half/{fp32store,bf16store}_{fetch,write,stats}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_half
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--preroll-steps 0 --no-cold --no-telemetry --no-calibration --precision 2 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-recipe-pass --sustain-seconds 0"
for mode in fp32store bf16store; do
  HS=""; [ $mode = bf16store ] && HS="--half-storage"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${mode}_fetch" -- python3 "$ROOT/bench.py" $ARGS $HS > "$OUT/${mode}_fetch.json" 2> "$OUT/${mode}_fetch.err" &&
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${mode}_write" -- python3 "$ROOT/bench.py" $ARGS $HS > "$OUT/${mode}_write.json" 2> "$OUT/${mode}_write.err" &&
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${mode}_stats" -- python3 "$ROOT/bench.py" --preroll-steps 0 --no-cold --no-telemetry --no-calibration --precision 2 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-recipe-pass --sustain-seconds 0 $HS > "$OUT/${mode}_stats.json" 2> "$OUT/${mode}_stats.err" || exit 1
  find "$OUT" -name "*kernel_trace.csv" -delete
  python3 "$ROOT/tools/pmc_traffic.py" $(find "$OUT/${mode}_fetch" -name "*counter_collection.csv") $(find "$OUT/${mode}_write" -name "*counter_collection.csv") "$OUT/${mode}_traffic.json" > /dev/null
  python3 -c "
import json; t=json.load(open('$OUT/${mode}_traffic.json')); d=json.loads(open('$OUT/${mode}_stats.json').read().strip().splitlines()[-1]); print('$mode', t['per_step'], d['ms_per_step'], d['value'])"
done
