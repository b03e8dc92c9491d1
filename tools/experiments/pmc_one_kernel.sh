#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the kernels one tool launches (two PMC passes, no trace domain):
#   tools/experiments/pmc_one_kernel.sh <tag> python3 tools/exp_f32p.py 177140 gather      (GPU box; CARTNET_LIB selects a variant)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PROG=$1; shift
ARGS=()
for a in "$@"; do case "$a" in tools/*|bench.py) ARGS+=("$ROOT/$a");; *) ARGS+=("$a");; esac; done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- "$PROG" "${ARGS[@]}" > "$OUT/fetch.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- "$PROG" "${ARGS[@]}" > "$OUT/write.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for which in ("fetch", "write"):
    f = glob.glob(f"{out}/{which}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(acc.items(), key=lambda x: -x[1][1])[:6]:
        per = v / n * (2 if which == "fetch" else 1) * 1024 / 1e6     # KB -> MB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
        print(f"{which:5s} {per:9.1f} MB per launch  x{n:4d}  {k}")
PY
