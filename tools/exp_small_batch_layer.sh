#!/bin/bash
# VERDICT r5 item 3 (GPU box): the one-launch layer forward (tools/exp_small_batch_layer.py) next to today's chain of the same
# layer inside the model's own step at configs[2] sizes, precision 2, fp32 storage -- from a kernel trace: per forward layer the
# time from the end of the previous layer's node update to the end of this one's (its seven launches and their gaps).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/small_batch_layer
mkdir -p "$OUT"
timeout -k 10 300 python3 "$ROOT/tools/exp_small_batch_layer.py" 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
JARVIS_ONLY=2,0 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/tools/bench_jarvis.py" > "$OUT/bench.txt" 2> "$OUT/err.txt"
grep precision "$OUT/bench.txt"
F=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
upd = [i for i, r in enumerate(rows) if "cn_node_update_fwd_kernel" in r[2]]
chains = []
for a, b in zip(upd, upd[1:]):
    names = [rows[i][2].split("(")[0].replace("void ", "").replace("cn_gemm::", "") for i in range(a + 1, b + 1)]
    if len(names) == 7 and not any("bwd" in n or "loss" in n or "head" in n for n in names):
        chains.append((rows[b][1] - rows[a][1], [rows[i][1] - rows[i][0] for i in range(a + 1, b + 1)], names))
chains = chains[len(chains) // 2:]                     # steady state
import statistics
print(f"today's chain of one layer forward (layers 1-3 of {len(chains)} forward passes): median {statistics.median(c[0] for c in chains) / 1e3:.1f} us "
      f"end to end; kernels " + " + ".join(f"{statistics.median(c[1][k] for c in chains) / 1e3:.1f}" for k in range(7)) + " us")
print("  " + " | ".join(n[:34] for n in chains[-1][2]))
PY
rm -f "$F"
