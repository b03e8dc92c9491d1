"""Per-stream timeline of ONE training step from a rocprofv3 --kernel-trace CSV (kernel_trace.csv).

usage: python tools/timeline.py <kernel_trace.csv> [step_index_from_end]

Splits the trace into steps at cn_adam_kernel, takes one steady-state step and prints: wall time, time during which
at least one kernel runs (union), per-queue busy time, idle gaps on the chip, and per kernel name the summed duration
plus how much of it ran ALONE (no other kernel in flight) -- the un-hidden part of the HBM-bound kernels."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
adam = [i for i, r in enumerate(rows) if "cn_adam_kernel" in r[2]]
if len(adam) < back + 1:
    raise SystemExit("not enough steps in the trace")
lo, hi = adam[-back - 1] + 1, adam[-back] + 1
step = rows[lo:hi]
t0, t1 = step[0][0], max(r[1] for r in step)
print(f"step of {len(step)} launches, wall {1e-6 * (t1 - t0):.3f} ms")


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("cn_gemm::", "")
    n = n.split("(")[0]
    return n[:70]


# sweep: count of kernels in flight
ev = []
for s, e, n, q in step:
    ev.append((s, 1, n))
    ev.append((e, -1, n))
ev.sort()
active = defaultdict(int)
nact = 0
last = t0
union = 0
alone = defaultdict(int)
total = defaultdict(int)
gaps = 0
for t, d, n in ev:
    if nact > 0:
        union += t - last
        if nact == 1:
            only = next(k for k, v in active.items() if v > 0)
            alone[short(only)] += t - last
    else:
        gaps += t - last
    active[n] += d
    nact += d
    last = t
for s, e, n, q in step:
    total[short(n)] += e - s
print(f"chip busy (union) {1e-6 * union:.3f} ms, idle gaps {1e-6 * gaps:.3f} ms")
byq = defaultdict(int)
for s, e, n, q in step:
    byq[q] += e - s
for q, v in sorted(byq.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: busy {1e-6 * v:.3f} ms")
print(f"\n{'kernel':72s} {'calls':>5s} {'sum ms':>8s} {'alone ms':>9s}")
calls = defaultdict(int)
for s, e, n, q in step:
    calls[short(n)] += 1
for k, v in sorted(total.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{k:72s} {calls[k]:5d} {1e-6 * v:8.3f} {1e-6 * alone.get(k, 0):9.3f}")
print(f"{'TOTAL':72s} {len(step):5d} {1e-6 * sum(total.values()):8.3f} {1e-6 * sum(alone.values()):9.3f}")

# --list: every launch of the step in start order, one line each: start offset, duration, queue, kernel -- the Gantt view
if "--list" in sys.argv:
    qs = sorted({q for _, _, _, q in step})
    print("\nstart_us   dur_us  queue  kernel")
    for s, e, n, q in step:
        print(f"{1e-3 * (s - t0):8.1f} {1e-3 * (e - s):8.1f}  q{qs.index(q)}  {short(n)}")
