"""Wave-cycle breakdown per kernel from one rocprofv3 PMC pass:
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \\
            SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer
(MI355X_MICROARCH.md, "rocprofv3 PMC slots": WAIT_ANY = wave parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall,
ACTIVE_INST_ANY = issuing; the three are disjoint and add up to WAVE_CYCLES.)
Usage: python tools/pmc_stalls.py <counter_collection.csv> [top_n]"""
import collections, csv, sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
big = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        calls[k] += 1
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
order = sorted(rows, key=lambda k: -rows[k]["SQ_WAVE_CYCLES"])[:top]
print("| kernel | launches | parked (waitcnt / barrier) | issue stall | issuing | of which LDS issue stall | LDS bank-conflict / LDS active |")
print("|---|---|---|---|---|---|---|")
for k in order:
    c = rows[k]
    w = c["SQ_WAVE_CYCLES"] or 1.0
    lds = c["SQ_LDS_IDX_ACTIVE"] or 1.0
    print(f"| `{k[:70]}` | {calls[k]} | {c['SQ_WAIT_ANY'] / w:.2f} | {c['SQ_WAIT_INST_ANY'] / w:.2f} | "
          f"{c['SQ_ACTIVE_INST_ANY'] / w:.2f} | {c['SQ_WAIT_INST_LDS'] / w:.2f} | {c['SQ_LDS_BANK_CONFLICT'] / lds:.3f} |")
