"""Matrix-pipe utilisation per kernel from a rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE).

busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the counter adds up the cycles in
which each SIMD's matrix pipe is busy (32 per v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32), GRBM_GUI_ACTIVE
is summed over the 8 XCDs (MI355X_MICROARCH.md).  The effective clock is GRBM_GUI_ACTIVE / 8 / kernel time.
Usage: python tools/pmc_mfma.py <counter_collection.csv> [top_n]  -> markdown table
"""
import collections, csv, sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
        dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 10
print("| kernel | launches | avg us | clock GHz | matrix pipe busy |")
print("|---|---|---|---|---|")
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:top]:
    n = cnt[k]
    if n == 0 or acc[k]["GRBM_GUI_ACTIVE"] == 0:
        continue
    gui = acc[k]["GRBM_GUI_ACTIVE"] / 8 / n
    busy = acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / n
    print(f"| `{k}` | {n} | {dur[k]/n/1e3:.1f} | {gui/(dur[k]/n):.2f} | {busy/(gui*1024):.3f} |")
