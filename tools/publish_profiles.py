"""Copy the outputs of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into profiles/ and write the summary.
Usage: python tools/publish_profiles.py gpurun_out/prof_r01b r01"""
import glob, json, os, shutil, subprocess, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
fetch = glob.glob(os.path.join(src, "fetch", "*", "*_counter_collection.csv"))[0]
write = glob.glob(os.path.join(src, "write", "*", "*_counter_collection.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_bench_n1_kernel_stats.csv"))
shutil.copy(fetch, os.path.join(dst, f"{tag}_pmc_fetch_size.csv"))
shutil.copy(write, os.path.join(dst, f"{tag}_pmc_write_size.csv"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, f"{tag}_bench_n1_under_rocprof.json"))
shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, f"{tag}_bench_n1.json"))
fetch3 = glob.glob(os.path.join(src, "fetch_x3", "*", "*_counter_collection.csv"))
write3 = glob.glob(os.path.join(src, "write_x3", "*", "*_counter_collection.csv"))
extra = []
if fetch3 and write3:        # the same two passes over bf16x3 steps (bench.py --precision 1)
    shutil.copy(fetch3[0], os.path.join(dst, f"{tag}_pmc_fetch_size_x3.csv"))
    shutil.copy(write3[0], os.path.join(dst, f"{tag}_pmc_write_size_x3.csv"))
    extra = [fetch3[0], write3[0]]
subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), fetch, write,
                os.path.join(dst, "traffic.json"), *extra], check=True, stdout=subprocess.DEVNULL)
# provenance of the constant bench.py quotes as path_hbm.counted_*: the commit published from and the digest of the kernel
# sources the passes ran on (collect_profiles.sh); publishing passes of OTHER sources than the tree's is refused
sys.path.insert(0, root)
import bench as _bench
dig_file = os.path.join(src, "csrc_digest.txt")
measured = open(dig_file).read().strip() if os.path.exists(dig_file) else None
if measured != _bench._csrc_digest() and "--force" not in sys.argv:
    raise SystemExit(f"{src} measured kernel sources {measured}, the tree is at {_bench._csrc_digest()}: re-run "
                     "tools/collect_profiles.sh on this tree (or pass --force)")
tj = json.load(open(os.path.join(dst, "traffic.json")))
tj["csrc_digest"] = measured
tj["commit"] = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
json.dump(tj, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
mfma_csv = glob.glob(os.path.join(src, "mfma", "*", "*_counter_collection.csv"))
mfma_table = ""
if mfma_csv:
    shutil.copy(mfma_csv[0], os.path.join(dst, f"{tag}_pmc_mfma_busy.csv"))
    mfma_table = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_mfma.py"), mfma_csv[0], "10"],
                                check=True, capture_output=True, text=True).stdout
table = subprocess.run([sys.executable, os.path.join(root, "tools", "summarize_stats.py"), stats, "36"],
                       check=True, capture_output=True, text=True).stdout
under = json.loads(open(os.path.join(src, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
plain = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
rf = plain["roofline"]
traffic = json.load(open(os.path.join(dst, "traffic.json")))
import csv, re
rows = list(csv.DictReader(open(stats)))
kname = re.search(r"\((cn_gemm_\w+)\)", rf["kernel"]).group(1)                  # kernel behind the dominant launch group
cands = [r for r in rows if kname in r["Name"]]
dom = max(cands, key=lambda r: float(r["TotalDurationNs"]))
vkey = rf["kernel"].split("variant ")[1].split(" (")[0]
tv = traffic["variants"]["fp32"].get(vkey.split("[")[0], {})
tbytes = tv.get("hbm_bytes_per_launch_edge_rows", tv.get("hbm_bytes_per_launch"))
md = f"""# Round {int(tag[1:3])} — rocprofv3 `--kernel-trace --stats` of the default bench command (1x MI355X)

Command (on the GPU box, `tools/collect_profiles.sh`): `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-recipe-pass --sustain-seconds 0`

Files: `{tag}_bench_n1_kernel_stats.csv` (raw per-kernel stats), `{tag}_bench_n1_under_rocprof.json` (the bench line printed under the profiler),
`{tag}_bench_n1.json` (un-profiled default run incl. `cpu_baseline` and `sustained`), `{tag}_pmc_fetch_size.csv` / `{tag}_pmc_write_size.csv` (separate `--pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-x3-pass`: fp32-MFMA steps only) and `traffic.json` (HBM bytes per launch per kernel derived from them by
`tools/pmc_traffic.py`: FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as is).

The profiled process runs, in this order: 13 fp32-MFMA training steps on two streams (3 warm-up + 10 timed), 3 single-stream fp32 steps (the `roofline.isolated` pass),
12 bf16x3 steps (2 + 10, the `bf16x3` object of the bench line) and the eval-forward pass; kernel names tell the GEMM families apart (`cn_gemm_f32nn / f32nn128 / f32nn_actout / f32tn_kernel`,
`cn_gemm_kernel<..., 0>` = fp32 MFMA; `cn_gemm_x3nn16 / x3nn / x3tn_kernel` = bf16x3).
Counted HBM traffic per optimiser step: {traffic['per_step']['hbm_bytes'] / 1e9:.1f} GB at fp32, {traffic.get('per_step_x3', {}).get('hbm_bytes', float('nan')) / 1e9:.1f} GB at bf16x3
(`{tag}_pmc_fetch_size_x3.csv` / `{tag}_pmc_write_size_x3.csv`: the same two passes with `--precision 1`; `traffic.json` -> `variants.x3`, `kernels_x3`, `per_step_x3`).

Bench line under the profiler: {under['value']} graphs/s, {under['ms_per_step']} ms/step (bf16x3 pass: {under['bf16x3']['value']} graphs/s, {under['bf16x3']['ms_per_step']} ms/step).
Un-profiled: {plain['value']} graphs/s, {plain['ms_per_step']} ms/step; sustained {plain.get('sustained', {}).get('value')} graphs/s over {plain.get('sustained', {}).get('seconds')} s; bf16x3 pass {plain['bf16x3']['value']} graphs/s, {plain['bf16x3']['ms_per_step']} ms/step; cpu_baseline {plain['cpu_baseline']['value']} graphs/s on {plain['cpu_baseline']['cores']} threads.

Dominant launch group (largest total time among the `cartnet_gemm` variants of the timed steps): `{rf['kernel']}`, i.e. kernel `{dom['Name'].split('(')[0]}`.
HIP events inside bench.py give {rf['avg_launch_us']} us per launch over the overlapped timed steps ({rf['achieved']} TFLOP/s, {rf['frac']} of the fp32 matrix peak) and
{rf.get('isolated', {}).get('avg_launch_us')} us isolated ({rf.get('isolated', {}).get('achieved')} TFLOP/s, {rf.get('isolated', {}).get('frac')});
rocprofv3 reports {float(dom['AverageNs'])/1e3:.1f} us averaged over all {dom['Calls']} launches of that kernel in the process -- every shape it runs, in the overlapped, warm-up and
isolated steps alike; bench.py's average over the same set of shapes is {rf.get('kernel_avg_launch_us_all_shapes', 'n/a')} us in the overlapped steps (taken in warm-up steps 2-3: inside the
timed steps only the dominant variant carries events) and {rf.get('kernel_avg_launch_us_all_shapes_isolated', 'n/a')} us isolated (the process mixes 13 overlapped with 3 isolated fp32 steps).
HBM traffic of that variant from the PMC passes: {(tbytes or float('nan'))/1e6:.0f} MB per launch (algorithmic: a weight-gradient product reads two [E, 256] x 2-group operands once = 726 MB;
an activation x weight product of the layer reads [E, 256 or 512] and writes [E, 512]: 1086 MB for dpre, 724 MB for dE).

Backward runs on two streams, so kernel durations of the two streams overlap in wall time (their sum exceeds the step time).

{table}

## Matrix-pipe utilisation (`{tag}_pmc_mfma_busy.csv`: `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`, same command as the other PMC passes)

Kernels run one at a time under the counter pass (no overlap between streams), so these are isolated figures; the clock column shows the chip
sustaining ~2.3 GHz under the fp32 MFMA and ~2.0 GHz under the bf16 MFMA (the peaks in MI355X_MICROARCH.md are quoted at 2.4 GHz).

{mfma_table}
"""
open(os.path.join(dst, f"{tag}_bench_n1_summary.md"), "w").write(md)
print(md[:1500])
