"""HBM bytes per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; tools/collect_profiles.sh).

FETCH_SIZE (KB) is doubled for gfx950 (wide coalesced 128-B reads are tallied at 64 B, MI355X_MICROARCH.md "HBM");
WRITE_SIZE (KB) is taken as is.  Output: JSON keyed by kernel name, averages over all launches of that kernel in the
profiled process (since round 2 the PMC passes run fp32 steps only, `bench.py --no-x3-pass`: 1 warm-up + 2 timed;
a process that also ran the bf16x3 pass is recognised by its x3 kernels and labelled accordingly).
Usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
                                    [<fetch csv of a --precision 1 run> <write csv of a --precision 1 run>]
With the last two (passes over bf16x3 steps only) the output gains variants["x3"], "kernels_x3" and "per_step_x3".
"""
import csv, json, sys, collections


BIG_GRID = 500_000      # threads: the edge-sized (E-row) GEMM launches of the benchmark batch have >= 1384 x 512


def per_kernel(path, counter, min_grid=0):
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or int(r["Grid_Size"]) < min_grid:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        cnt[r["Kernel_Name"]] += 1
    return tot, cnt


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def analyse(fpath, wpath, verbose=True):
    f, fc = per_kernel(fpath, "FETCH_SIZE")
    w, wc = per_kernel(wpath, "WRITE_SIZE")
    out = {"_about": __doc__.strip()}
    rows = []
    for k in f:
        n = fc[k]
        fetch = 2.0 * 1024.0 * f[k] / n
        write = 1024.0 * w.get(k, 0.0) / max(wc.get(k, 1), 1)
        rows.append((fetch * n + write * n, k, n, fetch, write))
    rows.sort(reverse=True)
    for _, k, n, fetch, write in rows:
        out[short(k)] = {"launches_profiled": n, "fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(write),
                         "hbm_bytes_per_launch": int(fetch + write)}
    # the same numbers under bench.py's variant keys (nn/tn + tile width + fused SiLU), launch-weighted
    import re
    variants = {"fp32": {}, "x3": {}}
    for _, k, n, fetch, write in rows:
        name = short(k)
        m = re.match(r"cn_gemm::cn_gemm_kernel<(\w+), (\w+), (\d+), (\w+), (\w+), (\w+), (\d)>", name)
        mode = "fp32"
        if m:
            a_ks, b_ks, bn, a_act, b_act, _fast, prec = m.groups()
            if prec != "0":
                mode = "x3"
        else:
            m3 = re.match(r"cn_gemm::cn_gemm_f32(nn|tn)_kernel<(\w+)>", name)
            m2 = re.match(r"cn_gemm::cn_gemm_x3(nn|tn)_kernel<(\w+), (\w+)>", name)
            m5 = re.match(r"cn_gemm::cn_gemm_x3nn16_kernel<(\w+)(?:, \d+)?>", name)      # gemm_x3s.h: 16x16x32 MFMA shape (round 5: <A_ACT, GSTK>)
            if m5:
                key = "nn256" + ("+silu(A)" if m5.group(1) == "true" else "")
                v = variants["x3"].setdefault(key, {"launches_profiled": 0, "fetch": 0.0, "write": 0.0})
                v["launches_profiled"] += n
                v["fetch"] += fetch * n
                v["write"] += write * n
                continue
            # gemm_f32p.h (round 6): the persistent kernel <A_ACT, ACT_OUT, KIND, NS> -> bench.py's "nn256p..." keys
            m6 = re.match(r"cn_gemm::cn_gemm_f32p_kernel<(\w+), (\w+), (\d+), (\d+)>", name)
            if m6:
                key = "nn256p" + ("+silu(A)" if m6.group(1) == "true" else "") + ("+out" if m6.group(2) == "true" else "")
                v = variants["fp32"].setdefault(key, {"launches_profiled": 0, "fetch": 0.0, "write": 0.0})
                v["launches_profiled"] += n
                v["fetch"] += fetch * n
                v["write"] += write * n
                continue
            m4 = re.match(r"cn_gemm::cn_gemm_f32nn128_kernel<(\w+)(?:, \d+)?>", name)
            if m4 or name == "cn_gemm::cn_gemm_f32nn_actout_kernel":
                key = ("nn128" + ("+silu(A)" if m4.group(1) == "true" else "")) if m4 else "nn256+silu(A)+out"
                v = variants["fp32"].setdefault(key, {"launches_profiled": 0, "fetch": 0.0, "write": 0.0})
                v["launches_profiled"] += n
                v["fetch"] += fetch * n
                v["write"] += write * n
                continue
            if m3:
                m2 = m3
            elif not m2 or m2.group(3) == "true":
                continue
            else:
                mode = "x3"
            a_ks, bn = ("true" if m2.group(1) == "tn" else "false"), "256"
            a_act = m2.group(2) if m2.group(1) == "nn" else "false"
            b_act = m2.group(2) if m2.group(1) == "tn" else "false"
        key = ("tn" if a_ks == "true" else "nn") + bn + ("+silu(A)" if a_act == "true" else "") + \
              ("+silu(B)" if b_act == "true" else "")
        v = variants[mode].setdefault(key, {"launches_profiled": 0, "fetch": 0.0, "write": 0.0})
        v["launches_profiled"] += n
        v["fetch"] += fetch * n
        v["write"] += write * n
    for mode in variants:
        for key, v in variants[mode].items():
            n = v["launches_profiled"]
            variants[mode][key] = {"launches_profiled": n, "fetch_bytes_per_launch": int(v["fetch"] / n),
                                   "write_bytes_per_launch": int(v["write"] / n),
                                   "hbm_bytes_per_launch": int((v["fetch"] + v["write"]) / n)}
    # the activation x weight kernels again, edge-sized launches only (the N-row node projections share the kernel)
    fb, fbc = per_kernel(fpath, "FETCH_SIZE", BIG_GRID)
    wb, wbc = per_kernel(wpath, "WRITE_SIZE", BIG_GRID)
    for k in fb:
        name = short(k)
        for pat, mode, key in ((r"cn_gemm::cn_gemm_f32nn_kernel<false>", "fp32", "nn256"),
                               (r"cn_gemm::cn_gemm_f32nn128_kernel<false>", "fp32", "nn128"),
                               (r"cn_gemm::cn_gemm_f32nn128_kernel<false, 0>", "fp32", "nn128"),
                               (r"cn_gemm::cn_gemm_f32tn_kernel<false>", "fp32", "tn256"),
                               (r"cn_gemm::cn_gemm_x3nn_kernel<false, false>", "x3", "nn256"),
                               (r"cn_gemm::cn_gemm_x3nn16_kernel<false>", "x3", "nn256"),
                               (r"cn_gemm::cn_gemm_x3nn16_kernel<false, 0>", "x3", "nn256")):
            if name == pat and key in variants[mode]:
                n = fbc[k]
                variants[mode][key]["hbm_bytes_per_launch_edge_rows"] = int(
                    (2.0 * 1024.0 * fb[k] + 1024.0 * wb.get(k, 0.0)) / n)
                variants[mode][key]["launches_profiled_edge_rows"] = n
    # The weight-gradient kernel: split-K gives its E-row and N-row launches the same grid (512 workgroups), so the
    # edge-sized launches are told apart by what they fetched (>= 400 MB: both operands are [E, 256] x groups); the
    # write pass is matched launch by launch through the kernel's dispatch ORDER (same command, same sequence).
    def per_dispatch(path, counter, kernel):
        rows_ = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(path))
                 if r["Counter_Name"] == counter and short(r["Kernel_Name"]) == kernel]
        rows_.sort()
        return [v for _, v in rows_]
    for tnk, mode in (("cn_gemm::cn_gemm_f32tn_kernel<false>", "fp32"), ("cn_gemm::cn_gemm_x3tn_kernel<false, false>", "x3")):
        fd, wd = per_dispatch(fpath, "FETCH_SIZE", tnk), per_dispatch(wpath, "WRITE_SIZE", tnk)
        if fd and len(fd) == len(wd) and "tn256" in variants[mode]:
            big = [i for i, v in enumerate(fd) if 2.0 * 1024.0 * v >= 4.0e8]
            if big:
                variants[mode]["tn256"]["hbm_bytes_per_launch_edge_rows"] = int(
                    sum(2.0 * 1024.0 * fd[i] + 1024.0 * wd[i] for i in big) / len(big))
                variants[mode]["tn256"]["launches_profiled_edge_rows"] = len(big)
    out["variants"] = variants
    # whole-step traffic: every dispatch of the profiled process / number of optimiser steps in it (cn_adam_kernel)
    steps = max(1, max((fc[k] for k in f if "adam" in k), default=1))
    tot_f = sum(2.0 * 1024.0 * f[k] for k in f)
    tot_w = sum(1024.0 * w[k] for k in w)
    out["per_step"] = {"steps_profiled": steps, "fetch_bytes": int(tot_f / steps), "write_bytes": int(tot_w / steps),
                       "hbm_bytes": int((tot_f + tot_w) / steps),
                       "note": "fabric-side bytes (Infinity-Cache hits included) per optimiser step, averaged over the "
                               + ("fp32 and bf16x3" if any("x3" in k for k in f) else "fp32-MFMA") +
                               " steps of the profiled bench process"}
    if verbose:
        for _, k, n, fetch, write in rows[:16]:
            print(f"{short(k)[:70]:70s} n={n:4d} fetch {fetch/1e6:8.1f} MB write {write/1e6:8.1f} MB")
    return out


def main():
    out = analyse(sys.argv[1], sys.argv[2])
    if len(sys.argv) > 5:
        x = analyse(sys.argv[4], sys.argv[5], verbose=False)
        out["variants"]["x3"] = x["variants"]["x3"]
        out["per_step_x3"] = x["per_step"]
        out["per_step_x3"]["note"] = "the same for bf16x3 steps (bench.py --precision 1 under the same two PMC passes)"
        out["kernels_x3"] = {k: v for k, v in x.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}
    json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
