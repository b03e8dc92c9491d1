"""CPU-only checks: the C-ABI library loads and exports every symbol include/cartnet_hip.h declares (no compute
calls without a GPU), host-side containers / collation / schedules behave like the reference's counterparts."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "cartnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cartnet_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from cartnet_amd import build, lib
    build.build(verbose=False)                      # hipcc cross-compiles for gfx950 without a GPU
    cdll = ctypes.CDLL(lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(cdll, name), f"{name} is declared in include/cartnet_hip.h but not exported"
    # the ctypes prototypes cover exactly the declared set
    assert sorted(lib.PROTOTYPES) == declared
    assert lib.load().cartnet_abi_version() == 12 == lib.ABI_VERSION


def test_ctypes_mirrors_have_the_c_struct_layouts():
    """lib.load() compares sizeof of every struct that crosses the ABI with the library's own (cartnet_abi_struct_sizes);
    here the same numbers once more plus two offsets that moved this round (fields appended to CartnetGemmArgs /
    CartnetModel must sit where the C side reads them)."""
    from cartnet_amd import lib
    l = lib.load()
    sizes = (ctypes.c_size_t * 16)()
    n = l.cartnet_abi_struct_sizes(sizes, 16)
    mirrors = [lib.GemmArgs, lib.Shard, lib.Collated, lib.GemmProfile, lib.Groups, lib.LayerParams, lib.LayerBuffers,
               lib.Params, lib.Model, lib.BatchDesc, lib.GateGemmArgs, lib.IcfConv, lib.IcfParams, lib.IcfModel]
    assert n == len(mirrors)
    assert [ctypes.sizeof(m) for m in mirrors] == list(sizes[:n])
    # round 5: the gst_* block sits between dact_half and tile_policy (five pointers + gst_ld; tile_policy closes the struct)
    assert lib.GemmArgs.gst_g.offset == lib.GemmArgs.dact_half.offset + 4
    assert lib.GemmArgs.gst_ld.offset == lib.GemmArgs.gst_g.offset + 40
    # ... and dact_kind (ABI 10) closes it: gst_ld, tile_policy, dact_kind + 4 bytes of tail padding
    assert lib.GemmArgs.dact_kind.offset == lib.GemmArgs.tile_policy.offset + 4
    assert lib.GemmArgs.gather_rows.offset + 4 == ctypes.sizeof(lib.GemmArgs)
    assert lib.Model.grad_ready_user.offset + 8 == ctypes.sizeof(lib.Model)                          # last field


def test_host_side_argument_validation_without_gpu():
    """Shape / null checks happen on the host before any launch: callable on a machine with no GPU."""
    from cartnet_amd import lib
    l = lib.load()
    args = lib.GemmArgs()
    args.M, args.N, args.K = 4, 4, -1
    args.ngroups = args.nsegs = args.splitk = 1
    assert l.cartnet_gemm(ctypes.byref(args), None) != 0
    assert b"negative" in l.cartnet_last_error()
    assert l.cartnet_segment_sum(None, 6, None, None, 3, 6, None, 6, None) != 0      # W not a multiple of 4
    assert b"multiples of 4" in l.cartnet_last_error()
    assert l.cartnet_gate_scatter_nparts(10) == 3 and l.cartnet_gate_scatter_nparts(10**7) == 1024
    assert l.cartnet_node_nparts(1) == 1


def test_gate_statistics_epilogue_is_only_accepted_where_a_kernel_carries_it():
    """CartnetGemmArgs.gst_* (round 5): the host-side predicate cartnet_gemm_gate_stats_ok decides from shapes and
    pointers alone (no launch), and cartnet_gemm refuses what it rejects instead of computing without the sums."""
    from cartnet_amd import lib
    l = lib.load()
    a = lib.GemmArgs()
    assert l.cartnet_gemm_gate_stats_ok(ctypes.byref(a)) == 0                    # nothing set
    D, E = 256, 64 * 128
    a.M, a.N, a.K = E, D, D
    a.lda, a.ldb, a.ldc, a.ldr = 2 * D, 3 * D, D, D
    a.ngroups, a.nsegs, a.splitk, a.b_kstrided = 1, 2, 1, 1
    fake = 1 << 20                                                                # 16-byte aligned, never dereferenced
    a.A[0], a.A[1], a.B[0], a.B[1], a.C[0], a.resid[0] = fake, fake + 4 * D, fake, fake, fake, fake
    a.b_split_folded = fake
    a.colsum[0], a.colsq[0] = fake, fake
    a.gst_g, a.gst_ld, a.gst_mean_rstd, a.gst_gamma, a.gst_beta = fake, 2 * D, fake, fake, fake
    assert l.cartnet_gemm_gate_stats_ok(ctypes.byref(a)) == 1                    # the dE product of a D = 256 layer, 64 row tiles
    a.precision = 1
    assert l.cartnet_gemm_gate_stats_ok(ctypes.byref(a)) == 0                    # bf16x3 takes the narrow tiles below 96 row tiles
    a.M = 96 * 128
    assert l.cartnet_gemm_gate_stats_ok(ctypes.byref(a)) == 1
    a.precision = 2
    assert l.cartnet_gemm_gate_stats_ok(ctypes.byref(a)) == 0                    # plain bf16: no such kernel
    a.precision, a.M = 0, 63 * 128
    assert l.cartnet_gemm_gate_stats_ok(ctypes.byref(a)) == 0                    # too few row tiles: the narrow general kernel
    a.M = E
    a.colsq[0] = None
    assert l.cartnet_gemm_gate_stats_ok(ctypes.byref(a)) == 0                    # both sums or none
    assert l.cartnet_gemm(ctypes.byref(a), None) != 0 and b"gst_g" in l.cartnet_last_error()
    a.colsq[0] = fake
    a.N = a.ldc = a.ldr = 512
    assert l.cartnet_gemm_gate_stats_ok(ctypes.byref(a)) == 0                    # D = 512: the K-segments do not fold


def test_weight_image_sizes_and_gemm_precision_validation_without_gpu():
    """Pure host functions of the GEMM ABI: image sizes (6 B per element for the three bf16 planes, 4 B for the swizzled
    fp32 rows; 0 for shapes without an image) and the precision range check."""
    from cartnet_amd import lib
    l = lib.load()
    assert l.cartnet_gemm_split_b_bytes(256, 256) == 6 * 256 * 256
    assert l.cartnet_gemm_pack_b_bytes(256, 512) == 4 * 256 * 512
    for K, N in ((250, 256), (256, 200), (0, 256), (16, 0)):
        assert l.cartnet_gemm_split_b_bytes(K, N) == 0 and l.cartnet_gemm_pack_b_bytes(K, N) == 0
    assert l.cartnet_colstats_nparts(1) == 1 and l.cartnet_colstats_nparts(10 ** 6) == 1024
    args = lib.GemmArgs()
    args.M = args.N = args.K = 16
    args.lda = args.ldb = args.ldc = 16
    args.ngroups = args.nsegs = args.splitk = 1
    args.precision = 3
    buf = ctypes.create_string_buffer(16 * 16 * 4 + 64)
    base = (ctypes.addressof(buf) + 63) & ~63
    args.A[0] = args.B[0] = args.C[0] = base
    assert l.cartnet_gemm(ctypes.byref(args), None) != 0 and b"precision" in l.cartnet_last_error()


def test_product_model_fails_loudly_off_gpu():
    from cartnet_amd.model import CartNet
    from cartnet_amd.synthetic import make_batch
    m = CartNet(16, 8, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(make_batch(1, 5))


def test_state_dict_layout_matches_reference_keys():
    """SURVEY.md §8b key list (probed from the reference) -- shapes at D=256, R=64, L=4 and 2,498,438 parameters."""
    from cartnet_amd.model import CartNet
    m = CartNet(256, 64, 4)
    sd = m.state_dict()
    assert sum(p.numel() for p in m.parameters()) == 2_498_438
    expect = {
        "encoder.embedding.weight": (119, 512), "encoder.temperature_proj_atom.weight": (512, 1),
        "encoder.temperature_proj_atom.bias": (512,), "encoder.encoder_atom.1.weight": (256, 512),
        "encoder.encoder_edge.0.weight": (512, 67), "encoder.encoder_edge.2.weight": (256, 512),
        "encoder.rbf.means": (64,), "encoder.rbf.betas": (64,),
        "layers.0.MLP_aggr.0.weight": (256, 768), "layers.3.MLP_gate.2.weight": (256, 256),
        "layers.2.norm.running_mean": (256,), "layers.1.norm2.num_batches_tracked": (),
        "head.MLP.0.weight": (128, 256), "head.MLP.2.weight": (6, 128),
    }
    for k, shp in expect.items():
        assert tuple(sd[k].shape) == shp, k
    keys = list(sd)
    assert keys[0] == "encoder.embedding.weight" and keys[-1] == "head.MLP.2.bias"
    assert keys.index("layers.0.MLP_aggr.0.weight") < keys.index("layers.0.MLP_gate.0.weight") < \
        keys.index("layers.0.norm.weight") < keys.index("layers.0.norm2.weight")
    scalar = CartNet(64, 64, 2, temperature=False, cholesky=False).state_dict()
    assert "encoder.bias" in scalar and tuple(scalar["head.MLP.2.weight"].shape) == (1, 32)


def test_create_model_contract():
    from cartnet_amd.config import cfg, set_cfg
    from cartnet_amd import master
    set_cfg()
    cfg.model = "nope"
    with pytest.raises(Exception, match="Model not implemented"):
        master.create_model()
    set_cfg()


def test_batch_collation_follows_pyg_rules():
    from cartnet_amd.data import Batch, DataLoader
    from cartnet_amd.synthetic import make_crystal
    items = [make_crystal(i, n) for i, n in enumerate((5, 9, 3))]
    b = Batch.from_data_list(items)
    assert b.num_graphs == 3 and b.ptr.tolist() == [0, 5, 14, 17]
    assert b.batch.tolist() == [0] * 5 + [1] * 9 + [2] * 3
    e0, e1 = items[0].edge_index.shape[1], items[1].edge_index.shape[1]
    assert torch.equal(b.edge_index[:, :e0], items[0].edge_index)
    assert torch.equal(b.edge_index[:, e0:e0 + e1], items[1].edge_index + 5)
    assert bool((b.edge_index[1][1:] >= b.edge_index[1][:-1]).all())          # stays sorted by target
    assert b.temperature.shape == (3,) and b.cell.shape == (3, 3, 3)
    assert b.y.shape[0] == int(b.non_H_mask.sum())
    # sharded loader: ranks see disjoint crystals, together all of them (n = 11 is not a multiple of the world size:
    # nothing is dropped), the same number of batches on every rank, per-rank edge totals balanced
    ds = [make_crystal(i, 4 + (i % 5)) for i in range(11)]
    seen, edges = [], []
    for r in range(2):
        dl = DataLoader(ds, batch_size=2, shuffle=True, seed=7, rank=r, world_size=2)
        assert len(dl) == 3
        mine = [j for chunk in dl._batches() for j in chunk]
        seen.append(mine)
        edges.append(sum(int(ds[j].edge_index.shape[1]) for j in mine))
        assert sum(int(bb.num_graphs) for bb in dl if bb is not None) == len(mine)
    assert sorted(seen[0] + seen[1]) == list(range(11))
    assert abs(edges[0] - edges[1]) <= max(int(d.edge_index.shape[1]) for d in ds)


def test_edge_balanced_partition_properties():
    """cartnet_amd.distributed.balanced_partition / rank_batches (contiguous shards balanced by edge
    count): exhaustive, disjoint, ordered, non-empty parts, near-equal weights; equal batch counts across ranks."""
    import random
    from cartnet_amd.distributed import balanced_partition, rank_batches
    rnd = random.Random(3)
    for n in (0, 1, 2, 7, 64, 1000):
        for parts in (1, 2, 3, 8):
            w = [rnd.randint(0, 5000) for _ in range(n)]
            P = balanced_partition(w, parts)
            assert len(P) == parts and [i for r in P for i in r] == list(range(n))
            if n >= parts:
                assert all(len(r) > 0 for r in P)
            if n >= 1000:
                sums = [sum(w[i] for i in r) for r in P]
                assert max(sums) <= 1.02 * sum(w) / parts
    # one rank's share of the ADP epoch (162,270 crystals of 64-324 atoms over 8 ranks, batch 64): every rank takes the
    # same number of optimiser steps and no step carries more than a few percent above the mean number of edges
    w = [int(14.3 * rnd.randint(64, 324)) for _ in range(162270 // 8)]
    per_rank = [rank_batches(w, 64, r, 8) for r in range(8)]
    assert len({len(b) for b in per_rank}) == 1 and len(per_rank[0]) == -(-len(w) // (8 * 64))
    assert [i for b in per_rank for r in b for i in r] == list(range(len(w)))
    step_edges = [[sum(w[i] for i in r) for r in b] for b in per_rank]
    mean = sum(w) / (8 * len(per_rank[0]))
    assert max(max(s) for s in step_edges) <= 1.03 * mean
    # fewer crystals than ranks x steps: trailing empty ranges, still exhaustive
    tiny = [rank_batches([5, 5, 5], 1, r, 2) for r in range(2)]
    assert [i for b in tiny for r in b for i in r] == [0, 1, 2] and len(tiny[0]) == len(tiny[1]) == 2


def test_one_cycle_schedule_matches_torch():
    from cartnet_amd.optim import one_cycle_lr, one_cycle_momentum
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    total = 57
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=total, pct_start=0.01)
    for step in range(total):
        assert abs(opt.param_groups[0]["lr"] - one_cycle_lr(step, total, 1e-3, 0.01)) < 1e-12
        assert abs(opt.param_groups[0]["betas"][0] - one_cycle_momentum(step, total, 0.01)) < 1e-12
        opt.step()
        if step + 1 < total:
            sch.step()
    # a warm-up long enough for both phases (the reference's default 0.01 puts every step of a short run in the second)
    opt = torch.optim.Adam([p], lr=1e-3)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=total, pct_start=0.3)
    for step in range(total):
        assert abs(opt.param_groups[0]["lr"] - one_cycle_lr(step, total, 1e-3, 0.3)) < 1e-12
        assert abs(opt.param_groups[0]["betas"][0] - one_cycle_momentum(step, total, 0.3)) < 1e-12
        opt.step()
        if step + 1 < total:
            sch.step()


def test_augmentation_rotates_targets_and_directions():
    """dataset/datasetADP.py:33-39: y <- R^T y R, cart_dir <- cart_dir R, cell <- cell R."""
    from cartnet_amd.synthetic import augment_data, make_crystal, random_rotation
    d = make_crystal(3, 8)
    y0, dir0 = d.y.clone(), d.cart_dir.clone()
    g = torch.Generator().manual_seed(1)
    R = random_rotation(torch.Generator().manual_seed(1))
    assert torch.allclose(R @ R.t(), torch.eye(3), atol=1e-6) and abs(torch.det(R).item() - 1) < 1e-6
    augment_data(d, g)
    assert torch.allclose(d.y, R.t() @ y0 @ R, atol=1e-7)
    assert torch.allclose(d.cart_dir, dir0 @ R, atol=1e-7)
    assert torch.allclose(d.cart_dir.norm(dim=-1), torch.ones(d.cart_dir.shape[0]), atol=1e-5)


def test_bench_self_launch_builds_the_torchrun_child(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset: bench.py must start N ranks itself as a CHILD
    torch.distributed.run (never an exec, before any GPU call) and return the child's exit code."""
    import argparse
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class FakeChild:
        pid = 424242

        def __init__(self, cmd, env=None, **kw):
            seen["cmd"], seen["env"], seen["kw"] = cmd, env, kw

        def wait(self, timeout=None):
            seen.setdefault("waits", []).append(timeout)
            return 7

    monkeypatch.setattr(bench.subprocess, "Popen", FakeChild)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--share-gpu"])
    rc = bench.self_launch(argparse.Namespace(gpus=4, share_gpu=True))
    assert rc == 7
    assert seen["kw"].get("start_new_session") is True and seen["waits"][0] and seen["waits"][0] > 0     # own group, bounded wait
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and "--master-addr" in cmd and "127.0.0.1" in cmd
    assert cmd[-5:] == ["--gpus", "4", "--steps", "3", "--share-gpu"] and cmd[-6].endswith("bench.py")

    # a child that never finishes: its whole process group is killed (exactly that group) and the exit code is non-zero
    killed = []

    class Hung(FakeChild):
        def wait(self, timeout=None):
            if not killed:
                raise subprocess.TimeoutExpired("x", timeout)
            return -15

    monkeypatch.setattr(bench.subprocess, "Popen", Hung)
    monkeypatch.setattr(bench.os, "killpg", lambda pid, sig: killed.append((pid, sig)))
    monkeypatch.setenv("BENCH_LAUNCH_TIMEOUT", "0.01")
    assert bench.self_launch(argparse.Namespace(gpus=4, share_gpu=True)) == 124
    assert killed and killed[0][0] == Hung.pid
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["CARTNET_DIST_BACKEND"] == "gloo"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_telemetry_reads_a_sysfs_tree(tmp_path, monkeypatch):
    """cartnet_amd.telemetry against a fake amdgpu sysfs directory (the layout of profiles/r03_sysfs_probe.txt): starred DPM
    level, hwmon power / cap / labelled temperatures, missing files -> None, the sampler's min / mean / max."""
    from cartnet_amd import telemetry as tele
    d = tmp_path / "card"
    h = d / "hwmon" / "hwmon4"
    h.mkdir(parents=True)
    (d / "pp_dpm_sclk").write_text("S: 108Mhz\n0: 500Mhz\n1: 2400Mhz *\n")
    (d / "pp_dpm_mclk").write_text("0: 2000Mhz *\n")
    (h / "power1_input").write_text("1265000000\n")
    (h / "power1_cap").write_text("1400000000\n")
    (h / "temp2_input").write_text("53000\n")
    (h / "temp2_label").write_text("junction\n")
    (h / "temp3_input").write_text("47000\n")
    (h / "temp3_label").write_text("mem\n")
    (h / "freq1_input").write_text("2374000000\n")
    (h / "freq1_label").write_text("sclk\n")
    monkeypatch.setattr(tele, "_DIR_CACHE", {0: str(d)})
    s = tele.read(0)
    assert s["sclk_mhz"] == 2400 and s["mclk_mhz"] == 2000 and s["power_w"] == 1265.0 and s["power_cap_w"] == 1400.0
    assert s["temp_c"] == {"junction": 53.0, "mem": 47.0}
    c = tele.compact(s)
    assert c["junction_c"] == 53.0 and c["mem_c"] == 47.0 and c["edge_c"] is None
    (d / "pp_dpm_sclk").unlink()                              # falls back to hwmon freq1_input
    assert tele.read(0)["sclk_mhz"] == 2374
    smp = tele.Sampler(0, period=0.005).start()              # clamped to 0.05 s: one period for every pass, never a busy loop
    assert smp.period == 0.05
    import time
    time.sleep(0.13)
    out = smp.stop()
    assert out["samples"] >= 2 and out["power_w"]["max"] == 1265.0 and out["junction_c_max"] == 53.0
    monkeypatch.setattr(tele, "_DIR_CACHE", {0: None})
    empty = tele.read(0)
    assert empty["sclk_mhz"] is None and empty["temp_c"] == {}
    assert tele._active_level_mhz("0: 500Mhz\n1: 2400Mhz") is None


def test_traffic_json_is_what_the_committed_pmc_passes_give():
    """profiles/traffic.json (bench.py's `roofline.traffic`, DESIGN.md's bytes per step) is DERIVED data: recompute it
    from the committed rocprofv3 counter CSVs with tools/pmc_traffic.py and compare -- fp32 and bf16x3 steps."""
    import importlib.util, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    import glob
    rounds = sorted(os.path.basename(f)[:3] for f in glob.glob(os.path.join(prof, "r[0-9][0-9]_pmc_fetch_size.csv")))
    tag = rounds[-1] if rounds else "r00"       # traffic.json is published from the newest round's passes
    need = [os.path.join(prof, f"{tag}_pmc_{k}.csv") for k in ("fetch_size", "write_size", "fetch_size_x3", "write_size_x3")]
    if not all(os.path.exists(f) for f in need):
        pytest.skip("counter CSVs of this round are not in the tree")
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(root, "tools", "pmc_traffic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ref = json.load(open(os.path.join(prof, "traffic.json")))
    out = mod.analyse(need[0], need[1], verbose=False)
    x3 = mod.analyse(need[2], need[3], verbose=False)
    assert out["per_step"]["hbm_bytes"] == ref["per_step"]["hbm_bytes"]
    assert x3["per_step"]["hbm_bytes"] == ref["per_step_x3"]["hbm_bytes"]
    assert out["variants"]["fp32"]["tn256"] == ref["variants"]["fp32"]["tn256"]
    assert x3["variants"]["x3"]["nn256"] == ref["variants"]["x3"]["nn256"]
    # the gfx950 correction is in: fetched bytes are the counter (KB) x 1024 x 2
    assert ref["per_step"]["fetch_bytes"] > ref["per_step"]["write_bytes"] > 0
