#!/usr/bin/env python3
"""graphs/sec of the CartNet hot path (forward + MAE loss + backward + gradient all-reduce + Adam step) on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W``.  N > 1 runs one rank per GPU over RCCL: either the caller
starts the ranks (``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``, WORLD_SIZE must equal
N) or -- when WORLD_SIZE is unset -- bench.py starts them itself as a child torch.distributed.run BEFORE anything
touches the GPU, relays rank 0's JSON line and exits with the child's code.  Rank 0 prints ONE JSON line.  Workload = BASELINE.json configs[1]: CartNet L=4, D=256, R=64, fp32,
ADP-shaped synthetic crystals of 194 atoms (periodic 5 A radius graph, ~2.8k edges each), ``--graphs`` crystals per
rank per step (weak scaling: every rank owns its own crystals; one 10 MB gradient all-reduce per step).
Inputs are resident in HBM before the timed region.

Extra objects on the line:
  roofline      the dominant kernel (the fp32-MFMA GEMM variant with the largest share of the step) priced live with HIP events on the
                launch stream over the timed steps (cartnet_profile_gemm): executed 2*M*N*K FLOPs / measured time vs
                157.3 TFLOP/s
  cpu_baseline  the oracle (CPU restatement of the reference forward + autograd backward) timed on the host cores
                on a bounded sample (rank 0, N = 1 only)
  sustained     >= --sustain-seconds of the same steps after the timed region (graphs/s over the whole stretch and the
                min / max ms per step over 100-step windows): clock / thermal steady state on the record
  telemetry     sclk / mclk / socket power / temperatures of THIS rank's card from sysfs (cartnet_amd/telemetry.py, plain
                file reads): a snapshot before the timed region, samples every 50 ms DURING it, during the sustained
                stretch and at its end -- so a slow box can be told from a slow build
  calibration   the box's own ceiling for the GEMM kernel: 50 back-to-back launches of the plain two-group layer product
                (E rows, K = N = 256) on cn_gemm_f32nn_kernel, alone on the chip, as TFLOP/s and fraction of 157.3
  jarvis_bf16   BASELINE configs[2] (scripts/train_cartnet_jarvis.sh shapes: batch 64, Scalar_head, no temperature; 64
                crystals of 2-20 atoms), bf16 operands + bf16 storage: >= 20 timed steps after warm-up
  icomformer    BASELINE configs[4] (models/comformer.py:115-132, D = 256) on the headline's 64 x 194-atom batch, fp32
  ragged        the headline model on a ragged batch (64 crystals of 64-324 atoms), fp32
                (the three carry ms_per_step, host_enqueue_ms_per_step, whole_step_frac and telemetry like the headline)
``--no-telemetry`` starts no sampler thread (tools and profiler runs); every pass samples at the same period.
"""
from __future__ import annotations

import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

import torch            # importing torch does not initialise the GPU; nothing below touches it before self_launch()

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
FLOPS_REF_PER_GRAPH = 38.0e9       # fwd+bwd, reference formulation (E=2800, N=194, D=256, L=4)
FLOPS_EXEC_PER_GRAPH = 21.8e9      # algebraically restructured (what this build executes)


def build_batch(n_graphs: int, first: int, atoms: int):
    from cartnet_amd.synthetic import make_batch
    return make_batch(n_graphs, atoms, first=first)


def cpu_baseline(seconds_budget: float = 12.0, which: str = "cartnet"):
    """Time the oracle (plain-torch CPU restatement of models/cartnet.py -- or models/comformer.py for
    ``--model icomformer`` -- with autograd backward) on 4 crystals; ``which="jarvis"``: the configs[2] workload itself
    (CartNet with the scalar head on the 64 crystals of 2-20 atoms the jarvis_bf16 pass times)."""
    n_graphs = 4
    batch = build_batch(n_graphs, 10_000, 194)
    if which == "jarvis":
        from cartnet_amd.data import Batch
        from cartnet_amd.model import make_state_dict
        from cartnet_amd.synthetic import make_crystal
        from oracle import cartnet_ref as orc
        gen = torch.Generator().manual_seed(7)
        sizes = torch.randint(2, 21, (64,), generator=gen).tolist()
        n_graphs = 64
        batch = Batch.from_data_list([make_crystal(5000 + i, n, adp=False) for i, n in enumerate(sizes)])
        sd = make_state_dict(256, 64, 4, seed=0, cholesky=False, temperature=False)
        fwd = lambda p: orc.cartnet_forward(p, batch, num_layers=4, training=True, use_temperature=False, cholesky=False)
    elif which == "icomformer":
        from cartnet_amd.comformer import make_icomformer_state_dict
        from oracle import icomformer_ref as orc
        sd = make_icomformer_state_dict(256, seed=0)
        fwd = lambda p: orc.icomformer_forward(p, batch, training=True)
    else:
        from cartnet_amd.model import make_state_dict
        from oracle import cartnet_ref as orc
        sd = make_state_dict(256, 64, 4, seed=0)
        fwd = lambda p: orc.cartnet_forward(p, batch, num_layers=4, training=True)
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "rbf" not in k
                  else v.clone()) for k, v in sd.items()}
    def step():
        for v in params.values():
            if v.requires_grad:
                v.grad = None
        pred = fwd(params)
        (pred - batch.y).abs().mean().backward()

    def median_time(iters):
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2]

    # torch's CPU kernels stop scaling well before all cores of a large host: take the best of a few thread counts
    ncpu = os.cpu_count() or 1
    best = None
    for nt in sorted({min(ncpu, 8), min(ncpu, 32), min(ncpu, 64)}):
        torch.set_num_threads(nt)
        step()
        step()
        t = median_time(3)
        if best is None or t < best[1]:
            best = (nt, t)
    cores, one = best
    torch.set_num_threads(cores)
    iters = max(5, min(60, int(seconds_budget / max(one, 1e-3))))
    med = median_time(iters)
    torch.set_num_threads(1)          # scalar figure: one thread, two runs after one warm-up
    step()
    one_thread = min(median_time(1), median_time(1))
    torch.set_num_threads(cores)
    cpu_model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(n_graphs / med, 3), "unit": "graphs/s", "cores": cores, "kind": "port",
            "one_thread_value": round(n_graphs / one_thread, 3), "cpu_model": cpu_model,
            "sample": f"oracle fwd+bwd (fp32, train-mode BN) on {n_graphs} crystals "
                      f"({'2-20' if which == 'jarvis' else '194'} atoms each, N={int(batch.x.shape[0])}, "
                      f"E={int(batch.edge_index.shape[1])}), median of {iters} runs after warm-up, best of 8/32/64 "
                      f"torch CPU threads = {cores} (host: {cpu_model}, {ncpu} logical CPUs); one_thread_value: same "
                      f"step on 1 thread"}


def _csrc_digest() -> str:
    """sha256 over the kernel sources: profiles/traffic.json records the digest of the build its PMC passes measured, so
    the bench line can say whether the constant it quotes belongs to THIS build (there is no .git on the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cartnet_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()[:16]


TIMED_EVERY = 4      # launches of the dominant GEMM variant between two timed ones inside the timed region


def timed_pass(step, fresh, warm: int, steps: int, graphs: int, sampler_factory, peak_tflops: float, ops, world: int = 1):
    """Warm-up, one extra step with the GEMM launch timer on (FLOPs per step), then `steps` timed steps bracketed by
    synchronises: {ms_per_step, host_enqueue_ms_per_step, value, whole_step_frac, telemetry}."""
    spin = 5                   # untimed steps right before the timed loop (see --spinup in main)
    bs = [fresh() for _ in range(warm + 1 + spin + steps)]
    for b in bs[:warm]:
        step(b)
    torch.cuda.synchronize()
    ops.profile_gemm(True)
    step(bs[warm])
    torch.cuda.synchronize()
    ops.profile_gemm(False)
    variants = ops.profile_gemm_read()
    gflop = sum(v["flops"] for v in variants.values())
    # host time to enqueue one step on an IDLE queue (median of 3): over back-to-back steps the host of a C++-sequenced
    # model runs ahead until the device queue is full and then waits, which is not host work
    idle = []
    for b in [fresh() for _ in range(3)]:
        torch.cuda.synchronize()
        th = time.perf_counter()
        step(b)
        idle.append(time.perf_counter() - th)
    idle.sort()
    sampler = sampler_factory()
    torch.cuda.synchronize()
    prof = None
    if os.environ.get("BENCH_PROFILE_SUB"):          # tools: where the host time of a sub-pass goes (stderr)
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    import gc
    gc.collect()               # a generation-2 collection inside a 30 ms host-bound region would double it
    gc.disable()
    for b in bs[warm + 1:warm + 1 + spin]:
        step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in bs[warm + 1 + spin:]:
        loss = step(b)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(14)
    tel = sampler.stop() if sampler is not None else None
    if not torch.isfinite(loss):
        return None
    out = {"value": round(graphs * world * steps / dt, 2), "unit": "graphs/s", "steps": steps, "warmup": warm,
           "ms_per_step": round(1e3 * dt / steps, 3), "host_enqueue_ms_per_step": round(1e3 * idle[1], 3),
           "host_loop_ms_per_step": round(1e3 * t_enq / steps, 3),
           "gemm_flops_per_step": int(gflop),
           "whole_step_frac": round(gflop / (dt / steps) / 1e12 / peak_tflops, 4), "peak_tflops": peak_tflops}
    if variants:
        # the dominant cartnet_gemm variant of this workload, priced with HIP events on its launch stream during ONE
        # instrumented step right after the warm-up (every GEMM launch of that step carries an event pair; the timed
        # steps above carry none): executed 2 M N K / event time, in-step (two streams share the chip in backward)
        key = max(variants, key=lambda k: variants[k]["ms"])
        d = variants[key]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak_tflops, "unit": "TFLOP/s",
                           "frac": round(ach / peak_tflops, 4), "traffic": None,
                           "kernel": f"cartnet_gemm variant {key}", "launches": d["launches"],
                           "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                           "share_of_step": round(d["ms"] / (1e3 * dt / steps), 3),
                           "measured": "one instrumented step after the warm-up (an event pair on every GEMM launch)",
                           "all_gemm_variants_ms_per_step": {k: round(v["ms"], 3) for k, v in sorted(variants.items())
                                                             if v["ms"] >= 0.02},
                           "all_gemm_variants_frac": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak_tflops, 3)
                                                      for k, v in sorted(variants.items()) if v["ms"] >= 0.02}}
    if tel is not None:
        out["telemetry_during"] = tel
    return out


def self_launch(args) -> int:
    """``python bench.py --gpus N`` without an outer launcher: start the N ranks as a CHILD torch.distributed.run (never
    an exec, and before this process has made any GPU call), pass the child's output through and return its exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.share_gpu:
        env.setdefault("CARTNET_DIST_BACKEND", "gloo")      # several ranks on one card: RCCL needs one device per rank
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # The ranks run in a process group of their own and under a watchdog: a rank that hangs (in the rendezvous, in a
    # collective) ends with the whole group killed and a non-zero exit here instead of a bench that never returns.  The
    # ranks give up on their own after CARTNET_DIST_TIMEOUT seconds per collective (distributed.init_from_env).
    limit = float(os.environ.get("BENCH_LAUNCH_TIMEOUT", "1500"))
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return child.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {args.gpus} ranks did not finish within {limit:.0f} s -- killing them", file=sys.stderr, flush=True)
    except KeyboardInterrupt:
        pass
    try:
        os.killpg(child.pid, signal.SIGTERM)          # exactly the group started above
        child.wait(timeout=20)
    except subprocess.TimeoutExpired:
        os.killpg(child.pid, signal.SIGKILL)
        child.wait()
    except ProcessLookupError:
        pass
    return 124


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preroll-steps", type=int, default=200,
                    help="untimed steps before the warm-up steps: the card settles over seconds (see the comment in main)")
    ap.add_argument("--no-cold", action="store_true",
                    help="skip the cold figure (W untimed + K timed steps as the first work of the process)")
    ap.add_argument("--spinup", type=int, default=2,
                    help="untimed steps right before the timed region, after the warm-up's bookkeeping (see the comment there)")
    ap.add_argument("--graphs", type=int, default=64, help="crystals per rank per step")
    ap.add_argument("--atoms", type=int, default=194)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--precision", type=int, default=0, help="0: fp32 MFMA GEMMs, 1: bf16x3 split-operand MFMA, 2: plain bf16 operands")
    ap.add_argument("--no-x3-pass", action="store_true", help="skip the extra bf16x3 timed pass")
    ap.add_argument("--half-storage", action="store_true",
                    help="with --precision 2: the layers' edge-sized intermediates live in HBM as bf16 (CartNet.half_storage)")
    ap.add_argument("--no-recipe-pass", action="store_true",
                    help="skip the extra timed pass with BatchNorm groups of 4 (the reference recipe's micro-batches)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: ranks beyond the visible GPUs share them (gloo transport unless "
                         "CARTNET_DIST_BACKEND says otherwise)")
    ap.add_argument("--model", choices=("cartnet", "icomformer"), default="cartnet",
                    help="cartnet: the headline (BASELINE configs[1]); icomformer: BASELINE configs[4], the alternative "
                         "message-passing path (models/comformer.py) on the same crystals")
    ap.add_argument("--bn-group-size", type=int, default=0,
                    help="> 0: BatchNorm statistics and loss per group of this many crystals (the reference recipe's "
                         "micro-batches of 4 inside one pass, CartnetGroups); 0: one BatchNorm batch (the headline)")
    ap.add_argument("--no-calibration", action="store_true", help="skip the isolated plain-GEMM calibration launches")
    ap.add_argument("--sustain-seconds", type=float, default=6.0,
                    help="length of the sustained stretch after the timed region (0 disables; N = 1 only)")
    ap.add_argument("--no-telemetry", action="store_true",
                    help="no sysfs sampler thread during any pass (the before / after snapshots stay)")
    ap.add_argument("--no-subconfigs", action="store_true",
                    help="skip the jarvis_bf16 / icomformer / ragged passes (BASELINE configs[2], [4] and the ragged batch)")
    ap.add_argument("--sub-steps", type=int, default=20, help="timed steps of each of those passes (after 5 warm-up steps)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    from cartnet_amd import distributed as cdist
    os.environ.setdefault("CARTNET_DIST_TIMEOUT", "300")     # a benchmark rank that never arrives must not hold the box
    rank, world, local = cdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU, the two must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the CartNet hot path has no CPU fallback")
    if args.share_gpu:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    seen = cdist.ranks_seen(dev)              # first collective: SUM of one 1 per rank
    if seen != world:
        raise SystemExit(f"the communicator carries {seen} rank(s), --gpus says {args.gpus}")

    from cartnet_amd import ops
    from cartnet_amd.config import cfg
    from cartnet_amd.model import CartNet
    from cartnet_amd.optim import FlatAdam

    cfg.radius = 5.0
    torch.manual_seed(0)
    icf = args.model == "icomformer"
    if icf:
        from cartnet_amd.comformer import iComformer
        if args.bn_group_size > 0:
            raise SystemExit("--bn-group-size applies to CartNet only")
        model = iComformer(256).to(dev).train()
    else:
        model = CartNet(dim_in=256, dim_rbf=64, num_layers=4).to(dev).train()
        model.bn_group_size = args.bn_group_size
    model.gemm_precision = args.precision
    if args.half_storage:
        model.half_storage = True
    opt = FlatAdam(model, lr=1e-3)
    opt.direct_grads = True      # (as train_epoch does: the loss reads the parameters through the model only)
    base = build_batch(args.graphs, 100_000 + rank * args.graphs, args.atoms).to(dev)
    N, E = int(base.x.shape[0]), int(base.edge_index.shape[1])

    def fresh():
        # forward overwrites batch.x / batch.edge_attr (reference semantics): re-arm the inputs, keep them in HBM
        b = base.clone()
        b.num_graphs = base.num_graphs
        b._cartnet_layout = getattr(base, "_cartnet_layout", None)
        b._cartnet_mask_index = getattr(base, "_cartnet_mask_index", None)
        return b

    from cartnet_amd.train import grouped_loss, compute_loss
    from cartnet_amd import train as ctrain

    # more than one rank (or CARTNET_DIST_FORCE): the gradient all-reduce goes out in buckets under backward (GradSync)
    gsync = cdist.GradSync(opt.flat_grad, measure=True) if (cdist._active() and not icf) else None

    def step(b):
        pred, true = model(b)
        if args.bn_group_size > 0:
            loss = grouped_loss(pred, true, b, args.bn_group_size)[0]
        else:
            loss = compute_loss(pred, true)[0]          # MAE (cfg.loss default), train/metrics.py:26
        if gsync is not None:
            model.grad_sync = gsync
            ctrain.backward(loss)
            model.grad_sync = None
            scale = gsync.finish()
        else:
            ctrain.backward(loss)
            scale = cdist.all_reduce_gradients(opt.flat_grad)
        opt.step(scale)
        opt.zero_grad()
        return loss

    batches = [fresh() for _ in range(args.warmup + args.spinup + args.steps)]
    # the CSR/CSC layout is part of the hot path: build it inside the steps (not cached) so it is timed
    for b in batches:
        b._cartnet_layout = None
        b._cartnet_mask_index = None
    # Kernel timer: every GEMM variant is priced with HIP events during the warm-up steps after the first (same
    # launches, same two-stream overlap as the timed steps); inside the timed region only the dominant variant carries
    # event pairs -- ~26 events per step instead of ~130, whose markers cost ~1.5 % of the step.
    timer = not args.no_kernel_timer
    warm_summary, only = {}, None
    # Pre-roll: the card's power management takes SECONDS to settle once work arrives (profiles/r04_sustained_60s.json:
    # 16.3 ms per step over the first 100 steps of a stretch, 14.01-14.02 for the 57 s after it; the K = 10 timed steps
    # of that process read 14.76).  `--preroll-steps` untimed steps (default 200, ~3 s; the same count on every rank) run
    # before the W warm-up steps, so that W and K mean what they say on a card that has been working.
    # Cold figure: the driver's flags taken literally -- W untimed steps, then K timed ones between a
    # barrier + synchronise on both sides, as the FIRST work of the process on a card that may have been idle.  Reported
    # next to the steady value, never instead of it; these W + K steps count towards the pre-roll.
    cold = None
    cold_steps = 0
    if not args.no_cold:
        def raw():
            b = fresh()
            b._cartnet_layout = None
            b._cartnet_mask_index = None
            return b
        for _ in range(args.warmup):
            step(raw())
        cbs = [raw() for _ in range(args.steps)]
        cdist.barrier()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for b in cbs:
            step(b)
        torch.cuda.synchronize()
        cdist.barrier()
        torch.cuda.synchronize()
        dtc = cdist.max_over_ranks(time.perf_counter() - tc, dev)
        cold_steps = args.warmup + args.steps
        cold = {"value": round(args.graphs * world * args.steps / dtc, 2), "unit": "graphs/s",
                "ms_per_step": round(1e3 * dtc / args.steps, 3), "steps": args.steps, "warmup": args.warmup,
                "note": "the first W + K steps of the process (W untimed, K timed), before any pre-roll: includes the "
                        "card's clock / power ramp from idle (profiles/r04_sustained_60s.json)"}
        del cbs
    for i in range(max(0, args.preroll_steps - cold_steps)):
        b = fresh()
        b._cartnet_layout = None
        b._cartnet_mask_index = None
        step(b)
        if i % 50 == 49:
            torch.cuda.synchronize()        # (bounds the allocator's backlog of batches in flight)
    for i in range(args.warmup):
        if timer and i == 1:
            torch.cuda.synchronize()
            ops.profile_gemm(True)
        step(batches[i])
    if timer and args.warmup >= 2:
        torch.cuda.synchronize()
        ops.profile_gemm(False)
        warm_summary = ops.profile_gemm_read()
        if warm_summary:
            only = warm_summary[max(warm_summary, key=lambda k: warm_summary[k]["ms"])]["variant"]
    from cartnet_amd import telemetry as tele
    telemetry = {"before_timed": tele.compact(tele.read(local))} if rank == 0 else None

    def new_sampler():          # one period for every pass; None with --no-telemetry or off rank 0
        return tele.Sampler(local).start() if (rank == 0 and not args.no_telemetry) else None
    cdist.barrier()
    torch.cuda.synchronize()
    # one in TIMED_EVERY launches of the dominant variant carries an event pair inside the timed region (4 is coprime with
    # the nine such launches of a step: the sample walks over all of them): the pairs cost the step ~0.7 % when every
    # launch had one (same-box A/B, tools/experiments/ab_timer.sh), which is the difference between 14.0 and 14.1 ms
    timed_every = TIMED_EVERY if (timer and only is not None) else 1
    sampler = new_sampler()
    # The bookkeeping above (reading ~260 event pairs, a sysfs snapshot, starting the sampler) leaves the card idle for
    # tens of milliseconds after the W warm-up steps and its clocks drop; the first timed steps then read ~1 % long
    # (K = 10: 14.16 ms against 14.01 sustained on the same box).  `--spinup` more untimed steps (default 2) run right
    # before the barrier + synchronise that open the timed region, so that the K timed steps see the card as the W
    # warm-up steps left it.
    for i in range(args.spinup):
        step(batches[args.warmup + i])
    cdist.barrier()
    torch.cuda.synchronize()
    ops.profile_gemm(timer, only=only, every=timed_every)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(batches[args.warmup + args.spinup + i])
    t_enq = time.perf_counter() - t0          # host time to enqueue all steps (GPU-bound when << dt)
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0        # this rank's own K steps (before it waits for the others)
    cdist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if sampler is not None:
        telemetry["during_timed"] = sampler.stop()
    ops.profile_gemm(False)
    timed_summary = ops.profile_gemm_read() if (rank == 0 and not args.no_kernel_timer) else {}
    dt = cdist.max_over_ranks(dt, dev)
    rank_lo, rank_hi = cdist.min_max_over_ranks(dt_rank, dev)
    if not torch.isfinite(loss):
        raise SystemExit("non-finite loss")

    # Sustained stretch (outside the timed region, same step, same batches re-armed): long enough for the clock and
    # the power management to settle under the fp32 matrix load; windows of 100 steps are timed with one host sync each.
    sustained = None
    if args.sustain_seconds > 0:
        win = 100
        est = max(cdist.max_over_ranks(dt, dev) / args.steps, 1e-4)      # the same window count on every rank
        n_win = max(1, int(args.sustain_seconds / (est * win) + 0.999))
        pool = [fresh() for _ in range(4)]
        ms = []
        sampler = new_sampler()
        cdist.barrier()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for wi in range(n_win):
            tw = time.perf_counter()
            for i in range(win):
                bx = pool[i & 3]
                bx.x, bx.edge_attr = base.x, None                 # forward replaced them: re-arm without a copy
                bx._cartnet_layout = None
                bx._cartnet_mask_index = None
                step(bx)
            torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - tw) / win)
        if sampler is not None:
            telemetry["during_sustained"] = sampler.stop()
        if telemetry is not None:
            telemetry["sustained_end"] = tele.compact(tele.read(local))
        cdist.barrier()
        tot = cdist.max_over_ranks(time.perf_counter() - ts0, dev)
        sustained = {"seconds": round(tot, 2), "steps": n_win * win,
                     "value": round(args.graphs * world * n_win * win / tot, 2), "unit": "graphs/s",
                     "ms_per_step_min_window": round(min(ms), 3), "ms_per_step_max_window": round(max(ms), 3),
                     "ms_per_step_first_window": round(ms[0], 3), "ms_per_step_last_window": round(ms[-1], 3),
                     "window_steps": win}

    # The timed steps run backward on two streams, so a GEMM's event-bracketed duration includes the time it shares
    # the chip with the weight-gradient stream.  Three extra single-stream steps (outside the timed region) give the
    # same launches undisturbed: kernel quality without the overlap.
    isolated = {}
    if world == 1 and not args.no_kernel_timer and not icf:   # single process only: the extra steps would need every rank
        model.overlap_weight_gradients = False
        extra = [fresh() for _ in range(3)]
        torch.cuda.synchronize()
        ops.profile_gemm(True)
        for bx in extra:
            step(bx)
        torch.cuda.synchronize()
        ops.profile_gemm(False)
        isolated = ops.profile_gemm_read()
        model.overlap_weight_gradients = True

    # Calibration: what THIS box gives the plain GEMM kernel, alone on the chip and warm (right after the sustained
    # stretch): the two-group layer product gs = h @ [W2g | W2a] at the benchmark's E rows, K = N = 256, weights as a
    # DMA image -> cn_gemm_f32nn_kernel<false>, 50 launches back to back between two events on the launch stream.
    calibration = None
    if not args.no_calibration:
        Dm = 256
        gcal = torch.Generator().manual_seed(7)
        hcal = torch.randn(E, 2 * Dm, generator=gcal).to(dev)
        wcal = [(torch.randn(Dm, Dm, generator=gcal) * 0.05).to(dev) for _ in range(2)]
        ocal = torch.empty(E, 2 * Dm, device=dev)
        img = ops.pack_b(wcal)

        def cal():
            ops.gemm([hcal[:, :Dm], hcal[:, Dm:]], wcal, [ocal[:, :Dm], ocal[:, Dm:]], b_kstrided=True, b_split=img,
                     precision=0)
        for _ in range(150):            # ~60 ms of the same launches first: a burst from an idle card reads up to 25 % low
            cal()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        sampler = new_sampler()
        ev0.record()
        for _ in range(50):
            cal()
        ev1.record()
        torch.cuda.synchronize()
        cal_tel = sampler.stop() if sampler is not None else None
        us = 1e3 * ev0.elapsed_time(ev1) / 50
        fl = 2.0 * E * Dm * Dm * 2
        calibration = {"kernel": "cn_gemm_f32nn_kernel<false>: plain two-group product [E, 256] x [256, 256] x 2, E = %d" % E,
                       "launches": 50, "avg_launch_us": round(us, 2), "achieved": round(fl / us / 1e6, 2),
                       "unit": "TFLOP/s", "peak": PEAK_FP32_MFMA_TFLOPS,
                       "frac": round(fl / us / 1e6 / PEAK_FP32_MFMA_TFLOPS, 4), "telemetry": cal_tel}
        del hcal, ocal, img

    # Second timed pass with the bf16x3 split-operand GEMMs (same parity budget, every parity test runs both): the
    # same K steps bracketed the same way, reported next to the headline fp32-MFMA number, never instead of it.
    x3 = None
    if args.precision == 0 and not args.no_x3_pass:
        model.gemm_precision = 1
        extra = [fresh() for _ in range(2 + args.steps)]
        for bx in extra:
            bx._cartnet_layout = None
            bx._cartnet_mask_index = None
        for bx in extra[:2]:
            step(bx)
        cdist.barrier()
        torch.cuda.synchronize()
        sampler3 = new_sampler()
        t1 = time.perf_counter()
        for bx in extra[2:]:
            loss3 = step(bx)
        torch.cuda.synchronize()
        cdist.barrier()
        torch.cuda.synchronize()
        dt3 = cdist.max_over_ranks(time.perf_counter() - t1, dev)
        tel3 = sampler3.stop() if sampler3 is not None else None
        # its own roofline: ONE more step, outside the timed region above, with an event pair on every
        # cartnet_gemm launch; six bf16 MFMA products per fp32 product -> the ceiling is the dense bf16 peak / 6
        x3_roof = None
        if not args.no_kernel_timer:
            bx = fresh()                        # (every rank runs the step: it carries the gradient all-reduce)
            torch.cuda.synchronize()
            if rank == 0:
                ops.profile_gemm(True)
            step(bx)
            torch.cuda.synchronize()
            v3 = None
            if rank == 0:
                ops.profile_gemm(False)
                v3 = ops.profile_gemm_read()
            if v3:
                peak3 = 2500.0 / 6.0
                k3 = max(v3, key=lambda k: v3[k]["ms"])
                d3 = v3[k3]
                ach3 = d3["flops"] / (d3["ms"] * 1e-3) / 1e12
                g3 = sum(v["flops"] for v in v3.values())
                x3_roof = {"bound": "mfma", "achieved": round(ach3, 2), "peak": round(peak3, 1), "unit": "TFLOP/s",
                           "frac": round(ach3 / peak3, 4), "traffic": None,
                           "kernel": f"cartnet_gemm variant {k3} (bf16x3: fp32-equivalent FLOPs against the dense bf16 peak / 6)",
                           "launches": d3["launches"], "avg_launch_us": round(1e3 * d3["ms"] / d3["launches"], 2),
                           "measured": "one instrumented step after the timed ones (an event pair on every GEMM launch), in-step",
                           "whole_step_frac": round(g3 / (dt3 / args.steps) / 1e12 / peak3, 4),
                           "all_gemm_variants_ms_per_step": {k: round(v["ms"], 3) for k, v in sorted(v3.items()) if v["ms"] >= 0.05},
                           "all_gemm_variants_frac": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak3, 3)
                                                      for k, v in sorted(v3.items()) if v["ms"] >= 0.05}}
        model.gemm_precision = 0
        if torch.isfinite(loss3):
            x3 = {"value": round(args.graphs * world * args.steps / dt3, 2), "unit": "graphs/s",
                  "ms_per_step": round(1e3 * dt3 / args.steps, 3),
                  # the bf16x3 products are bound by the clock the chip holds under dense bf16 matrix work (in-kernel:
                  # 1.53-1.70 GHz against 2.3 GHz under the fp32 MFMA loop, profiles/r03_exp_x3_power.md)
                  "telemetry_during": tel3,
                  "note": "same step with gemm_precision=1: every fp32 product rebuilt from six bf16 MFMA products "
                          "(operands split exactly into three bf16 pieces), fp32 accumulate; same 1e-5 parity tests"}
            if x3_roof is not None:
                x3["roofline"] = x3_roof

    # bf16 mode with bf16 storage (BASELINE configs[2]'s arithmetic at this batch: DESIGN.md 4c), same bracketing
    bf16s = None
    if args.precision == 0 and not args.no_x3_pass and not icf and args.bn_group_size == 0:
        model.gemm_precision, model.half_storage = 2, True
        extra = [fresh() for _ in range(2 + args.steps)]
        for bx in extra:
            bx._cartnet_layout = None
            bx._cartnet_mask_index = None
        for bx in extra[:2]:
            step(bx)
        cdist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for bx in extra[2:]:
            loss4 = step(bx)
        torch.cuda.synchronize()
        cdist.barrier()
        torch.cuda.synchronize()
        dt4 = cdist.max_over_ranks(time.perf_counter() - t1, dev)
        model.gemm_precision, model.half_storage = 0, False
        if torch.isfinite(loss4):
            bf16s = {"value": round(args.graphs * world * args.steps / dt4, 2), "unit": "graphs/s",
                     "ms_per_step": round(1e3 * dt4 / args.steps, 3), "dtype": "bf16",
                     "note": "same step with gemm_precision=2 and half_storage: plain bf16 MFMA operands, fp32 accumulate, "
                             "pre / gs / dpre kept in HBM as bf16 -- bf16 tolerances (3e-2), NOT the 1e-5 parity budget"}

    # Third timed pass (CartNet, default BatchNorm batching only): the reference's ADP recipe -- micro-batches of 4
    # crystals, 16 accumulated per optimiser step (scripts/train_cartnet_adp.sh:4) -- carried as BatchNorm groups of 4
    # inside the same 64-crystal pass: reference-recipe semantics, reported next to the headline.
    recipe = None
    if not icf and args.precision == 0 and args.bn_group_size == 0 and not args.no_recipe_pass and args.graphs > 4:
        model.bn_group_size = 4

        def gstep(b):
            pred, true = model(b)
            loss = grouped_loss(pred, true, b, 4)[0]
            ctrain.backward(loss)
            scale = cdist.all_reduce_gradients(opt.flat_grad)
            opt.step(scale)
            opt.zero_grad()
            return loss
        extra = [fresh() for _ in range(2 + args.steps)]
        for bx in extra:
            bx._cartnet_layout = None
            bx._cartnet_mask_index = None
        for bx in extra[:2]:
            gstep(bx)
        cdist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for bx in extra[2:]:
            loss4 = gstep(bx)
        torch.cuda.synchronize()
        cdist.barrier()
        torch.cuda.synchronize()
        dt4 = cdist.max_over_ranks(time.perf_counter() - t1, dev)
        model.bn_group_size = 0
        if torch.isfinite(loss4):
            recipe = {"value": round(args.graphs * world * args.steps / dt4, 2), "unit": "graphs/s",
                      "ms_per_step": round(1e3 * dt4 / args.steps, 3),
                      "note": f"same step with BatchNorm statistics, running-statistics updates and loss per group of 4 "
                              f"crystals ({-(-args.graphs // 4)} groups): the reference recipe batch 4 x accumulation "
                              f"{-(-args.graphs // 4)} in one pass; literal micro-batches of 4 run at ~1.6k graphs/s"}

    # Inference rate (eval mode, no_grad: what --inference / --montecarlo of main.py run), same batch: untimed region
    eval_fwd = None
    if world == 1 and not args.no_recipe_pass:
        model.eval()
        with torch.no_grad():
            for _ in range(2):
                model(fresh())
            evb = [fresh() for _ in range(args.steps)]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for bx in evb:
                model(bx)
            torch.cuda.synchronize()
            dte = time.perf_counter() - t1
        eval_fwd = {"value": round(args.graphs * args.steps / dte, 2), "unit": "graphs/s",
                    "ms_per_batch": round(1e3 * dte / args.steps, 3),
                    "note": "forward only, eval mode (running BatchNorm statistics), torch.no_grad: nothing kept for backward"}
        if args.precision == 0 and not args.no_x3_pass:      # the same pass with bf16x3 products (same parity budget)
            model.gemm_precision = 1
            with torch.no_grad():
                for _ in range(2):
                    model(fresh())
                evb = [fresh() for _ in range(args.steps)]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for bx in evb:
                    model(bx)
                torch.cuda.synchronize()
                dte3 = time.perf_counter() - t1
            model.gemm_precision = 0
            eval_fwd["bf16x3_ms_per_batch"] = round(1e3 * dte3 / args.steps, 3)
            eval_fwd["bf16x3_value"] = round(args.graphs * args.steps / dte3, 2)
        model.train()

    # BASELINE configs[2] / configs[4] and the ragged batch, driver-timed: compact sub-objects with the
    # same bracketing as the headline (warm-up, then >= 20 steps between two synchronises), single rank only
    sub_cfg = {}
    if world == 1 and not args.no_subconfigs and not icf and args.precision == 0 and args.bn_group_size == 0:
        from cartnet_amd.data import Batch
        from cartnet_amd.synthetic import make_crystal, make_batch

        def clone_of(b0):
            def f():
                b = b0.clone()
                b.num_graphs = b0.num_graphs
                b._cartnet_layout = None
                b._cartnet_mask_index = None
                return b
            return f

        def train_step_of(mdl, optim):
            def f(b):
                pred, true = mdl(b)
                loss = compute_loss(pred, true)[0]
                ctrain.backward(loss)
                optim.step(cdist.all_reduce_gradients(optim.flat_grad))
                optim.zero_grad()
                return loss
            return f
        # (i) configs[2]: scripts/train_cartnet_jarvis.sh:5-6 -- batch 64, accumulation 1, Scalar_head (models/cartnet.py:
        # 323-327), no temperature (main.py:183); the reference does not state graph sizes: 2-20 atoms
        gen = torch.Generator().manual_seed(7)
        sizes = torch.randint(2, 21, (64,), generator=gen).tolist()
        jb = Batch.from_data_list([make_crystal(5000 + i, n, adp=False) for i, n in enumerate(sizes)]).to(dev)
        jm = CartNet(256, 64, 4, temperature=False, cholesky=False).to(dev).train()
        jm.gemm_precision, jm.half_storage = 2, True
        jopt = FlatAdam(jm, lr=1e-3)
        jopt.direct_grads = True
        r = timed_pass(train_step_of(jm, jopt), clone_of(jb), 5, 5 * args.sub_steps, 64, new_sampler, 2500.0, ops)
        if r is not None:
            r.update({"workload": f"BASELINE configs[2]: CartNet L=4 D=256 Scalar_head, no temperature, 64 crystals of 2-20 "
                                  f"atoms per step (N={int(jb.x.shape[0])}, E={int(jb.edge_index.shape[1])}), bf16 MFMA "
                                  "operands, bf16 storage of pre / gs / dpre, fp32 accumulate", "dtype": "bf16"})
            if not args.no_cpu_baseline:
                r["cpu_baseline"] = cpu_baseline(4.0, which="jarvis")
            sub_cfg["jarvis_bf16"] = r
        del jm, jopt, jb
        # (ii) configs[4]: iComformer (models/comformer.py:115-132) on the headline's batch, fp32 MFMA
        from cartnet_amd.comformer import iComformer
        im = iComformer(256).to(dev).train()
        im.gemm_precision = 0
        iopt = FlatAdam(im, lr=1e-3)
        r = timed_pass(train_step_of(im, iopt), fresh, 5, args.sub_steps, args.graphs, new_sampler,
                       PEAK_FP32_MFMA_TFLOPS, ops)
        if r is not None:
            r.update({"workload": f"BASELINE configs[4]: iComformer D=256 (4 attention layers + edge-update layer, Cholesky "
                                  f"head) fp32 train step on the headline's batch (N={N}, E={E})", "dtype": "f32"})
            if not args.no_cpu_baseline:
                r["cpu_baseline"] = cpu_baseline(5.0, which="icomformer")
            sub_cfg["icomformer"] = r
        del im, iopt
        torch.cuda.empty_cache()
        # (iii) the headline model on a ragged batch: crystals of 64-324 atoms (uniform, mean 194)
        rb = make_batch(args.graphs, None, first=300_000).to(dev)
        r = timed_pass(step, clone_of(rb), 3, args.sub_steps, args.graphs, new_sampler, PEAK_FP32_MFMA_TFLOPS, ops)
        if r is not None:
            r.update({"workload": f"headline model and step on {args.graphs} crystals of 64-324 atoms "
                                  f"(N={int(rb.x.shape[0])}, E={int(rb.edge_index.shape[1])})", "dtype": "f32"})
            sub_cfg["ragged"] = r
        del rb
        torch.cuda.empty_cache()

    graphs_total = args.graphs * world * args.steps
    value = graphs_total / dt
    out = {
        "metric": "graphs/sec (iComformer D=256 on ADP shapes, ~194 atoms/~2.8k edges), forward+backward+Adam" if icf
        else "graphs/sec (CartNet 4x256, ~194 atoms/~2.8k edges), forward+backward+Adam",
        "value": round(value, 2), "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup": args.spinup, "preroll_steps": args.preroll_steps,
        # every step run before the timed region opened: the cold W + K, the rest of the pre-roll, W warm-up steps, the spin-up
        "untimed_steps_total": max(args.preroll_steps, cold_steps) + args.warmup + args.spinup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if args.precision == 2 else "f32", "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[4]: iComformer D=256 (4 attention layers + edge-update layer, Cholesky "
                                f"head) fp32 train step, " if icf else
                                f"BASELINE configs[1]: CartNet L=4 D=256 R=64 fp32 train step, ") +
                               f"{args.graphs} synthetic "
                               f"ADP crystals x {args.atoms} atoms per GPU per step (N={N} atoms, E={E} edges per GPU)",
                   "graphs_per_gpu_per_step": args.graphs, "parallelism": f"graph-sharded dp{world}",
                   "batchnorm_groups": (f"{-(-args.graphs // args.bn_group_size)} groups of {args.bn_group_size} crystals "
                                        "(reference-recipe micro-batches in one pass)") if args.bn_group_size > 0
                   else "one (the whole per-GPU batch)",
                   "gemm_precision": "fp32 MFMA" if args.precision == 0 else
                   ("bf16x3 split-operand MFMA (six bf16 MFMA products per fp32 product, fp32 accumulate) for all 256-wide GEMMs"
                    if args.precision == 1 else
                    "plain bf16 MFMA operands, fp32 accumulate; " +
                    ("bf16 storage of pre / gs / dpre" if args.half_storage else "fp32 storage"))},
        "host_enqueue_ms_per_step": round(1e3 * t_enq / args.steps, 3),
    }
    if not icf:
        out["path_tflops_executed"] = round(value * FLOPS_EXEC_PER_GRAPH * (E / args.graphs / 2800.0) / 1e12 / world, 2)
        out["path_tflops_reference_equiv"] = round(value * FLOPS_REF_PER_GRAPH * (E / args.graphs / 2800.0) / 1e12 / world, 2)
    # whole-path HBM figures (north_star asks for the fraction of the HBM roofline): compulsory bytes of the reference
    # formulation (68 MB per 2800-edge crystal, fwd+bwd, perfect fusion) and the bytes the PMC passes
    # counted for one step of this build (profiles/traffic.json), both over the measured step time, against 8 TB/s
    try:
        tjson = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        tjson = {}
    per_step = tjson.get("per_step", {})
    step_s = dt / args.steps
    alg = 68.0e6 * (E / 2800.0)
    if not icf:
        out["path_hbm"] = {"peak_TBps": 8.0,
                           "algorithmic_bytes_per_step": int(alg), "algorithmic_TBps": round(alg / step_s / 1e12, 3),
                           "algorithmic_frac": round(alg / step_s / 8.0e12, 4),
                           "counted_from": "profiles/traffic.json (committed rocprofv3 PMC passes of the fp32 step; a constant "
                                           "of the build, NOT measured by this run)",
                           # which build the constant was measured on (tools/collect_profiles.sh stamps both), and whether
                           # the kernel sources have changed since: a stale constant is visible in the line
                           "counted_from_commit": tjson.get("commit"),
                           "counted_from_csrc_digest": tjson.get("csrc_digest"),
                           "counted_is_stale": (tjson.get("csrc_digest") != _csrc_digest()) if tjson else None,
                           "counted_bytes_per_step": per_step.get("hbm_bytes"),
                           "counted_TBps": round(per_step["hbm_bytes"] / step_s / 1e12, 3) if per_step.get("hbm_bytes") else None,
                           "counted_frac": round(per_step["hbm_bytes"] / step_s / 8.0e12, 4) if per_step.get("hbm_bytes") else None}
    if cold is not None:
        out["cold"] = cold
    if world > 1 or cdist._active():
        import torch.distributed as _dist
        out["ranks_seen"] = cdist.ranks_seen(dev)
        out["backend"] = _dist.get_backend() if _dist.is_initialized() else None
        # every rank's own K steps (before the closing barrier): min / max over the ranks
        out["rank_ms_per_step"] = {"min": round(1e3 * rank_lo / args.steps, 3), "max": round(1e3 * rank_hi / args.steps, 3)}
        out["allreduce"] = ("bucketed under backward (head, layers L-1..0, encoder: distributed.GradSync)"
                            if gsync is not None else "one flat all-reduce after backward")
        out["allreduce_exposed_ms_per_step"] = (round(gsync.exposed_ms(), 4) if gsync is not None and
                                                gsync.exposed_ms() is not None else None)
    if sustained is not None:
        out["sustained"] = sustained
    if telemetry is not None:
        telemetry["end_of_run"] = tele.compact(tele.read(local))
        out["telemetry"] = telemetry
    if calibration is not None:
        out["calibration"] = calibration
    if x3 is not None:
        out["bf16x3"] = x3
    if bf16s is not None:
        out["bf16_with_bf16_storage"] = bf16s
    if recipe is not None:
        out["reference_recipe_groups_of_4"] = recipe
    if eval_fwd is not None:
        out["eval_forward"] = eval_fwd
    out.update(sub_cfg)
    if rank == 0:
        summ = timed_summary
        if summ:
            key = max(summ, key=lambda k: summ[k]["ms"])
            d = summ[key]
            # per-variant table: from the warm-up steps when the timed region only carried the dominant variant
            vsumm, vsteps = (warm_summary, max(1, args.warmup - 1)) if warm_summary else (summ, args.steps)
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            # HBM bytes per launch of this variant from the committed rocprofv3 PMC passes (profiles/traffic.json:
            # FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate passes); null if never profiled
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tv = json.load(open(tpath)).get("variants", {}).get("fp32" if args.precision == 0 else "x3", {})
                    tv = tv.get(key.split("[")[0], {})
                    traffic = tv.get("hbm_bytes_per_launch_edge_rows" if "[E-rows" in key else "", tv.get("hbm_bytes_per_launch"))
                except Exception:
                    traffic = None
            # bf16x3 kernels retire an fp32 product with six bf16 MFMAs: their ceiling is the dense bf16 peak / 6
            peak = PEAK_FP32_MFMA_TFLOPS if args.precision == 0 else (2500.0 / 6.0 if args.precision == 1 else 2500.0)
            fam = "f32" if args.precision == 0 else "x3"
            kernel_name = f"cn_gemm_{fam}{'tn' if key.startswith('tn') else 'nn'}_kernel"
            if "+out" in key:
                kernel_name = f"cn_gemm_{fam}nn_actout_kernel"      # gemm_f32ao.h / gemm_x3ao.h
            elif key.startswith("nn128") and args.precision == 0:
                kernel_name = "cn_gemm_f32nn128_kernel"             # gemm_f32w128.h
            if key.split("+")[0].split("[")[0].endswith("p"):
                kernel_name = "cn_gemm_f32p_kernel"                 # gemm_f32p.h: the persistent kernel
            # traffic.json aggregates every launch of the kernel template (all shapes): a per-launch average
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1),
                               "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                               "kernel": f"cartnet_gemm variant {key} ({kernel_name})", "launches": d["launches"],
                               "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                               "sampled_every": timed_every,
                               "share_of_step": round(d["ms"] * timed_every / (1e3 * dt), 3),
                               "all_gemm_variants_ms_per_step": {k: round(v["ms"] / vsteps, 3)
                                                                 for k, v in sorted(vsumm.items())
                                                                 if v["ms"] / vsteps >= 0.05},
                               "all_gemm_variants_from": "warm-up steps 2.." if warm_summary else "timed steps"}
            # the whole step against the same peak: every cartnet_gemm FLOP of a step over the step's wall time (two
            # co-running GEMMs each read ~2x their isolated duration in `frac` above; this figure does not care)
            gflop = sum(v["flops"] for v in vsumm.values()) / vsteps
            out["roofline"]["whole_step"] = {"gemm_flops_per_step": int(gflop),
                                             "achieved": round(gflop / step_s / 1e12, 2),
                                             "frac": round(gflop / step_s / 1e12 / peak, 4)}
            # the same figures as scalars of `roofline` itself (a parser that keeps only scalar keys keeps them)
            out["roofline"]["whole_step_achieved"] = out["roofline"]["whole_step"]["achieved"]
            out["roofline"]["whole_step_frac"] = out["roofline"]["whole_step"]["frac"]
            if calibration is not None:
                out["roofline"]["calibration_frac"] = calibration["frac"]
                out["roofline"]["calibration_avg_launch_us"] = calibration["avg_launch_us"]
            # all launches of the same kernel template (every shape), for comparison with rocprofv3's per-kernel average
            base = key.split("[")[0]
            same = [v for k, v in vsumm.items() if k.split("[")[0] == base]
            out["roofline"]["kernel_avg_launch_us_all_shapes"] = round(
                1e3 * sum(v["ms"] for v in same) / max(1, sum(v["launches"] for v in same)), 2)
            if isolated:
                same_i = [v for k, v in isolated.items() if k.split("[")[0] == base]
                if same_i:
                    out["roofline"]["kernel_avg_launch_us_all_shapes_isolated"] = round(
                        1e3 * sum(v["ms"] for v in same_i) / max(1, sum(v["launches"] for v in same_i)), 2)
            if key in isolated:
                di = isolated[key]
                achi = di["flops"] / (di["ms"] * 1e-3) / 1e12
                out["roofline"]["isolated"] = {
                    "note": "same launches in 3 extra single-stream steps (no overlap with the weight-gradient stream)",
                    "achieved": round(achi, 2), "frac": round(achi / peak, 4),
                    "avg_launch_us": round(1e3 * di["ms"] / di["launches"], 2)}
                out["roofline"]["isolated_achieved"] = out["roofline"]["isolated"]["achieved"]
                out["roofline"]["isolated_frac"] = out["roofline"]["isolated"]["frac"]
                out["roofline"]["isolated_avg_launch_us"] = out["roofline"]["isolated"]["avg_launch_us"]
            if isolated:
                # the step as the sum of its GEMM launches' isolated durations (what two streams would cost run serially)
                out["roofline"]["isolated_gemm_ms_per_step"] = round(sum(v["ms"] for v in isolated.values()) / 3, 3)
                out["roofline"]["isolated_gemm_variants_us_per_launch"] = {
                    k: round(1e3 * v["ms"] / max(1, v["launches"]), 1) for k, v in sorted(isolated.items())
                    if v["ms"] / 3 >= 0.05}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(which=args.model)
        elif world > 1:
            # (a SCALE record explains itself)
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = "the CPU oracle is timed on rank 0 of the N = 1 line only"
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        bad = out.get("ranks_seen") != world
        dist.destroy_process_group()
        if bad:
            raise SystemExit(f"ranks_seen = {out.get('ranks_seen')} but --gpus {args.gpus}")


if __name__ == "__main__":
    main()
