#!/usr/bin/env python3
"""Benchmark of the YOLOv1 hot path on MI355X.  Prints ONE JSON line (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--no-train] [--no-cpu] [--no-resnet]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

metric / value : images/sec of the forward pass of YOLOv1 (24-conv backbone + FC head) on synthetic
                 448x448 batches of `--batch` (64) images per GPU, inputs resident in HBM, bf16 storage /
                 fp32 MFMA accumulation -- BASELINE.json configs[1].  A "step" = one forward over one batch.
extra keys     : "train" = forward + YOLOLoss + backward + clip + Adam step (configs[2]; with N > 1 the
                 gradients are all-reduced over RCCL: configs[3]);  "nms" = decode + per-image NMS boxes/sec
                 (configs[4]'s post-processing);  "roofline" for the dominant kernel (implicit-GEMM MFMA
                 kernel, forward launches);  "cpu_baseline" = the stock-torch CPU restatement of the
                 reference network (oracle/torch_ref.py) timed on this box's host cores.
Multi-GPU: one process per GPU, batches shard (weak scaling), forward has no collective.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "yolo-v1_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def _sync_all(world):
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


def timed_steps(fn, steps, warmup, world):
    """W untimed + K timed steps bracketed by barrier + synchronize on both sides; max over ranks."""
    for _ in range(warmup):
        fn()
    _sync_all(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    _sync_all(world)
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def kernel_rooflines(fn, reps=2, alone=True):
    """run `fn` reps times with every MFMA launch bracketed by events on ITS launch stream (engine.TIMERS) and return
    {kernel kind: roofline block} for the MFMA kernels: achieved = algorithmic FLOPs of the launches / their summed durations.
    ``alone``: the backward pass on ONE stream, so that every launch has the chip to itself (the step itself runs the weight gradients
    on a second stream beside the data-gradient chain; with alone=False the events sit on both streams and the durations are those of
    the schedule that actually runs -- overlapping launches share the CUs, so a launch reads longer and the sums exceed the step)."""
    from yolo import engine
    keep = engine.WGRAD_STREAM
    if alone:
        engine.WGRAD_STREAM = False
    engine.TIMERS = []
    n0 = engine.IGEMM_LAUNCHES
    try:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    finally:
        timers, engine.TIMERS = engine.TIMERS, None
        engine.WGRAD_STREAM = keep
    launches = (engine.IGEMM_LAUNCHES - n0) // reps
    agg = {}
    for tag, kern, flops, e0, e1 in timers:
        d = agg.setdefault(kern, [0.0, 0.0, 0])
        d[0] += e0.elapsed_time(e1) / reps
        d[1] += flops / reps
        d[2] += 1
    how = "each launch timed alone -- one stream" if alone else "as scheduled: weight gradients on the second stream beside the data gradients, events on both streams"
    names = {"igemm": "igemm_persist_kernel / igemm_pipe_kernel / igemm_kernel / igemm_stream_kernel (implicit GEMM: conv / Linear forward and data gradient; " + how + ")",
             "wgrad": "wgrad_kernel / wgrad_wide_kernel / wgrad_pipe_kernel (weight gradient, ds_read_b64_tr_b16 operands; " + how + ")", "stem": "stem_fwd_kernel (7x7/s2 stem)"}
    out = {}
    for kern, (ms, fl, n) in agg.items():
        if fl <= 0 or kern not in names:
            continue
        ach = fl / (ms * 1e-3) / 1e12
        out[kern] = {"kernel": names[kern], "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None, "timed_sections_per_step": n // reps,
                     "kernel_ms_per_step": round(ms, 3), "flops_per_step": fl}
    if "igemm" in out:
        out["igemm"]["launches_per_step"] = launches
    return out


def _profile_traffic(suffix):
    """HBM bytes per launch from the newest committed profiles/*<suffix> (PMC passes cannot run inside a bench run)"""
    names = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(suffix))
    if not names:
        return None, "-"
    return round(json.load(open(os.path.join(ROOT, "profiles", names[-1])))["hbm_bytes_per_launch"]), names[-1]


def _launch_ranks(n: int) -> None:
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: start the N ranks ourselves (one process per GPU under
    torch.distributed.run, RCCL over xGMI) -- BEFORE this process makes any GPU call -- relay rank 0's JSON line and exit with the
    children's status.  (Run under `python -m torch.distributed.run ... bench.py --gpus N` the ranks already exist and this is
    never reached.)"""
    import socket
    import subprocess
    have = torch.cuda.device_count()          # counts devices without initialising the GPU
    if os.environ.get("BENCH_REHEARSE") == "1":
        have = max(have * n, n) if have else 0  # rehearsal of the N-rank path on ONE GPU (gloo; see main): every rank uses device 0
    if have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        raise SystemExit(2)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = pr.stdout.decode(errors="replace").splitlines()
    js = [ln for ln in lines if ln.startswith("{") and ln.rstrip().endswith("}")]
    for ln in lines:
        if not js or ln is not js[-1]:
            print(ln, file=sys.stderr)
    if js:
        print(js[-1])
    sys.stdout.flush()
    raise SystemExit(pr.returncode if pr.returncode else (0 if js else 1))


def _physical_cores() -> int:
    """distinct (physical id, core id) pairs of /proc/cpuinfo, capped by what this process may run on"""
    pairs, phys = set(), None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                pairs.add((phys, ln.split(":")[1].strip()))
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(len(pairs) or avail, avail))


def _cpu_quota() -> float | None:
    """CPU bandwidth limit of this cgroup in cores (None: unlimited / unknown) -- threads beyond it only queue"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-nms", action="store_true")
    ap.add_argument("--layers", action="store_true", help="print the per-layer kernel table to stderr")
    ap.add_argument("--no-resnet", action="store_true", help="skip BASELINE configs[4] (ResNet50 variant: batch 64 inference + NMS, and its training step)")
    ap.add_argument("--sustain-s", type=float, default=2.0, help="length of the sustained forward measurement in seconds (>= 200 steps)")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        _launch_ranks(a.gpus)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # BENCH_REHEARSE=1: the whole N-rank code path (launcher, sharded inputs, gradient arena + bucketed all-reduce, max-over-ranks timing,
    # rank-0 JSON) on a box with ONE GPU -- every rank on device 0, gloo instead of RCCL (which refuses to use a device twice).  The numbers
    # mean nothing (the ranks share the GPU); the run shows that the path works.  Marked in the JSON line.
    rehearse = os.environ.get("BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ     # launched by torch.distributed.run
    if use_dist and rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    elif use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on STDOUT when its communicator is created; stdout must carry only the
        # JSON line, so fd 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)                # backend "nccl" = RCCL on ROCm
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import synth
    from yolo import YOLOLoss, YOLOv1, engine, ops
    from yolo.parallel import make_grad_reducer

    B = a.batch
    torch.manual_seed(0)
    model = YOLOv1().to(dev)
    if use_dist:
        for p_ in model.parameters():
            dist.broadcast(p_.data, 0)
    # synthetic inputs, resident in HBM before any timed region
    rng = np.random.Generator(np.random.PCG64([0, 5 + rank]))
    x = torch.from_numpy(rng.standard_normal((B, 3, 448, 448), dtype=np.float32)).to(dev)
    tgt = torch.from_numpy(synth.synth_targets(B, 1 + rank)).to(dev)

    # ---------------------------------------------------------------- forward (the headline value)
    model.eval()

    def fwd():
        with torch.no_grad():
            return model(x)

    # first forward of the process: weight packing, workspace allocation, library load -- no tuning (the launch plans ship
    # with the package); reported, not part of any timed region
    _sync_all(world)
    t0 = time.perf_counter()
    fwd()
    torch.cuda.synchronize()
    first_forward_ms = 1e3 * (time.perf_counter() - t0)

    dt_f = timed_steps(fwd, a.steps, a.warmup, world)
    fwd_ips = world * B * a.steps / dt_f
    # the driver's K steps last ~0.06 s; the same measurement sustained over >= 200 steps and >= --sustain-s seconds
    n_sus = max(200, int(a.sustain_s / max(dt_f / a.steps, 1e-6)))
    dt_s = timed_steps(fwd, n_sus, 0, world)
    sustained = {"value": round(world * B * n_sus / dt_s, 1), "unit": "images/s", "steps": n_sus, "seconds": round(dt_s, 3), "ms_per_step": round(1e3 * dt_s / n_sus, 3)}

    # ---------------------------------------------------------------- roofline of the dominant kernel
    roof = None
    layer_rows = []
    if rank == 0:
        engine.TIMERS = []
        reps = 3
        launches0 = engine.IGEMM_LAUNCHES
        # marker launches (the single-tensor sumsq kernel is used nowhere else in this script): tools/trace_roofline.py finds
        # this pass between them in a rocprofv3 kernel trace
        from yolo._hip import lib as _lib, ptr as _ptr, stream as _stream
        mark_x, mark_acc = torch.zeros(4, device=dev), torch.zeros((), dtype=torch.float64, device=dev)
        _lib().yolo_sumsq_f32(_ptr(mark_x), 4, _ptr(mark_acc), _stream())
        for _ in range(reps):
            fwd()
        _lib().yolo_sumsq_f32(_ptr(mark_x), 4, _ptr(mark_acc), _stream())
        torch.cuda.synchronize()
        launches = (engine.IGEMM_LAUNCHES - launches0) // reps
        agg = {}
        for tag, kern, flops, e0, e1 in engine.TIMERS:
            d = agg.setdefault(tag, [kern, flops, 0.0])
            d[2] += e0.elapsed_time(e1) / reps
        engine.TIMERS = None
        ig_ms = sum(v[2] for v in agg.values() if v[0] == "igemm")
        ig_fl = sum(v[1] for v in agg.values() if v[0] == "igemm")
        # kernel launches, not layers: the tuner may run a layer as two pixel-range launches or as a split-K launch (whose
        # bracket also holds the 12.8 MB scratch fill and the finishing pass of those three 7x7 layers, ~3 % of ig_ms)
        n_launch = launches
        layer_rows = [(k, v[0], v[1], v[2]) for k, v in agg.items()]
        ach = ig_fl / (ig_ms * 1e-3) / 1e12
        # HBM bytes per launch from PMC counters: measured by tools/collect_traffic.sh (two separate
        # rocprofv3 --pmc passes of this very command) and committed under profiles/ -- a bench run
        # cannot profile itself
        traffic, tname = _profile_traffic("igemm_traffic.json")
        roof = {"kernel": "igemm_persist_kernel / igemm_pipe_kernel / igemm_kernel / igemm_stream_kernel (implicit-GEMM conv/FC, v_mfma_f32_16x16x32_bf16)", "bound": "mfma",
                "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE + WRITE_SIZE with the guide's gfx950 corrections, profiles/" + tname + ")",
                "launches_per_step": n_launch, "avg_launch_ms": round(ig_ms / n_launch, 4),
                "flops_per_launch": ig_fl / n_launch, "flops_per_step": ig_fl, "kernel_ms_per_step": round(ig_ms, 3)}
        if a.layers:
            for key, plan_ in engine._TUNED.items():
                print(f"tuned {key} -> {plan_}", file=sys.stderr)
            for k, kern, fl, ms in layer_rows:
                print(f"{k:14s} {kern:14s} {ms:8.3f} ms {fl / max(ms, 1e-9) / 1e9:9.1f} TFLOP/s", file=sys.stderr)

    # ---------------------------------------------------------------- train step
    train = None
    if not a.no_train:
        model.train()
        crit = YOLOLoss()
        from yolo.optim import Adam
        opt = Adam(model.parameters(), lr=1e-4, weight_decay=5e-4, max_grad_norm=10.0)  # clip_grad_norm_(10) + Adam fused
        # Adam also refreshes the bf16 operands of the Linear layers in the same pass; their update (822 MB of FC1 state) runs as a
        # background pass on 64 CUs beside the next forward's conv stack.  The persistent conv kernels draw their tiles from a queue
        # (igemm_persist.hip), so the CUs the background pass holds cost them a share of the chip, not a whole round of tiles:
        # 11.40 ms per step with the overlap, 11.58 without, same box (with statically assigned tiles it was 11.62 / 11.55).
        opt.attach_plan(model.hip_plan(), overlap=os.environ.get("BENCH_ADAM_OVERLAP", "1") == "1")
        # data parallel: gradient arena + all-reduce overlapped with the backward pass (FC1's 822 MB first)
        ar = make_grad_reducer(model, dev) if (use_dist and world > 1) else None      # the reducer the shipped training loop uses (nothing to reduce in a world of one)

        ar_events = []      # (before, after) events around the reducer's finish(): what the all-reduce adds to the main stream's timeline

        def step():
            opt.zero_grad(set_to_none=True)
            loss, _ = crit(model(x), tgt)
            loss.backward()
            if ar is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ar.all_reduce_mean()
                e1.record()
                ar_events.append((e0, e1))
            opt.step()

        ksteps = max(3, a.steps // 2)
        kwarm = max(2, a.warmup // 2)
        t_loc = [0.0]

        def timed_local():
            """as timed_steps, and keeps this rank's own time too"""
            for _ in range(kwarm):
                step()
            _sync_all(world)
            ar_events.clear()
            t0 = time.perf_counter()
            for _ in range(ksteps):
                step()
            torch.cuda.synchronize()
            t_loc[0] = time.perf_counter() - t0
            _sync_all(world)
            dt = time.perf_counter() - t0
            if dist.is_initialized():
                t = torch.tensor([dt], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return dt

        dt_t = timed_local()
        dp = None
        if ar is not None:
            # exposed = time the main stream spends between "backward is queued" and "every bucket has been averaged", per step
            exposed = sum(e0.elapsed_time(e1) for e0, e1 in ar_events) / max(1, len(ar_events))
            mine = torch.tensor([1e3 * t_loc[0] / ksteps, exposed], dtype=torch.float64, device="cuda")
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            dp = {"all_reduce_ms": round(max(float(t[1]) for t in allr), 3), "all_reduce_ms_note": "exposed on the main stream (not overlapped with the backward pass), max over ranks",
                  "rank_ms_per_step": [round(float(t[0]), 3) for t in allr], "max_rank_ms_per_step": round(max(float(t[0]) for t in allr), 3),
                  "payload_bytes": int(ar.arena.numel() * 4) if hasattr(ar, "arena") else None, "reducer": type(ar).__name__}
        if a.layers and rank == 0 and world == 1:
            engine.TIMERS = []
            step()
            torch.cuda.synchronize()
            for tag, kern, flops, e0, e1 in engine.TIMERS:
                ms = e0.elapsed_time(e1)
                print(f"train {tag:16s} {kern:14s} {ms:8.3f} ms {flops / max(ms, 1e-9) / 1e9:9.1f} TFLOP/s", file=sys.stderr)
            engine.TIMERS = None
        def step_alone():                   # for the per-launch rooflines: every step behind a full synchronisation
            opt.synchronize()
            torch.cuda.synchronize()
            step()

        troof = kernel_rooflines(step_alone)      # on every rank: the step holds a collective when N > 1
        sroof = kernel_rooflines(step, alone=False)
        wtraffic, wname = _profile_traffic("wgrad_traffic.json")
        if troof.get("wgrad") is not None:
            troof["wgrad"]["traffic"] = wtraffic
            troof["wgrad"]["traffic_unit"] = "HBM bytes per launch (PMC FETCH_SIZE + WRITE_SIZE with the guide's gfx950 corrections, profiles/" + wname + ")"
        train = {"value": round(world * B * ksteps / dt_t, 1), "unit": "images/s", "ms_per_step": round(1e3 * dt_t / ksteps, 3),
                 "steps": ksteps, "global_batch": world * B,
                 "step_tflops": round(120.8e9 * world * B / (dt_t / ksteps) / 1e12, 1),
                 "roofline": troof.get("wgrad"), "roofline_igemm": troof.get("igemm"),
                 "roofline_as_scheduled": {"wgrad": sroof.get("wgrad"), "igemm": sroof.get("igemm")},
                 "includes": "zero_grad, forward, YOLOLoss fwd+bwd (HIP), backward (HIP), " + ("RCCL grad all-reduce overlapped with backward, " if ar is not None else "") + "clip_grad_norm_(10), Adam(lr 1e-4, wd 5e-4) fused in one multi-tensor pass"
                             + (" (the Linear layers' share as a background pass beside the next step's forward; the last step's pass ends inside the timed region)"
                                if os.environ.get("BENCH_ADAM_OVERLAP", "1") == "1" else ""),
                 "flops_per_image": 120.8e9, "first_layer_dgrad": "skipped (input needs no gradient)"}
        if dp is not None:
            train.update(dp)
        model.eval()

    # ---------------------------------------------------------------- decode + NMS
    nms = None
    if not a.no_nms and rank == 0:
        p01 = torch.from_numpy(np.random.Generator(np.random.PCG64([0, 99])).uniform(0, 1, size=(64, 7, 7, 30)).astype(np.float32)).to(dev)

        def post():
            rec, cnt = ops.decode(p01, 0.3, 7, 2, 20)
            return ops.nms(rec, cnt, 0.4, 1)

        for _ in range(5):
            post()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            post()
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / 200
        nms = {"value": round(64 * 98 / dtp, 1), "unit": "raw boxes/s", "us_per_batch64": round(dtp * 1e6, 1), "config": "64x(7,7,30)~U(0,1), conf 0.3, nms 0.4, metrics variant"}

    # ---------------------------------------------------------------- preprocessing on the device
    pre = None
    if not a.no_nms and rank == 0:
        from yolo.preprocess import preprocess_u8
        u8 = torch.from_numpy(np.random.Generator(np.random.PCG64([0, 5])).integers(0, 256, size=(64, 375, 500, 3), dtype=np.uint8)).to(dev)

        def timeit(fn, reps):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps

        dt_p = timeit(lambda: preprocess_u8(u8, (448, 448), nchw=False, nhwc4_halo=3), 50)
        dt_e = timeit(lambda: model.forward_uint8(u8), 10)
        pre = {"value": round(64 / dt_p, 1), "unit": "images/s", "us_per_batch64": round(dt_p * 1e6, 1),
               "config": "64 decoded uint8 RGB images 375x500 resident in HBM -> Resize(448,448) (Pillow BILINEAR, bit-exact) + ToTensor + Normalize -> NHWC4 bf16",
               "forward_from_uint8": {"value": round(64 / dt_e, 1), "unit": "images/s", "ms_per_batch64": round(dt_e * 1e3, 3)}}

    # ---------------------------------------------------------------- small batches: predict.py's single image (configs[0] runs it on the CPU), evaluate.py's 16
    small = None
    if rank == 0 and world == 1 and not a.no_nms:
        small = {"unit": "ms per forward (YOLOv1Backbone + FC head, inputs resident, back-to-back forwards)",
                 "note": "yolo/plans/gfx950.json holds measured plans for batches 1, 2, 4, 8, 16, 32 and 64; few-pixel deep-K layers split their K range over the chip (DESIGN.md section 7)"}
        model.eval()
        for nb in (1, 16):
            xb = x[:nb].contiguous()
            with torch.no_grad():
                for _ in range(10):
                    model(xb)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    model(xb)
                torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 100
            small[f"batch{nb}"] = {"ms": round(ms, 4), "images_per_s": round(nb / ms * 1e3, 1)}

    # ---------------------------------------------------------------- ResNet50 variant (configs[4])
    resnet = None
    if not a.no_resnet and rank == 0 and world == 1:
        from yolo import ResNetBackbone
        del model
        torch.cuda.empty_cache()
        rm = YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=True)).to(dev).eval()

        def rfwd():
            with torch.no_grad():
                pr = rm(x)
                rec, cnt = ops.decode(pr, 0.3, 7, 2, 20)
                return ops.nms(rec, cnt, 0.4, 0)

        rsteps = max(10, a.steps)
        dtr = timed_steps(rfwd, rsteps, 3, world) / rsteps
        rroof = kernel_rooflines(rfwd)
        resnet = {"value": round(B / dtr, 1), "unit": "images/s", "ms_per_batch": round(dtr * 1e3, 3), "steps": rsteps,
                  "config": "configs[4]: YOLOv1(ResNetBackbone) batch 64 inference (BN folded) + decode + NMS conf 0.3 / nms 0.4, random init",
                  "roofline": rroof.get("igemm")}
        # the reference's default TRAINING model (src/train.py:144): the same network with the trunk trainable, BatchNorm on batch statistics
        del rm
        torch.cuda.empty_cache()
        from yolo.optim import Adam as HipAdam
        tm = YOLOv1(backbone=ResNetBackbone(pretrained=False, freeze=False)).to(dev).train()
        topt = HipAdam(tm.parameters(), lr=1e-4, weight_decay=5e-4, max_grad_norm=10.0)
        topt.attach_plan(tm.head.hip_plan())
        tcrit = YOLOLoss()

        def rstep():
            topt.zero_grad(set_to_none=True)
            ls, _ = tcrit(tm(x), tgt)
            ls.backward()
            topt.step()

        tsteps = max(5, a.steps // 4)
        dtt = timed_steps(rstep, tsteps, 2, world) / tsteps
        trroof = kernel_rooflines(rstep, reps=1)
        resnet["train"] = {"value": round(B / dtt, 1), "unit": "images/s", "ms_per_step": round(dtt * 1e3, 3), "steps": tsteps,
                           "config": "YOLOv1(ResNetBackbone(freeze=False)) batch 64: forward (batch-statistics BatchNorm) + loss + backward + clip + Adam, random init",
                           "roofline": trroof.get("wgrad"), "roofline_igemm": trroof.get("igemm")}
        del tm, topt
        torch.cuda.empty_cache()

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        from oracle import oracle as O
        from oracle.torch_ref import RefYOLOv1
        ref = RefYOLOv1().eval()
        phys, quota = _physical_cores(), _cpu_quota()
        # the host's best, not its default: torch starts with one thread per logical CPU (128 on this box), which oversubscribes a
        # container whose CPU share is smaller; sweep thread counts x batch sizes on a bounded budget and report the fastest
        cands = sorted({t for t in (phys, 64, 32, 16, int(quota) if quota else 0) if 0 < t <= max(phys, 1) * 2}, reverse=True)
        x64 = x.cpu()
        sweep = []
        t_budget = time.perf_counter() + 16.0

        def run_cfg(threads, batch, min_s):
            torch.set_num_threads(threads)
            xb = x64[:batch]
            with torch.no_grad():
                ref(xb[:1])                                   # thread pool + primitive caches of this setting
                t0 = time.perf_counter()
                n = 0
                while True:
                    ref(xb)
                    n += batch
                    if time.perf_counter() - t0 >= min_s:
                        break
                return n, time.perf_counter() - t0

        for th in cands:
            if time.perf_counter() > t_budget:
                break
            n, dtc = run_cfg(th, 8, 0.8)
            sweep.append({"threads": th, "batch": 8, "images_per_s": round(n / dtc, 2)})
        for th in [r["threads"] for r in sorted(sweep, key=lambda r: -r["images_per_s"])[:2]]:
            if time.perf_counter() > t_budget:
                break
            n, dtc = run_cfg(th, 64, 0.0)
            sweep.append({"threads": th, "batch": 64, "images_per_s": round(n / dtc, 2)})
        best = max(sweep, key=lambda r: r["images_per_s"])
        n_img, dtc = run_cfg(best["threads"], best["batch"], 6.0)       # the reported sample at the best setting
        torch.set_num_threads(best["threads"])
        with torch.no_grad():
            ref(x64[:1])
            t0 = time.perf_counter()
            for _ in range(5):
                ref(x64[:1])
            lat1 = (time.perf_counter() - t0) / 5
        # forward + YOLOLoss + backward on the host (stock autograd; the loss's CPU formulation is the reference's arithmetic)
        from yolo import YOLOLoss as _HostLoss
        ref.train()
        tg8 = tgt[:8].cpu()
        crit_c = _HostLoss()
        t0 = time.perf_counter()
        n_tr = 0
        while n_tr < 16 and time.perf_counter() - t0 < 6.0:
            ref.zero_grad(set_to_none=True)
            ls, _ = crit_c(ref(x64[:8]), tg8)
            ls.backward()
            n_tr += 8
        dt_tr = time.perf_counter() - t0
        ref.eval()
        cpu = {"value": round(n_img / dtc, 2), "unit": "images/s", "cores": best["threads"], "kind": "port",
               "sample": f"forward of oracle/torch_ref.RefYOLOv1 (stock torch.nn fp32, the reference's layer table) on batches of {best['batch']} with "
                         f"{best['threads']} threads (fastest of the sweep), {n_img} images in {dtc:.1f} s",
               "cpu_model": _cpu_model(), "nproc": os.cpu_count(), "physical_cores": phys, "cgroup_cpu_quota": quota, "sweep": sweep,
               "latency_n1_s": round(lat1, 4),
               "train": {"value": round(n_tr / dt_tr, 2), "unit": "images/s", "sample": f"forward + YOLOLoss + backward (stock autograd), batches of 8, {n_tr} images in {dt_tr:.1f} s, {best['threads']} threads"}}
        if pre is not None:
            from PIL import Image
            from oracle import preprocess_ref as PR
            hu8 = u8[:8].cpu().numpy()
            t0 = time.perf_counter()
            n_pp = 0
            while time.perf_counter() - t0 < 3.0:
                for i in range(8):
                    PR.to_tensor_normalize(np.asarray(Image.fromarray(hu8[i]).resize((448, 448), Image.BILINEAR)))
                n_pp += 8
            pre["cpu_baseline"] = {"value": round(n_pp / (time.perf_counter() - t0), 1), "unit": "images/s", "cores": 1, "kind": "port",
                                   "sample": f"PIL.Image.resize(BILINEAR) + NumPy ToTensor/Normalize (the reference's host transform), {n_pp} images"}
        if nms is not None:
            pn = p01.cpu().numpy()
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < 3.0:
                for n in range(64):
                    O.nms(O.decode(pn[n], 0.3), 0.4, 1)
                reps += 1
            nms["cpu_baseline"] = {"value": round(reps * 64 * 98 / (time.perf_counter() - t0), 1), "unit": "raw boxes/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle/yolo_oracle.c decode+nms, {reps} x 64 images"}
            # the reference's own way: one .item() per scalar, Python lists (oracle/post_py.py restates metrics.py:185-341)
            from oracle import post_py as PY
            pt = p01.cpu()
            t0 = time.perf_counter()
            n_py = 0
            while n_py < 64 and time.perf_counter() - t0 < 3.0:
                PY.nms_py(PY.decode_py(pt[n_py % 64], 0.3), 0.4)
                n_py += 1
            nms["cpu_baseline_python"] = {"value": round(n_py * 98 / (time.perf_counter() - t0), 1), "unit": "raw boxes/s", "cores": 1, "kind": "port",
                                          "sample": f"oracle/post_py.py (per-scalar .item() + Python lists, as the reference's mAPMetric runs), {n_py} images"}

    if rank == 0:
        out = {
            "metric": "images/sec at 448x448, forward (YOLOv1Backbone + FC head)", "value": round(fwd_ips, 1), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt_f / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: batch=64/GPU 448x448 forward-only, YOLOv1Backbone + FC head, 1xMI355X per rank",
                       "global_batch": world * B, "per_gpu_batch": B, "weights": "random init (torch default, seed 0)",
                       "flops_per_image": 40.57e9, "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": roof, "cpu_baseline": cpu, "sustained": sustained, "first_forward_ms": round(first_forward_ms, 1),
            **({"rehearsal": "BENCH_REHEARSE=1: all ranks on ONE GPU over gloo -- a test of the N-rank code path, not a measurement"} if rehearse else {}),
            "train": train, "nms": nms, "preprocess": pre, "small_batches": small, "resnet50_variant": resnet,
        }
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
