"""CPU, world_size 2 over gloo: the data-parallel gradient path (yolo.parallel) reproduces the
single-process gradients of the global batch ('N-way split == full batch')."""

import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_model():
    import torch.nn as nn
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 8, 3, 1, 1), nn.LeakyReLU(0.1), nn.Flatten(), nn.Linear(8 * 14 * 14, 7 * 7 * 30))


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "yolo-v1_amd"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from yolo import YOLOLoss
    from yolo.parallel import GradAllReduce, broadcast_parameters, shard_batch
    torch.manual_seed(100 + rank)          # different init per rank: broadcast must fix it
    model = _tiny_model()
    if rank != 0:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    broadcast_parameters(model)
    N = 8
    x = torch.from_numpy(synth.synth_normal((N, 3, 14, 14), 7))
    t = torch.from_numpy(synth.synth_targets(N, 3))
    sl = shard_batch(N, rank, world)
    loss, _ = YOLOLoss()(model(x[sl]).view(-1, 7, 7, 30), t[sl])
    loss.backward()
    GradAllReduce(model.parameters(), big_bytes=1 << 12).all_reduce_mean()   # exercises both the big and the packed path
    # the reducer the shipped training loop picks (training.train_epoch): plain GradAllReduce for CPU tensors / custom modules;
    # averaging already-identical gradients must leave them unchanged
    from yolo.parallel import make_grad_reducer
    red = make_grad_reducer(model, "cpu")
    assert type(red).__name__ == "GradAllReduce"
    before = [p.grad.clone() for p in model.parameters()]
    red.all_reduce_mean()
    for a, p in zip(before, model.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-6, atol=1e-8)
    q.put((rank, [p.grad.numpy().copy() for p in model.parameters()], [p.detach().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_full_batch():
    sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synth
    from yolo import YOLOLoss
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference on the global batch with rank 0's parameters
    model = _tiny_model()
    x = torch.from_numpy(synth.synth_normal((8, 3, 14, 14), 7))
    t = torch.from_numpy(synth.synth_targets(8, 3))
    loss, _ = YOLOLoss()(model(x).view(-1, 7, 7, 30), t)
    loss.backward()
    for r in res:
        for g, p_now, p_ref in zip(r[1], r[2], model.parameters()):
            assert torch.equal(torch.from_numpy(p_now), p_ref.detach())                       # broadcast worked
            torch.testing.assert_close(torch.from_numpy(g), p_ref.grad, rtol=1e-5, atol=1e-6)  # averaged shards == global batch
    for a, b in zip(res[0][1], res[1][1]):
        assert (a == b).all()                                               # ranks agree bit for bit


def test_shard_batch():
    from yolo.parallel import shard_batch
    assert [shard_batch(512, r, 8) for r in (0, 7)] == [slice(0, 64), slice(448, 512)]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


class _FakePlan:
    """stands in for engine.Plan on the CPU: an arena plus the two callbacks"""

    def __init__(self, n):
        self.arena = torch.zeros(n)
        self.on_grad_ready = None
        self.on_backward_done = None
        self.on_stream_wait = None
        self.grad_norm_sq = {}


def _overlap_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "yolo-v1_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yolo.parallel import OverlappedGradAllReduce
    plan = _FakePlan(1000)
    ar = OverlappedGradAllReduce(plan, "cpu", bucket_bytes=4 * 300)
    out = []
    for step in range(2):
        plan.arena.copy_(torch.arange(1000, dtype=torch.float32) * (rank + 1 + step))
        for lo, hi in ((0, 100), (100, 450), (450, 700), (700, 900)):       # layers finishing in arena order
            plan.on_grad_ready(lo, hi)
        plan.on_backward_done()                                              # tail 900..1000 = bias region
        ar.finish()
        out.append(plan.arena.numpy().copy())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucket_allreduce_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    base = torch.arange(1000, dtype=torch.float32).numpy()
    for step in range(2):
        want = base * ((1 + step) + (2 + step)) / 2          # mean over the two ranks
        for r in res:
            assert (abs(r[1][step] - want) < 1e-3).all()


def _train_worker(rank, world, port, ckdir, q):
    for p in (ROOT, os.path.join(ROOT, "yolo-v1_amd"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pathlib import Path
    import synth
    from torch.utils.data import DataLoader, TensorDataset
    from yolo import YOLOLoss, training
    from yolo.parallel import broadcast_parameters, shard_batch

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = _tiny_model()

        def forward(self, x):
            return self.body(x).view(-1, 7, 7, 30)

    torch.manual_seed(50 + rank)
    model = Net()
    broadcast_parameters(model)
    x = torch.from_numpy(synth.synth_normal((8, 3, 14, 14), 9))
    t = torch.from_numpy(synth.synth_targets(8, 4))
    sl = shard_batch(8, rank, world)
    loader = DataLoader(TensorDataset(x[sl], t[sl]), batch_size=2)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10], gamma=0.1)
    res = training.train(model, loader, loader, YOLOLoss(), opt, sched, "cpu", 2, Path(ckdir), save_frequency=1)
    q.put((rank, float(res["final_train_loss"]), [p.detach().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_two_ranks_write_checkpoints_once(tmp_path):
    """the shipped training loop under torch.distributed (gloo, world 2): gradients are averaged by the reducer the loop picks,
    ranks stay bit-identical, and ONLY rank 0 writes the checkpoint files -- atomically, no temp files left, loadable with
    weights_only=True (round 1: every rank wrote the same paths concurrently)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for a, b in zip(res[0][2], res[1][2]):
        assert (a == b).all()
    names = sorted(os.listdir(tmp_path))
    assert names == ["yolo_best.pth", "yolo_epoch_1.pth", "yolo_epoch_2.pth", "yolo_latest.pth"], names
    ck = torch.load(tmp_path / "yolo_latest.pth", weights_only=True)
    assert ck["epoch"] == 2 and "optimizer_state_dict" in ck


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 3: the clip norm under data parallelism, and the stream a gradient bucket is all-reduced from
# ---------------------------------------------------------------------------------------------------------------------------------
class _ArenaPlan:
    """engine.Plan as the reducer and the optimizer see it, on the CPU: a gradient arena holding every gradient of a tiny model in
    the order backward produces them (last layer first), the callbacks, and the squared-norm hint engine.Plan.backward leaves for the
    gradient of the big Linear layer (computed by the kernel that stored it, i.e. from the LOCAL gradient)."""

    def __init__(self, model):
        self.params = list(model.parameters())
        order = list(reversed(self.params))
        self.arena = torch.zeros(sum(p.numel() for p in order))
        self.views, off = {}, 0
        for p in order:
            self.views[id(p)] = (off, off + p.numel())
            off += p.numel()
        self.on_grad_ready = self.on_backward_done = self.on_stream_wait = None
        self.grad_norm_sq = {}

    def backward_into_arena(self, loss, hint_for):
        """what Plan.backward does with an arena: gradients are WRITTEN into the arena, .grad = views, hint for one weight, callbacks"""
        grads = torch.autograd.grad(loss, self.params)
        for p, g in zip(self.params, grads):
            lo, hi = self.views[id(p)]
            self.arena[lo:hi].copy_(g.reshape(-1))
            p.grad = self.arena[lo:hi].view_as(p)
        nsq = hint_for.grad.double().pow(2).sum()                # = yolo_wgrad_desc.dw_sumsq
        self.grad_norm_sq[id(hint_for)] = ((hint_for.grad.data_ptr(), tuple(hint_for.grad.shape)), hint_for.grad._version, nsq)
        for p in reversed(self.params):
            self.on_grad_ready(*self.views[id(p)])
        self.on_backward_done()


class _RawAvgReducer:
    """mixed into OverlappedGradAllReduce below: the RCCL branch (ReduceOp.AVG in place, no mul_ afterwards) on gloo, which has no AVG --
    sum, then divide through a NumPy view: like the collective itself, that does not touch the tensor's version counter"""

    def _reduce(self, lo, hi):
        if hi > lo:
            self._check_ordered(lo, hi, self._stream_id())
            dist.all_reduce(self.arena[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            self.arena.numpy()[lo:hi] /= dist.get_world_size(self.group)


def _clip_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "yolo-v1_amd"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from yolo import YOLOLoss
    from yolo import optim as yopt
    from yolo.parallel import OverlappedGradAllReduce, broadcast_parameters, shard_batch

    class AvgReducer(_RawAvgReducer, OverlappedGradAllReduce):
        pass

    model = _tiny_model()
    broadcast_parameters(model)
    plan = _ArenaPlan(model)
    red = AvgReducer(plan, "cpu", bucket_bytes=4 * 2000)
    red._avg = True                                            # the RCCL branch of finish(): no mul_, hence no version bump
    # max_grad_norm far below |g|: the clip is active; eps far above the clipped gradient's elements: the update is ~ lr * g / eps, i.e.
    # proportional to the clip coefficient (with the default eps Adam's m / sqrt(v) would cancel a wrong coefficient)
    opt = yopt.Adam(model.parameters(), lr=1e-2, eps=1e-3, weight_decay=5e-4, max_grad_norm=0.05)
    opt.plans.append(plan)
    x = torch.from_numpy(synth.synth_normal((8, 3, 14, 14), 7))
    t = torch.from_numpy(synth.synth_targets(8, 3))
    sl = shard_batch(8, rank, world)
    big = model[3].weight
    stale = None
    for step in range(2):
        loss, _ = YOLOLoss()(model(x[sl]).view(-1, 7, 7, 30), t[sl])
        plan.backward_into_arena(loss, big)
        hint = dict(plan.grad_norm_sq)
        v0 = big.grad._version
        red.all_reduce_mean()
        if step == 0:
            # the hint describes the local gradient; after the averaging it must not be taken any more
            grads, extra = yopt._split_known(model.parameters(), plan.grad_norm_sq)
            stale = (len(extra), len(plan.grad_norm_sq), big.grad._version > v0, len(yopt._split_known(model.parameters(), hint)[1]))
        opt.step()
    q.put((rank, stale, [p.detach().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_clip_norm_after_allreduce_is_the_global_one():
    """the plan's |g|^2 hint for the big Linear gradient is computed before the all-reduce; RCCL's in-place AVG
    does not bump the version counter, so the optimizer would clip with a per-rank norm.  After finish() the hint is gone (and the
    version bumped, so that even a kept copy of the hint no longer matches); two ranks through OverlappedGradAllReduce +
    yolo.optim.Adam(max_grad_norm) end bit-identical and equal to the single-process step on the whole batch
    (clip_grad_norm_ + torch.optim.Adam)."""
    sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synth
    from yolo import YOLOLoss
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_clip_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] == (0, 0, True, 0), r[1]           # no hint taken, table cleared, version bumped, a kept copy does not match either
    for a, b in zip(res[0][2], res[1][2]):
        assert (a == b).all()
    model = _tiny_model()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, eps=1e-3, weight_decay=5e-4)
    x = torch.from_numpy(synth.synth_normal((8, 3, 14, 14), 7))
    t = torch.from_numpy(synth.synth_targets(8, 3))
    for step in range(2):
        opt.zero_grad()
        loss, _ = YOLOLoss()(model(x).view(-1, 7, 7, 30), t)
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.05)
        assert float(norm) > 0.5                        # the clip really scales
        opt.step()
    for got, want in zip(res[0][2], model.parameters()):
        torch.testing.assert_close(torch.from_numpy(got), want.detach(), rtol=2e-4, atol=2e-6)


class _FakeStream:
    def __init__(self, rec, handle):
        self.rec, self.cuda_stream = rec, handle

    def wait_stream(self, other):
        self.rec.events.append(("wait", self.cuda_stream, other.cuda_stream))

    def wait_event(self, ev):
        pass


class _FakeStreams:
    """engine.STREAMS stand-in: two streams, `use()` switches the current one; every launch of the fake library is logged with it"""

    def __init__(self):
        self.events = []
        self.main, self.side_s = _FakeStream(self, 0x1000), _FakeStream(self, 0x2000)
        self.cur = self.main

    def current(self, dev):
        return self.cur

    def side(self, dev, low):
        return self.side_s

    def use(self, s):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self.cur = self.cur, s
            try:
                yield
            finally:
                self.cur = prev
        return ctx()


class _FakeLib:
    """libyolo_hip stand-in: every entry point succeeds and is logged with its stream argument (always the last one)"""

    def __init__(self, rec):
        self.rec = rec

    def __getattr__(self, name):
        def call(*args):
            st = args[-1]
            self.rec.events.append(("launch", name, getattr(st, "value", st)))
            return 0
        return call


def test_plan_backward_buckets_are_reduced_from_a_stream_that_has_their_gradients(monkeypatch):
    """the weight gradients of the conv layers, their unpack passes and the gradient-ready callbacks run on a
    low-priority side stream, the Linear layers' on the main stream.  Drives engine.Plan.backward of the REAL YOLOv1 layer table
    (fake kernels, recording streams, the gloo backend at world size 1) with the shipped OverlappedGradAllReduce and checks for
    every bucket that each range in it was produced on the stream current at the all-reduce call, or on one that stream has
    waited for since (OverlappedGradAllReduce._check_ordered raises otherwise -- shown with a broken schedule at the end)."""
    sys.path.insert(0, os.path.join(ROOT, "yolo-v1_amd"))
    import ctypes
    from yolo import YOLOv1, engine
    from yolo.parallel import OverlappedGradAllReduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(37500 + os.getpid() % 2000)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        rec = _FakeStreams()
        fake = _FakeLib(rec)
        monkeypatch.setattr(engine, "STREAMS", rec)
        monkeypatch.setattr(engine, "lib", lambda: fake)
        monkeypatch.setattr(engine, "stream", lambda: ctypes.c_void_p(rec.cur.cuda_stream))
        monkeypatch.setattr(engine, "_SIDE_STREAMS", {})
        monkeypatch.setattr(engine, "_splitk_scratch", lambda n, zero: torch.zeros(n))
        torch.manual_seed(0)
        model = YOLOv1()
        plan = model.hip_plan()
        red = OverlappedGradAllReduce(plan, "cpu", stream_id=lambda: rec.cur.cuda_stream)
        red.log = []
        x = torch.zeros(1, 3, 448, 448)
        out, saved = plan.forward(x, True, False)
        rec.events.clear()
        plan.backward(saved, torch.zeros_like(out), False)
        # the schedule as shipped: every weight gradient (conv and Linear) on the side stream; FC1's data gradient -- a yolo_wgrad launch too -- on main
        wg = [(e[1], e[2]) for e in rec.events if e[0] == "launch" and e[1].startswith("yolo_wgrad")]
        assert sum(1 for n, s in wg if s == 0x2000) == 26 and sum(1 for n, s in wg if s == 0x1000) == 1
        assert ("wait", 0x1000, 0x2000) in rec.events                                # the join in front of on_backward_done
        assert len(red.log) >= 4
        covered = 0
        for lo, hi, cur, pieces in red.log:
            assert lo == covered and pieces
            covered = hi
            for (a, b, s, t) in pieces:
                assert s == cur or red._waits.get((cur, s), -1) >= t
        assert covered == plan.arena.numel()
        first = red.log[0]
        assert first[2] == 0x2000 and all(p[2] == 0x2000 for p in first[3])          # FC2 + FC1 (822 MB) leave from the stream that produced them
        assert any(cur == 0x2000 for _, _, cur, _ in red.log[1:-1])                  # conv buckets leave from the side stream
        assert red.log[-1][2] == 0x1000                                             # the tail (+ bias region) after the join, from main
        red.finish()
        # a broken schedule: a range produced on the side stream, all-reduced from the main stream without a wait in between
        with engine.STREAMS.use(rec.side_s):
            red._ready(0, 10)
        with pytest.raises(RuntimeError, match="has not waited"):
            red._reduce(0, 10)
        red._stream_wait(0x1000, 0x2000)
        red._reduce(0, 10)
        red.finish()
    finally:
        dist.destroy_process_group()
