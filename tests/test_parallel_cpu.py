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
