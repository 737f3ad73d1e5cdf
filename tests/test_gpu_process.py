"""Process-level properties on the GPU: (1) two FRESH processes produce bit-identical batch-64 outputs -- the launch plans
are shipped data, nothing is timed at run time; (2) the data-parallel training path users run (trainer -> make_grad_reducer
-> gradient arena + OverlappedGradAllReduce) under torch.distributed.run with RCCL at world size 1 (a dev box has one GPU;
BASELINE configs[3] needs eight) equals the single-process step.  Children are ordinary subprocesses."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "dist_child.py")

pytestmark = pytest.mark.gpu


def _run(args, timeout=600, **extra_env):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env)
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{args}\n--- stdout\n{r.stdout[-2000:]}\n--- stderr\n{r.stderr[-4000:]}"
    return r


def test_two_fresh_processes_give_bit_identical_batch64_outputs(tmp_path):
    """the per-process wall-clock tuner could pick different plans (hence different bf16 roundings) process to
    process and rank to rank.  Plans now come from yolo/plans/gfx950.json: same launches everywhere."""
    outs = []
    for k in range(2):
        f = tmp_path / f"h{k}.txt"
        _run([CHILD, "hash", str(f), "64"])
        outs.append(f.read_text().split())
    assert outs[0][0] == outs[1][0], outs
    assert int(outs[0][1]) > 20, "the shipped plan table was not loaded"


def test_rccl_world1_overlapped_path_equals_single_process_step(tmp_path):
    plain, rccl = tmp_path / "plain.pt", tmp_path / "rccl.pt"
    _run([CHILD, "plain", str(plain)])
    port = 29600 + os.getpid() % 300
    _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
          CHILD, "rccl", str(rccl)])
    a = torch.load(plain, weights_only=True)
    b = torch.load(rccl, weights_only=True)
    assert b["reducer"] == "OverlappedGradAllReduce" and a["reducer"] is None
    assert abs(a["loss"] - b["loss"]) <= 1e-6 * abs(a["loss"])
    # gradients: equal up to the order of fp32 atomics in the weight-gradient kernels
    for n, g in a["grads"].items():
        torch.testing.assert_close(b["grads"][n], g, rtol=2e-4, atol=1e-6 * float(g.abs().max()) + 1e-9, msg=n)
    for n, v in a["norms"].items():
        assert abs(b["norms"][n] - v) <= 1e-4 * v + 1e-12, n
    for n, p in a["params"].items():
        torch.testing.assert_close(b["params"][n], p, rtol=1e-5, atol=1e-7, msg=n)


@pytest.mark.parametrize("small_split", ["0", "1"])
def test_two_ranks_on_one_gpu_equal_the_single_process_step_on_the_whole_batch(tmp_path, small_split):
    """at world size 1 a mean over ranks is the identity.  Two ranks (gloo; both on this box's one GPU) each take half of
    the batch through the shipped path -- gradient arena, bucketed all-reduce overlapped with the two-stream backward pass, fused
    clip + Adam: the ranks end bit-identical, and equal to the single-process step on all four images up to what a different batch
    size changes (other launch plans -> other fp32 summation orders -> a few bf16 roundings)."""
    plain, two = tmp_path / "plain.pt", tmp_path / "two.pt"
    env = {"YOLO_AMD_SMALL_SPLIT": small_split, "YOLO_AMD_PLAN_TABLE": small_split}
    _run([CHILD, "plain", str(plain)], **env)
    port = 29900 + os.getpid() % 300 + 300 * int(small_split)
    _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
          CHILD, "gloo2", str(two)], timeout=900, **env)
    a = torch.load(plain, weights_only=True)
    r0 = torch.load(str(two) + ".r0", weights_only=True)
    r1 = torch.load(str(two) + ".r1", weights_only=True)
    assert r0["reducer"] == r1["reducer"] == "OverlappedGradAllReduce"
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), n            # replicas stay identical (same averaged gradient, same clip norm)
    for n in r0["grads"]:
        assert torch.equal(r0["grads"][n], r1["grads"][n]), n
    # mean of the two shard losses = loss of the whole batch (YOLOLoss divides by the local N, shards are equal)
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - a["loss"]) <= 2e-3 * abs(a["loss"])
    # small_split = "0" (config.SMALL_SPLIT and config.PLAN_TABLE off: every layer is ONE plain launch of the deterministic default plan at 2 and at 4
    # images -> the same fp32 summation orders): the plumbing is compared at the tolerance of a few bf16 roundings.  "1" (the defaults: the shipped
    # table has measured plans for batches 2 and 4, and problems without an entry split by rule): the deep layers split their K range by the number
    # of pixel tiles, so their fp32 sums are taken in another order at 2 and at 4 images -- 0.1-0.3 % of a gradient's norm at the head end of
    # the backward pass, growing to 1-2.6 % (cosine 0.975) at the first layers of this random-init LeakyReLU network, whose gates flip under a
    # rounding; the split itself is pinned against fp64 in test_gpu_layers.py::test_k_range_slabs_of_one_layer_equal_the_fp64_product
    tol, cos_min = (2e-2, 0.995) if small_split == "0" else (5e-2, 0.95)
    print("norm deviations:", {n: round(abs(r0["norms"][n] - v) / (v + 1e-30), 4) for n, v in a["norms"].items()})
    for n, v in a["norms"].items():
        assert abs(r0["norms"][n] - v) <= tol * v + 1e-9, (n, r0["norms"][n], v)
    for n, g in a["grads"].items():
        gg = r0["grads"][n].double().flatten()
        cos = float((gg * g.double().flatten()).sum() / (gg.norm() * g.double().norm() + 1e-30))
        assert cos > cos_min, (n, cos)
        if n.startswith("head.4."):
            assert cos > 0.9995, (n, cos)          # behind no LeakyReLU gate: rounding level either way
