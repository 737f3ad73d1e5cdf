"""Child process of tests/test_gpu_process.py (not a test module): one training step of YOLOv1 on 4 images, either
plain (single process) or under torch.distributed.run with RCCL at world size 1 through the SHIPPED data-parallel path
(yolo.parallel.make_grad_reducer -> gradient arena + OverlappedGradAllReduce), or just a forward whose output hash is
printed.  Writes a small .pt with what the parent compares."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "yolo-v1_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import synth  # noqa: E402


def build():
    from yolo import YOLOv1
    m = YOLOv1()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.yolov1_state_dict().items()}, strict=True)
    return m.cuda()


def main():
    mode, out = sys.argv[1], sys.argv[2]
    if mode == "hash":
        N = int(sys.argv[3])
        m = build().eval()
        x = torch.from_numpy(synth.synth_images(N, 17)).cuda()
        with torch.no_grad():
            y = m(x)
            y2 = m(x)
        assert torch.equal(y, y2)
        from yolo import engine
        open(out, "w").write(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest() + f" {len(engine._TUNED)}\n")
        return
    use_dist = mode in ("rccl", "gloo2")
    rank, world = 0, 1
    if mode == "gloo2":
        # two ranks on ONE GPU (a dev box has one; RCCL refuses to use a device twice, gloo stages through the host): the shipped
        # data-parallel path at a world size where averaging is not the identity
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
    elif use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    from yolo import YOLOLoss
    from yolo.optim import Adam
    from yolo.parallel import broadcast_parameters, make_grad_reducer
    m = build().eval()                          # eval: no dropout, so that both runs see the same network
    if use_dist:
        broadcast_parameters(m)
    opt = Adam(m.parameters(), lr=1e-4, weight_decay=5e-4, max_grad_norm=10.0)
    opt.attach_plan(m.hip_plan())
    red = make_grad_reducer(m, "cuda") if use_dist else None
    x = torch.from_numpy(synth.synth_images(4, 23)).cuda()
    t = torch.from_numpy(synth.synth_targets(4, 41, max_obj=3)).cuda()
    if world > 1:
        from yolo.parallel import shard_batch
        sl = shard_batch(4, rank, world)
        x, t = x[sl].contiguous(), t[sl].contiguous()
    opt.zero_grad(set_to_none=True)
    loss, parts = YOLOLoss()(m(x), t)
    loss.backward()
    if red is not None:
        red.all_reduce_mean()
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if p.dim() == 1 or p.numel() < (1 << 20)}
    norms = {n: float(p.grad.double().norm()) for n, p in m.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    params = {n: p.detach().float().cpu().clone() for n, p in m.named_parameters() if p.dim() == 1}
    torch.save({"loss": float(parts["total"]), "grads": grads, "norms": norms, "params": params,
                "reducer": type(red).__name__ if red is not None else None}, out if world == 1 else f"{out}.r{rank}")
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
