"""Worker for the world_size>1 tests (launched by torch.distributed.run).

--mode grads   --device cpu : the data-parallel plumbing (FlatParams + GradAllReduce: module-boundary buckets, readiness
                              bookkeeping, exchange overlapped with "backward") on CPU tensors over gloo -- no kernels.
--mode shards  --device cpu : the sampling partition / gather helpers (shard_range, gather_u8) over gloo.
--mode train   --device cuda: every rank runs the real HIP train step with its shard of the golden batch.
--mode sample  --device cuda: Diffusion.sample_sharded and sample_rotation_sweep_sharded, gathered on rank 0.
With --backend gloo all ranks share cuda:0 (exchanges staged through host memory); with --backend nccl (= RCCL) rank r
uses cuda:r.
"""
import argparse
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FSET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}


def grads_cpu(args, rank, world):
    import afdm
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(13, 7), torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    fp = afdm.FlatParams(net)
    assert all(p.data_ptr() == fp.flat.data_ptr() + 4 * o for p, o in zip(fp.params, fp.offsets))
    ddp = afdm.GradAllReduce(fp, n_buckets=3, model=net)
    # 156 parameters, target 52 per bucket, cut at module boundaries walking backwards: [0, 98) and [98, 156)
    ok = ddp.slices == [(0, 98), (98, 156)]

    def local_grads(r):
        g = torch.Generator().manual_seed(100 + r)
        return [torch.randn(p.shape, generator=g) for p in fp.params]

    want = torch.zeros_like(fp.grad)
    for r in range(world):
        want += torch.cat([l.reshape(-1) for l in local_grads(r)])
    want /= world
    # (1) overlapped: "backward" writes the gradients last module first and reports them as it goes
    fp.zero_grad()
    ddp.begin_step()
    mine = local_grads(rank)
    launched = []
    for idx in reversed(range(len(fp.params))):
        fp.params[idx].grad.copy_(mine[idx])                  # what the in-place weight-gradient kernels do
        ddp.wrote([fp.params[idx]])
        launched.append(len(ddp._launched))
    # the second bucket (modules 1 and 2 = parameters 2..5) starts when parameter 2 is written, the first at the very end
    ok = ok and launched == [0, 0, 0, 1, 1, 2]
    scale = ddp.finish()
    ok = ok and ddp.overlapped_last_step == 2 and scale == 1.0 / world
    ok = ok and torch.allclose(fp.grad * scale, want, rtol=1e-6, atol=1e-7)
    # (2) nothing reported (a replayed hipGraph): finish() exchanges everything
    fp.zero_grad()
    for p, l in zip(fp.params, mine):
        p.grad.copy_(l)
    scale = ddp()
    ok = ok and ddp.overlapped_last_step == 0 and torch.allclose(fp.grad * scale, want, rtol=1e-6, atol=1e-7)
    # (3) plain flat tensor, equal slices (the round-1 form)
    flat = torch.cat([l.reshape(-1) for l in mine]).clone()
    d3 = afdm.GradAllReduce(flat, n_buckets=3)
    ok = ok and len(d3.slices) == 3 and torch.allclose(flat * d3(), want, rtol=1e-6, atol=1e-7)
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        torch.save({"ok": all(flags), "world": world}, args.out)


def shards_cpu(args, rank, world):
    from afdm.diffusion import gather_u8, shard_range
    ok = True
    for n in (1, 2, 5, 9, 36):
        counts = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
        ok = ok and sum(counts) == n and max(counts) - min(counts) <= 1
        lo, hi = shard_range(n, rank, world)
        full = (torch.arange(n * 6, dtype=torch.int64) % 251).to(torch.uint8).reshape(n, 1, 2, 3)
        got = gather_u8(full[lo:hi], counts)
        if rank == 0:
            ok = ok and torch.equal(got, full)
        else:
            ok = ok and got is None
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        torch.save({"ok": all(flags), "world": world}, args.out)


def train_cuda(args, rank, world, dev):
    import afdm
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_step.npz"), allow_pickle=False)
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(FSET), device=dev, variant=3).to(dev)
    diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
    step = afdm.TrainStep(model, diff, lr=3e-4, graph={"0": False, "1": True, "lanes": "lanes"}[os.environ.get("AFD_TEST_GRAPH", "0")], distributed=True)
    B = g["images"].shape[0] // world
    sl = slice(rank * B, (rank + 1) * B)
    T = lambda a: torch.from_numpy(np.asarray(a))
    loss = step(T(g["images"][sl]).to(dev), t=T(g["t0"][sl]), eps=T(g["eps0"][sl]).to(dev))
    overlapped, started = step.ddp.overlapped_last_step, step.ddp.started_before_finish
    lt = loss.detach().cpu().reshape(1)
    dist.all_reduce(lt)
    if rank == 0:
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
        torch.save({"loss_mean": (lt / world).item(), "params": flat, "overlapped_buckets": overlapped, "started_before_finish": started,
                    "n_buckets": len(step.ddp.slices), "slices": torch.tensor(step.ddp.slices)}, args.out)


def sample_cuda(args, rank, world, dev):
    import afdm
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(FSET), device=dev, variant=3).to(dev)
    diff = afdm.Diffusion(noise_steps=201, img_size=32, device=dev)
    afdm.set_seed(5)
    xq, rq = diff.sample_sharded(model, n=5, image_channels=3)
    thetas = [-90.0, 0.0, 45.0]
    afdm.set_seed(5)
    xs, rs = diff.sample_rotation_sweep_sharded(model, 2, 3, thetas)
    if rank == 0:
        torch.save({"x": xq.cpu(), "result": rq.cpu(), "rot_x": torch.stack([t.cpu() for t in xs]),
                    "rot_result": torch.stack([t.cpu() for t in rs])}, args.out)
    else:
        assert xq is None and rq is None and xs is None and rs is None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--mode", default="grads", choices=["grads", "shards", "train", "sample"])
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(args.backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    if args.device == "cpu":
        (grads_cpu if args.mode == "grads" else shards_cpu)(args, rank, world)
    else:
        dev = torch.device("cuda", rank if args.backend == "nccl" else 0)
        torch.cuda.set_device(dev)
        (train_cuda if args.mode == "train" else sample_cuda)(args, rank, world, dev)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
