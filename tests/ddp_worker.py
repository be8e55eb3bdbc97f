"""Worker for the world_size>1 tests (launched by torch.distributed.run, gloo backend).

--device cpu : exercises the data-parallel plumbing only (FlatParams + GradAllReduce bucketing) on
               CPU tensors -- no kernels are called, so it runs in the GPU-less build container.
--device cuda: every rank runs the real HIP train step on cuda:0 with its shard of the golden batch
               (the gradient exchange is staged through host memory because the ranks share one GPU).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import afdm

    if args.device == "cpu":
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(13, 7), torch.nn.Linear(7, 5))
        fp = afdm.FlatParams(net)
        assert all(p.data_ptr() == fp.flat.data_ptr() + 4 * o for p, o in zip(fp.params, fp.offsets))
        ddp = afdm.GradAllReduce(fp.grad, n_buckets=3)
        fp.zero_grad()
        g = torch.Generator().manual_seed(100 + rank)
        local = [torch.randn(p.shape, generator=g) for p in fp.params]
        for p, l in zip(fp.params, local):
            p.grad.copy_(l)                                    # what autograd's in-place accumulation does
        scale = ddp()
        mean = fp.grad * scale
        # reference: gather every rank's grads the slow way
        want = torch.zeros_like(fp.grad)
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            want += torch.cat([torch.randn(p.shape, generator=gr).reshape(-1) for p in fp.params])
        want /= world
        ok = torch.allclose(mean, want, rtol=1e-6, atol=1e-7) and len(ddp.slices) == 3 and scale == 1.0 / world
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok))
        if rank == 0:
            torch.save({"ok": all(flags), "world": world}, args.out)
    else:
        dev = torch.device("cuda:0")
        g = np.load(os.path.join(ROOT, "tests", "golden", "train_step.npz"), allow_pickle=False)
        fset = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
        afdm.set_seed(42)
        model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=fset, device=dev, variant=3).to(dev)
        diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
        step = afdm.TrainStep(model, diff, lr=3e-4, graph=os.environ.get("AFD_TEST_GRAPH", "0") == "1", distributed=True)
        B = g["images"].shape[0] // world
        sl = slice(rank * B, (rank + 1) * B)
        T = lambda a: torch.from_numpy(np.asarray(a))
        loss = step(T(g["images"][sl]).to(dev), t=T(g["t0"][sl]), eps=T(g["eps0"][sl]).to(dev))
        lt = loss.detach().cpu().reshape(1)
        dist.all_reduce(lt)
        if rank == 0:
            flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
            torch.save({"loss_mean": (lt / world).item(), "params": flat}, args.out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
