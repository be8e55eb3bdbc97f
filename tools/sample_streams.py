"""Sampling throughput against the number of independent trajectories in flight (Diffusion.sample_concurrent)."""
import sys, os, time, math
sys.path.insert(0, "/root/repo")
import torch, afdm
dev = torch.device("cuda:0")
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=201, img_size=32, device=dev)
for graph in (False, True):
    for streams, batch in [(1, 256), (2, 256), (4, 256), (6, 256), (2, 512), (4, 128), (8, 128)]:
        n = streams * batch
        diff.sample_concurrent(model, n, 3, batch=batch, streams=streams, graph=graph)
        torch.cuda.synchronize(); t0 = time.time()
        diff.sample_concurrent(model, n, 3, batch=batch, streams=streams, graph=graph)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f"graph={graph} streams={streams} batch={batch}: {n / (dt * 999 / 200):.1f} img/s (T=1000 equivalent)", flush=True)
