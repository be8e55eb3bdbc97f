"""F12-F17: the DDPM process (modules/ddpm_models.py:301-436) over the HIP kernels.

Bit-exactness: the schedule tables are built on the host with the reference's own torch calls (fp32
linspace; torch.cumprod, which on the CPU accumulates fp32 inputs in double -- DESIGN.md section 4); `sample_timesteps` draws from torch's CPU generator exactly like
the reference; noise_images / the denoise update / the uint8 quantisation are bit-exact HIP
restatements (csrc/ddpm.hip).  The sampling loop stays on the device: no per-step H2D copy of `t`
(the reference does one per step, :362) and no per-step host sync.
"""
import logging

import numpy as np
import torch
import torch.distributed as dist

from . import ops


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank` of `world` (sizes differ by at most one; empty when n < world)."""
    return n * rank // world, n * (rank + 1) // world


def gather_u8(t, counts, group=None, dst=0):
    """Collect every rank's (k_r, ...) uint8 tensor on rank `dst`, concatenated in rank order; `counts[r]` = k_r (known to
    every rank: shard sizes are a function of n and the world size).  Shards are padded to the largest so one all_gather
    does it; with 'gloo' the bytes are staged through host memory.  Returns the tensor on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    via_host = dist.get_backend(group) == "gloo"
    kmax = max(max(counts), 1)
    pad = torch.zeros((kmax,) + tuple(t.shape[1:]), dtype=torch.uint8, device="cpu" if via_host else t.device)
    if t.shape[0]:
        pad[:t.shape[0]].copy_(t)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    if rank != dst:
        return None
    return torch.cat([parts[r][:counts[r]] for r in range(world)]).to(t.device)


class Diffusion:
    def __init__(self, noise_steps=1000, beta_start=1e-4, beta_end=0.02, img_size=256, device="cuda"):
        self.noise_steps, self.beta_start, self.beta_end = noise_steps, beta_start, beta_end
        self.img_size, self.device = img_size, device
        beta = self.prepare_noise_schedule()                    # host fp32
        alpha = 1.0 - beta
        alpha_hat = torch.cumprod(alpha, dim=0)                 # host ATen cumprod, as the reference (:309): accumulates in double, rounds to fp32
        self.beta, self.alpha, self.alpha_hat = beta.to(device), alpha.to(device), alpha_hat.to(device)
        self.filter = None
        self._t_cache = {}

    def prepare_noise_schedule(self):
        return torch.linspace(self.beta_start, self.beta_end, self.noise_steps)

    # F14 ---------------------------------------------------------------------------------
    def noise_images(self, x, t, eps=None):
        """-> (x_t, eps).  `eps` may be injected (parity mode); default = device RNG like the reference."""
        if eps is None:
            eps = torch.randn_like(x)
        return ops.noise_images(x, eps, t.to(x.device), self.alpha_hat), eps

    # F13 ---------------------------------------------------------------------------------
    def sample_timesteps(self, n):
        return torch.randint(low=1, high=self.noise_steps, size=(n,))     # CPU global generator, t in [1, T-1]

    # F16 ---------------------------------------------------------------------------------
    def _t_full(self, n, i, device):
        """Device tensor full((n,), i): built once per (n, stream) as arange and sliced -- no per-step H2D.
        Keyed by the current HIP stream as well: the table is filled by kernels on the stream that first asks, so a
        trajectory of `sample_concurrent` on another stream builds (and later frees) its own copy instead of reading one
        whose fill it is not ordered behind."""
        key = (n, str(device), ops._stream())
        tab = self._t_cache.get(key)
        if tab is None:
            tab = torch.arange(self.noise_steps, device=device, dtype=torch.long)[:, None].repeat(1, n).contiguous()
            if len(self._t_cache) >= 16:
                self._t_cache.pop(next(iter(self._t_cache)))
            self._t_cache[key] = tab
        return tab[i]

    def _initial_noise(self, n, c, noise_source, shard=None):
        """shard = (lo, hi): draw the noise of all n images (every rank consumes the identical generator stream) and keep
        rows lo:hi -- a sharded run then produces exactly the images of the one-rank run."""
        shape = (n, c, self.img_size, self.img_size)
        if noise_source == "device":
            x = torch.randn(shape, device=self.device)
        else:
            x = torch.randn(shape).to(self.device)                         # reference: CPU draw, then copy (:360)
        return x if shard is None else x[shard[0]:shard[1]].contiguous()

    def _step_noise(self, x, noise_source, shard=None, n=None):
        shape = x.shape if shard is None else (n,) + tuple(x.shape[1:])
        if noise_source == "cpu":      # parity mode: replay the reference's CPU-path stream
            z = torch.randn(shape).to(x.device)
        else:
            z = torch.randn(shape, device=x.device)
        return z if shard is None else z[shard[0]:shard[1]].contiguous()

    def _loop(self, model, n, image_channels, theta=None, noise_source="reference", graph=None, shard=None):
        """Shared body of sample / revert.  noise_source: 'reference' (x_T from the CPU generator,
        per-step noise from the device generator -- what the reference does on a GPU), 'cpu'
        (everything from the CPU generator: reproduces the reference's CPU run), 'device'.
        graph: capture ONE denoise step (UNet forward + device noise + update) into a hipGraph and replay
        it for i = T-1 .. 2 (the step index lives in device memory).  Off by default: measured on MI355X the
        loop is bound by the ~330 dependent kernel boundaries per step on the device (2.18 ms/step at n=6
        with or without replay), not by host launches."""
        theta_step = None if theta is None else theta / self.noise_steps
        if graph is None:
            graph = False
        self._hint(model)
        model.eval()
        snaps = []
        with torch.no_grad():
            x = self._initial_noise(n, image_channels, noise_source, shard)
            n_all, n = n, x.shape[0]
            if graph and theta is None and noise_source != "cpu" and shard is None:
                x = self._graph_steps(model, x, snaps)
                first_eager = 1
            else:
                first_eager = self.noise_steps - 1
            for i in reversed(range(1, first_eager + 1)):
                if n == 0:                               # an empty shard still consumes the shared noise stream
                    if i > 1:
                        self._step_noise(x, noise_source, shard, n_all)
                    if i % 100 == 0:
                        snaps.append(x)
                    continue
                t = self._t_full(n, i, x.device)
                eps = model(x, t)
                noise = self._step_noise(x, noise_source, shard, n_all) if i > 1 else None
                x = ops.denoise_step(x, eps, noise, self.alpha, self.alpha_hat, self.beta, i)
                if theta_step is not None:
                    x = self.rotate_2d_matrix(x, theta_step, self.filter)
                if i % 100 == 0:
                    snaps.append(x)
        model.train()        # the reference leaves the model in train mode (:379)
        self._unhint(model)
        snaps.append(x)
        return x, snaps

    def _hint(self, model):
        """Tell the model the range of the timesteps this process will pass (all of them in [0, noise_steps)): lets the
        UNet tabulate its time embeddings once per trajectory (unet.UNet._timestep_tables); cleared again when the loop
        ends (_unhint), so a direct call of the model may pass any t."""
        if hasattr(model, "_t_range"):
            model._t_range = self.noise_steps

    @staticmethod
    def _unhint(model):
        """The loop is over: a later direct call of the model may pass any t again."""
        if hasattr(model, "_t_range"):
            model._t_range = None

    def _graph_steps(self, model, x, snaps):
        """Steps i = T-1 .. 2 by replaying one captured step; returns x after step 2."""
        n = x.shape[0]
        t_dev = torch.full((n,), self.noise_steps - 1, device=x.device, dtype=torch.long)
        xs = x.clone()

        def one_step():
            eps = model(xs, t_dev)
            noise = torch.randn_like(xs)
            ops.denoise_step_dev(xs, eps, noise, self.alpha, self.alpha_hat, self.beta, t_dev, xs)   # in place (elementwise)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        keep = xs.clone()
        rng = torch.cuda.get_rng_state(x.device)
        with torch.cuda.stream(side):
            one_step()                                   # warm-up (allocator, lazy init); its effect is undone below
        torch.cuda.current_stream().wait_stream(side)
        xs.copy_(keep)
        torch.cuda.set_rng_state(rng, x.device)          # ... including its noise draw: the replays consume the stream the eager loop would
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one_step()
        for i in reversed(range(2, self.noise_steps)):
            t_dev.fill_(i)
            g.replay()
            if i % 100 == 0:
                snaps.append(xs.clone())
        return xs.clone()

    def sample(self, model, n, image_channels, theta=None, noise_source="reference", return_float=False, graph=None):
        logging.info(f"Sampling {n} new images....")
        if theta is not None:
            logging.info(f"Theta {theta} provided. Rotation will be applied.")
        x, snaps = self._loop(model, n, image_channels, theta, noise_source, graph)
        self.last_float_snapshots = snaps          # pre-quantisation x at i % 100 == 0 and the final x (parity tests)
        xq = ops.quantize_u8(x)
        rq = ops.quantize_u8(torch.cat(snaps))
        if return_float:
            return xq, rq, x
        return xq, rq

    def sample_sharded(self, model, n, image_channels, theta=None, noise_source="reference", group=None, dst=0):
        """`sample` with the n images partitioned over the ranks of `group` (sampling is embarrassingly parallel per
        image: replicas only, no collective in the loop) and the uint8 results gathered on rank `dst`.
        Every rank must have been seeded alike (as the reference's scripts do with set_seed): each rank draws the noise
        of ALL n images from the same generator stream and keeps its rows, so the gathered (x, result) equal what one rank
        computes for the same seed -- whatever the world size.  Returns (x_u8, result_u8) on `dst`, (None, None) elsewhere;
        without an initialised process group it is `sample`."""
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return self.sample(model, n, image_channels, theta=theta, noise_source=noise_source)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi = shard_range(n, rank, world)
        x, snaps = self._loop(model, n, image_channels, theta, noise_source, None, shard=(lo, hi))
        counts = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
        xq = gather_u8(ops.quantize_u8(x) if hi > lo else torch.empty((0,) + tuple(x.shape[1:]), dtype=torch.uint8, device=x.device),
                       counts, group, dst)
        # result = cat(snapshots): (S * n_r, ...) on each rank -> (S, n, ...) snapshot-major like `sample`
        S = len(snaps)
        rq_local = ops.quantize_u8(torch.cat(snaps)) if hi > lo else torch.empty((0,) + tuple(x.shape[1:]), dtype=torch.uint8, device=x.device)
        rq = gather_u8(rq_local, [S * c for c in counts], group, dst)
        if rank != dst:
            return None, None
        offs = np.cumsum([0] + [S * c for c in counts])
        per_rank = [rq[offs[r]:offs[r + 1]].reshape(S, counts[r], *rq.shape[1:]) for r in range(world)]
        return xq, torch.cat(per_rank, dim=1).reshape(S * n, *rq.shape[1:])

    def sample_rotation_sweep_sharded(self, model, n, image_channels, thetas, group=None, dst=0):
        """Config E sweep (ddpm_tasks.py:346-369) with the ANGLES partitioned over the ranks (BASELINE config 5): each rank
        runs `sample_rotation_sweep` on its contiguous share of `thetas`; since that draws the n-image noise of every step
        once and shares it between its angles, all ranks (seeded alike, ddpm_tasks.py:365) consume the identical noise
        stream however many angles they hold -- "same seed per theta" holds across the shards.  Gathers on `dst`:
        ([x_u8 per angle], [result_u8 per angle]) in the order of `thetas`; (None, None) elsewhere."""
        thetas = list(thetas)
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return self.sample_rotation_sweep(model, n, image_channels, thetas)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        K = len(thetas)
        lo, hi = shard_range(K, rank, world)
        xs, rs = self.sample_rotation_sweep(model, n, image_channels, thetas[lo:hi])
        S = (self.noise_steps - 1) // 100 + 1                       # snapshots at i % 100 == 0 (0 < i < T) + the final x
        shape = (image_channels, self.img_size, self.img_size)
        empty = torch.empty((0,) + shape, dtype=torch.uint8, device=self.device)
        counts = [shard_range(K, r, world)[1] - shard_range(K, r, world)[0] for r in range(world)]
        xa = gather_u8(torch.cat(xs) if xs else empty, [c * n for c in counts], group, dst)
        ra = gather_u8(torch.cat(rs) if rs else empty, [c * n * S for c in counts], group, dst)
        if rank != dst:
            return None, None
        return list(xa.reshape(K, n, *shape)), list(ra.reshape(K, S * n, *shape))

    def sample_rotation_sweep(self, model, n, image_channels, thetas):
        """Config E sweep (ddpm_tasks.py:346-369) as ONE batched trajectory.  The reference re-seeds before every angle,
        so every angle consumes the identical noise stream and only the per-step rotation differs: here the
        len(thetas) * n images ride one batch, every noise draw (x_T from the CPU generator, the per-step noise from the
        device generator, n images each: the same draws in the same order as one `sample(n, theta)` call after the same
        seeding) is shared by all angles, and each angle's slice is rotated by its own theta / T after every step.
        One UNet forward per step instead of len(thetas): at n = 4 the sweep is launch-bound, so this is ~9x faster for
        the reference's 9 angles.  Returns ([x_u8 per angle], [result_u8 per angle]) like `rotation_results`."""
        K = len(thetas)
        if K == 0:                                       # (a rank of the sharded sweep that holds no angle)
            return [], []
        self._hint(model)
        model.eval()
        snaps = [[] for _ in range(K)]
        with torch.no_grad():
            x = self._initial_noise(n, image_channels, "reference").repeat(K, 1, 1, 1)
            for i in reversed(range(1, self.noise_steps)):
                eps = model(x, self._t_full(K * n, i, x.device))
                noise = torch.randn(n, *x.shape[1:], device=x.device).repeat(K, 1, 1, 1) if i > 1 else None
                x = ops.denoise_step(x, eps, noise, self.alpha, self.alpha_hat, self.beta, i)
                for k, th in enumerate(thetas):
                    if th:                                   # the reference rotates only for a truthy theta (:375)
                        x[k * n:(k + 1) * n] = self.rotate_2d_matrix(x[k * n:(k + 1) * n], th / self.noise_steps, self.filter)
                if i % 100 == 0:
                    for k in range(K):
                        snaps[k].append(x[k * n:(k + 1) * n].clone())
        model.train()
        self._unhint(model)
        xs, results = [], []
        for k in range(K):
            xk = x[k * n:(k + 1) * n]
            xs.append(ops.quantize_u8(xk))
            results.append(ops.quantize_u8(torch.cat(snaps[k] + [xk])))
        return xs, results

    def sample_concurrent(self, model, n, image_channels, batch=256, streams=4, noise_fn=None, graph=False):
        """Throughput form of `sample` for many images: the n images are cut into batches of `batch` and `streams`
        of those trajectories run CONCURRENTLY, each on its own HIP stream (the trajectories are independent: sampling
        is embarrassingly parallel per image).  One 256-image forward leaves CUs idle in its small layers; a second
        trajectory in flight fills them: measured on MI355X (tools/sample_streams.py) 77 / 96 / 103 / 107 images/s with
        1 / 2 / 3 / 4 trajectories of 256 images in flight (a single 512-image batch: 91).  Returns (x_u8, result_u8) like `sample`, batches
        concatenated.  Noise comes from the device generator (or noise_fn(batch_index, i, x) for tests).
        graph=True: every trajectory's denoise step (UNet forward + update; the noise is drawn into a static buffer just
        before) is captured once into its own hipGraph and replayed on its stream, which takes the host's ~170 launches
        per forward out of the loop."""
        self._hint(model)
        model.eval()
        if graph:
            return self._sample_concurrent_graphs(model, n, image_channels, batch, streams, noise_fn)
        sizes = [min(batch, n - o) for o in range(0, n, batch)]
        pool = [torch.cuda.Stream() for _ in range(max(1, min(streams, len(sizes))))]
        cur = torch.cuda.current_stream()
        xs_out, snaps_out = [None] * len(sizes), [None] * len(sizes)
        with torch.no_grad():
            for g0 in range(0, len(sizes), len(pool)):
                group = list(range(g0, min(g0 + len(pool), len(sizes))))
                xs, snaps = {}, {k: [] for k in group}
                for k in group:
                    st = pool[k - g0]
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        xs[k] = torch.randn(sizes[k], image_channels, self.img_size, self.img_size, device=self.device) \
                            if noise_fn is None else noise_fn(k, self.noise_steps, (sizes[k], image_channels, self.img_size, self.img_size))
                for i in reversed(range(1, self.noise_steps)):
                    for k in group:                                  # one denoise step of every trajectory of the group
                        with torch.cuda.stream(pool[k - g0]):
                            x = xs[k]
                            eps = model(x, self._t_full(x.shape[0], i, x.device))
                            noise = None if i == 1 else (torch.randn_like(x) if noise_fn is None else noise_fn(k, i, x.shape))
                            x = ops.denoise_step(x, eps, noise, self.alpha, self.alpha_hat, self.beta, i)
                            if i % 100 == 0:
                                snaps[k].append(x)
                            xs[k] = x
                for k in group:
                    with torch.cuda.stream(pool[k - g0]):
                        snaps[k].append(xs[k])
                        xs_out[k] = ops.quantize_u8(xs[k])
                        snaps_out[k] = ops.quantize_u8(torch.cat(snaps[k]))
                    cur.wait_stream(pool[k - g0])
        model.train()
        self._unhint(model)
        for t in xs_out + snaps_out:
            t.record_stream(cur)
        return torch.cat(xs_out), torch.cat(snaps_out)

    def _sample_concurrent_graphs(self, model, n, image_channels, batch, streams, noise_fn):
        sizes = [min(batch, n - o) for o in range(0, n, batch)]
        pool = [torch.cuda.Stream() for _ in range(max(1, min(streams, len(sizes))))]
        cur = torch.cuda.current_stream()
        xs_out, snaps_out = [None] * len(sizes), [None] * len(sizes)
        shape = (image_channels, self.img_size, self.img_size)
        slots = {}                                           # (batch size, stream) -> (x, t, noise, graph), captured once
        with torch.no_grad():
            for g0 in range(0, len(sizes), len(pool)):
                group = list(range(g0, min(g0 + len(pool), len(sizes))))
                state, snaps = {}, {k: [] for k in group}
                for k in group:
                    st = pool[k - g0]
                    st.wait_stream(cur)
                    key = (sizes[k], k - g0)
                    with torch.cuda.stream(st):
                        if key not in slots:
                            xs = torch.zeros(sizes[k], *shape, device=self.device)
                            t_dev = torch.full((sizes[k],), self.noise_steps - 1, device=self.device, dtype=torch.long)
                            nz = torch.zeros_like(xs)

                            def one_step(xs=xs, t_dev=t_dev, nz=nz):
                                eps = model(xs, t_dev)
                                ops.denoise_step_dev(xs, eps, nz, self.alpha, self.alpha_hat, self.beta, t_dev, xs)
                            one_step()                       # warm-up outside capture (allocator, cached weight transforms)
                            st.synchronize()
                            g = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g, stream=st):
                                one_step()
                            slots[key] = (xs, t_dev, nz, g)
                        xs = slots[key][0]
                        xs.copy_(torch.randn(sizes[k], *shape, device=self.device) if noise_fn is None
                                 else noise_fn(k, self.noise_steps, (sizes[k], *shape)))
                    state[k] = slots[key]
                for i in reversed(range(1, self.noise_steps)):
                    for k in group:
                        xs, t_dev, nz, g = state[k]
                        with torch.cuda.stream(pool[k - g0]):
                            if i > 1:
                                t_dev.fill_(i)
                                if noise_fn is None:
                                    nz.normal_()
                                else:
                                    nz.copy_(noise_fn(k, i, xs.shape))
                                g.replay()
                            else:                            # the last step adds no noise (ddpm_models.py:370-373)
                                eps = model(xs, self._t_full(xs.shape[0], 1, xs.device))
                                xs.copy_(ops.denoise_step(xs, eps, None, self.alpha, self.alpha_hat, self.beta, 1))
                            if i % 100 == 0:
                                snaps[k].append(xs.clone())
                for k in group:
                    with torch.cuda.stream(pool[k - g0]):
                        xs = state[k][0]
                        xs_out[k] = ops.quantize_u8(xs)
                        snaps_out[k] = ops.quantize_u8(torch.cat(snaps[k] + [xs]))
                    cur.wait_stream(pool[k - g0])
        model.train()
        self._unhint(model)
        for t in xs_out + snaps_out:
            t.record_stream(cur)
        return torch.cat(xs_out), torch.cat(snaps_out)

    def revert(self, model, n, image_channels, noise_source="reference", graph=None):
        logging.info(f"Sampling {n} new images....")
        _, snaps = self._loop(model, n, image_channels, None, noise_source, graph)
        return ops.quantize_u8(torch.cat(snaps))

    # under development in the reference (:388-419); kept as a host-driven loop over the HIP step
    def sample_shift(self, model, n, image_channels, shift=None, noise_source="reference"):
        logging.info(f"Sampling {n} new images....")
        if shift == 0:
            shift = None
        idx = None
        if shift is not None:
            dur = np.abs(shift) / self.noise_steps
            idx = set(np.round(np.arange(0, self.noise_steps, dur)).astype(int)[1:].tolist())
        self._hint(model)
        model.eval()
        with torch.no_grad():
            x = self._initial_noise(n, image_channels, noise_source)
            for i in reversed(range(1, self.noise_steps)):
                eps = model(x, self._t_full(n, i, x.device))
                noise = self._step_noise(x, noise_source) if i > 1 else None
                x = ops.denoise_step(x, eps, noise, self.alpha, self.alpha_hat, self.beta, i)
                if idx is not None and i in idx:
                    x = self.shift_2d_matrix(x, 1 * np.sign(shift), 0, self.device)
        model.train()
        self._unhint(model)
        return ops.quantize_u8(x)

    # F17 (Config E): the reference rotates on the CPU with scipy every step (D2H, single-threaded spline,
    # H2D).  For device tensors the same order-3 / grid-wrap / prefiltered spline runs in a HIP kernel (fp64).
    @staticmethod
    def rotate_2d_matrix(matrix, degrees, filter=None):
        if matrix.is_cuda:
            return ops.rotate_spline3_wrap(matrix, degrees)
        from scipy import ndimage
        r = ndimage.rotate(input=matrix.numpy(), angle=degrees, axes=(2, 3), reshape=False, mode="grid-wrap")
        return torch.from_numpy(r)

    @staticmethod
    def shift_2d_matrix(matrix, hshift, vshift, device):
        """ndimage.shift(x, (0,0,v,h), mode='grid-wrap') (ddpm_models.py:431-436).  The reference only ever
        shifts by whole pixels (:415), where the periodic spline shift is exactly a roll: done on the device."""
        if matrix.is_cuda and float(hshift).is_integer() and float(vshift).is_integer():
            return torch.roll(matrix, shifts=(int(vshift), int(hshift)), dims=(2, 3)).to(device)
        from scipy import ndimage
        r = ndimage.shift(input=matrix.cpu().numpy(), shift=(0, 0, vshift, hshift), mode="grid-wrap")
        return torch.from_numpy(r).to(device)
