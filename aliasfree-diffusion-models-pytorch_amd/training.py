"""F15: the train step (modules/ddpm_utils.py:483-518) -- AdamW + MSE on the HIP engine, optional
data parallelism (one process per GPU, RCCL all-reduce of one flat fp32 gradient buffer), and
hipGraph replay of the whole step.
"""
import logging
import os

import torch
import torch.distributed as dist

from . import ops
from ._lib import lib


class argument:
    """Attribute bag of run settings (ddpm_utils.py:11-23)."""

    def __init__(self, run_name=None, epochs=None, batch_size=None, image_size=None, image_channels=3,
                 dataset_path=None, device=None, lr=None, noise_steps=None, image_gen_n=4):
        self.run_name, self.epochs, self.batch_size, self.image_size = run_name, epochs, batch_size, image_size
        self.image_channels, self.dataset_path, self.device, self.lr = image_channels, dataset_path, device, lr
        self.noise_steps, self.image_gen_n = noise_steps, image_gen_n


def set_seed(seed):
    """modules/utils.py:98-105."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def setup_logging(run_name):
    """modules/utils.py:84-88."""
    os.makedirs("models", exist_ok=True)
    os.makedirs("results", exist_ok=True)
    os.makedirs(os.path.join("models", run_name), exist_ok=True)
    os.makedirs(os.path.join("results", run_name), exist_ok=True)


class FlatParams:
    """Re-homes a model's parameters (and their .grad) as views of two flat fp32 buffers, in
    `parameters()` order, so the optimiser is one kernel launch and the DDP exchange is a handful of
    large all-reduces.  state_dict()/load_state_dict() keep working (the Parameters are the same
    objects; only their storage moved)."""

    def __init__(self, model):
        params = [p for p in model.parameters()]
        assert params and all(p.dtype == torch.float32 for p in params)
        dev = params[0].device
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.empty(self.numel, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.offsets = []
        o = 0
        for p in params:
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            self.offsets.append(o)
            o += n

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # autograd may have replaced .grad; re-attach the views
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class FusedAdamW:
    """torch.optim.AdamW(params, lr) semantics (betas .9/.999, eps 1e-8, weight_decay 0.01 --
    the defaults the reference relies on, ddpm_utils.py:489) as ONE kernel over the flat buffers.
    The step counter and bias corrections live on the device so a captured hipGraph replays correctly."""

    def __init__(self, model_or_flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.fp = model_or_flat if isinstance(model_or_flat, FlatParams) else FlatParams(model_or_flat)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        dev = self.fp.flat.device
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.state = torch.zeros(4, device=dev, dtype=torch.float32)

    def zero_grad(self, set_to_none=False):
        self.fp.zero_grad()

    def step(self, grad_scale=1.0):
        L, s = lib(), torch.cuda.current_stream().cuda_stream
        L.afd_adamw_tick(self.state.data_ptr(), self.betas[0], self.betas[1], s)
        L.afd_adamw_step(self.fp.flat.data_ptr(), self.fp.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                         self.fp.numel, self.state.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps,
                         self.weight_decay, grad_scale, s)
        ops.bump_param_epoch()                          # the parameters moved under raw pointers: cached transforms are stale


class GradAllReduce:
    """Data-parallel gradient exchange: SUM all-reduce of the flat gradient buffer in `n_buckets`
    contiguous slices (a few MB each: xGMI rings are per-link bound, so few large messages), the mean
    is folded into AdamW's grad_scale.  Works on any torch.distributed backend: 'nccl' (= RCCL over
    xGMI) on GPUs; with 'gloo' the slices are staged through host memory (used by the CPU tests and
    when several ranks share one GPU)."""

    def __init__(self, flat_grad, n_buckets=4, group=None):
        self.g, self.group = flat_grad, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = flat_grad.numel()
        nb = max(1, min(n_buckets, n))
        edges = [n * i // nb for i in range(nb + 1)]
        self.slices = [(edges[i], edges[i + 1]) for i in range(nb) if edges[i + 1] > edges[i]]
        self.via_host = dist.is_initialized() and dist.get_backend(group) == "gloo" and flat_grad.is_cuda

    def __call__(self):
        if self.world == 1:
            return 1.0
        if self.via_host:
            h = self.g.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            self.g.copy_(h)
        else:
            works = [dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                     for a, b in reversed(self.slices)]          # last layers' grads are ready first
            for w in works:
                w.wait()
        return 1.0 / self.world


class TrainStep:
    """The reference's per-batch body (ddpm_utils.py:499-507) as one callable:
         t -> noise_images -> UNet -> MSE -> zero_grad -> backward -> [all-reduce] -> AdamW.
    `graph=True` captures the device work into a hipGraph on first use (static shapes): the whole step on
    one GPU; with data parallelism everything up to and including backward -- the gradient all-reduce
    (one 23.6 MB exchange) and AdamW then run after the replay.  The CPU-generator timestep draw and the
    H2D copies always stay outside the graph."""

    def __init__(self, model, diffusion, lr, graph=False, distributed=None, n_buckets=4, overlap_wgrad=None):
        self.model, self.diffusion = model, diffusion
        # weight-gradient kernels on a second stream (ops._GradMode.side): off the critical path of backward, they fill
        # the CUs the dependent chain of small kernels leaves idle.  Measured on MI355X (B=256): eager 12.0 -> 11.1
        # ms/step, captured graph 11.45 -> 11.3 (forks batched 16 layers at a time: every fork is a cross-stream edge
        # in the graph, and 57 of them cost more than the overlap returns).  Layers per fork, eager: 1 / 2 / 4 / 8 / 16 ->
        # 9.77 / 9.67 / 9.66 / 9.87 / 9.95 ms.  Stream priorities (side high, or main high) both measured slower.
        if overlap_wgrad is None:
            overlap_wgrad = True
        self.wgrad_stream = torch.cuda.Stream() if overlap_wgrad else None
        self.wgrad_batch = int(os.environ.get("AFD_WGRAD_BATCH", 16 if graph else 4))      # layers per fork (tuning hook)
        self.opt = FusedAdamW(model, lr=lr)
        want_ddp = distributed if distributed is not None else dist.is_initialized()
        self.ddp = GradAllReduce(self.opt.fp.grad, n_buckets) if want_ddp else None
        self.use_graph = graph
        self._graph = None
        self._static = None
        self._wino_plan, self._wino_requests = None, None      # ops.WinoStepPlan after the first (recording) step

    def _fwd_bwd(self, images, t, eps):
        W = ops._WinoWeights
        if self._wino_plan is not None and self._wino_plan.valid():
            self._wino_plan.launch()                   # every transformed-weight image of the step, one launch
            W.active_plan = self._wino_plan
        elif self._wino_requests is None:
            self._wino_requests = W.recording = {}     # first step: note which images the dispatch asks for
        try:
            x_t, noise = self.diffusion.noise_images(images, t, eps)
            pred = self.model(x_t, t)
            loss = ops.mse_loss(noise, pred)
            self.opt.zero_grad()
            with ops.inplace_param_grads(self.wgrad_stream, self.wgrad_batch):   # weight gradients add straight into the flat .grad views
                loss.backward()
        finally:
            if W.recording is not None and W.recording is self._wino_requests:
                W.recording = None
                self._wino_plan = ops.WinoStepPlan(self._wino_requests)
            W.active_plan = None
        return loss.detach()

    def _update(self):
        scale = self.ddp() if self.ddp is not None else 1.0
        self.opt.step(grad_scale=scale)

    def _body(self, images, t, eps):
        loss = self._fwd_bwd(images, t, eps)
        self._update()
        return loss

    def __call__(self, images, t=None, eps=None):
        """images (B,C,S,S) on the device; t (B,) int64 [default: diffusion.sample_timesteps];
        eps: injected noise or None (device RNG).  Returns the loss as a 0-d device tensor."""
        if t is None:
            t = self.diffusion.sample_timesteps(images.shape[0])
        t = t.to(images.device, non_blocking=True)
        if not self.use_graph:
            return self._body(images, t, eps)
        whole = self.ddp is None                    # single GPU: AdamW is captured too
        if self._graph is None:
            self._static = {"images": images.clone(), "t": t.clone(), "eps": None if eps is None else eps.clone()}
            st = self._static
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):                      # warm-up outside capture (allocator, lazy init)
                    self._body(st["images"], st["t"], st["eps"])
            torch.cuda.current_stream().wait_stream(s)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                st["loss"] = (self._body if whole else self._fwd_bwd)(st["images"], st["t"], st["eps"])
        st = self._static
        st["images"].copy_(images)
        st["t"].copy_(t)
        if eps is not None:
            st["eps"].copy_(eps)
        self._graph.replay()
        ops.bump_param_epoch()                          # the replayed AdamW moved the parameters
        if not whole:
            self._update()
        return st["loss"]


def train(args, model_path=None, dataloader=None, model=None, diffusion=None):
    """Drop-in for modules/ddpm_utils.py:483-518: returns the list of per-epoch mean losses; saves a
    preview grid and the state_dict every epoch."""
    from tqdm import tqdm
    setup_logging(args.run_name)
    device = args.device
    step = TrainStep(model, diffusion, lr=args.lr, graph=False)
    n_batches = len(dataloader)
    loss_all = []
    for epoch in range(args.epochs):
        logging.info(f"Starting epoch {epoch}:")
        pbar = tqdm(dataloader)
        epoch_loss = torch.zeros((), device=device)
        for i, (images, _) in enumerate(pbar):
            loss = step(images.to(device))
            epoch_loss += loss
            if i % 50 == 0:
                pbar.set_postfix(MSE=loss.item())
        loss_all.append(epoch_loss.item() / n_batches)
        sampled, _ = diffusion.sample(model, n=args.image_gen_n, image_channels=args.image_channels)
        try:
            from .imageio_utils import save_images
            save_images(sampled, os.path.join("results", args.run_name, f"{epoch}.jpg"))
        except Exception as e:                                     # preview only; never fail a run on I/O
            logging.warning(f"preview not saved: {e}")
        if model_path:
            torch.save(model.state_dict(), model_path)
    return loss_all
