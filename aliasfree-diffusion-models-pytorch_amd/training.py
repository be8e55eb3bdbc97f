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
    """Re-homes a model's parameters (and their .grad) as views of two flat fp32 buffers so the optimiser is one
    kernel launch and the DDP exchange is a handful of large all-reduces.  Order: `parameters()` order, except that
    parameters the forward never touches (`model.unused_parameters()`, e.g. variant 4's stage-level `norm1` or
    `label_emb` of an unconditionally trained net) sit at the tail, beyond `n_active`: the reference's
    `torch.optim.AdamW` skips parameters whose grad is None (ddpm_utils.py:489,504-506 after `zero_grad()`), so they
    must see neither the update nor the weight decay.  state_dict()/load_state_dict() keep working (the Parameters are
    the same objects; only their storage moved)."""

    def __init__(self, model, conditional=False):
        """conditional: the train loop will pass class labels (UNet.forward(x, t, y)), so `label_emb` is a live parameter
        (updated and exchanged); the reference's own loop never does (ddpm_utils.py:502), hence the default."""
        params = [p for p in model.parameters()]
        assert params and all(p.dtype == torch.float32 for p in params)
        unused = getattr(model, "unused_parameters", None)
        skip = set()
        if callable(unused):
            try:
                skip = {id(p) for p in unused(conditional=conditional)}
            except TypeError:
                skip = {id(p) for p in unused()}
        self.conditional = conditional
        params = [p for p in params if id(p) not in skip] + [p for p in params if id(p) in skip]
        dev = params[0].device
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.n_active = sum(p.numel() for p in params if id(p) not in skip)
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
    the defaults the reference relies on, ddpm_utils.py:489) as ONE kernel over the flat buffers (their first
    `n_active` elements: see FlatParams).
    The step counter and bias corrections live on the device so a captured hipGraph replays correctly."""

    def __init__(self, model_or_flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, conditional=False):
        self.fp = model_or_flat if isinstance(model_or_flat, FlatParams) else FlatParams(model_or_flat, conditional=conditional)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        dev = self.fp.flat.device
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.state = torch.zeros(4, device=dev, dtype=torch.float32)

    def zero_grad(self, set_to_none=False):
        self.fp.zero_grad()

    def step(self, grad_scale=1.0):
        L, s = lib(), ops._stream()
        L.afd_adamw_tick(self.state.data_ptr(), self.betas[0], self.betas[1], s)
        L.afd_adamw_step(self.fp.flat.data_ptr(), self.fp.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                         self.fp.n_active, self.state.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps,
                         self.weight_decay, grad_scale, s)
        ops.bump_param_epoch()                          # the parameters moved under raw pointers: cached transforms are stale


class GradAllReduce:
    """Data-parallel gradient exchange: SUM all-reduce of the flat gradient buffer in a few contiguous buckets (a few MB
    each: xGMI rings are per-link bound, so few large messages), the mean folded into AdamW's grad_scale.

    Overlap with backward: backward produces parameter gradients in reverse layer order, i.e. from the END of the flat
    buffer.  Every op that writes a parameter's gradient reports it (`wrote`, wired to ops._GradMode.on_write by
    TrainStep; launches deferred to the weight-gradient stream report when they are actually launched); when the last
    parameter of a bucket has been written, a communication stream waits for the streams that wrote into the bucket
    (events) and the bucket's all-reduce starts there while backward goes on with the earlier layers.  `finish()` joins
    them before AdamW (and exchanges whatever was not reported, e.g. after a hipGraph replay of the backward).
    Buckets are cut at top-level module boundaries walking the model backwards (for the UNet: [up1..outc], [bot2,bot3],
    [down3,sa3,bot1], [inc..sa2]: the last one to become ready is the smallest, 2.2 MB).

    Works on any torch.distributed backend: 'nccl' (= RCCL over xGMI) on GPUs; with 'gloo' and device tensors each
    bucket is staged through host memory (CPU tests, several ranks sharing one GPU)."""

    def __init__(self, flat, n_buckets=4, group=None, model=None):
        fp = flat if isinstance(flat, FlatParams) else None
        self.g = fp.grad[:fp.n_active] if fp is not None else flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = self.g.numel()
        nb = max(1, min(n_buckets, n))
        edges = None
        if fp is not None and model is not None:
            edges = self._module_edges(fp, model, nb)
        if edges is None:
            edges = [n * i // nb for i in range(nb + 1)]
        self.slices = [(edges[i], edges[i + 1]) for i in range(len(edges) - 1) if edges[i + 1] > edges[i]]
        self.via_host = dist.is_initialized() and dist.get_backend(group) == "gloo" and self.g.is_cuda
        self.comm = torch.cuda.Stream() if self.g.is_cuda else None
        self.side = None                                # the weight-gradient stream (set by TrainStep)
        # parameter -> bucket, for the readiness bookkeeping
        self._bucket_of, self._count = {}, [0] * len(self.slices)
        if fp is not None:
            for p_, o in zip(fp.params, fp.offsets):
                if o >= fp.n_active:
                    continue
                k = next(i for i, (a, b) in enumerate(self.slices) if a <= o < b)
                assert o + p_.numel() <= self.slices[k][1], "a parameter straddles two buckets"
                self._bucket_of[id(p_)] = k
                self._count[k] += 1
        self._left, self._seen, self._works, self._launched = list(self._count), set(), [], set()
        self.overlapped_last_step = 0          # buckets whose all-reduce started while backward was still running
        self.started_before_finish = 0         # ... plus those started by the final flush of the weight-gradient queue
        self._in_backward = None

    @staticmethod
    def _module_edges(fp, model, nb):
        """Bucket edges at top-level module boundaries, accumulating from the last module backwards."""
        kids = [m for m in model.children() if any(True for _ in m.parameters())]
        off = {id(p_): o for p_, o in zip(fp.params, fp.offsets)}
        starts = []
        for m in kids:
            os_ = [off[id(p_)] for p_ in m.parameters() if off[id(p_)] < fp.n_active]
            if os_:
                starts.append(min(os_))
        if len(starts) < 2 or starts != sorted(starts) or starts[0] != 0:
            return None
        target, edges, hi = fp.n_active / nb, [fp.n_active], fp.n_active
        for st in reversed(starts[1:]):
            if hi - st >= target and len(edges) < nb:
                edges.append(st)
                hi = st
        edges.append(0)
        return sorted(set(edges))

    # -- readiness bookkeeping (called during backward) ------------------------------------------
    def begin_step(self):
        self._left, self._seen, self._works, self._launched = list(self._count), set(), [], set()
        self._in_backward = None

    def backward_done(self):
        """Called when autograd's backward has returned (before the last queued weight-gradient launches are flushed): the
        buckets started up to here really overlapped backward."""
        self._in_backward = len(self._launched)

    def wrote(self, params):
        """The gradient of each given parameter has been fully written by work already ENQUEUED on the current stream or on
        the weight-gradient stream."""
        if self.world == 1:
            return
        for p_ in params:
            k = self._bucket_of.get(id(p_))
            if k is None or id(p_) in self._seen:
                continue
            self._seen.add(id(p_))
            self._left[k] -= 1
            if self._left[k] == 0:
                self._launch(k)

    def would_complete(self, params):
        """True if reporting `params` as written would complete (and so start the exchange of) some bucket: the hint that
        makes ops.flush_wgrads launch its queued gradient folds now rather than batch them further."""
        if self.world == 1:
            return False
        need = {}
        for p_ in params:
            k = self._bucket_of.get(id(p_))
            if k is not None and id(p_) not in self._seen and k not in self._launched:
                need.setdefault(k, set()).add(id(p_))
        return any(len(v) == self._left[k] for k, v in need.items())

    def _launch(self, k):
        if k in self._launched:
            return
        self._launched.add(k)
        a, b = self.slices[k]
        buf = self.g[a:b]
        if self.comm is None:                            # CPU tensors (gloo): plain async all-reduce
            self._works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None))
            return
        self.comm.wait_stream(torch.cuda.current_stream())
        for st in (self.side or ()):
            self.comm.wait_stream(st)
        with torch.cuda.stream(self.comm):
            if self.via_host:
                h = buf.to("cpu", non_blocking=False)
                self._works.append((dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group, async_op=True), h, buf))
            else:
                self._works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None))

    def finish(self):
        """Exchange every bucket not started yet, wait for all of them; returns the factor that turns the summed gradient
        into the mean (handed to AdamW as grad_scale)."""
        if self.world == 1:
            return 1.0
        self.started_before_finish = len(self._launched)
        self.overlapped_last_step = self._in_backward if self._in_backward is not None else len(self._launched)
        for k in reversed(range(len(self.slices))):      # last layers first
            self._launch(k)
        for w, h, buf in self._works:
            w.wait()                                     # nccl: the current stream waits for the collective
            if h is not None:
                buf.copy_(h)
        self._works = []
        return 1.0 / self.world

    def __call__(self):
        """Whole exchange at once (no overlap): every bucket, last layers first."""
        self.begin_step()
        return self.finish()


class TrainStep:
    """The reference's per-batch body (ddpm_utils.py:499-507) as one callable:
         t -> noise_images -> UNet -> MSE -> zero_grad -> backward -> [all-reduce] -> AdamW.
    `graph=True` captures the device work into a hipGraph on first use (static shapes): the whole step on
    one GPU; with data parallelism everything up to and including backward -- the gradient all-reduce
    (one 23.6 MB exchange) and AdamW then run after the replay.  The CPU-generator timestep draw and the
    H2D copies always stay outside the graph.
    `graph="lanes"` captures the same graph but never launches it: csrc/replay.hip walks its nodes once and re-issues
    them on TWO REAL STREAMS from a C++ loop (weight-gradient kernels on the side stream, cross-lane dependencies as
    events) -- the eager step's stream semantics with ~0.7 instead of 5.6 ms of host time per step and none of a
    hipGraph launch's cross-branch cost: 6.79 / 6.84 / 7.00 ms (lanes / eager / hipGraph) at B = 256, 3.7 / 5.2 / 4.0 at
    B = 16, bit-identical to the eager step.  The noise is drawn outside the replayed list (no generator state in it)."""

    def __init__(self, model, diffusion, lr, graph=False, distributed=None, n_buckets=4, overlap_wgrad=None, conditional=False):
        """conditional=True: the step takes class labels (`step(images, y=labels)`, UNet.forward(x, t, y): ddpm_models.py:276-277)
        and `label_emb` is optimised and exchanged like every other parameter.  With the default (the reference's loop,
        ddpm_utils.py:502, never passes labels) `label_emb` stays untouched, as under the reference's AdamW, and passing y raises."""
        self.model, self.diffusion = model, diffusion
        self.conditional = conditional
        # weight-gradient kernels on a second stream (ops._GradMode.side): off the critical path of backward, they fill
        # the CUs the dependent chain of small kernels leaves idle.  Measured on MI355X (B=256): eager 12.0 -> 11.1
        # ms/step, captured graph 11.45 -> 11.3 (forks batched 16 layers at a time: every fork is a cross-stream edge
        # in the graph, and 57 of them cost more than the overlap returns).  Layers per fork, eager: 1 / 2 / 4 / 8 / 16 ->
        # 9.77 / 9.67 / 9.66 / 9.87 / 9.95 ms.  (Stream priorities: see _main_hi below.)
        if overlap_wgrad is None:
            overlap_wgrad = True
        n_side = int(os.environ.get("AFD_WGRAD_STREAMS", 1))                                # side streams (tuning hook)
        prio = int(os.environ.get("AFD_WGRAD_PRIO", 0))                                   # side-stream priority (tuning hook; larger = lower)
        self.wgrad_stream = [torch.cuda.Stream(priority=prio) for _ in range(max(1, n_side))] if overlap_wgrad else None
        if overlap_wgrad and torch.cuda.is_available() and os.environ.get("AFD_WGRAD_INSITU", "0") == "1":
            # opt-in: the 3x3 weight gradients on 160 workgroups per launch instead of one per CU -- 26 % slower alone, but beside the
            # dependent chain they leave CUs to it and write fewer slabs (step 7.14 -> 7.07 ms: csrc/bf3_wgrad.hip).  Off by default so
            # that the step and the per-kernel roofline table of bench.py run ONE plan; process-wide, like every plan switch.
            lib().afd_debug_conv_path(49)
        # layers per fork (AFD_WGRAD_BATCH overrides, read per step: tuning hook).  Round 3, after the convolutions moved to the
        # fp16 matrix pipe (tools/ab_env.py AFD_WGRAD_BATCH, same box): 2 / 4 / 8 / 12 / 16 -> 7.49 / 7.46 / 7.32 / 7.34 / 7.34 ms
        # Re-measured with medians over windows (tools/step_median.py, separate processes): eager 4 / 6 / 8 / 10 / 12 / 16 -> 7.264 / 7.109 /
        # 7.097 / 7.133 / 7.147 / 7.182; captured 8 / 16 / 24 / 32 / 40 / 48 / 64 / 128 -> 7.343 / 7.285 / 7.220 / 7.208 / 7.299 / 7.375 / 7.361 / 7.552
        self.wgrad_batch = 8 if (not graph or graph == "lanes") else 32      # ("lanes" has the eager step's stream semantics)
        self.opt = FusedAdamW(model, lr=lr, conditional=conditional)
        want_ddp = distributed if distributed is not None else dist.is_initialized()
        self.ddp = GradAllReduce(self.opt.fp, n_buckets, model=model) if want_ddp else None
        if self.ddp is not None:
            self.ddp.side = self.wgrad_stream
        self._loss_work = None
        # Tuning hook: the dependent chain (forward, dgrads, norms, attention, AdamW) on a HIGH-priority stream, the weight gradients on a
        # normal one -- the idea being that the dispatcher hands free CUs to the chain first.  It does not pay.
        # Measured in separate processes (tools/step_median.py): 7.22 against 7.16 eager, 12.3 against 7.13 with the two-lane replay, 13.3 with a
        # graph captured on it; the weight-gradient stream at high priority instead: 14.0 -- mixed priorities make every cross-stream
        # wait expensive.  Off by default (AFD_MAIN_PRIO=1 turns it on for eager launches).
        self._main_hi = torch.cuda.Stream(priority=-1) if int(os.environ.get("AFD_MAIN_PRIO", 0)) and torch.cuda.is_available() else None
        self.use_graph = bool(graph)
        self.lanes = graph == "lanes"     # the captured step re-issued on two real streams from C++ (csrc/replay.hip)
        self._lanes_handle = None
        self._graph = None
        self._static = None
        self._wino_plan, self._wino_requests = None, None      # ops.WinoStepPlan after the first (recording) step

    def __del__(self):
        h = getattr(self, "_lanes_handle", None)           # the replay list points into the captured graph: free it first
        if h is not None:
            try:
                lib().afd_replay_free(h)
            except Exception:
                pass
            self._lanes_handle = None

    def _fwd_bwd(self, images, t, eps, y=None):
        W = ops._WinoWeights
        if self._wino_plan is not None and self._wino_plan.valid():
            self._wino_plan.launch()                   # every transformed-weight image of the step, one launch
            W.active_plan = self._wino_plan
        elif self._wino_requests is None:
            self._wino_requests = W.recording = {}     # first step: note which images the dispatch asks for
        try:
            x_t, noise = self.diffusion.noise_images(images, t, eps)
            pred = self.model(x_t, t) if y is None else self.model(x_t, t, y)
            loss = ops.mse_loss(noise, pred)
            self.opt.zero_grad()
            overlap = self.ddp is not None and self.ddp.world > 1 and not self.use_graph      # (a captured backward cannot hold the exchange)
            if overlap:
                self.ddp.begin_step()                  # bucket all-reduces start during backward, as their gradients complete
            with ops.inplace_param_grads(self.wgrad_stream, int(os.environ.get("AFD_WGRAD_BATCH", self.wgrad_batch)),   # weight gradients add straight into the flat .grad views
                                         on_write=self.ddp.wrote if overlap else None,
                                         fold_hint=self.ddp.would_complete if overlap else None):
                # backward on the calling thread instead of the autograd engine's device worker thread: no hand-over of the
                # interpreter lock per node (host time of a step 6.34 -> 5.6 ms, tools/step_median.py at B = 8; AFD_BWD_THREAD=1: the engine's thread)
                if os.environ.get("AFD_BWD_THREAD", "0") == "1":
                    loss.backward()
                else:
                    with torch.autograd.set_multithreading_enabled(False):
                        loss.backward()
                if overlap:
                    self.ddp.backward_done()
        finally:
            if W.recording is not None and W.recording is self._wino_requests:
                W.recording = None
                self._wino_plan = ops.WinoStepPlan(self._wino_requests)
            W.active_plan = None
        return loss.detach()

    def _update(self):
        scale = self.ddp.finish() if self.ddp is not None else 1.0
        self.opt.step(grad_scale=scale)

    def _body(self, images, t, eps, y=None):
        loss = self._fwd_bwd(images, t, eps, y)
        self._update()
        return loss

    def __call__(self, images, t=None, eps=None, y=None):
        """images (B,C,S,S) on the device; t (B,) int64 [default: diffusion.sample_timesteps];
        eps: injected noise or None (device RNG); y (B,) int64 class labels (only with conditional=True, eager launches).
        Returns the loss as a 0-d device tensor."""
        if y is not None and not self.conditional:
            raise ValueError("TrainStep: class labels were passed but the step was built with conditional=False: label_emb sits "
                             "outside the optimised range (FlatParams) and would never be updated; build TrainStep(..., conditional=True)")
        if y is not None and self.use_graph:
            raise ValueError("TrainStep(graph=True) does not take class labels; use eager launches for conditional training")
        if t is None:
            t = self.diffusion.sample_timesteps(images.shape[0])
        t = t.to(images.device, non_blocking=True)
        if self.lanes and eps is None:
            eps = torch.randn_like(images)          # drawn outside: the replayed list holds no generator state
        if not self.use_graph:
            if self._main_hi is not None:             # the dependent chain on the high-priority stream, joined with the caller's on both sides
                cur = torch.cuda.current_stream()
                self._main_hi.wait_stream(cur)
                with torch.cuda.stream(self._main_hi):
                    loss = self._body(images, t, eps, y)
                cur.wait_stream(self._main_hi)
                return loss
            return self._body(images, t, eps, y)
        whole = self.ddp is None                    # single GPU: AdamW is captured too
        if self._graph is None:
            self._static = {"images": images.clone(), "t": t.clone(), "eps": None if eps is None else eps.clone()}
            st = self._static
            # warm-up outside capture (allocator, lazy init, the Winograd plan's recording step).  These are real steps on
            # the first batch, so everything they change is put back afterwards -- parameters, AdamW moments and step
            # counter, the device generator -- and under data parallel they stop before the exchange: the first graph
            # call is then exactly one step, like the eager one (and like the reference's).
            fp, opt = self.opt.fp, self.opt
            keep = [b.clone() for b in (fp.flat, opt.m, opt.v, opt.state)]
            rng = torch.cuda.get_rng_state(images.device)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    (self._body if whole else self._fwd_bwd)(st["images"], st["t"], st["eps"])
            torch.cuda.current_stream().wait_stream(s)
            for b, k in zip((fp.flat, opt.m, opt.v, opt.state), keep):
                b.copy_(k)
            torch.cuda.set_rng_state(rng, images.device)
            ops.bump_param_epoch()
            self._graph = torch.cuda.CUDAGraph(keep_graph=True) if self.lanes else torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                st["loss"] = (self._body if whole else self._fwd_bwd)(st["images"], st["t"], st["eps"])
            if self.lanes:
                import ctypes
                h, counts = ctypes.c_void_p(), (ctypes.c_int * 4)()
                lib().afd_replay_build(self._graph.raw_cuda_graph(), ctypes.byref(h), counts)
                self._lanes_handle, self.lanes_counts = h, tuple(counts)      # (work nodes, main lane, side lane, cross-lane waits)
        st = self._static
        if (eps is None) != (st["eps"] is None):
            raise ValueError("TrainStep(graph=True): the step was captured " + ("without" if st["eps"] is None else "with") +
                             " injected noise; `eps` must be passed (or omitted) on every call alike")
        st["images"].copy_(images)
        st["t"].copy_(t)
        if eps is not None:
            st["eps"].copy_(eps)
        if self.lanes:
            lib().afd_replay_run(self._lanes_handle, torch.cuda.current_stream().cuda_stream, self.wgrad_stream[0].cuda_stream)
        else:
            self._graph.replay()
        ops.bump_param_epoch()                          # the replayed AdamW moved the parameters
        if not whole:
            self.ddp.begin_step()                       # the replayed backward reported nothing: exchange everything now
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
