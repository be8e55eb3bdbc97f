"""torch.autograd.Function shells around the C ABI (include/afd.h).

torch is used for device memory, the current HIP stream and the autograd graph -- every FLOP and
every byte moved on the device goes through libafd_hip.so.  All tensors are fp32, NCHW,
contiguous, on a HIP device; anything else raises (there is no CPU / eager fallback).
"""
import weakref

import numpy as np
import os
import torch

from ._lib import AfdError, lib

GN_EPS = 1e-5
LN_EPS = 1e-5


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _stream():
    """Raw handle of torch's current HIP stream.  torch.cuda.current_stream() builds a Stream object and walks the
    device-index helpers on every call: 2.5 ms of host time per train step over ~470 launches (tools/host_profile.py);
    the raw getter is one C call."""
    if _raw_stream is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise AfdError("afdm: the HIP engine needs tensors on a HIP device (got a CPU tensor); "
                           "there is no CPU fallback -- move the model and its inputs to 'cuda'")
        if t.dtype != torch.float32:
            raise AfdError(f"afdm: fp32 only (got {t.dtype})")


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


class Taps:
    """Host copy of an N x N filter: the kernels take taps as launch arguments, so there is no
    per-call H2D copy (the reference does one per call: filtrs.py:73,91)."""

    def __init__(self, k):
        a = k.detach().cpu().numpy() if isinstance(k, torch.Tensor) else np.asarray(k)
        self.arr = np.ascontiguousarray(a, dtype=np.float32)
        assert self.arr.ndim == 2 and self.arr.shape[0] == self.arr.shape[1]
        self.N = int(self.arr.shape[0])
        self.ptr = self.arr.ctypes.data

    @staticmethod
    def of(k):
        return k if isinstance(k, Taps) else Taps(k)


# ---------------------------------------------------------------------------------------------
# F2 / F3 filtered resampling
# ---------------------------------------------------------------------------------------------

class _NoCtx:
    """Stand-in for the autograd context when gradients are off (sampling): forward() may set attributes and call
    save_for_backward; nothing is kept."""
    needs_input_grad = (False,) * 16

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def mark_dirty(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


class _Fn(torch.autograd.Function):
    """autograd.Function whose apply() skips the autograd machinery under no_grad: Function.apply costs ~8 us of host
    time per call, and with four sampling trajectories in flight (4 x ~130 calls per round) the host is the bound."""

    @classmethod
    def apply(cls, *args):
        if torch.is_grad_enabled():
            return super().apply(*args)
        return cls.forward(_NoCtx(), *args)

class FiltUp2(_Fn):
    @staticmethod
    def forward(ctx, x, taps):
        _chk(x)
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty(B, C, 2 * H, 2 * W, device=x.device, dtype=torch.float32)
        lib().afd_filt_up2_fwd(_p(x), _p(y), B, C, H, W, 0, 0, taps.ptr, taps.N, _stream())
        ctx.taps, ctx.shape = taps, (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = _c(dy)
        dx = torch.empty(B, C, H, W, device=dy.device, dtype=torch.float32)
        lib().afd_filt_up2_bwd(_p(dy), _p(dx), B, C, H, W, 0, 0, ctx.taps.ptr, ctx.taps.N, _stream())
        return dx, None


class FiltDown2(_Fn):
    @staticmethod
    def forward(ctx, x, taps):
        _chk(x)
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty(B, C, (H + 1) // 2, (W + 1) // 2, device=x.device, dtype=torch.float32)
        lib().afd_filt_down2_fwd(_p(x), _p(y), B, C, H, W, 0, 0, taps.ptr, taps.N, _stream())
        ctx.taps, ctx.shape = taps, (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = _c(dy)
        dx = torch.empty(B, C, H, W, device=dy.device, dtype=torch.float32)
        lib().afd_filt_down2_bwd(_p(dy), _p(dx), B, C, H, W, 0, 0, ctx.taps.ptr, ctx.taps.N, _stream())
        return dx, None


def _act_ws(B, C, H, W, N, backward, device):
    nbytes = lib().afd_filt_act_workspace_bytes(B, C, H, W, N, backward)
    return torch.empty(nbytes // 4, device=device, dtype=torch.float32) if nbytes else None


class FiltAct(_Fn):
    """y = down2(GELU(up2(x)))   (ddpm_utils.py:123-125) with no fused prologue."""

    @staticmethod
    def forward(ctx, x, tu, td):
        _chk(x)
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        ws = _act_ws(B, C, H, W, tu.N, 0, x.device)
        lib().afd_filt_act_fwd(_p(x), _p(y), B, C, H, W, None, None, None, None, tu.ptr, td.ptr, tu.N, _p(ws), _stream())
        ctx.save_for_backward(x)
        ctx.tu, ctx.td = tu, td
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        dy = _c(dy)
        dx = torch.empty_like(x)
        ws = _act_ws(B, C, H, W, ctx.tu.N, 1, x.device)
        lib().afd_filt_act_bwd(_p(x), _p(dy), _p(dx), B, C, H, W, None, None, None, None,
                               ctx.tu.ptr, ctx.td.ptr, ctx.tu.N, _p(ws), None, _stream())
        return dx, None, None


# ---------------------------------------------------------------------------------------------
# F6 GroupNorm(1, C) with fused epilogues / fused filtered GELU
# ---------------------------------------------------------------------------------------------
class _GradMode:
    """When `inplace` is set (by TrainStep around backward) the weight-gradient kernels ADD straight into
    the parameters' preallocated `.grad` (FlatParams views) and backward returns None for them: this skips
    ~180 tiny autograd accumulate-adds and as many allocations per step.  Off by default, so
    torch.autograd.grad / plain .backward() keep their usual semantics.

    `side` (a HIP stream, set by `inplace_param_grads(side_stream=...)`): the weight-gradient kernels of the
    convolutions are not on the critical path of backward (nothing downstream consumes dW), so they are launched on
    that stream -- forked after dY exists, joined by the caller before the optimizer -- and fill the CUs that the
    dependent chain of small dgrad / norm kernels leaves idle.  Legal under stream capture (fork / join by events)."""
    inplace = False
    side = None           # first side stream (None: no side stream) -- what the ops test to decide whether to defer
    sides = []            # all side streams: queued launches are dealt to them in turn (one in-order stream runs ONE
    #                       weight-gradient kernel at a time; two let a layer's slab reduce overlap the next layer's multiplies)
    turn = 0
    pending = []          # (launch closure, tensors it reads, params): parameter-gradient kernels not launched yet
    launched = []         # the same triples after launch, held until the join (their buffers are in use on the side stream)
    batch = 8             # fork the side stream once per this many layers (few cross-stream edges in a captured graph)
    on_write = None       # callable(list of Parameters): their .grad is complete as far as ENQUEUED work goes -- the
    #                       data-parallel exchange's readiness hook (training.GradAllReduce.wrote)
    fold_hint = None      # callable(list of Parameters) -> bool: would reporting these complete a data-parallel bucket?
    folds = []            # per side stream: [packed afd_fold_desc bytes, ...] of producers already launched there
    fold_writes = []      # params whose .grad is complete once the queued folds have been launched
    flushes = 0           # flushes since the last fold launch
    fold_every = 2        # launch the queued folds every this many flushes (and at the join, and when a bucket completes)


def _wrote(*params):
    """Report parameters whose gradient the kernels enqueued so far (current stream + side stream) have fully written."""
    cb = _GradMode.on_write
    if cb is not None:
        cb([q for q in params if q is not None])


FOLD_DESC = "<QQqqqqii"      # afd_fold_desc: part, dst, n, stride, inner, rstride, splits, accumulate (56 bytes)


def fold_desc(part_ptr, dst_ptr, n, splits, stride=None, inner=None, rstride=1, accumulate=1):
    import struct
    return struct.pack(FOLD_DESC, part_ptr, dst_ptr, n, n if stride is None else stride, n if inner is None else inner, rstride,
                       splits, accumulate)


def fold_now(descs, stream=None):
    """One afd_fold_batched launch for packed descriptors (bytes)."""
    import ctypes
    buf = ctypes.create_string_buffer(descs, len(descs))
    lib().afd_fold_batched(ctypes.addressof(buf), len(descs) // 56, _stream() if stream is None else stream)


def defer_to_side_stream(fn, *keep, writes=()):
    """Queue fn(stream_handle) -- a launch whose result nothing downstream of backward consumes (a parameter gradient) --
    for the side stream.  `keep`: the tensors it reads; they are held until the join, so autograd cannot add into them
    in place and the allocator cannot recycle them while the side stream still reads them.  `writes`: the Parameters
    whose .grad the launch completes (reported to `on_write` once it is launched).
    fn may RETURN packed afd_fold_desc bytes: the deterministic fold that finishes its gradient.  Those are not launched
    per layer: they queue up on the producer's stream and ONE afd_fold_batched launch folds every `fold_every` flushes'
    worth of them (csrc/fold.hip; `writes` are then reported when that launch is enqueued).  fn = None with
    keep[0] = descriptor bytes queues a fold whose partials a main-stream kernel has already produced."""
    _GradMode.pending.append((fn, keep, writes))
    if len(_GradMode.pending) >= _GradMode.batch:
        flush_wgrads()


def flush_wgrads(final=False):
    """Launch the queued parameter-gradient kernels on the side stream (one fork for the whole batch), then the queued
    folds if it is time: every `fold_every` flushes, at the join (final) and whenever they would complete a bucket."""
    G = _GradMode
    sides, q = G.sides, G.pending
    if not q and not (final and any(G.folds)):
        return
    cur = torch.cuda.current_stream()
    lanes = [[] for _ in sides]
    for item in q:
        lanes[G.turn % len(sides)].append(item)
        G.turn += 1
    if len(G.folds) != len(sides):
        G.folds = [[] for _ in sides]
    done = []
    for k, (side, items) in enumerate(zip(sides, lanes)):
        if not items:
            continue
        side.wait_stream(cur)                                                 # fork: every queued dY (and the zeroed .grad) exists
        with torch.cuda.stream(side):
            h = side.cuda_stream
            for fn, keep, ws in items:
                descs = fn(h) if fn is not None else keep[0]
                if descs and G.fold_every == 0:                               # (A/B hook: one fold launch per producer, as in round 2)
                    fold_now(descs, h)
                    done.extend(ws)
                elif descs:
                    G.folds[k].append(descs)
                    G.fold_writes.extend(ws)
                else:
                    done.extend(ws)
    G.launched.extend(q)
    G.pending = []
    G.flushes += 1
    if any(G.folds) and (final or G.flushes >= max(1, G.fold_every) or (G.fold_hint is not None and G.fold_hint(G.fold_writes + done))):
        for k, side in enumerate(sides):
            if G.folds[k]:
                fold_now(b"".join(G.folds[k]), side.cuda_stream)
                G.folds[k] = []
        done.extend(G.fold_writes)
        G.fold_writes, G.flushes = [], 0
    if G.on_write is not None and done:
        _wrote(*done)


class inplace_param_grads:
    def __init__(self, side_stream=None, batch=8, on_write=None, fold_hint=None, fold_every=None):
        self.side_stream, self.batch, self.on_write, self.fold_hint = side_stream, batch, on_write, fold_hint
        self.fold_every = int(os.environ.get("AFD_FOLD_EVERY", 2)) if fold_every is None else fold_every

    def __enter__(self):
        G = _GradMode
        self.prev = (G.inplace, G.side, G.sides, G.on_write, G.fold_hint, G.fold_every)
        ss = self.side_stream
        self.streams = [] if ss is None else (list(ss) if isinstance(ss, (list, tuple)) else [ss])
        G.inplace, G.batch, G.on_write, G.fold_hint, G.fold_every = True, self.batch, self.on_write, self.fold_hint, max(0, self.fold_every)
        G.sides, G.side, G.turn = self.streams, (self.streams[0] if self.streams else None), 0
        G.folds, G.fold_writes, G.flushes = [[] for _ in self.streams], [], 0

    def __exit__(self, exc_type, *a):
        G = _GradMode
        if self.streams:
            if exc_type is not None:
                G.pending, G.folds, G.fold_writes = [], [[] for _ in self.streams], []      # backward failed: drop what was queued
            flush_wgrads(final=True)
            for st in self.streams:
                torch.cuda.current_stream().wait_stream(st)                  # join: every dW is in .grad
            G.launched = []                                                  # buffers may be freed now (main-stream order)
        G.inplace, G.side, G.sides, G.on_write, G.fold_hint, G.fold_every = self.prev


def _direct(*params):
    """True if every given parameter has a contiguous preallocated .grad we may accumulate into."""
    if not _GradMode.inplace:
        return False
    for q in params:
        if q is None:
            continue
        g = q.grad
        if g is None or not g.is_contiguous() or g.dtype != torch.float32:
            return False
    return True


def _gn_param_targets(C, gamma, beta, device):
    """Where a normalisation backward's tail writes dgamma / dbeta: straight into the parameters' .grad (in-place
    mode, accumulate = 1) or into a fresh (2, C) pair returned to autograd.  -> (dg, db, accumulate, returned pair)"""
    if gamma is not None and beta is not None and _direct(gamma, beta):
        return gamma.grad, beta.grad, 1, (None, None)
    out = torch.empty(2, C, device=device, dtype=torch.float32)
    return out[0], out[1], 0, (out[0], out[1])


class GroupNorm1(_Fn):
    """y = act(GroupNorm(1,C)(x)*gamma + beta + res) + emb[b,c]      (one HBM round trip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, emb, act):
        _chk(x, gamma, beta, res, emb)
        x, res, emb = _c(x), _c(res), _c(emb)
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(B, 2, device=x.device, dtype=torch.float32)
        lib().afd_groupnorm1_fwd(_p(x), _p(y), _p(stats), B, C, H * W, GN_EPS, _p(gamma), _p(beta), _p(res), act,
                                 _p(emb), _stream())
        ctx.save_for_backward(x, gamma, beta, res, stats)
        ctx.act, ctx.has_emb = act, emb is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, res, stats = ctx.saved_tensors
        B, C, H, W = x.shape
        dy = _c(dy)
        dx = torch.empty_like(x)
        dres = None
        if res is not None:
            dres = dy if ctx.act == 0 else torch.empty_like(x)      # act == 0: d(res) is dy itself
        part = torch.empty(B * C * 2, device=x.device, dtype=torch.float32)
        demb = torch.empty(B, C, device=x.device, dtype=torch.float32) if ctx.has_emb else None
        dg, db, acc, (dgamma, dbeta) = _gn_param_targets(C, gamma, beta, x.device)
        defer = acc and _GradMode.side is not None          # the two column sums of part (B, 2, C): folded with the other layers'
        lib().afd_groupnorm1_bwd(_p(x), _p(dy), _p(stats), B, C, H * W, _p(gamma), _p(beta), _p(res), ctx.act,
                                 _p(dx), _p(dres) if (res is not None and ctx.act == 1) else None, _p(part), _p(demb), 0,
                                 None if defer else _p(dg), None if defer else _p(db), acc, _stream())
        if defer:
            pp = _p(part)
            defer_to_side_stream(None, fold_desc(pp, _p(dg), C, B, stride=2 * C) + fold_desc(pp + 4 * C, _p(db), C, B, stride=2 * C),
                                 part, writes=(gamma, beta))
        elif acc:
            _wrote(gamma, beta)
        return dx, dgamma, dbeta, dres, demb, None


class GroupNormFiltAct(_Fn):
    """y = down2(GELU(up2(GroupNorm(1,C)(x)*gamma + beta + res)))    (ddpm_utils.py:122-125,127-131).

    Forward: a stats-only pass over x (4 B/elem) then the fused kernel applies the normalisation
    and the residual inside its load (8 B/elem [+4 with res]); the normalised tensor and the 2x
    intermediate never exist in memory."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, tu, td):
        _chk(x, gamma, beta, res)
        x, res = _c(x), _c(res)
        B, C, H, W = x.shape
        stats = torch.empty(B, 2, device=x.device, dtype=torch.float32)
        L = lib()
        y = torch.empty_like(x)
        if L.afd_filt_act_fwd_gn_supported(C, H, W, tu.N):
            # small samples: one workgroup holds a whole sample and computes the statistics itself (no statistics launch)
            L.afd_filt_act_fwd_gn(_p(x), _p(y), B, C, H, W, GN_EPS, _p(stats), _p(gamma), _p(beta), _p(res), tu.ptr, td.ptr, tu.N,
                                  _stream())
        else:
            L.afd_groupnorm1_fwd(_p(x), None, _p(stats), B, C, H * W, GN_EPS, None, None, None, 0, None, _stream())
            ws = _act_ws(B, C, H, W, tu.N, 0, x.device)
            L.afd_filt_act_fwd(_p(x), _p(y), B, C, H, W, _p(stats), _p(gamma), _p(beta), _p(res), tu.ptr, td.ptr, tu.N,
                               _p(ws), _stream())
        ctx.save_for_backward(x, gamma, beta, res, stats)
        ctx.tu, ctx.td = tu, td
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, res, stats = ctx.saved_tensors
        B, C, H, W = x.shape
        dy = _c(dy)
        L = lib()
        if L.afd_filt_act_fwd_gn_supported(C, H, W, ctx.tu.N):
            # small samples: one workgroup holds a sample, the same launch finishes GroupNorm's backward (dx leaves instead of dv)
            dx = torch.empty_like(x)
            dres = torch.empty_like(x) if res is not None else None
            part = torch.empty(B * C * 2, device=x.device, dtype=torch.float32)
            L.afd_filt_act_bwd_gn(_p(x), _p(dy), _p(dx), _p(dres), B, C, H, W, _p(stats), _p(gamma), _p(beta), _p(res),
                                  ctx.tu.ptr, ctx.td.ptr, ctx.tu.N, _p(part), _stream())
            dg, db, acc, (dgamma, dbeta) = _gn_param_targets(C, gamma, beta, x.device)
            if acc and _GradMode.side is not None:                     # the two column sums: folds, batched with the other layers'
                pp = _p(part)                                          # part is (B, 2, C)
                defer_to_side_stream(None, fold_desc(pp, _p(dg), C, B, stride=2 * C) + fold_desc(pp + 4 * C, _p(db), C, B, stride=2 * C),
                                     part, writes=(gamma, beta))
            else:
                L.afd_colsum2(_p(part), _p(dg), _p(db), B, C, acc, _stream())
                if acc:
                    _wrote(gamma, beta)
            return dx, dgamma, dbeta, dres, None, None
        dv = torch.empty_like(x)
        ws = _act_ws(B, C, H, W, ctx.tu.N, 1, x.device)
        part = torch.empty(B * C * 2, device=x.device, dtype=torch.float32)
        fused = ws is None                 # fast path: the filtered-GELU backward also emits the GroupNorm plane sums
        L.afd_filt_act_bwd(_p(x), _p(dy), _p(dv), B, C, H, W, _p(stats), _p(gamma), _p(beta), _p(res),
                           ctx.tu.ptr, ctx.td.ptr, ctx.tu.N, _p(ws), _p(part) if fused else None, _stream())
        dx = torch.empty_like(x)
        dg, db, acc, (dgamma, dbeta) = _gn_param_targets(C, gamma, beta, x.device)
        L.afd_groupnorm1_bwd(_p(x), _p(dv), _p(stats), B, C, H * W, _p(gamma), _p(beta), None, 0,
                             _p(dx), None, _p(part), None, 1 if fused else 0, _p(dg), _p(db), acc, _stream())
        if acc:
            _wrote(gamma, beta)
        return dx, dgamma, dbeta, (dv if res is not None else None), None, None


# ---------------------------------------------------------------------------------------------
# F5 / F10 convolution (3x3 pad 1, 1x1)
# ---------------------------------------------------------------------------------------------
def _kind_of(kinds, dgrad):
    """Form of one pass's weight image in an afd_conv3x3_weight_kinds value: 0 = Winograd, 1 = direct f16x2, 5 = direct bf16x3."""
    direct = kinds >> dgrad & 1
    return direct | (kinds & 4 if direct else 0)


class _WinoWeights:
    """Transformed 3x3 weights (G g G^T, forward and dgrad forms) kept per weight tensor while the weights do not
    change: keyed by the tensor object AND the HIP stream that asks (the transform launch and every kernel reading the
    image are then ordered on that one stream: trajectories of `Diffusion.sample_concurrent` running on other streams
    fill and read their own copies, and an evicted buffer returns to the pool of the only stream that used it),
    stamped with its autograd version and a global epoch that every raw in-place update of the parameters
    (FusedAdamW.step) bumps.  Never used under stream capture: a captured step must contain its own transform
    launches, because a replay runs after the weights moved."""
    epoch = 0
    store = {}
    max_entries = 4096
    recording = None          # dict filled with the requests of one step (TrainStep's first step), see WinoStepPlan
    active_plan = None        # WinoStepPlan whose buffers were filled at the start of the current step

    @classmethod
    def get(cls, w, nbytes, dgrad, kinds=0):
        plan = cls.active_plan
        if plan is not None:
            u = plan.lookup(w, nbytes, dgrad, kinds)
            if u is not None:
                return u, 1
        if cls.recording is not None:
            rec = cls.recording.setdefault(id(w), [w, 0, 0, 0])
            rec[1 + dgrad] = nbytes
            rec[3] = kinds
        capturing = torch.cuda.is_current_stream_capturing()
        key = (id(w), dgrad, _stream())
        stamp = (w._version, cls.epoch, w.data_ptr(), nbytes, _kind_of(kinds, dgrad))   # (the image's form follows the batch size)
        if not capturing:
            hit = cls.store.get(key)
            if hit is not None and hit[0]() is w and hit[2] == stamp:
                return hit[1], 1
        u = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
        if not capturing:
            if len(cls.store) > cls.max_entries:
                cls.store.clear()
            try:
                cls.store[key] = (weakref.ref(w), u, stamp)
            except TypeError:
                pass
        return u, 0


class WinoStepPlan:
    """Every transformed-weight image a train step needs, produced by ONE launch at the start of the step
    (afd_conv3x3_wino_weights_batched) instead of one launch per layer on the critical path.  Built from the requests
    recorded during a first step ({id(w): [w, forward bytes, dgrad bytes]}); valid while the weights keep their
    addresses (FlatParams views do)."""

    def __init__(self, requests):
        import struct
        reqs = [r for r in requests.values() if r[1] or r[2]]
        self.ok = bool(reqs)
        if not self.ok:
            return
        dev = reqs[0][0].device
        self.buf = torch.empty(sum(r[1] + r[2] for r in reqs) // 4, device=dev, dtype=torch.float32)
        self.map, descs, wg, off = {}, b"", [], 0
        for i, (w, nf, nd, kinds) in enumerate(reqs):
            Cout, Cin = w.shape[0], w.shape[1]
            uf = self.buf[off:off + nf // 4] if nf else None
            off += nf // 4
            ud = self.buf[off:off + nd // 4] if nd else None
            off += nd // 4
            self.map[id(w)] = (weakref.ref(w), w.data_ptr(), uf, ud, kinds)
            n_wg = (Cin * Cout // 64 + 3) // 4
            descs += struct.pack("<QQQiiii", w.data_ptr(), uf.data_ptr() if nf else 0, ud.data_ptr() if nd else 0, Cin, Cout, len(wg), kinds)
            wg += [i] * n_wg
        self.n_wg = len(wg)
        self.descs = torch.frombuffer(bytearray(descs), dtype=torch.uint8).to(dev)
        self.wg = torch.tensor(wg, dtype=torch.int32).to(dev)

    def valid(self):
        return self.ok and all(r() is not None and r().data_ptr() == ptr for (r, ptr, _, _, _) in self.map.values())

    def launch(self):
        lib().afd_conv3x3_wino_weights_batched(_p(self.descs), _p(self.wg), self.n_wg, _stream())

    def lookup(self, w, nbytes, dgrad, kinds=0):
        e = self.map.get(id(w))
        if e is None or e[0]() is not w or _kind_of(e[4], dgrad) != _kind_of(kinds, dgrad):   # (another batch size may want the other form)
            return None
        u = e[3] if dgrad else e[2]
        return u if (u is not None and u.numel() * 4 == nbytes) else None


def bump_param_epoch():
    """Call after parameters were modified through raw device pointers (the fused optimizer).  Every cached weight image is
    stale from here on: the store is emptied (each buffer goes back to the pool of the one stream that used it), so a
    sampling run after training does not pin images of streams that no longer exist."""
    _WinoWeights.epoch += 1
    if _WinoWeights.store:
        _WinoWeights.store.clear()


def _conv_fwd(x, w, bias, res, y, B, Cin, Cout, H, W, ks, act, want_dgrad=False):
    """3x3 layers the Winograd kernel covers go there (the library says which: a non-zero workspace size);
    everything else takes the direct implicit-GEMM kernels.  With want_dgrad the transformed weights of the
    backward pass are produced by the same launch as the forward ones and returned for Conv.backward."""
    L = lib()
    nb = L.afd_conv3x3_wino_workspace_bytes(B, Cin, Cout, H, W, 0) if ks == 3 else 0
    nd = L.afd_conv3x3_wino_workspace_bytes(B, Cin, Cout, H, W, 1) if (ks == 3 and want_dgrad) else 0
    ud = None
    if nb:
        kinds = L.afd_conv3x3_weight_kinds(B, Cin, Cout, H, W)     # which passes run the direct bf16x3 form (their image differs)
        u, ready = _WinoWeights.get(w, nb, 0, kinds)
        if nd:
            ud, dready = _WinoWeights.get(w, nd, 1, kinds)
            if not (ready and dready):
                L.afd_conv3x3_wino_weights(_p(w), _p(u), _p(ud), Cin, Cout, kinds, _stream())
                ready = 1
        L.afd_conv3x3_wino_fwd(_p(x), _p(w), _p(bias), _p(res), _p(y), B, Cin, Cout, H, W, act, _p(u), ready, kinds, _stream())
    else:
        L.afd_conv_fwd(_p(x), _p(w), _p(bias), _p(res), _p(y), B, Cin, Cout, H, W, ks, act, _stream())
    return (ud, kinds) if ud is not None else None


class Conv(_Fn):
    """y = conv(x, w) + bias + res.  (The GELU epilogue is only used by `conv_infer`.)

    fork=True returns (y, x again): the residual branch of a block that STARTS with this convolution takes x from
    here, so backward receives both gradients of x and the dgrad kernel's epilogue adds them (no autograd add)."""

    @staticmethod
    def forward(ctx, x, w, bias, res, w_param=None, b_param=None, fork=False):
        # w may be a reshaped VIEW of a parameter (Linear weights used as 1x1 kernels); w_param / b_param are
        # the leaf Parameters whose .grad the in-place mode accumulates into
        _chk(x, w, bias, res)
        x, w, res = _c(x), _c(w), _c(res)
        B, Cin, H, W = x.shape
        Cout, ks = w.shape[0], w.shape[-1]
        y = torch.empty(B, Cout, H, W, device=x.device, dtype=torch.float32)
        ctx.u_dgrad = _conv_fwd(x, w, bias, res, y, B, Cin, Cout, H, W, ks, 0, want_dgrad=x.requires_grad)
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.has_res = bias is not None, res is not None
        ctx.w_param, ctx.b_param = w_param, b_param
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, dy, dfork=None):
        x, w = ctx.saved_tensors
        B, Cin, H, W = x.shape
        Cout, ks = w.shape[0], w.shape[-1]
        dy = _c(dy)
        L = lib()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            dfork = _c(dfork)
            nb = L.afd_conv3x3_wino_workspace_bytes(B, Cin, Cout, H, W, 1) if ks == 3 else 0
            if nb:
                if ctx.u_dgrad is not None:            # produced together with the forward image
                    (u, kinds), ready = ctx.u_dgrad, 1
                else:
                    kinds = L.afd_conv3x3_weight_kinds(B, Cin, Cout, H, W)
                    u, ready = _WinoWeights.get(w, nb, 1, kinds)
                L.afd_conv3x3_wino_dgrad(_p(dy), _p(w), _p(dx), _p(dfork), B, Cin, Cout, H, W, _p(u), ready, kinds, _stream())
            else:
                L.afd_conv_dgrad(_p(dy), _p(w), _p(dx), B, Cin, Cout, H, W, ks, _stream())
                if dfork is not None:                      # (the direct kernels have no add-to-dx epilogue)
                    L.afd_add(_p(dx), _p(dfork), _p(dx), dx.numel(), _stream())
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            nbytes = L.afd_conv_wgrad_workspace_bytes(B, Cin, Cout, H, W, ks)
            ws = torch.empty(max(nbytes // 4, 1), device=x.device, dtype=torch.float32)
            wp, bp = ctx.w_param, ctx.b_param
            if _direct(wp, bp) and wp.grad.numel() == w.numel():
                side = _GradMode.side
                if side is None:
                    L.afd_conv_wgrad(_p(x), _p(dy), _p(wp.grad), _p(bp.grad) if bp is not None else None,
                                     B, Cin, Cout, H, W, ks, 1, _p(ws), _stream())
                    _wrote(wp, bp)
                else:
                    dwp, dbp = _p(wp.grad), (_p(bp.grad) if bp is not None else None)
                    defer_to_side_stream(lambda st, x=x, dy=dy, ws=ws: _wgrad_partials(x, dy, dwp, dbp, B, Cin, Cout, H, W, ks, ws, st),
                                         x, dy, ws, writes=(wp, bp))
            else:
                dw = torch.empty_like(w)
                db = torch.empty(Cout, device=x.device, dtype=torch.float32) if ctx.has_bias else None
                L.afd_conv_wgrad(_p(x), _p(dy), _p(dw), _p(db), B, Cin, Cout, H, W, ks, 0, _p(ws), _stream())
        return dx, dw, db, (dy if ctx.has_res else None), None, None, None


def conv(x, w, bias=None, res=None, w_param=None, b_param=None, fork=False):
    """w_param / b_param: the leaf Parameters behind `w` / `bias` (enables in-place grad accumulation).
    fork: also return x (for the residual of the block this convolution opens), see Conv."""
    if w_param is None and isinstance(w, torch.nn.Parameter):
        w_param = w
    if b_param is None and isinstance(bias, torch.nn.Parameter):
        b_param = bias
    return Conv.apply(x, w, bias, res, w_param, b_param, fork)


def conv_infer(x, w, bias=None, res=None, act=0):
    """No-grad convolution with the fused GELU epilogue."""
    _chk(x, w, bias, res)
    x, w, res = _c(x), _c(w), _c(res)
    B, Cin, H, W = x.shape
    Cout, ks = w.shape[0], w.shape[-1]
    y = torch.empty(B, Cout, H, W, device=x.device, dtype=torch.float32)
    _conv_fwd(x, w, bias, res, y, B, Cin, Cout, H, W, ks, act)
    return y


# ---------------------------------------------------------------------------------------------
# F10 attention block pieces
# ---------------------------------------------------------------------------------------------
class LayerNormC(_Fn):
    """nn.LayerNorm([C]) applied to the (B, L, C) token view of an NCHW tensor, without the transposes.

    Returns (y, x_res): x_res is x again, for the residual branch that every LayerNorm of the attention block sits
    beside (y -> ... -> + x).  Routing the residual through this node means backward receives BOTH gradients of x
    and the kernel writes LN'(dy) + d(x_res) in one pass, instead of autograd launching a separate add."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        _chk(x, gamma, beta)
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(B, H * W, 2, device=x.device, dtype=torch.float32)
        lib().afd_layernorm_c_fwd(_p(x), _p(y), _p(stats), B, C, H * W, LN_EPS, _p(gamma), _p(beta), _stream())
        ctx.save_for_backward(x, gamma, stats)
        ctx.beta_param = beta
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, gamma, stats = ctx.saved_tensors
        B, C, H, W = x.shape
        if dy is None:                                   # only the residual output was used
            return dres, None, None
        dy, dres = _c(dy), _c(dres)
        dx = torch.empty_like(x)
        L = lib()
        if _direct(gamma, ctx.beta_param):
            # in-place mode: dx now; dgamma / dbeta (plane sums + fold) as a separate call, on the side stream when
            # there is one (the same arithmetic either way: the two-stream step stays bit-identical)
            L.afd_layernorm_c_bwd(_p(x), _p(dy), _p(stats), B, C, H * W, _p(gamma), _p(dx), _p(dres), None, None, None, 0, _stream())
            _ln_param_grads(x, dy, stats, gamma, ctx.beta_param, B, C, H * W)
            return dx, None, None
        part = torch.empty(B, 2, C, device=x.device, dtype=torch.float32)
        dg, db, acc, (dgamma, dbeta) = _gn_param_targets(C, gamma, ctx.beta_param, x.device)
        L.afd_layernorm_c_bwd(_p(x), _p(dy), _p(stats), B, C, H * W, _p(gamma), _p(dx), _p(dres), _p(part), _p(dg), _p(db), acc,
                              _stream())
        return dx, dgamma, dbeta


def layernorm_c(x, gamma, beta):
    """y only (no residual routed through the node)."""
    return LayerNormC.apply(x, gamma, beta)[0]


class Attention(_Fn):
    """softmax(QK^T/sqrt(d))V on qkv (B, 3C, H, W) -> (B, C, H, W); scores never materialised."""

    @staticmethod
    def forward(ctx, qkv, heads):
        _chk(qkv)
        qkv = _c(qkv)
        B, C3, H, W = qkv.shape
        C, L = C3 // 3, H * W
        o = torch.empty(B, C, H, W, device=qkv.device, dtype=torch.float32)
        lse = torch.empty(B, heads, L, device=qkv.device, dtype=torch.float32)
        lib().afd_attn_fwd(_p(qkv), _p(o), _p(lse), B, heads, C // heads, L, _stream())
        ctx.save_for_backward(qkv, o, lse)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        B, C3, H, W = qkv.shape
        C, L = C3 // 3, H * W
        do = _c(do)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        lib().afd_attn_bwd(_p(qkv), _p(o), _p(do), _p(lse), _p(dqkv), _p(delta), B, ctx.heads, C // ctx.heads, L, _stream())
        return dqkv, None


def tok_supported(C):
    """True if the fused token-chain kernels (csrc/tok.hip) cover channel width C."""
    return bool(lib().afd_tok_supported(int(C)))


def _wgrad_partials(x, dy, dwp, dbp, B, Cin, Cout, H, W, ks, ws, st):
    """afd_conv_wgrad_partials: the slab producer now, its fold descriptor(s) returned for the batched fold."""
    import ctypes
    buf = ctypes.create_string_buffer(112)
    n = ctypes.c_int(0)
    lib().afd_conv_wgrad_partials(_p(x), _p(dy), dwp, dbp, B, Cin, Cout, H, W, ks, 1, _p(ws), ctypes.addressof(buf), ctypes.addressof(n), st)
    return buf.raw[:56 * n.value]


def _linear_grads(x, dy, w, b, B, C_in, C_out, H, W, need_w=True):
    """Weight / bias gradients of a token-wise Linear (= 1x1 convolution) y = w x + b from its input x (B,C_in,H,W) and
    dy (B,C_out,H,W).  In-place mode: accumulated straight into w.grad / b.grad, on the side stream when there is one
    (returns (None, None)); otherwise returns fresh (dw, db) for autograd."""
    L = lib()
    nbytes = L.afd_conv_wgrad_workspace_bytes(B, C_in, C_out, H, W, 1)
    ws = torch.empty(max(nbytes // 4, 1), device=x.device, dtype=torch.float32)
    if _direct(w, b):
        dwp, dbp = _p(w.grad), (_p(b.grad) if b is not None else None)
        if _GradMode.side is not None:
            defer_to_side_stream(lambda st, x=x, dy=dy, ws=ws: _wgrad_partials(x, dy, dwp, dbp, B, C_in, C_out, H, W, 1, ws, st),
                                 x, dy, ws, writes=(w, b))
        else:
            L.afd_conv_wgrad(_p(x), _p(dy), dwp, dbp, B, C_in, C_out, H, W, 1, 1, _p(ws), _stream())
            _wrote(w, b)
        return None, None
    dw = torch.empty_like(w)
    db = torch.empty(C_out, device=x.device, dtype=torch.float32) if b is not None else None
    L.afd_conv_wgrad(_p(x), _p(dy), _p(dw), _p(db), B, C_in, C_out, H, W, 1, 0, _p(ws), _stream())
    return dw, db


def _ln_param_grads(x, dy, stats, gamma, beta, B, C, HW):
    """LayerNorm-over-channels parameter gradients from the LayerNorm input x, the gradient at its output dy and the
    saved statistics; same in-place / side-stream convention as _linear_grads."""
    L = lib()
    part = torch.empty(B, 2, C, device=x.device, dtype=torch.float32)
    dg, db, acc, (dgamma, dbeta) = _gn_param_targets(C, gamma, beta, x.device)
    pdg, pdb = _p(dg), _p(db)
    if acc and _GradMode.side is not None:
        def fn(st, x=x, dy=dy, stats=stats, part=part):                  # the plane pass now, the two column sums with the batched fold
            L.afd_layernorm_c_bwd_partials(_p(x), _p(dy), _p(stats), B, C, HW, _p(part), st)
            pp = _p(part)
            return fold_desc(pp, pdg, C, B, stride=2 * C) + fold_desc(pp + 4 * C, pdb, C, B, stride=2 * C)
        defer_to_side_stream(fn, x, dy, stats, part, writes=(gamma, beta))
    else:
        L.afd_layernorm_c_bwd_params(_p(x), _p(dy), _p(stats), B, C, HW, _p(part), pdg, pdb, acc, _stream())
        if acc:
            _wrote(gamma, beta)
    return dgamma, dbeta


class AttnHead(_Fn):
    """qkv = in_proj(LayerNorm(x)) in ONE launch (ddpm_utils.py:70-71, csrc/tok.hip); also returns x again for the residual
    branch, so backward receives both gradients of x and the fused backward kernel adds them."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w_in, b_in):
        _chk(x, gamma, beta, w_in, b_in)
        x = _c(x)
        B, C, H, W = x.shape
        train = not isinstance(ctx, _NoCtx)             # (forward itself always runs with grad mode off)
        qkv = torch.empty(B, 3 * C, H, W, device=x.device, dtype=torch.float32)
        h = torch.empty_like(x) if train else None
        stats = torch.empty(B, H * W, 2, device=x.device, dtype=torch.float32) if train else None
        lib().afd_tok_head_fwd(_p(x), _p(gamma), _p(beta), _p(w_in), _p(b_in), _p(h), _p(stats), _p(qkv), B, C, H * W, LN_EPS, _stream())
        ctx.save_for_backward(x, h, stats, gamma, w_in)
        ctx.params = (beta, b_in)
        return qkv, x.view_as(x)

    @staticmethod
    def backward(ctx, dqkv, dres):
        x, h, stats, gamma, w_in = ctx.saved_tensors
        beta, b_in = ctx.params
        B, C, H, W = x.shape
        dqkv = _c(dqkv)
        if dres is None:
            dres = torch.zeros_like(x)
        dres = _c(dres)
        dx = torch.empty_like(x)
        dh = torch.empty_like(x)
        lib().afd_tok_head_bwd(_p(dqkv), _p(x), _p(stats), _p(gamma), _p(w_in), _p(dres), _p(dh), _p(dx), B, C, H * W, _stream())
        dw, db = _linear_grads(h, dqkv, w_in, b_in, B, C, 3 * C, H, W)
        dgamma, dbeta = _ln_param_grads(x, dh, stats, gamma, beta, B, C, H * W)
        return dx, dgamma, dbeta, dw, db


class AttnTail(_Fn):
    """out = FF(LayerNorm(a)) + a with a = out_proj(att) + x, in ONE launch (ddpm_utils.py:71-73, csrc/tok.hip)."""

    @staticmethod
    def forward(ctx, att, x, wo, bo, gamma, beta, w1, b1, w2, b2):
        _chk(att, x, wo, bo, gamma, beta, w1, b1, w2, b2)
        att, x = _c(att), _c(x)
        B, C, H, W = att.shape
        out = torch.empty_like(att)
        if not isinstance(ctx, _NoCtx):
            a, f, u, g = (torch.empty_like(att) for _ in range(4))
            stats = torch.empty(B, H * W, 2, device=att.device, dtype=torch.float32)
        else:
            a = f = u = g = stats = None
        lib().afd_tok_tail_fwd(_p(att), _p(x), _p(wo), _p(bo), _p(gamma), _p(beta), _p(w1), _p(b1), _p(w2), _p(b2),
                               _p(a), _p(stats), _p(f), _p(u), _p(g), _p(out), B, C, H * W, LN_EPS, _stream())
        ctx.save_for_backward(att, a, stats, f, u, g, gamma, wo, w1, w2)
        ctx.params = (bo, beta, b1, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        att, a, stats, f, u, g, gamma, wo, w1, w2 = ctx.saved_tensors
        bo, beta, b1, b2 = ctx.params
        B, C, H, W = att.shape
        dout = _c(dout)
        du, df, da, datt = (torch.empty_like(att) for _ in range(4))
        lib().afd_tok_tail_bwd(_p(dout), _p(u), _p(a), _p(stats), _p(gamma), _p(w2), _p(w1), _p(wo),
                               _p(du), _p(df), _p(da), _p(datt), B, C, H * W, _stream())
        dw2, db2 = _linear_grads(g, dout, w2, b2, B, C, C, H, W)
        dw1, db1 = _linear_grads(f, du, w1, b1, B, C, C, H, W)
        dwo, dbo = _linear_grads(att, da, wo, bo, B, C, C, H, W)
        dgamma, dbeta = _ln_param_grads(a, df, stats, gamma, beta, B, C, H * W)
        return datt, da, dwo, dbo, dgamma, dbeta, dw1, db1, dw2, db2


class Gelu(_Fn):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x = _c(x)
        y = torch.empty_like(x)
        lib().afd_gelu_fwd(_p(x), _p(y), x.numel(), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        lib().afd_gelu_bwd(_p(x), _p(dy), _p(dx), x.numel(), _stream())
        return dx


# ---------------------------------------------------------------------------------------------
# resampling of variants 0 / 2, concat
# ---------------------------------------------------------------------------------------------
class MaxPool2(_Fn):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty(B, C, H // 2, W // 2, device=x.device, dtype=torch.float32)
        lib().afd_maxpool2_fwd(_p(x), _p(y), B, C, H, W, _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        dy = _c(dy)
        dx = torch.empty_like(x)
        lib().afd_maxpool2_bwd(_p(x), _p(dy), _p(dx), B, C, H, W, _stream())
        return dx


class UpCat(_Fn):
    """cat([skip, up2x(x)], dim=1) in one buffer: the upsampler writes straight into the channel
    slice (ddpm_utils.py:241-242 / :354-355 / :413-414).  mode 'filt' = custom_upsample,
    'bilinear' = nn.Upsample(2, bilinear, align_corners=True)."""

    @staticmethod
    def forward(ctx, x, skip, taps, mode):
        _chk(x, skip)
        x, skip = _c(x), _c(skip)
        B, C, H, W = x.shape
        Cs = skip.shape[1]
        assert skip.shape[2] == 2 * H and skip.shape[3] == 2 * W
        plane = 4 * H * W
        out = torch.empty(B, Cs + C, 2 * H, 2 * W, device=x.device, dtype=torch.float32)
        L = lib()
        up_ptr = out.data_ptr() + 4 * Cs * plane
        bs = (Cs + C) * plane
        if mode == "filt":
            L.afd_filt_up2_fwd(_p(x), up_ptr, B, C, H, W, 0, bs, taps.ptr, taps.N, _stream())
        else:
            L.afd_bilinear_up2_fwd(_p(x), up_ptr, B, C, H, W, bs, _stream())
        L.afd_copy_batched(_p(skip), _p(out), B, Cs * plane, 0, bs, _stream())
        ctx.taps, ctx.mode, ctx.dims = taps, mode, (B, C, Cs, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, Cs, H, W = ctx.dims
        dout = _c(dout)
        plane = 4 * H * W
        bs = (Cs + C) * plane
        L = lib()
        dx = torch.empty(B, C, H, W, device=dout.device, dtype=torch.float32)
        up_ptr = dout.data_ptr() + 4 * Cs * plane
        if ctx.mode == "filt":
            L.afd_filt_up2_bwd(up_ptr, _p(dx), B, C, H, W, bs, 0, ctx.taps.ptr, ctx.taps.N, _stream())
        else:
            L.afd_bilinear_up2_bwd(up_ptr, _p(dx), B, C, H, W, bs, _stream())
        # the skip's gradient is a channel slice of dout: returned as a VIEW.  Every skip tensor has a second consumer (the next
        # encoder stage), so autograd adds the two gradients anyway -- its add reads the slice in place and writes a contiguous
        # sum; the copy that used to materialise the slice first (3 launches, 2 tensor passes per step) is gone (8.54 -> 8.48 ms/step, same box).
        return dx, dout[:, :Cs], None, None


# ---------------------------------------------------------------------------------------------
# F9 time embedding
# ---------------------------------------------------------------------------------------------
def pos_encoding(t, inv_freq):
    """(B,) int64 timesteps -> (B, 2*half) fp32 [sin | cos]   (ddpm_models.py:261-269)."""
    if not t.is_cuda:
        raise AfdError("afdm: timesteps must live on the HIP device")
    t = t.contiguous()
    B, half = t.shape[0], inv_freq.shape[0]
    out = torch.empty(B, 2 * half, device=t.device, dtype=torch.float32)
    lib().afd_pos_encoding(_p(t), _p(inv_freq), _p(out), B, half, _stream())
    return out


class EmbedAdd(_Fn):
    """t_emb + label_emb(y)  (ddpm_models.py:276-277): nn.Embedding lookup and the add in one launch; backward is the
    deterministic row scatter into the table's gradient (the positional encoding itself carries no gradient)."""

    @staticmethod
    def forward(ctx, temb, table, y):
        _chk(temb, table)
        if not y.is_cuda or y.dtype != torch.long:
            raise AfdError("afdm: class labels must be an int64 tensor on the HIP device")
        temb, table, y = _c(temb), _c(table), _c(y)
        B, D = temb.shape
        if table.shape[1] != D or y.shape[0] != B:
            raise AfdError(f"afdm: label embedding shape mismatch (temb {tuple(temb.shape)}, table {tuple(table.shape)}, y {tuple(y.shape)})")
        out = torch.empty_like(temb)
        lib().afd_embed_add_fwd(_p(temb), _p(table), _p(y), _p(out), B, D, table.shape[0], _stream())
        ctx.save_for_backward(y)
        ctx.table = table if isinstance(table, torch.nn.Parameter) else None
        ctx.K = table.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        (y,) = ctx.saved_tensors
        dout = _c(dout)
        B, D = dout.shape
        dtemb = dout if ctx.needs_input_grad[0] else None
        if not ctx.needs_input_grad[1]:
            return dtemb, None, None
        tp = ctx.table
        if tp is not None and _direct(tp):
            lib().afd_embed_add_bwd(_p(dout), _p(y), _p(tp.grad), B, D, ctx.K, 1, _stream())
            _wrote(tp)
            return dtemb, None, None
        dtable = torch.empty(ctx.K, D, device=dout.device, dtype=torch.float32)
        lib().afd_embed_add_bwd(_p(dout), _p(y), _p(dtable), B, D, ctx.K, 0, _stream())
        return dtemb, dtable, None


class SiluLinear(_Fn):
    """emb_layer = nn.Sequential(nn.SiLU(), nn.Linear(emb_dim, C))   (ddpm_utils.py:208-214)."""

    @staticmethod
    def forward(ctx, temb, w, bias):
        _chk(temb, w, bias)
        temb, w = _c(temb), _c(w)
        B, K = temb.shape
        N = w.shape[0]
        out = torch.empty(B, N, device=temb.device, dtype=torch.float32)
        lib().afd_silu_linear_fwd(_p(temb), _p(w), _p(bias), _p(out), B, K, N, _stream())
        ctx.save_for_backward(temb, w)
        ctx.params = (w if isinstance(w, torch.nn.Parameter) else None, bias if isinstance(bias, torch.nn.Parameter) else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        temb, w = ctx.saved_tensors
        B, K = temb.shape
        N = w.shape[0]
        dout = _c(dout)
        dtemb = torch.zeros_like(temb) if ctx.needs_input_grad[0] else None
        wp, bp = ctx.params
        if _direct(wp, bp):
            if dtemb is None and _GradMode.side is not None:      # parameter gradients only: off the critical path
                pw, pb = _p(wp.grad), _p(bp.grad)
                defer_to_side_stream(lambda st, temb=temb, w=w, dout=dout: lib().afd_silu_linear_bwd(
                    _p(temb), _p(w), _p(dout), pw, pb, None, B, K, N, 1, st), temb, w, dout, writes=(wp, bp))
                return None, None, None
            lib().afd_silu_linear_bwd(_p(temb), _p(w), _p(dout), _p(wp.grad), _p(bp.grad), _p(dtemb), B, K, N, 1, _stream())
            _wrote(wp, bp)
            return dtemb, None, None
        dw = torch.empty_like(w)
        db = torch.empty(N, device=w.device, dtype=torch.float32)
        lib().afd_silu_linear_bwd(_p(temb), _p(w), _p(dout), _p(dw), _p(db), _p(dtemb), B, K, N, 0, _stream())
        return dtemb, dw, db


def silu_linear_batched(temb, layers):
    """The emb_layer of several stages (each nn.Sequential(nn.SiLU(), nn.Linear(emb_dim, C_i)), ddpm_utils.py:208-214) on the SAME
    time embedding: ONE forward launch for all of them (the input exists as soon as the UNet forward starts, so one launch
    replaces six small dependent ones) -> [out_0, out_1, ...] (plain tensors, no autograd history).  `layers` = [(weight,
    bias), ...]; temb must not need a gradient (the conditional model keeps the per-stage op).  Backward stays PER STAGE:
    the stage wraps its output in its own autograd node WHEN ITS FORWARD RUNS (SiluLinearPre, see blocks._Stage._emb).
    One node for all six would hold every stage's gradient (and with it three of the four data-parallel buckets) back
    until the first encoder stage's backward; and six nodes created here, at the start of the forward, would carry the
    lowest sequence numbers of the graph -- autograd runs ready nodes highest-sequence first, so they would all run at
    the very END of backward (measured: the decoder bucket became ready at flush 24 of 25)."""
    import ctypes, struct
    ws, bs = [w for w, _ in layers], [b for _, b in layers]
    _chk(temb, *ws, *bs)
    temb = _c(temb)
    B, K = temb.shape
    with torch.no_grad():
        outs = [torch.empty(B, w.shape[0], device=temb.device, dtype=torch.float32) for w in ws]
    desc = b"".join(struct.pack("<QQQi4x", _p(w) or 0, _p(b) or 0, _p(o), w.shape[0]) for w, b, o in zip(ws, bs, outs))
    buf = ctypes.create_string_buffer(desc, len(desc))
    lib().afd_silu_linear_fwd_batched(_p(temb), ctypes.addressof(buf), len(ws), B, K, _stream())
    return outs


def gather_rows_batched(idx, tables):
    """[table_i[idx] for table_i in tables] in one launch (idx (B,) int64 on the device; tables (rows, N_i) fp32)."""
    import ctypes, struct
    _chk(*tables)
    B, rows = idx.shape[0], tables[0].shape[0]
    outs = [torch.empty(B, t.shape[1], device=t.device, dtype=torch.float32) for t in tables]
    desc = b"".join(struct.pack("<QQQi4x", _p(t), 0, _p(o), t.shape[1]) for t, o in zip(tables, outs))
    buf = ctypes.create_string_buffer(desc, len(desc))
    lib().afd_gather_rows_batched(_p(idx), ctypes.addressof(buf), len(tables), B, rows, _stream())
    return outs


class SiluLinearPre(_Fn):
    """Autograd node of ONE stage's emb_layer whose forward value was computed by silu_linear_batched: forward hands the
    precomputed output on, backward is SiluLinear's parameter-gradient part."""

    @staticmethod
    def forward(ctx, out, temb, w, bias):
        ctx.save_for_backward(temb, w)
        ctx.params = (w if isinstance(w, torch.nn.Parameter) else None, bias if isinstance(bias, torch.nn.Parameter) else None)
        return out.view_as(out)

    @staticmethod
    def backward(ctx, dout):
        temb, w = ctx.saved_tensors
        B, K = temb.shape
        N = w.shape[0]
        dout = _c(dout)
        wp, bp = ctx.params
        if _direct(wp, bp):
            if _GradMode.side is not None:                         # parameter gradients only: off the critical path
                pw, pb = _p(wp.grad), _p(bp.grad)
                defer_to_side_stream(lambda st, temb=temb, w=w, dout=dout: lib().afd_silu_linear_bwd(
                    _p(temb), _p(w), _p(dout), pw, pb, None, B, K, N, 1, st), temb, w, dout, writes=(wp, bp))
            else:
                lib().afd_silu_linear_bwd(_p(temb), _p(w), _p(dout), _p(wp.grad), _p(bp.grad), None, B, K, N, 1, _stream())
                _wrote(wp, bp)
            return None, None, None, None
        dw = torch.empty_like(w)
        db = torch.empty(N, device=w.device, dtype=torch.float32)
        lib().afd_silu_linear_bwd(_p(temb), _p(w), _p(dout), _p(dw), _p(db), None, B, K, N, 0, _stream())
        return None, None, dw, db


# ---------------------------------------------------------------------------------------------
# F14 / F15 / F16
# ---------------------------------------------------------------------------------------------
def noise_images(x, eps, t, alpha_hat):
    _chk(x, eps, alpha_hat)
    x, eps = _c(x), _c(eps)
    out = torch.empty_like(x)
    lib().afd_noise_images(_p(x), _p(eps), _p(t.contiguous()), _p(alpha_hat), _p(out), x.shape[0],
                           x.numel() // x.shape[0], _stream())
    return out


def denoise_step(x, eps_pred, noise, alpha, alpha_hat, beta, i, out=None):
    _chk(x, eps_pred, noise)
    x, eps_pred, noise = _c(x), _c(eps_pred), _c(noise)
    out = torch.empty_like(x) if out is None else out
    lib().afd_denoise_step(_p(x), _p(eps_pred), _p(noise), _p(alpha), _p(alpha_hat), _p(beta), int(i), _p(out),
                           x.numel(), _stream())
    return out


def denoise_step_dev(x, eps_pred, noise, alpha, alpha_hat, beta, t_dev, out):
    """Denoise update whose step index is t_dev[0] on the device (graph-replayable)."""
    _chk(x, eps_pred, noise)
    lib().afd_denoise_step_dev(_p(x), _p(eps_pred), _p(noise), _p(alpha), _p(alpha_hat), _p(beta), _p(t_dev), _p(out),
                               x.numel(), _stream())
    return out


def quantize_u8(x):
    _chk(x)
    x = _c(x)
    out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    lib().afd_quantize_u8(_p(x), _p(out), x.numel(), _stream())
    return out


def rotate_spline3_wrap(x, degrees):
    """scipy.ndimage.rotate(x, degrees, axes=(2,3), reshape=False, order=3, mode='grid-wrap') on the device."""
    import math
    _chk(x)
    x = _c(x)
    n, c, H, W = x.shape
    a = np.deg2rad(degrees)
    cs, sn = math.cos(a), math.sin(a)
    m = np.array([[cs, sn], [-sn, cs]], dtype=np.float64)                  # scipy.ndimage.rotate's rot_matrix
    plane = np.array([H, W], dtype=np.float64)
    off = (plane - 1) / 2 - m @ ((plane - 1) / 2)                           # in_center - rot @ out_center
    m = np.ascontiguousarray(m)
    off = np.ascontiguousarray(off)
    y = torch.empty_like(x)
    ws = torch.empty(n * c * H * W, device=x.device, dtype=torch.float64)
    lib().afd_affine_spline3_wrap(_p(x), _p(y), n * c, H, W, m.ctypes.data, off.ctypes.data, _p(ws), _stream())
    return y


class MseLoss(_Fn):
    """nn.MSELoss() (mean reduction), deterministic two-stage sum  (ddpm_utils.py:490,503)."""

    @staticmethod
    def forward(ctx, target, pred):
        _chk(target, pred)
        target, pred = _c(target), _c(pred)
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        ws = torch.empty(4096, device=pred.device, dtype=torch.float32)
        lib().afd_mse_fwd(_p(pred), _p(target), _p(loss), _p(ws), pred.numel(), _stream())
        ctx.save_for_backward(target, pred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        target, pred = ctx.saved_tensors
        dloss = dloss.reshape(1).contiguous()
        dpred = torch.empty_like(pred)
        lib().afd_mse_bwd(_p(pred), _p(target), _p(dloss), _p(dpred), pred.numel(), _stream())
        return None, dpred


def mse_loss(target, pred):
    return MseLoss.apply(target, pred)
