"""GPU parity tests, model level: blocks, whole UNet, train step and sampling loop on the HIP engine
against golden vectors produced by the reference's CPU path (tests/golden/make_golden.py)."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import check, golden_json, load_golden, note, rel_l2

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.asarray(a))
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}


@pytest.fixture(scope="module")
def A(gpu):
    import afdm
    return afdm, gpu


def _blocks(afdm):
    fs = dict(F_SET)
    return {
        "dc_4_8": lambda: afdm.DoubleConv(4, 8),
        "dc_res_8": lambda: afdm.DoubleConv(8, 8, residual=True),
        "dc_8_4_mid6": lambda: afdm.DoubleConv(8, 4, 6),
        "dcf_4_8": lambda: afdm.DoubleConv_F(4, 8, f_settings=fs),
        "dcf_res_8": lambda: afdm.DoubleConv_F(8, 8, residual=True, f_settings=fs),
        "sa_8_8": lambda: afdm.SelfAttention(8, 8),
        "sa_16_4": lambda: afdm.SelfAttention(16, 4),
        "down_4_8": lambda: afdm.Down(4, 8),
        "downF_4_8": lambda: afdm.Down_F(4, 8, f_settings=fs),
        "downFF_4_8": lambda: afdm.Down_FF(4, 8, f_settings=fs),
        "downFFF_4_8": lambda: afdm.Down_FFF(4, 8, f_settings=fs),
        "up_8_4": lambda: afdm.Up(8, 4),
        "upF_8_4": lambda: afdm.Up_F(8, 4, f_settings=fs),
        "upFF_8_4": lambda: afdm.Up_FF(8, 4, f_settings=fs),
        "upFFF_8_4": lambda: afdm.Up_FFF(8, 4, f_settings=fs),
    }


BLOCK_BWD_TOL = 1e-5
BLOCK_NAMES = ["dc_4_8", "dc_res_8", "dc_8_4_mid6", "dcf_4_8", "dcf_res_8", "sa_8_8", "sa_16_4", "down_4_8", "downF_4_8",
               "downFF_4_8", "downFFF_4_8", "up_8_4", "upF_8_4", "upFF_8_4", "upFFF_8_4"]


@pytest.mark.parametrize("name", BLOCK_NAMES)
def test_block_fwd_bwd_vs_reference(A, name):
    afdm, dev = A
    g = load_golden("blocks.npz")
    mod = _blocks(afdm)[name]()
    p = name + ".sd."
    sd = {k[len(p):]: T(g[k]) for k in g.files if k.startswith(p)}
    assert set(sd) == set(mod.state_dict().keys())
    mod.load_state_dict(sd)
    mod = mod.to(dev)
    ins, j = [], 0
    while f"{name}.in{j}" in g.files:
        ins.append(T(g[f"{name}.in{j}"]).to(dev).requires_grad_(True))
        j += 1
    if name.startswith(("down", "up")):
        # golden stage signature: Down(x, t) / Up(x, skip, t) with t already the (B,256) embedding
        y = mod(*ins)
    else:
        y = mod(ins[0])
    check("blocks fwd vs reference golden", y.detach().cpu(), g[f"{name}.y"], 1e-5, name)
    params = list(mod.named_parameters())
    grads = torch.autograd.grad(y, ins + [q for _, q in params], T(g[f"{name}.dy"]).to(dev), allow_unused=True)
    # the golden gradients are the reference's own fp32 CPU results (their rounding error is part of the difference)
    for j in range(len(ins)):
        check("blocks input grads vs reference golden (fp32 vs fp32)", grads[j].cpu(), g[f"{name}.din{j}"], BLOCK_BWD_TOL, (name, "din", j))
    for (kn, _), gr in zip(params, grads[len(ins):]):
        check("blocks param grads vs reference golden (fp32 vs fp32)", gr.cpu(), g[f"{name}.dsd.{kn}"], BLOCK_BWD_TOL, (name, kn))


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("c", [1, 3])
def test_unet_forward_vs_reference(A, variant, c):
    afdm, dev = A
    g = load_golden("unet_fwd.npz")
    tag = f"v{variant}_c{c}"
    afdm.set_seed(42)
    net = afdm.UNet(c_in=c, c_out=c, image_size=32, f_settings=dict(F_SET) if variant else None, device=dev, variant=variant)
    meta = golden_json(g, "meta")[tag]
    assert list(net.state_dict().keys()) == meta["keys"]
    assert [list(v.shape) for v in net.state_dict().values()] == meta["shapes"]
    assert sum(p.numel() for p in net.parameters()) == meta["n_params"] and len(list(net.buffers())) == 0
    cs = np.array([[v.double().sum().item(), v.double().abs().sum().item()] for v in net.state_dict().values()])
    assert np.allclose(cs, g[f"{tag}.param_checksums"], rtol=1e-12, atol=1e-12), "seeded init differs from the reference"
    net = net.to(dev)
    with torch.no_grad():
        y = net(T(g[f"{tag}.x"]).to(dev), T(g[f"{tag}.t"]).to(dev))
    err = check("UNet fwd vs reference golden", y.cpu(), g[f"{tag}.y"], 1e-5, tag)
    print(tag, "UNet fwd rel-L2 vs reference:", err)
    # grad mode takes the unfused FF path; it must agree with the no-grad (fused epilogue) path
    y2 = net(T(g[f"{tag}.x"]).to(dev), T(g[f"{tag}.t"]).to(dev))
    assert rel_l2(y2.detach().cpu(), y.cpu()) < 1e-6


@pytest.mark.parametrize("variant", [3, 0])
def test_unet_image_size_64_default_constructor_shapes(A, variant):
    """The reference's default UNet(image_size=64): 64x64 maps, 64..512 channels, L=4096 attention, GroupNorm
    samples beyond the register path.  Forward and one parameter gradient vs the CPU oracle (B=1)."""
    afdm, dev = A
    from oracle import ref_ops as R
    afdm.set_seed(1)
    net = afdm.UNet(c_in=3, c_out=3, image_size=64, f_settings=dict(F_SET) if variant else None, device=dev, variant=variant)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(dev)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 3, 64, 64, generator=g)
    t = torch.tensor([321])
    y = net(x.to(dev), t.to(dev))
    with torch.no_grad():
        ref = R.unet_forward(sd, x, t, variant, F_SET)
    err = rel_l2(y.detach().cpu(), ref)
    print(f"image_size=64 variant {variant}: fwd rel-L2 vs oracle {err:.2e}")
    assert err < 1e-5
    dy = torch.randn(y.shape, generator=g)
    (gw,) = torch.autograd.grad(y, net.inc.conv1.weight if variant else net.inc.double_conv[0].weight, dy.to(dev))
    key = "inc.conv1.weight" if variant else "inc.double_conv.0.weight"
    sdg = {k: v.clone().requires_grad_(k == key) for k, v in sd.items()}
    (gref,) = torch.autograd.grad(R.unet_forward(sdg, x, t, variant, F_SET), sdg[key], dy)
    assert rel_l2(gw.cpu(), gref) < 1e-4


def test_unet_variant4_runs_and_matches_reference(A):
    afdm, dev = A
    g = load_golden("unet_fwd.npz")
    afdm.set_seed(42)
    net = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=4)
    meta = golden_json(g, "meta")["v4_c3"]
    assert list(net.state_dict().keys()) == meta["keys"]
    net = net.to(dev)
    with torch.no_grad():
        y = net(T(g["v4_c3.x"]).to(dev), T(g["v4_c3.t"]).to(dev))
    assert rel_l2(y.cpu(), g["v4_c3.y"]) < 1e-5


def test_constructor_errors_and_cpu_forward_fails_loudly(A):
    afdm, dev = A
    with pytest.raises(ValueError, match="f_settings is empty"):
        afdm.UNet(variant=1)
    with pytest.raises(ValueError, match="variant value must be between 0 and 4"):
        afdm.UNet(variant=7)
    net = afdm.UNet(c_in=1, c_out=1, image_size=32, device="cpu", variant=0)
    with pytest.raises(RuntimeError):
        net(torch.randn(1, 1, 32, 32), torch.tensor([5]))


def _train_setup(afdm, dev, variant=3):
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET) if variant else None, device=dev, variant=variant).to(dev)
    diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
    return model, diff


@pytest.fixture
def conv_path(A, request):
    """'winograd' forces the Winograd kernels on every 3x3 layer they cover (at these small batches the by-rule choice
    would keep most layers on the direct kernels); 'direct' switches them off."""
    afdm, _ = A
    L = afdm.lib()
    for m in {"auto": (64, 96), "winograd": (67, 98), "direct": (65, 97)}[request.param]:
        L.afd_debug_conv_path(m)
    yield request.param
    L.afd_debug_conv_path(64)
    L.afd_debug_conv_path(96)


@pytest.mark.parametrize("variant,conv_path", [(3, "auto"), (3, "winograd"), (3, "direct"), (0, "auto"), (1, "auto"), (1, "winograd"),
                                               (2, "auto")], indirect=["conv_path"])
def test_train_step_vs_reference(A, variant, conv_path):
    """ddpm_utils.py:499-507 with t / eps injected from the reference's own CPU run (B=4): Config D (variant 3, under
    each convolution path) and Configs A / B / C (variants 0 / 1 / 2, Train.ipynb:106 sweeps them)."""
    afdm, dev = A
    g = load_golden("train_step.npz" if variant == 3 else f"train_step_v{variant}.npz")
    model, diff = _train_setup(afdm, dev, variant)
    names = [n for n, _ in model.named_parameters()]
    step = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
    images = T(g["images"]).to(dev)
    l0 = step(images, t=T(g["t0"]), eps=T(g["eps0"]).to(dev))
    e_loss0 = abs(l0.item() - g["losses"][0]) / abs(g["losses"][0])
    params = dict(model.named_parameters())
    l2 = np.array([params[n].grad.double().pow(2).sum().sqrt().item() for n in names])
    e_norm = float(np.max(np.abs(l2 - g["grad_checksums0"][:, 2]) / (np.abs(g["grad_checksums0"][:, 2]) + 1e-8)))
    e_grad = max(rel_l2(params[k[6:]].grad.cpu(), g[k]) for k in g.files if k.startswith("grad0."))
    # post-AdamW parameters.  Adam's first step is -lr * g / (|g| + 1e-8): where |g| is within 100x of eps the update
    # amplifies a 1e-9 absolute gradient difference into percents (zero-initialised biases with |g| ~ 1e-7: even the CPU
    # oracle, the same ATen ops in another graph, is 8e-5 off the reference on variant 0's bot2 bias), so the 1e-4 gate
    # is taken over the elements with |g_ref| >= 1e-6 and the whole tensor is gated at 2e-3.
    e_par, e_par_all, over = 0.0, 0.0, {}
    for k in g.files:
        if k.startswith("param1."):
            got, want, gref = params[k[7:]].detach().cpu().numpy(), g[k], np.abs(g["grad0." + k[7:]])
            m = gref >= 1e-6
            assert m.mean() > 0.9, k
            e_par = max(e_par, rel_l2(got[m], want[m]))
            e_all = rel_l2(got, want)
            e_par_all = max(e_par_all, e_all)
            if e_all > 1e-4:
                over[k[7:]] = e_all
    if over:
        # the evidence for the relaxed whole-tensor gate: the CPU oracle (the same ATen ops in another autograd graph) takes the
        # same step from the same initial parameters; its own whole-tensor error against the reference on exactly these tensors
        from oracle import ref_ops as R
        afdm.set_seed(42)
        ref_net = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET) if variant else None, device="cpu", variant=variant)
        sd = {k: v.clone().requires_grad_(True) for k, v in ref_net.state_dict().items()}
        _, _, ah = R.noise_schedule(1000)
        _, _, og = R.train_step_loss_and_grads(sd, T(g["images"]), T(g["t0"]), T(g["eps0"]), variant, F_SET, ah)
        for k, e_hip in sorted(over.items(), key=lambda kv: -kv[1]):
            with torch.no_grad():
                p1, _, _ = R.adamw_step(sd[k].detach(), og[k], torch.zeros_like(og[k]), torch.zeros_like(og[k]), 1, 3e-4)
            e_orc = rel_l2(p1, g["param1." + k])
            small = float((np.abs(g["grad0." + k]) < 1e-6).mean())
            print(f"  post-AdamW whole tensor > 1e-4: {k}: HIP {e_hip:.2e}, CPU oracle {e_orc:.2e} vs the reference; "
                  f"{100 * small:.1f} % of its elements have |g_ref| < 1e-6, max |g_ref| {np.abs(g['grad0.' + k]).max():.2e}")
            note("train step: post-AdamW whole-tensor error of the CPU ORACLE on the tensors where HIP exceeds 1e-4", e_orc, (variant, k))
    l1 = step(images, t=T(g["t1"]), eps=T(g["eps1"]).to(dev))
    e_loss1 = abs(l1.item() - g["losses"][1]) / abs(g["losses"][1])
    print(f"train step v{variant} [{conv_path}]: loss0 rel {e_loss0:.2e}, worst grad rel-L2 {e_grad:.2e}, worst grad-norm rel {e_norm:.2e}, "
          f"worst post-AdamW param rel-L2 {e_par:.2e} (all elements {e_par_all:.2e}), loss1 rel {e_loss1:.2e}")
    for fam, e in (("train step: loss", max(e_loss0, e_loss1)), ("train step: gradients", e_grad), ("train step: gradient norms", e_norm),
                   ("train step: post-AdamW params (|g| >= 1e-6)", e_par), ("train step: post-AdamW params (all elements)", e_par_all)):
        note(fam, e, (variant, conv_path))
    assert e_loss0 < 2e-5 and e_loss1 < 1e-4
    assert np.allclose(l2, g["grad_checksums0"][:, 2], rtol=3e-4, atol=1e-8)
    assert e_grad < 1e-4 and e_par < 1e-4 and e_par_all < 2e-3                      # SURVEY 8d gates
    cs = np.array([[v.double().sum().item(), v.double().abs().sum().item()] for v in model.state_dict().values()])
    assert np.allclose(cs[:, 1], g["param_checksums_after2"][:, 1], rtol=1e-3, atol=0.1)   # |.|-sums incl. zero-init biases after 2 Adam steps


def test_overlapped_weight_gradients_are_bitwise_equal(A):
    """The weight-gradient kernels on the second stream (TrainStep(overlap_wgrad=True)) change WHEN they run, not what
    they compute: after two steps every parameter is bit-identical to the single-stream run, and a repeat of the same
    run is bit-identical to itself (no atomics anywhere)."""
    afdm, dev = A
    g = load_golden("train_step.npz")
    images, t0, e0 = T(g["images"]).to(dev), T(g["t0"]), T(g["eps0"]).to(dev)
    t1, e1 = T(g["t1"]), T(g["eps1"]).to(dev)
    outs = []
    for overlap in (False, True, True):
        model, diff = _train_setup(afdm, dev)
        step = afdm.TrainStep(model, diff, lr=3e-4, graph=False, overlap_wgrad=overlap)
        step(images, t=t0, eps=e0)
        step(images, t=t1, eps=e1)
        torch.cuda.synchronize()
        outs.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu())
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[1], outs[2])


def test_full_batch_step_matrix_core_kernels_agree_with_fp32_kernels(A):
    """BASELINE size (Config D, B = 256): one forward + backward of the whole net with every layer on the kernels the rule
    picks (f16x2 forward / dgrad / 3x3 weight gradients and bf16x3 1x1 weight gradients on the matrix cores) against the same
    step with those switched off (fp32 Winograd / direct kernels, an independent algorithm): loss and every parameter
    gradient agree far inside the 1e-5 contract.  The rule must really have picked the matrix-core forms at this size."""
    afdm, dev = A
    from afdm import ops
    L = afdm.lib()
    assert L.afd_conv3x3_weight_kinds(256, 128, 128, 16, 16) == 3 and L.afd_conv_wgrad_form(256, 128, 128, 16, 16, 3) == 4
    assert L.afd_conv_wgrad_form(256, 32, 96, 32, 32, 1) == 4
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=3).to(dev)
    diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
    g = torch.Generator().manual_seed(11)
    images = torch.randn(256, 3, 32, 32, generator=g).to(dev)
    t = torch.randint(1, 1000, (256,), generator=g)
    eps = torch.randn(256, 3, 32, 32, generator=g).to(dev)
    outs = []
    try:
        for modes in ((80, 84, 88), (81, 85, 89)):
            for m in modes:
                L.afd_debug_conv_path(m)
            model.zero_grad(set_to_none=True)
            x_t, _ = diff.noise_images(images, t.to(dev), eps=eps)
            loss = ops.mse_loss(eps, model(x_t, t.to(dev)))
            loss.backward()
            torch.cuda.synchronize()
            outs.append((loss.item(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    finally:
        for m in (80, 84, 88):
            L.afd_debug_conv_path(m)
    assert abs(outs[0][0] - outs[1][0]) <= 2e-6 * abs(outs[1][0])
    assert outs[0][1].keys() == outs[1][1].keys() and len(outs[0][1]) > 150
    worst = 0.0
    for n in outs[0][1]:
        a, b = outs[0][1][n].double(), outs[1][1][n].double()
        if b.norm() > 0:
            worst = max(worst, ((a - b).norm() / b.norm()).item())
            check("full-batch step: matrix-core (f16x2 / bf16x3) vs fp32 kernels, parameter gradients", outs[0][1][n].cpu(), outs[1][1][n].cpu(), 1e-5, n)
    assert worst < 1e-5


def test_full_batch_step_vs_cpu_oracle(A):
    """BASELINE size against the ORACLE (not HIP vs HIP): Config D, B = 256, one forward + backward
    (ddpm_utils.py:489,498-509) on the HIP engine with the kernels the rule picks at this size, against
    oracle.ref_ops.train_step_loss_and_grads on the box's host cores with the same images / t / eps:
    loss <= 2e-5, EVERY parameter-gradient tensor <= 1e-4, recorded in the ledger."""
    afdm, dev = A
    from afdm import ops
    from oracle import ref_ops as R
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device="cpu", variant=3)
    sd = {k: v.clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to(dev)
    diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
    g = torch.Generator().manual_seed(23)
    images = torch.rand(256, 3, 32, 32, generator=g) * 2 - 1
    t = torch.randint(1, 1000, (256,), generator=g)
    eps = torch.randn(256, 3, 32, 32, generator=g)
    step = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
    loss = step._fwd_bwd(images.to(dev), t.to(dev), eps.to(dev))           # forward + backward of the product step, no AdamW
    torch.cuda.synchronize()
    got = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
    _, _, ah = R.noise_schedule(1000)
    nt = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    try:
        loss_o, _, grads_o = R.train_step_loss_and_grads(sd, images, t, eps, 3, F_SET, ah)
    finally:
        torch.set_num_threads(nt)
    e_loss = abs(loss.item() - loss_o.item()) / abs(loss_o.item())
    note("full-batch (B=256) step vs CPU oracle: loss", e_loss)
    assert e_loss < 2e-5, e_loss
    assert set(grads_o) == set(got) and len(got) == 182
    worst = max(check("full-batch (B=256) step vs CPU oracle: parameter gradients (fp32 vs fp32)", got[k], grads_o[k], 1e-4, k) for k in got)
    print(f"B=256 Config-D step vs CPU oracle: loss rel {e_loss:.2e}, worst parameter-gradient rel-L2 {worst:.2e} over {len(got)} tensors")


def test_graph_replay_equals_eager(A):
    afdm, dev = A
    g = load_golden("train_step.npz")
    images, t0, e0 = T(g["images"]).to(dev), T(g["t0"]), T(g["eps0"]).to(dev)
    t1, e1 = T(g["t1"]), T(g["eps1"]).to(dev)
    outs = []
    for use_graph in (False, True, "lanes"):                              # "lanes": the captured step re-issued on two real streams (csrc/replay.hip)
        model, diff = _train_setup(afdm, dev)
        step = afdm.TrainStep(model, diff, lr=3e-4, graph=use_graph)      # the capture warm-up must leave no trace: the
        la = step(images, t=t0, eps=e0).item()                            # first graph call is exactly one step
        lb = step(images, t=t1, eps=e1).item()
        outs.append((la, lb, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()))
        if use_graph is True:
            graph_step = step
        if use_graph == "lanes":
            n, n_main, n_side, n_wait = step.lanes_counts
            print("two-lane replay:", step.lanes_counts)
            assert n == n_main + n_side and n_side >= 40 and 2 <= n_wait <= n_side      # the weight gradients sit on the side lane, joined by a few events
    for k, tag in ((1, "graph"), (2, "lanes")):
        print(tag, "vs eager after 2 steps: loss diffs", abs(outs[0][0] - outs[k][0]), abs(outs[0][1] - outs[k][1]),
              "param rel-L2", rel_l2(outs[k][2], outs[0][2]))
        assert abs(outs[0][0] - outs[k][0]) < 1e-6 and abs(outs[0][1] - outs[k][1]) < 1e-5
        assert rel_l2(outs[k][2], outs[0][2]) < 1e-6
    assert torch.equal(outs[2][2], outs[1][2])                            # the same captured launches, issued differently: bit-identical
    with pytest.raises(ValueError, match="injected noise"):
        graph_step(images, t=t0)                                          # captured with eps: omitting it later is an error
    assert torch.isfinite(step(images, t=t0)).item()                      # ("lanes" draws the noise outside the replayed list: either way)


def test_unused_parameters_stay_untouched_like_the_reference(A):
    """Variant 4 (+ num_classes, no labels passed): the reference's AdamW never touches the stage-level norm1 nor
    label_emb (their grad is None); two steps with the reference's t / eps, then the same per-tensor checksums."""
    afdm, dev = A
    g = load_golden("conditional.npz")
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=4, num_classes=10).to(dev)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
    step = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
    images = T(g["v4.images"]).to(dev)
    for i in range(2):
        loss = step(images, t=T(g["v4.t"][i]), eps=T(g["v4.eps"][i]).to(dev))
        assert abs(loss.item() - g["v4.losses"][i]) < 1e-4 * abs(g["v4.losses"][i])
    untouched = [k for k, v in model.state_dict().items() if torch.equal(v, before[k])]
    assert untouched == golden_json(g, "v4.untouched")
    cs = np.array([[v.double().sum().item(), v.double().abs().sum().item()] for v in model.state_dict().values()])
    assert np.allclose(cs[:, 1], g["v4.param_checksums_after2"][:, 1], rtol=1e-3, atol=0.1)


def test_conditional_unet_forward_backward_vs_reference(A):
    """UNet(num_classes=10).forward(x, t, y) (ddpm_models.py:254-258,276-277): the label embedding goes through the C ABI
    (afd_embed_add_*); forward, the gradient reaching label_emb and two other parameters vs the reference."""
    afdm, dev = A
    g = load_golden("conditional.npz")
    afdm.set_seed(42)
    net = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=3, num_classes=10)
    assert list(net.state_dict().keys()) == golden_json(g, "keys")
    net = net.to(dev)
    x, t, y = T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["labels"]).to(dev)
    pred = net(x, t, y)
    e_f = rel_l2(pred.detach().cpu(), g["y"])
    dl, dw, de = torch.autograd.grad(pred, [net.label_emb.weight, net.outc.weight, net.down1.emb_layer[1].weight], T(g["dy"]).to(dev))
    errs = (rel_l2(dl.cpu(), g["d_label_emb"]), rel_l2(dw.cpu(), g["d_outc_weight"]), rel_l2(de.cpu(), g["d_down1_emb_weight"]))
    print(f"conditional UNet: fwd rel-L2 {e_f:.2e}; d label_emb / outc.weight / down1.emb weight rel-L2 {errs[0]:.2e} {errs[1]:.2e} {errs[2]:.2e}")
    assert e_f < 1e-5 and max(errs) < 1e-4
    assert float(dl[1].abs().max()) == 0.0                              # class 1 is absent from y
    with torch.no_grad():
        assert rel_l2(net(x, t).cpu(), g["y_uncond"]) < 1e-5 and rel_l2(net(x, t, y).cpu(), g["y"]) < 1e-5
    with pytest.raises(RuntimeError, match="int64"):
        net(x, t, y.float())


@pytest.mark.parametrize("conv_path", ["auto", "winograd"], indirect=True)
@pytest.mark.parametrize("variant,c", [(3, 3), (0, 1)])
def test_sample_100_steps_vs_reference(A, variant, c, conv_path):
    """Diffusion.sample / revert, T=101, replaying the reference CPU run's noise stream."""
    afdm, dev = A
    g = load_golden("sample.npz")
    tag = f"v{variant}_c{c}"
    afdm.set_seed(42)
    model = afdm.UNet(c_in=c, c_out=c, image_size=32, f_settings=dict(F_SET) if variant else None, device=dev, variant=variant).to(dev)
    diff = afdm.Diffusion(noise_steps=101, img_size=32, device=dev)
    afdm.set_seed(7)
    xq, rq, xf = diff.sample(model, n=2, image_channels=c, noise_source="cpu", return_float=True)
    assert model.training
    err = check("100-step sample (pre-quantisation x) vs reference", xf.cpu(), g[f"{tag}.float_x_after_i1"], 1e-4, (tag, conv_path))
    print(tag, "100-step sample rel-L2 (pre-quantisation):", err)
    assert xq.dtype == torch.uint8 and tuple(rq.shape) == g[f"{tag}.sample_result"].shape
    for got, want in ((xq, g[f"{tag}.sample_x"]), (rq, g[f"{tag}.sample_result"])):
        d = got.cpu().numpy().astype(int) - want.astype(int)
        assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.01, "uint8 images differ by more than rounding-boundary flips"
    afdm.set_seed(7)
    rv = diff.revert(model, n=2, image_channels=c, noise_source="cpu")
    d = rv.cpu().numpy().astype(int) - g[f"{tag}.revert"].astype(int)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.01
    if variant == 3:   # Config E: per-step rotation (device spline rotate, bit-identical to scipy's on the golden angles); uint8 gate:
        #                  99 rotations by 90/100 degrees amplify rounding-boundary flips, so <= 2 grey levels on < 2 % of the pixels
        afdm.set_seed(7)
        xr, _ = diff.sample(model, n=2, image_channels=c, theta=90.0, noise_source="cpu")
        d = xr.cpu().numpy().astype(int) - g[f"{tag}.sample_theta90_x"].astype(int)
        assert np.abs(d).max() <= 2 and (d != 0).mean() < 0.02


def test_sample_999_steps_vs_reference(A):
    """SURVEY 8d gate: the FULL-length trajectory (T=1000, 999 UNet evaluations, Config D, c=3, n=2), replaying the
    reference CPU run's noise stream; pre-quantisation x within 1e-4 relative L2 at i = 900, 500, 100 and 1."""
    afdm, dev = A
    g = load_golden("sample_full.npz")
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=3).to(dev)
    diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
    afdm.set_seed(7)
    xq, rq, xf = diff.sample(model, n=2, image_channels=3, noise_source="cpu", return_float=True)
    fs = diff.last_float_snapshots                 # x after i = 900, 800, ..., 100, then the final x
    snaps = {900: fs[0], 500: fs[4], 100: fs[8], 1: xf}
    for i in (900, 500, 100, 1):
        err = check("999-step sample (pre-quantisation x) vs reference", snaps[i].cpu(), g[f"float_x_after_i{i}"], 1e-4, f"i={i}")
        print(f"999-step sample, after i={i}: rel-L2 {err:.3e}")
    for got, want in ((xq, g["sample_x"]), (rq, g["sample_result"])):
        d = got.cpu().numpy().astype(int) - want.astype(int)
        assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.01


def test_rotation_sweep_equals_one_angle_at_a_time(A):
    """Config E: the batched sweep against the reference's procedure (re-seed, sample(n, theta)) for each angle."""
    afdm, dev = A
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=3).to(dev)
    diff = afdm.Diffusion(noise_steps=201, img_size=32, device=dev)
    thetas = [-90.0, 0.0, 33.0, 90.0]
    afdm.set_seed(5)
    xs, rs = diff.sample_rotation_sweep(model, 3, 3, thetas)
    for k, th in enumerate(thetas):
        afdm.set_seed(5)
        x1, r1 = diff.sample(model, n=3, image_channels=3, theta=th)
        assert rs[k].shape == r1.shape
        for got, want in ((xs[k], x1), (rs[k], r1)):
            d = got.cpu().numpy().astype(int) - want.cpu().numpy().astype(int)
            assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.01, th


def test_sample_concurrent_equals_one_trajectory_at_a_time(A):
    """Diffusion.sample_concurrent: the same batches with the same (per batch, per step) noise, run two at a time on two
    streams or one after the other, give identical images."""
    afdm, dev = A
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=3).to(dev)
    diff = afdm.Diffusion(noise_steps=21, img_size=32, device=dev)

    def noise_fn(k, i, shape):
        g = torch.Generator().manual_seed(1000 * k + i)
        return torch.randn(shape, generator=g).to(dev)

    xa, ra = diff.sample_concurrent(model, n=10, image_channels=3, batch=4, streams=2, noise_fn=noise_fn)
    xb, rb = diff.sample_concurrent(model, n=10, image_channels=3, batch=4, streams=1, noise_fn=noise_fn)
    assert xa.shape == (10, 3, 32, 32) and xa.dtype == torch.uint8 and model.training
    assert torch.equal(xa, xb) and torch.equal(ra, rb)
    # every trajectory's step replayed from its own hipGraph (batches of 4, 4 and 2: two graphs per stream slot)
    xc, rc = diff.sample_concurrent(model, n=10, image_channels=3, batch=4, streams=2, noise_fn=noise_fn, graph=True)
    assert model.training and torch.equal(xa, xc) and torch.equal(ra, rc)


def test_graph_sampling_equals_eager_sampling(A):
    """One captured denoise step replayed T-2 times must equal the eager loop (same device RNG seed)."""
    afdm, dev = A
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=3).to(dev)
    diff = afdm.Diffusion(noise_steps=31, img_size=32, device=dev)
    outs = []
    for use_graph in (False, True):
        afdm.set_seed(5)
        xq, rq, xf = diff.sample(model, n=3, image_channels=3, noise_source="device", return_float=True, graph=use_graph)
        outs.append((xq.cpu(), rq.cpu(), xf.cpu()))
    # the capture warm-up's effects (x, the device generator's state) are undone, so the replays draw the very noise the
    # eager loop draws: same images bit for bit
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_sample_concurrent_cold_cache_winograd_unequal_batches(A):
    """Trajectories on several streams, every 3x3 layer on the Winograd kernels (cached transformed weights), starting
    from a COLD cache (as after any optimiser step) with unequal batch sizes: each stream must fill and read its own
    transformed-weight images and timestep tables; the result equals the one-stream run bit for bit."""
    afdm, dev = A
    from afdm import ops
    L = afdm.lib()
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=3).to(dev)
    diff = afdm.Diffusion(noise_steps=9, img_size=32, device=dev)

    def noise_fn(k, i, shape):
        g = torch.Generator().manual_seed(77 * k + i)
        return torch.randn(shape, generator=g).to(dev)

    for m in (67, 98):
        L.afd_debug_conv_path(m)                        # force Winograd (the by-rule choice keeps small batches direct)
    try:
        outs = []
        for streams in (3, 1, 3):
            ops.bump_param_epoch()                      # cold cache: every cached transform is stale
            diff._t_cache.clear()
            outs.append(diff.sample_concurrent(model, n=37, image_channels=3, batch=16, streams=streams, noise_fn=noise_fn))
            torch.cuda.synchronize()
    finally:
        L.afd_debug_conv_path(64)
        L.afd_debug_conv_path(96)
    assert outs[0][0].shape == (37, 3, 32, 32)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[2][0], outs[1][0]) and torch.equal(outs[2][1], outs[1][1])
    assert len({k[2] for k in ops._WinoWeights.store}) >= 3           # one set of images per stream


def test_graph_step_matches_eager_step_bitwise(A):
    """Determinism of the captured step itself: replaying with injected noise == eager kernels."""
    afdm, dev = A
    from afdm import ops
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=3).to(dev).eval()
    diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 32, 32, generator=g).to(dev)
    nz = torch.randn(2, 3, 32, 32, generator=g).to(dev)
    t = torch.full((2,), 437, device=dev, dtype=torch.long)
    with torch.no_grad():
        eps = model(x, t)
        a = ops.denoise_step(x, eps, nz, diff.alpha, diff.alpha_hat, diff.beta, 437)
        b = ops.denoise_step_dev(x, eps, nz, diff.alpha, diff.alpha_hat, diff.beta, t, torch.empty_like(x))
    assert torch.equal(a, b)


def test_ddpm_run_drop_in_end_to_end(A, tmp_path, monkeypatch):
    """Train.ipynb's entry point with its 21 params keys on a tiny synthetic MNIST CSV: train 1 epoch, reload
    the checkpoint, sample, write PNGs + collage, settings text and loss CSV in the reference's layout."""
    afdm, dev = A
    rng = np.random.default_rng(0)
    csvp = tmp_path / "mnist.csv"
    arr = np.concatenate([rng.integers(0, 10, (16, 1)), rng.integers(0, 256, (16, 784))], axis=1)
    np.savetxt(csvp, arr, fmt="%d", delimiter=",", header=",".join(["label"] + [f"p{i}" for i in range(784)]), comments="")
    monkeypatch.chdir(tmp_path)
    params = {"unet_v": 3, "dataset": "MNIST", "epochs": 1, "batchsize": 8, "image_size": 32, "image_channels": 1,
              "device": "cuda", "lr": 3e-4, "noise_steps": 12, "image_gen_per_epoch": 2, "dataset_dir": str(csvp),
              "f_kernel": 3, "f_beta": 2, "f_down": math.pi / 2, "f_up": math.pi / 2, "save_trining": False,
              "gen_per_batch": 4, "gen_total": 4, "collage_n_per_image": 4, "collage_n": 4, "seed": 42}
    out = afdm.ddpm_run(params)
    run = "DDPM_Uncondtional_MNIST_3"
    assert len(out["loss_all"]) == 1 and np.isfinite(out["loss_all"][0])
    assert (tmp_path / "models" / run / "ckpt_MNIST_3.pt").exists()
    assert (tmp_path / "results" / run / "0.jpg").exists()
    txt = (tmp_path / "runs" / run / "settings_MNIST_3.txt").read_text()
    assert txt.splitlines()[0] == "unet_v: 3" and "batch_size : 8" in txt and "kernel_size: 3" in txt
    assert (tmp_path / "runs" / run / "trining_loss_MNIST_3.csv").exists()
    assert (tmp_path / "images" / "generated" / "MNIST_3" / "image_3.png").exists()
    assert os.path.exists(str(tmp_path / "images" / "generated" / "MNIST_3") + "_collage_0.png")
    assert tuple(out["sample"].shape) == (6, 1, 32, 32) and tuple(out["revert"].shape) == (1, 1, 32, 32)
    sd = torch.load(tmp_path / "models" / run / "ckpt_MNIST_3.pt", weights_only=True)
    assert len(sd) == 182        # reference checkpoint wire format: same 182 keys


def _run_ranks(tmp_path, mode, port, name, backend="gloo", extra_env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / name
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0",
               **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "ddp_worker.py"), "--device", "cuda", "--mode", mode,
           "--backend", backend, "--out", str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=900)
    return torch.load(out, weights_only=True)


def _single_rank_step(afdm, dev):
    g = load_golden("train_step.npz")
    model, diff = _train_setup(afdm, dev)
    step = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
    loss = step(T(g["images"]).to(dev), t=T(g["t0"]), eps=T(g["eps0"]).to(dev)).item()
    return loss, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()


def test_two_rank_data_parallel_equals_single_rank(A, tmp_path):
    """2 processes on the one GPU (gloo, buckets staged through the host): B=2+2 must equal one rank with B=4, and the
    bucket all-reduces must have started DURING backward: counted when autograd's backward returns, BEFORE the final flush
    of the weight-gradient queue -- three of the four (the first-layers bucket, the smallest, completes with backward
    itself and starts at that flush).  The six stages' time-embedding gradients leave through per-stage nodes, so they do
    not hold the decoder buckets back."""
    afdm, dev = A
    got = _run_ranks(tmp_path, "train", 29533, "ddp.pt")
    loss, flat = _single_rank_step(afdm, dev)
    print("2-rank vs 1-rank: loss rel", abs(got["loss_mean"] - loss) / abs(loss), "param rel-L2", rel_l2(got["params"], flat),
          "buckets", got["slices"].tolist(), "started before the end of backward:", got["overlapped_buckets"])
    assert abs(got["loss_mean"] - loss) < 1e-5 * abs(loss)
    assert rel_l2(got["params"], flat) < 1e-6
    assert got["n_buckets"] == 4 and got["overlapped_buckets"] == 3 and got["started_before_finish"] == 4
    sizes = [b - a for a, b in got["slices"].tolist()]
    assert sizes[0] == min(sizes)                           # the bucket that cannot overlap is the smallest (inc .. sa2)


def test_two_rank_data_parallel_graph_mode(A, tmp_path):
    """TrainStep(graph=True, distributed=True): the captured forward + backward, then the whole exchange + AdamW."""
    afdm, dev = A
    got = _run_ranks(tmp_path, "train", 29534, "ddp_graph.pt", extra_env={"AFD_TEST_GRAPH": "1"})
    loss, flat = _single_rank_step(afdm, dev)
    assert abs(got["loss_mean"] - loss) < 1e-5 * abs(loss) and rel_l2(got["params"], flat) < 1e-6
    assert got["overlapped_buckets"] == 0


def test_two_rank_data_parallel_lanes_mode(A, tmp_path):
    """TrainStep(graph="lanes", distributed=True): forward + backward re-issued on two real streams (the replayed list ends
    with the joined weight gradients: the executor's own join orders the exchange behind them), then the exchange + AdamW."""
    afdm, dev = A
    got = _run_ranks(tmp_path, "train", 29538, "ddp_lanes.pt", extra_env={"AFD_TEST_GRAPH": "lanes"})
    loss, flat = _single_rank_step(afdm, dev)
    assert abs(got["loss_mean"] - loss) < 1e-5 * abs(loss) and rel_l2(got["params"], flat) < 1e-6
    assert got["overlapped_buckets"] == 0


def test_two_rank_rccl_data_parallel(A, tmp_path):
    """The 'nccl' (= RCCL) branch: one rank per GPU, bucket all-reduces on the collective stream behind events.
    Needs two GPUs: skipped on the one-GPU test box."""
    afdm, dev = A
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL path; unmeasured on hardware so far)")
    got = _run_ranks(tmp_path, "train", 29535, "ddp_rccl.pt", backend="nccl")
    loss, flat = _single_rank_step(afdm, dev)
    assert abs(got["loss_mean"] - loss) < 1e-5 * abs(loss) and rel_l2(got["params"], flat) < 1e-6
    assert got["overlapped_buckets"] == 3 and got["started_before_finish"] == 4


def test_two_rank_sharded_sampling_equals_single_rank(A, tmp_path):
    """Diffusion.sample_sharded / sample_rotation_sweep_sharded over 2 ranks (n = 5 -> shards of 2 and 3 images; 3 angles
    -> 1 and 2): gathered on rank 0, bit-identical to the one-rank calls under the same seed (ddpm_tasks.py:346-369, :365)."""
    afdm, dev = A
    got = _run_ranks(tmp_path, "sample", 29536, "sample.pt")
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=3).to(dev)
    diff = afdm.Diffusion(noise_steps=201, img_size=32, device=dev)
    afdm.set_seed(5)
    xq, rq = diff.sample(model, n=5, image_channels=3)
    assert got["x"].shape == (5, 3, 32, 32) and got["result"].shape == (15, 3, 32, 32)
    assert torch.equal(got["x"], xq.cpu()) and torch.equal(got["result"], rq.cpu())
    afdm.set_seed(5)
    xs, rs = diff.sample_rotation_sweep(model, 2, 3, [-90.0, 0.0, 45.0])
    assert torch.equal(got["rot_x"], torch.stack([t.cpu() for t in xs]))
    assert torch.equal(got["rot_result"], torch.stack([t.cpu() for t in rs]))
    # and the unsharded entry points are what the sharded ones fall back to without a process group
    afdm.set_seed(5)
    x1, r1 = diff.sample_sharded(model, n=5, image_channels=3)
    assert torch.equal(x1, xq) and torch.equal(r1, rq)


def test_bench_entry_point_two_ranks_dry_run(A, tmp_path):
    """The driver's multi-GPU command line, rehearsed: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`
    with both ranks on the one GPU over gloo (AFD_DIST_BACKEND).  One JSON line from rank 0 with the whole-job fields right.
    (No hardware scaling number comes out of this: two ranks share one card.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0", AFD_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--no-sample", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 512 and d["config"]["parallelism"] == "dp2"
    assert math.isfinite(d["final_loss"]) and d["value"] > 0 and abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    assert d["metric"] == "train_step_images_per_sec" and d["unit"] == "images/s" and d["vs_baseline"] is None


def test_conditional_train_step_updates_label_embedding(A):
    """TrainStep(conditional=True): labels go to UNet.forward(x, t, y) (ddpm_models.py:276-277) and label_emb is inside the
    optimised / exchanged range; with the default (the reference's loop passes no labels, ddpm_utils.py:502) it stays outside
    and passing y is refused instead of silently never training the table."""
    afdm, dev = A
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device=dev, variant=3, num_classes=10).to(dev)
    diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
    g = torch.Generator().manual_seed(3)
    images = (torch.rand(4, 3, 32, 32, generator=g) * 2 - 1).to(dev)
    y = torch.tensor([1, 7, 7, 3], device=dev)
    step = afdm.TrainStep(model, diff, lr=3e-4, graph=False, conditional=True)
    assert step.opt.fp.n_active == step.opt.fp.numel                       # nothing sits outside the optimised range
    before = model.label_emb.weight.detach().clone()
    loss = step(images, y=y)
    assert math.isfinite(loss.item())
    # AdamW with a dense gradient: every row decays by (1 - lr * wd); the rows of the labels seen also take the Adam step (~lr)
    decay_only = before * (1 - 3e-4 * 0.01)
    dev_from_decay = (model.label_emb.weight.detach() - decay_only).abs().max(dim=1).values
    assert [bool(v > 1e-4) for v in dev_from_decay.tolist()] == [i in (1, 3, 7) for i in range(10)]
    assert float(dev_from_decay[[0, 2, 4, 5, 6, 8, 9]].max()) < 1e-6
    step2 = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
    assert step2.opt.fp.n_active == step2.opt.fp.numel - model.label_emb.weight.numel()
    with pytest.raises(ValueError, match="conditional=False"):
        step2(images, y=y)
