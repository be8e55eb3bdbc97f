#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE on CPU.

Run (in the build container only -- /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it does: imports /root/reference/modules/{filtrs,ddpm_utils,ddpm_models} with empty
stand-in modules for the two imports the image lacks and the hot path never calls
(`torchvision`, `imageio` -- SURVEY.md section 8c), seeds exactly like the reference's
`set_seed`, and dumps inputs + expected outputs as small .npz files.  Only DATA is stored
(inputs, outputs, checksums, key/shape lists) -- never reference source.

Versions are recorded inside every fixture (`_versions`): parity is pinned against
torch / scipy / numpy of THIS image, because the reference has no tests of its own.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import scipy
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    import matplotlib
    matplotlib.use("Agg")
    for name in ("torchvision", "torchvision.transforms", "torchvision.utils",
                 "torchvision.datasets", "imageio"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.transforms"].ToPILImage = object
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.path.insert(0, REF)
    import modules.filtrs as rf
    import modules.ddpm_utils as ru
    import modules.ddpm_models as rm
    import modules.utils as rutil
    return rf, ru, rm, rutil


VERSIONS = json.dumps({"torch": torch.__version__, "numpy": np.__version__,
                       "scipy": scipy.__version__, "threads": torch.get_num_threads()})


def save(name, **arrs):
    arrs["_versions"] = np.array(VERSIONS)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def n(t):
    return t.detach().cpu().clone().numpy()


F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": np.pi / 2, "omega_c_up": np.pi / 2}


def gen_filters(rf):
    out = {}
    grid = []
    for wi, omega in enumerate((np.pi / 2, np.pi / 4, np.pi, 2.0)):
        for N in (3, 4, 6, 11):
            for bi, beta in enumerate((None, 0, 1, 2, 14)):
                k = rf.circularLowpassKernel(omega_c=omega, N=N, beta=beta)
                key = f"k_w{wi}_N{N}_b{bi}"
                out[key] = n(k)
                grid.append([key, float(omega), N, -1.0 if beta is None else float(beta)])
    out["grid"] = np.array(json.dumps(grid))
    save("filters.npz", **out)


def gen_resample(rf):
    out = {}
    cases = []
    g = torch.Generator().manual_seed(1234)
    shapes = [(2, 3, 8, 8), (1, 2, 6, 10), (1, 1, 5, 7), (2, 2, 4, 4), (1, 4, 32, 32)]
    for N, beta in ((3, 2), (4, 2), (6, None), (11, 1)):
        for omega_i, omega in enumerate((np.pi / 2, np.pi / 3)):
            k = rf.circularLowpassKernel(omega_c=omega, N=N, beta=beta)
            for si, shp in enumerate(shapes):
                if shp[-1] == 32 and N != 3:
                    continue
                tag = f"N{N}_w{omega_i}_s{si}"
                x = torch.randn(shp, generator=g)
                for op, fn in (("up", rf.custom_upsample), ("down", rf.custom_downsample)):
                    xi = x.clone().requires_grad_(True)
                    y = fn(xi, k)
                    dy = torch.randn(y.shape, generator=g)
                    (dx,) = torch.autograd.grad(y, xi, dy)
                    out[f"{op}_{tag}_y"] = n(y)
                    out[f"{op}_{tag}_dy"] = n(dy)
                    out[f"{op}_{tag}_dx"] = n(dx)
                # the filtered nonlinearity as DoubleConv_F composes it (up -> exact GELU -> down)
                k2 = rf.circularLowpassKernel(omega_c=omega * 0.9, N=N, beta=beta)
                xi = x.clone().requires_grad_(True)
                y = rf.custom_downsample(torch.nn.functional.gelu(rf.custom_upsample(xi, k)), k2)
                dy = torch.randn(y.shape, generator=g)
                (dx,) = torch.autograd.grad(y, xi, dy)
                out[f"act_{tag}_y"] = n(y)
                out[f"act_{tag}_dy"] = n(dy)
                out[f"act_{tag}_dx"] = n(dx)
                out[f"x_{tag}"] = n(x)
                out[f"ku_{tag}"] = n(k)
                out[f"kd_{tag}"] = n(k2)
                cases.append(tag)
    out["cases"] = np.array(json.dumps(cases))
    save("resample.npz", **out)


def _dump_module(prefix, mod, out):
    for kname, v in mod.state_dict().items():
        out[f"{prefix}.sd.{kname}"] = n(v)


def _fwd_bwd(mod, inputs, out, prefix, g):
    ins = [i.clone().requires_grad_(i.dtype.is_floating_point) for i in inputs]
    y = mod(*ins)
    dy = torch.randn(y.shape, generator=g)
    params = [p for p in mod.parameters()]
    fl = [i for i in ins if i.requires_grad]
    grads = torch.autograd.grad(y, fl + params, dy, allow_unused=True)
    out[f"{prefix}.y"] = n(y)
    out[f"{prefix}.dy"] = n(dy)
    for j, i in enumerate(inputs):
        out[f"{prefix}.in{j}"] = n(i)
    for j in range(len(fl)):
        out[f"{prefix}.din{j}"] = n(grads[j])
    for (kname, _), gr in zip(mod.named_parameters(), grads[len(fl):]):
        out[f"{prefix}.dsd.{kname}"] = n(gr)
    _dump_module(prefix, mod, out)


def gen_blocks(ru):
    out = {}
    g = torch.Generator().manual_seed(777)
    torch.manual_seed(777)
    B, H = 2, 8
    x4 = torch.randn(B, 4, H, H, generator=g)
    x8 = torch.randn(B, 8, H, H, generator=g)
    temb = torch.randn(B, 256, generator=g)
    fs = dict(F_SET)
    _fwd_bwd(ru.DoubleConv(4, 8), [x4], out, "dc_4_8", g)
    _fwd_bwd(ru.DoubleConv(8, 8, residual=True), [x8], out, "dc_res_8", g)
    _fwd_bwd(ru.DoubleConv(8, 4, 6), [x8], out, "dc_8_4_mid6", g)
    _fwd_bwd(ru.DoubleConv_F(4, 8, f_settings=fs), [x4], out, "dcf_4_8", g)
    _fwd_bwd(ru.DoubleConv_F(8, 8, residual=True, f_settings=fs), [x8], out, "dcf_res_8", g)
    _fwd_bwd(ru.SelfAttention(8, H), [x8], out, "sa_8_8", g)
    _fwd_bwd(ru.SelfAttention(16, 4), [torch.randn(B, 16, 4, 4, generator=g)], out, "sa_16_4", g)
    _fwd_bwd(ru.Down(4, 8), [x4, temb], out, "down_4_8", g)
    _fwd_bwd(ru.Down_F(4, 8, f_settings=fs), [x4, temb], out, "downF_4_8", g)
    _fwd_bwd(ru.Down_FF(4, 8, f_settings=fs), [x4, temb], out, "downFF_4_8", g)
    _fwd_bwd(ru.Down_FFF(4, 8, f_settings=fs), [x4, temb], out, "downFFF_4_8", g)
    xs = torch.randn(B, 4, 4, 4, generator=g)      # low-res input to Up*
    skip = torch.randn(B, 4, 8, 8, generator=g)
    _fwd_bwd(ru.Up(8, 4), [xs, skip, temb], out, "up_8_4", g)
    _fwd_bwd(ru.Up_F(8, 4, f_settings=fs), [xs, skip, temb], out, "upF_8_4", g)
    _fwd_bwd(ru.Up_FF(8, 4, f_settings=fs), [xs, skip, temb], out, "upFF_8_4", g)
    _fwd_bwd(ru.Up_FFF(8, 4, f_settings=fs), [xs, skip, temb], out, "upFFF_8_4", g)
    save("blocks.npz", **out)


def _param_checksums(model):
    rows = []
    for _, p in model.state_dict().items():
        p64 = p.double()
        rows.append([p64.sum().item(), p64.abs().sum().item()])
    return np.array(rows, dtype=np.float64)


def gen_unet(rm, rutil):
    out = {}
    meta = {}
    for variant in (0, 1, 2, 3):
        for c in (1, 3):
            tag = f"v{variant}_c{c}"
            rutil.set_seed(42)
            net = rm.UNet(c_in=c, c_out=c, image_size=32, f_settings=dict(F_SET) if variant else None,
                          device="cpu", variant=variant)
            g = torch.Generator().manual_seed(100 + variant * 10 + c)
            x = torch.randn(2, c, 32, 32, generator=g)
            t = torch.tensor([500, 37], dtype=torch.long)
            with torch.no_grad():
                y = net(x, t)
                pe = net.pos_encoding(t.unsqueeze(-1).float(), 256)
            out[f"{tag}.x"] = n(x)
            out[f"{tag}.t"] = n(t)
            out[f"{tag}.y"] = n(y)
            out[f"{tag}.posenc"] = n(pe)
            out[f"{tag}.param_checksums"] = _param_checksums(net)
            meta[tag] = {"keys": list(net.state_dict().keys()),
                         "shapes": [list(v.shape) for v in net.state_dict().values()],
                         "n_params": sum(p.numel() for p in net.parameters()),
                         "n_buffers": len(list(net.buffers()))}
    # variant 4 is constructed only for its key list / parameter count (kernels out of scope)
    rutil.set_seed(42)
    net = rm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device="cpu", variant=4)
    g = torch.Generator().manual_seed(144)
    x = torch.randn(2, 3, 32, 32, generator=g)
    t = torch.tensor([500, 37], dtype=torch.long)
    with torch.no_grad():
        y = net(x, t)
    out["v4_c3.x"], out["v4_c3.t"], out["v4_c3.y"] = n(x), n(t), n(y)
    out["v4_c3.param_checksums"] = _param_checksums(net)
    meta["v4_c3"] = {"keys": list(net.state_dict().keys()),
                     "shapes": [list(v.shape) for v in net.state_dict().values()],
                     "n_params": sum(p.numel() for p in net.parameters()), "n_buffers": 0}
    out["meta"] = np.array(json.dumps(meta))
    save("unet_fwd.npz", **out)


def gen_schedule(rm, rutil):
    out = {}
    for T in (1000, 300, 101, 11):
        d = rm.Diffusion(noise_steps=T, img_size=32, device="cpu")
        out[f"beta_{T}"], out[f"alpha_{T}"], out[f"alpha_hat_{T}"] = n(d.beta), n(d.alpha), n(d.alpha_hat)
    d = rm.Diffusion(noise_steps=1000, img_size=32, device="cpu")
    rutil.set_seed(42)
    out["t_seed42_n8"] = n(d.sample_timesteps(8))
    rutil.set_seed(42)
    out["t_seed42_n256"] = n(d.sample_timesteps(256))
    # noise_images with the CPU generator: x_t and the eps it drew
    g = torch.Generator().manual_seed(5)
    x = torch.rand(8, 3, 32, 32, generator=g) * 2 - 1
    rutil.set_seed(42)
    t = d.sample_timesteps(8)
    x_t, eps = d.noise_images(x, t)
    out["noise_x"], out["noise_t"], out["noise_xt"], out["noise_eps"] = n(x), n(t), n(x_t), n(eps)
    # one denoise update (ddpm_models.py:367-374) per step-index class, with fixed inputs.  The
    # expression below is the reference's line :374 verbatim in meaning; gen_sample() checks that
    # replaying it reproduces Diffusion.sample()'s own uint8 output.
    g = torch.Generator().manual_seed(8)
    xx, ee, nz = (torch.randn(2, 3, 8, 8, generator=g) for _ in range(3))
    out["den_x"], out["den_eps"], out["den_noise"] = n(xx), n(ee), n(nz)
    for i in (999, 500, 2, 1):
        tt = (torch.ones(2) * i).long()
        a, ah, b = d.alpha[tt][:, None, None, None], d.alpha_hat[tt][:, None, None, None], d.beta[tt][:, None, None, None]
        noise = nz if i > 1 else torch.zeros_like(xx)
        out[f"den_out_{i}"] = n(1 / torch.sqrt(a) * (xx - ((1 - a) / (torch.sqrt(1 - ah))) * ee) + torch.sqrt(b) * noise)
    v = torch.cat([torch.linspace(-1.5, 1.5, 100001), torch.tensor([-1.0, 1.0, 0.0, 0.999999, -0.999999])])
    out["quant_in"] = n(v)
    out["quant_out"] = n(((v.clamp(-1, 1) + 1) / 2 * 255).type(torch.uint8))
    save("schedule.npz", **out)


TRAIN_KEYS = {     # parameters whose full gradient / post-step value is stored (the rest: checksums); per key scheme
    True: ("outc.weight", "outc.bias", "inc.conv1.weight", "inc.norm1.weight", "sa3.mha.in_proj_bias", "sa6.ln.weight",
           "down1.emb_layer.1.bias", "up3.conv.1.conv2.weight", "bot2.norm2.bias", "sa6.mha.out_proj.weight"),
    False: ("outc.weight", "outc.bias", "inc.double_conv.0.weight", "inc.double_conv.1.weight", "sa3.mha.in_proj_bias",
            "sa6.ln.weight", "down1.emb_layer.1.bias", "up3.conv.1.double_conv.3.weight", "bot2.double_conv.4.bias",
            "sa6.mha.out_proj.weight"),
}


def gen_train_step(rm, rutil, variant=3):
    """Two reference train-step bodies (ddpm_utils.py:498-509) on CPU, B=4, c=3, for one UNet variant.
    variant 3 -> train_step.npz (Config D); variants 0 / 1 / 2 -> train_step_v{0,1,2}.npz (Configs A / B / C)."""
    out = {}
    rutil.set_seed(42)
    model = rm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET) if variant else None, device="cpu", variant=variant)
    diff = rm.Diffusion(noise_steps=1000, img_size=32, device="cpu")
    opt = torch.optim.AdamW(model.parameters(), lr=3e-4)
    mse = torch.nn.MSELoss()
    g = torch.Generator().manual_seed(42)
    images = torch.rand(4, 3, 32, 32, generator=g) * 2 - 1
    keys = TRAIN_KEYS[variant in (2, 3)]
    losses = []
    for step in range(2):
        t = diff.sample_timesteps(images.shape[0])
        x_t, noise = diff.noise_images(images, t)
        pred = model(x_t, t)
        loss = mse(noise, pred)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            out["t0"], out["eps0"], out["pred0"] = n(t), n(noise), n(pred)
            gsum = []
            for kname, p in model.named_parameters():
                g64 = p.grad.double()
                gsum.append([g64.sum().item(), g64.abs().sum().item(), g64.pow(2).sum().sqrt().item()])
            out["grad_checksums0"] = np.array(gsum)
            for kname in keys:
                out[f"grad0.{kname}"] = n(dict(model.named_parameters())[kname].grad)
        else:
            out["t1"], out["eps1"] = n(t), n(noise)
        opt.step()
        losses.append(loss.item())
        if step == 0:
            for kname in keys[:3] + (keys[5], keys[7], keys[8]):
                out[f"param1.{kname}"] = n(dict(model.named_parameters())[kname])
    out["images"] = n(images)
    out["losses"] = np.array(losses, dtype=np.float64)
    out["param_checksums_after2"] = _param_checksums(model)
    save("train_step.npz" if variant == 3 else f"train_step_v{variant}.npz", **out)


def gen_conditional(rm, rutil):
    """Conditional UNet (num_classes=10, ddpm_models.py:254-258,276-277): forward with labels and the gradient that
    reaches label_emb; plus the reference AdamW's treatment of parameters that never receive a gradient (variant 4's
    stage-level norm1, and label_emb when no labels are passed): they keep their initial value."""
    out = {}
    rutil.set_seed(42)
    net = rm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device="cpu", variant=3, num_classes=10)
    g = torch.Generator().manual_seed(321)
    x = torch.randn(4, 3, 32, 32, generator=g)
    t = torch.tensor([500, 37, 999, 1], dtype=torch.long)
    y = torch.tensor([3, 9, 3, 0], dtype=torch.long)
    out["x"], out["t"], out["labels"] = n(x), n(t), n(y)
    out["n_params"] = np.array(sum(p.numel() for p in net.parameters()))
    out["keys"] = np.array(json.dumps(list(net.state_dict().keys())))
    out["label_emb"] = n(net.label_emb.weight)
    pred = net(x, t, y)
    dy = torch.randn(pred.shape, generator=g)
    pred.backward(dy)
    out["y"], out["dy"] = n(pred), n(dy)
    out["d_label_emb"] = n(net.label_emb.weight.grad)
    out["d_outc_weight"] = n(net.outc.weight.grad)
    out["d_down1_emb_weight"] = n(net.down1.emb_layer[1].weight.grad)
    with torch.no_grad():
        out["y_uncond"] = n(net(x, t))
    # variant 4, two AdamW steps without labels: which parameters does the reference leave untouched?
    rutil.set_seed(42)
    net4 = rm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device="cpu", variant=4, num_classes=10)
    before = {k: v.clone() for k, v in net4.state_dict().items()}
    opt = torch.optim.AdamW(net4.parameters(), lr=3e-4)
    diff = rm.Diffusion(noise_steps=1000, img_size=32, device="cpu")
    imgs = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    ts, es, ls = [], [], []
    for _ in range(2):
        tt = diff.sample_timesteps(2)
        x_t, noise = diff.noise_images(imgs, tt)
        loss = torch.nn.functional.mse_loss(noise, net4(x_t, tt))
        opt.zero_grad()
        loss.backward()
        opt.step()
        ts.append(n(tt)); es.append(n(noise)); ls.append(loss.item())
    untouched = [k for k, v in net4.state_dict().items() if torch.equal(v, before[k])]
    out["v4.images"], out["v4.t"], out["v4.eps"], out["v4.losses"] = n(imgs), np.stack(ts), np.stack(es), np.array(ls)
    out["v4.untouched"] = np.array(json.dumps(untouched))
    out["v4.param_checksums_after2"] = _param_checksums(net4)
    save("conditional.npz", **out)


def gen_sample(rm, rutil):
    """Reference Diffusion.sample / revert with T=101 (100 denoise steps; snapshot at i=100) on CPU.

    All noise comes from the torch CPU generator under set_seed(7): x_T = randn(n,c,S,S), then one
    randn_like per step with i>1 -- the parity test replays exactly that stream."""
    out = {}
    for variant, c in ((3, 3), (0, 1)):
        tag = f"v{variant}_c{c}"
        rutil.set_seed(42)
        model = rm.UNet(c_in=c, c_out=c, image_size=32, f_settings=dict(F_SET) if variant else None,
                        device="cpu", variant=variant)
        diff = rm.Diffusion(noise_steps=101, img_size=32, device="cpu")
        rutil.set_seed(7)
        x, result = diff.sample(model, n=2, image_channels=c)
        out[f"{tag}.sample_x"], out[f"{tag}.sample_result"] = n(x), n(result)
        rutil.set_seed(7)
        rv = diff.revert(model, n=2, image_channels=c)
        out[f"{tag}.revert"] = n(rv)
        # pre-quantisation float trajectory, replaying the same noise stream by hand
        rutil.set_seed(7)
        with torch.no_grad():
            model.eval()
            xx = torch.randn((2, c, 32, 32))
            traj = {}
            for i in reversed(range(1, 101)):
                t = (torch.ones(2) * i).long()
                eps = model(xx, t)
                a, ah, b = diff.alpha[t][:, None, None, None], diff.alpha_hat[t][:, None, None, None], diff.beta[t][:, None, None, None]
                nz = torch.randn_like(xx) if i > 1 else torch.zeros_like(xx)
                xx = 1 / torch.sqrt(a) * (xx - ((1 - a) / (torch.sqrt(1 - ah))) * eps) + torch.sqrt(b) * nz
                if i in (100, 91, 51, 1):
                    traj[i] = xx.clone()
            model.train()
        for i, v in traj.items():
            out[f"{tag}.float_x_after_i{i}"] = n(v)
        hand = (((xx.clamp(-1, 1) + 1) / 2) * 255).type(torch.uint8)
        assert torch.equal(hand, x), "hand replay of ddpm_models.py:367-374 diverged from Diffusion.sample"
        # rotation (Config E) on the same model: theta=90 over T=101 -> 100 rotations of 90/101 deg
        if variant == 3:
            rutil.set_seed(7)
            xr, rr = diff.sample(model, n=2, image_channels=c, theta=90.0)
            out[f"{tag}.sample_theta90_x"] = n(xr)
    g = torch.Generator().manual_seed(9)
    m = torch.randn(4, 3, 32, 32, generator=g)
    out["rot_in"] = n(m)
    for ai, ang in enumerate((0.09, -0.09, 0.0225, 7.5)):
        out[f"rot_out_{ai}"] = n(rm.Diffusion.rotate_2d_matrix(m, ang))
        out[f"rot_angle_{ai}"] = np.array(ang)
    out["shift_out"] = n(rm.Diffusion.shift_2d_matrix(m, 1, 0, "cpu"))
    save("sample.npz", **out)


def gen_sample_full(rm, rutil):
    """The reference's full-length run: Diffusion.sample with T=1000 (999 denoise steps), Config D, c=3, n=2, on CPU,
    noise from the torch CPU generator under set_seed(7) as in gen_sample().  Stores the uint8 outputs and the
    pre-quantisation float x at i = 900, 500, 100 and 1 (hand replay of ddpm_models.py:367-374, checked against
    Diffusion.sample's own uint8 result)."""
    out = {}
    variant, c = 3, 3
    rutil.set_seed(42)
    model = rm.UNet(c_in=c, c_out=c, image_size=32, f_settings=dict(F_SET), device="cpu", variant=variant)
    diff = rm.Diffusion(noise_steps=1000, img_size=32, device="cpu")
    rutil.set_seed(7)
    x, result = diff.sample(model, n=2, image_channels=c)
    out["sample_x"], out["sample_result"] = n(x), n(result)
    rutil.set_seed(7)
    with torch.no_grad():
        model.eval()
        xx = torch.randn((2, c, 32, 32))
        for i in reversed(range(1, 1000)):
            t = (torch.ones(2) * i).long()
            eps = model(xx, t)
            a, ah, b = diff.alpha[t][:, None, None, None], diff.alpha_hat[t][:, None, None, None], diff.beta[t][:, None, None, None]
            nz = torch.randn_like(xx) if i > 1 else torch.zeros_like(xx)
            xx = 1 / torch.sqrt(a) * (xx - ((1 - a) / (torch.sqrt(1 - ah))) * eps) + torch.sqrt(b) * nz
            if i in (900, 500, 100, 1):
                out[f"float_x_after_i{i}"] = n(xx.clone())
        model.train()
    hand = (((xx.clamp(-1, 1) + 1) / 2) * 255).type(torch.uint8)
    assert torch.equal(hand, x), "hand replay diverged from Diffusion.sample"
    save("sample_full.npz", **out)


def main():
    torch.set_num_threads(8)
    rf, ru, rm, rutil = _import_reference()
    which = sys.argv[1:] or ["filters", "resample", "blocks", "unet", "schedule", "train", "train_variants", "conditional", "sample"]
    if "filters" in which:
        gen_filters(rf)
    if "resample" in which:
        gen_resample(rf)
    if "blocks" in which:
        gen_blocks(ru)
    if "unet" in which:
        gen_unet(rm, rutil)
    if "schedule" in which:
        gen_schedule(rm, rutil)
    if "train" in which:
        gen_train_step(rm, rutil)
    for v in (0, 1, 2):
        if f"train_v{v}" in which or "train_variants" in which:
            gen_train_step(rm, rutil, v)
    if "conditional" in which:
        gen_conditional(rm, rutil)
    if "sample" in which:
        gen_sample(rm, rutil)
    if "sample_full" in which:                  # ~10 minutes of CPU: not part of the default set
        gen_sample_full(rm, rutil)
    assert not os.path.exists(os.path.join(REF, "modules", "__pycache__")), "reference tree was written to"


if __name__ == "__main__":
    main()
