"""CPU-side tests: the C-ABI library loads and exports every symbol include/afd.h declares, the
host-side entry point (filter design) is bit-exact vs the reference, the module shells reproduce the
reference's state_dict / seeded init, and the N>1 data-parallel plumbing works (gloo, world_size 2)."""
import ctypes
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_json, load_golden

F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}


def test_library_exports_every_declared_symbol():
    import afdm
    from afdm._lib import LIBPATH, parse_header
    sigs = parse_header()
    assert len(sigs) >= 40
    cdll = ctypes.CDLL(LIBPATH)
    for name in sigs:
        assert hasattr(cdll, name), f"{name} declared in include/afd.h but not exported"
    assert afdm.lib().afd_version().startswith(b"afd-hip")


def test_entry_points_reject_bad_arguments_without_a_gpu():
    import afdm
    L = afdm.lib()
    with pytest.raises(afdm.AfdError, match="N=99"):
        L.afd_lowpass_kernel(1.0, 99, 0, 0.0, np.empty(4, dtype=np.float32).ctypes.data)
    with pytest.raises(afdm.AfdError, match="NULL"):
        L.afd_filt_up2_fwd(None, None, 1, 1, 4, 4, 0, 0, None, 3, None)
    with pytest.raises(afdm.AfdError, match="ksize"):
        L.afd_conv_fwd(1, 1, None, None, 1, 1, 8, 8, 4, 4, 5, 0, None)


def test_conv_dispatch_rules_are_consistent_on_the_host():
    """The library's per-shape kernel choice (host code only): the matrix-core forms (f16x2; bf16x3 behind switches 77 / 79) take the BASELINE layers by rule, small
    batches and uncovered shapes stay on the fp32 kernels, every chosen form has a workspace to live in, and the debug
    switches turn each form off and on again."""
    import afdm
    L = afdm.lib()
    layers = [(3, 32, 32), (32, 32, 32), (64, 64, 32), (64, 32, 32), (128, 128, 16), (64, 128, 8), (256, 256, 8), (64, 64, 8),
              (128, 128, 4), (256, 256, 4), (24, 40, 8)]
    for (ci, co, S) in layers:
        for B in (1, 4, 16, 256):
            kinds = L.afd_conv3x3_weight_kinds(B, ci, co, S, S)
            assert 0 <= kinds <= 3
            for bit, dgrad in ((1, 0), (2, 1)):
                if kinds & bit:                                   # a direct pass keeps its split weights (+ row scales) in the shared workspace
                    assert L.afd_conv3x3_wino_workspace_bytes(B, ci, co, S, S, dgrad) >= 54 * ci * co >= 36 * ci * co + 4 * max(ci, co)
            form = L.afd_conv_wgrad_form(B, ci, co, S, S, 3)
            assert form in (0, 1, 3, 4)
            assert L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) >= 4 * co * ci * 9
    assert L.afd_conv3x3_weight_kinds(256, 128, 128, 16, 16) == 3 and L.afd_conv3x3_weight_kinds(4, 128, 128, 16, 16) == 0
    assert L.afd_conv3x3_weight_kinds(256, 128, 128, 4, 4) == 3 and L.afd_conv3x3_weight_kinds(256, 24, 40, 8, 8) == 0   # (4x4: the f16x2 split-K kernel)
    assert L.afd_conv3x3_weight_kinds(16, 128, 128, 4, 4) == 0                                                                # too few slices: Winograd split-K
    try:
        L.afd_debug_conv_path(75)
        assert L.afd_conv3x3_weight_kinds(256, 128, 128, 4, 4) == 0 and L.afd_conv3x3_wino_workspace_bytes(256, 128, 128, 4, 4, 0) > 0
    finally:
        L.afd_debug_conv_path(74)
    assert L.afd_conv_wgrad_form(256, 128, 128, 16, 16, 3) == 4 and L.afd_conv_wgrad_form(256, 128, 128, 4, 4, 3) == 4
    assert L.afd_conv_wgrad_form(256, 3, 32, 32, 32, 3) == 3 and L.afd_conv_wgrad_form(256, 24, 40, 8, 8, 3) == 0
    assert L.afd_conv_wgrad_form(256, 32, 96, 32, 32, 1) == 4 and L.afd_conv_wgrad_form(256, 32, 3, 32, 32, 1) == 0
    try:
        L.afd_debug_conv_path(81); L.afd_debug_conv_path(85)
        assert L.afd_conv3x3_weight_kinds(256, 128, 128, 16, 16) == 0
        assert L.afd_conv_wgrad_form(256, 128, 128, 16, 16, 3) == 1 and L.afd_conv_wgrad_form(256, 32, 96, 32, 32, 1) == 0
        assert L.afd_conv3x3_wino_workspace_bytes(256, 128, 128, 16, 16, 0) > 0        # the Winograd kernels take over
        L.afd_debug_conv_path(82)
        assert L.afd_conv3x3_weight_kinds(4, 128, 128, 16, 16) == 3                    # forced wherever the shape is covered
        L.afd_debug_conv_path(77); L.afd_debug_conv_path(79); L.afd_debug_conv_path(84)
        assert L.afd_conv3x3_weight_kinds(4, 128, 128, 16, 16) == 7                    # ... in round 2's bf16x3 arithmetic
        assert L.afd_conv_wgrad_form(256, 128, 128, 16, 16, 3) == 2
    finally:
        L.afd_debug_conv_path(80); L.afd_debug_conv_path(84); L.afd_debug_conv_path(88); L.afd_debug_conv_path(76); L.afd_debug_conv_path(78)
    assert L.afd_conv3x3_weight_kinds(256, 128, 128, 16, 16) == 3


def test_filter_design_in_the_library_is_bit_exact_vs_reference():
    import afdm
    g = load_golden("filters.npz")
    for key, omega, N, beta in golden_json(g, "grid"):
        k = afdm.circularLowpassKernel(omega, N, None if beta < 0 else beta)
        assert k.dtype == torch.float32 and np.array_equal(k.numpy(), g[key]), key
    k = afdm.circularLowpassKernel(omega_c=math.pi / 2, N=3, beta=2).numpy()
    assert abs(k.sum() - 1.0) < 1e-6 and abs(k[1, 1] - 0.37743083) < 1e-8


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_state_dict_and_seeded_init_match_reference(variant):
    import afdm
    g = load_golden("unet_fwd.npz")
    meta = golden_json(g, "meta")
    for c in ((1, 3) if variant < 4 else (3,)):
        tag = f"v{variant}_c{c}"
        afdm.set_seed(42)
        net = afdm.UNet(c_in=c, c_out=c, image_size=32, f_settings=dict(F_SET) if variant else None, device="cpu", variant=variant)
        sd = net.state_dict()
        assert list(sd.keys()) == meta[tag]["keys"]
        assert [list(v.shape) for v in sd.values()] == meta[tag]["shapes"]
        assert sum(p.numel() for p in net.parameters()) == meta[tag]["n_params"]
        cs = np.array([[v.double().sum().item(), v.double().abs().sum().item()] for v in sd.values()])
        assert np.allclose(cs, g[f"{tag}.param_checksums"], rtol=1e-12, atol=1e-12)
    assert meta["v3_c1"]["n_params"] == 5896513 and meta["v3_c3"]["n_params"] == 5897155     # Results.ipynb:121


def test_constructor_contract_and_loud_cpu_failure():
    import afdm
    with pytest.raises(ValueError, match="f_settings is empty"):
        afdm.UNet(variant=3)
    with pytest.raises(ValueError, match="variant value must be between 0 and 4"):
        afdm.UNet(variant=5)
    net = afdm.UNet(c_in=1, c_out=1, image_size=32, device="cpu", variant=0)
    with pytest.raises(RuntimeError, match="no CPU"):
        net(torch.randn(2, 1, 32, 32), torch.tensor([500, 500]))
    with pytest.raises(RuntimeError, match="HIP device"):
        afdm.custom_upsample(torch.randn(1, 1, 4, 4), afdm.circularLowpassKernel(1.0, 3))


def test_schedule_and_timesteps_bit_exact_on_host():
    import afdm
    g = load_golden("schedule.npz")
    for T in (1000, 300, 101, 11):
        d = afdm.Diffusion(noise_steps=T, img_size=32, device="cpu")
        assert np.array_equal(d.beta.numpy(), g[f"beta_{T}"])
        assert np.array_equal(d.alpha.numpy(), g[f"alpha_{T}"])
        assert np.array_equal(d.alpha_hat.numpy(), g[f"alpha_hat_{T}"])
    d = afdm.Diffusion(noise_steps=1000, img_size=32, device="cpu")
    afdm.set_seed(42)
    assert d.sample_timesteps(8).tolist() == [790, 618, 251, 50, 64, 435, 413, 942]
    afdm.set_seed(42)
    assert np.array_equal(d.sample_timesteps(256).numpy(), g["t_seed42_n256"])


def test_modules_shim_resolves_reference_import_paths():
    sys.path.insert(0, ROOT)
    from modules.ddpm_models import UNet, Diffusion          # Results.ipynb:33-39
    from modules.ddpm_utils import train, argument, DoubleConv_F, Down_FFF, SelfAttention   # noqa: F401
    from modules.filtrs import circularLowpassKernel, custom_upsample, custom_downsample     # noqa: F401
    from modules.utils import set_seed, setup_logging                                      # noqa: F401
    from modules.ddpm_tasks import ddpm_run, rotation_results                               # noqa: F401  (Train.ipynb:23)
    import afdm
    assert UNet is afdm.UNet and Diffusion is afdm.Diffusion


def test_stage_classes_are_real_module_subclasses():
    """ddpm_utils.py:199-480 define Down / Up / *_F / *_FF / *_FFF as nn.Module classes: isinstance and subclassing work."""
    import afdm
    fs = dict(F_SET)
    d, u = afdm.Down_FFF(4, 8, f_settings=fs), afdm.Up_FF(8, 4, f_settings=fs)
    assert isinstance(d, afdm.Down_FFF) and isinstance(d, torch.nn.Module) and not isinstance(d, afdm.Down_FF)
    assert isinstance(u, afdm.Up_FF) and not isinstance(u, afdm.Up_FFF)
    assert isinstance(afdm.Down(4, 8), afdm.Down) and isinstance(afdm.Up(8, 4), afdm.Up)

    class MyDown(afdm.Down_F):
        pass
    assert isinstance(MyDown(4, 8, f_settings=fs), afdm.Down_F)
    with pytest.raises(ValueError, match="f_settings is empty"):
        afdm.Down_F(4, 8)
    net = afdm.UNet(c_in=1, c_out=1, image_size=32, f_settings=fs, device="cpu", variant=3)
    assert isinstance(net.down1, afdm.Down_FFF) and isinstance(net.up3, afdm.Up_FFF) and isinstance(net.inc, afdm.DoubleConv_F)


def test_flat_params_keep_unused_parameters_out_of_the_optimiser_range():
    """The reference's AdamW skips parameters whose grad is None (variant 4's stage-level norm1, label_emb without labels):
    FlatParams puts them behind n_active, where the fused AdamW launch never reaches."""
    import afdm
    afdm.set_seed(0)
    net = afdm.UNet(c_in=1, c_out=1, image_size=32, f_settings=dict(F_SET), device="cpu", variant=4, num_classes=5)
    unused = net.unused_parameters()
    n_unused = sum(p.numel() for p in unused)
    assert n_unused == 2 * (32 + 64 + 128 + 128 + 64 + 32) + 5 * 256
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    fp = afdm.FlatParams(net)
    assert fp.n_active == fp.numel - n_unused
    tail = {id(p) for p in fp.params[len(fp.params) - len(unused):]}
    assert tail == {id(p) for p in unused}
    assert all(p.data_ptr() == fp.flat.data_ptr() + 4 * o for p, o in zip(fp.params, fp.offsets))
    assert all(o + p.numel() <= fp.n_active for p, o in zip(fp.params, fp.offsets) if id(p) not in tail)
    assert all(torch.equal(v, sd0[k]) for k, v in net.state_dict().items())       # re-homing moved storage, not values
    assert afdm.FlatParams(torch.nn.Linear(3, 2)).n_active == 8                    # models without the hook: everything active


def _write_mnist_csv(path, n=24):
    rng = np.random.default_rng(0)
    arr = np.concatenate([rng.integers(0, 10, (n, 1)), rng.integers(0, 256, (n, 784))], axis=1)
    header = ",".join(["label"] + [f"p{i}" for i in range(784)])
    np.savetxt(path, arr, fmt="%d", delimiter=",", header=header, comments="")


def test_loaders_without_torchvision(tmp_path):
    import afdm
    from PIL import Image
    csvp = tmp_path / "mnist.csv"
    _write_mnist_csv(csvp)
    a = afdm.argument(batch_size=8, dataset_path=str(csvp))
    dl, ds = afdm.get_data_MNIST(a)
    xb, yb = next(iter(dl))
    assert tuple(xb.shape) == (8, 1, 32, 32) and xb.min() >= -1.0 - 1e-6 and xb.max() <= 1.0 + 1e-6 and len(ds) == 24
    root = tmp_path / "imgs"
    for c in ("a", "b"):
        (root / c).mkdir(parents=True)
        for i in range(3):
            Image.fromarray(np.random.default_rng(i).integers(0, 255, (40, 48, 3), dtype=np.uint8)).save(root / c / f"{i}.png")
    a = afdm.argument(batch_size=4, dataset_path=str(root), image_size=32)
    dl, ds = afdm.get_data(a)
    assert len(ds) == 6 and ds[0][0].shape[0] == 3 and min(ds[0][0].shape[1:]) == 32
    out = tmp_path / "gen"
    afdm.save_gen_images(str(out), torch.randint(0, 255, (4, 3, 32, 32), dtype=torch.uint8), np.arange(4))
    afdm.make_collage(str(out), str(out), 4, 4, 32)
    assert (out / "image_3.png").exists() and os.path.exists(str(out) + "_collage_0.png")


def _run_workers(tmp_path, mode, port, name):
    out = tmp_path / name
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_worker.py"), "--device", "cpu", "--mode", mode,
           "--out", str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=300)
    return torch.load(out, weights_only=True)


def test_two_rank_gloo_gradient_exchange(tmp_path):
    """world_size 2: module-boundary buckets, readiness bookkeeping (a bucket's all-reduce starts when its last
    gradient is reported, i.e. during backward), the everything-at-the-end form, and the mean."""
    res = _run_workers(tmp_path, "grads", 29531, "ddp_cpu.pt")
    assert res["ok"] and res["world"] == 2


def test_two_rank_gloo_shard_and_gather(tmp_path):
    """world_size 2: the partition / gather-on-rank-0 helpers of the sharded sampling paths (uneven and empty shards)."""
    res = _run_workers(tmp_path, "shards", 29532, "shards_cpu.pt")
    assert res["ok"] and res["world"] == 2


def test_unet_gradient_buckets_follow_the_layer_order():
    """Config D: 4 buckets cut at top-level module boundaries from the output end; the bucket that becomes ready last
    (inc .. sa2) is the smallest."""
    import afdm
    afdm.set_seed(0)
    net = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device="cpu", variant=3)
    fp = afdm.FlatParams(net)
    ddp = afdm.GradAllReduce(fp, n_buckets=4, model=net)
    assert ddp.slices == [(0, 554144), (554144, 2163232), (2163232, 3786784), (3786784, 5897155)]
    assert sum(ddp._count) == len(fp.params) and ddp.world == 1 and ddp.finish() == 1.0
