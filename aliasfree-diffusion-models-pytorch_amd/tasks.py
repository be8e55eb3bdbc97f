"""Thin counterpart of the reference's task orchestration (modules/ddpm_tasks.py) so `Train.ipynb`'s
`ddpm_run(params)` runs unmodified on the HIP engine.  Host-side glue only:
same `params` keys, run-directory layout, settings text and file names; the plotting / visual-check
cells of the reference are not reproduced (out of scope: visualisation)."""
import csv
import gc
import logging
import os

import numpy as np
import torch

from .data import get_data, get_data_MNIST, make_collage, save_dataset_MNIST, save_gen_images
from .diffusion import Diffusion
from .training import argument, set_seed, train
from .unet import UNet


def _f_settings(params):
    if params["f_kernel"] is None:
        return None
    return {"kernel_size": params["f_kernel"], "kaiser_beta": params["f_beta"],
            "omega_c_down": params["f_down"], "omega_c_up": params["f_up"]}


def _loader(dataset_name, args):
    return get_data_MNIST(args) if dataset_name == "MNIST" else get_data(args)


def ddpm_run(params):
    v, name = params["unet_v"], params["dataset"]
    args = argument()
    args.run_name = f"DDPM_Uncondtional_{name}_{v}"           # (sic) spelling is part of the directory contract
    args.epochs, args.batch_size, args.image_size = params["epochs"], params["batchsize"], params["image_size"]
    args.image_channels, args.device, args.lr = params["image_channels"], params["device"], params["lr"]
    args.noise_steps, args.image_gen_n = params["noise_steps"], params["image_gen_per_epoch"]
    args.dataset_path = params["dataset_dir"]
    cwd = os.getcwd()
    modelpath = os.path.join(cwd, f"models/DDPM_Uncondtional_{name}_{v}/ckpt_{name}_{v}.pt")
    f_settings = _f_settings(params)
    tr_dir = os.path.join(cwd, f"images/original/{name}")
    gen_dir = os.path.join(cwd, f"images/generated/{name}_{v}")
    logging.basicConfig(format="%(asctime)s - %(levelname)s: %(message)s", level=logging.INFO, datefmt="%I:%M:%S")
    seed = params["seed"]
    set_seed(seed)
    if torch.cuda.is_available():
        print("CUDA is available. Device:", torch.cuda.get_device_name(0))
    else:
        print("CUDA is not available.")

    settings = {
        "unet_v": v, "run_name": args.run_name, "epochs": args.epochs, "batch_size ": args.batch_size,
        "image_size": args.image_size, "image_channels": args.image_channels, "device": args.device, "lr": args.lr,
        "noise_steps": args.noise_steps, "image_gen_n": args.image_gen_n, "datapath": args.dataset_path,
        "modelpath": modelpath, "save_tr_data": params["save_trining"], "tr_save_path": tr_dir,
        "gen_savepath": gen_dir, "gen_per_batch": params["gen_per_batch"], "total_gen": params["gen_total"],
        "seed": seed, "collage_n_per_image": params["collage_n_per_image"], "collage_n": params["collage_n"],
        "dataset": name,
    }
    for k_out, k_in in (("kernel_size", "kernel_size"), ("kaiser_beta", "kaiser_beta"),
                        ("omega_c_down", "omega_c_down"), ("omega_c_up", "omega_c_up")):
        settings[k_out] = f_settings[k_in] if f_settings is not None else "None"
    text = "\n".join(f"{k}: {val}" for k, val in settings.items())
    print(text)
    run_dir = os.path.join(cwd, f"runs/DDPM_Uncondtional_{name}_{v}")
    os.makedirs(run_dir, exist_ok=True)
    with open(os.path.join(run_dir, f"settings_{name}_{v}.txt"), "w") as fh:
        fh.write(text)

    # smoke forward (the reference does this on the CPU; the HIP engine has no CPU path)
    net = UNet(c_in=args.image_channels, c_out=args.image_channels, image_size=args.image_size, f_settings=f_settings,
               device=args.device, variant=v).to(args.device)
    print(sum(p.numel() for p in net.parameters()))
    x = torch.randn(2, args.image_channels, args.image_size, args.image_size, device=args.device)
    with torch.no_grad():
        print(net(x, x.new_tensor([500] * x.shape[0]).long()).shape)
    del net

    # train
    set_seed(seed)
    dataloader, dataset = _loader(name, args)
    model = UNet(c_in=args.image_channels, c_out=args.image_channels, image_size=args.image_size, f_settings=f_settings,
                 device=args.device, variant=v).to(args.device)
    diffusion = Diffusion(noise_steps=args.noise_steps, img_size=args.image_size, device=args.device)
    loss_all = train(args, model_path=modelpath, dataloader=dataloader, model=model, diffusion=diffusion)
    with open(os.path.join(run_dir, f"trining_loss_MNIST_{v}.csv"), "w", newline="") as fh:     # (sic) reference file name
        csv.writer(fh).writerow(loss_all)
    torch.cuda.empty_cache()
    gc.collect()

    # reload, sample, revert
    set_seed(seed)
    model = UNet(c_in=args.image_channels, c_out=args.image_channels, image_size=args.image_size, f_settings=f_settings,
                 device=args.device, variant=v).to(args.device)
    model.load_state_dict(torch.load(modelpath, weights_only=True))
    diffusion = Diffusion(noise_steps=args.noise_steps, img_size=args.image_size, device=args.device)
    x, _ = diffusion.sample(model, n=6, image_channels=args.image_channels)
    set_seed(seed)
    denoise_img = diffusion.revert(model, n=1, image_channels=args.image_channels)

    if params["save_trining"] and name == "MNIST":
        save_dataset_MNIST(tr_dir, _loader(name, args)[1])
    else:
        print("skipped saving training dataset")
    for start in np.arange(0, params["gen_total"], params["gen_per_batch"]):
        fileno = np.arange(start, start + params["gen_per_batch"], 1)
        xg, _ = diffusion.sample(model, n=params["gen_per_batch"], image_channels=args.image_channels)
        save_gen_images(gen_dir, xg, fileno)
    make_collage(gen_dir, gen_dir, params["collage_n_per_image"], params["collage_n"], args.image_size)
    torch.cuda.empty_cache()
    gc.collect()
    return {"loss_all": loss_all, "sample": x, "revert": denoise_img, "modelpath": modelpath, "gen_dir": gen_dir}


def _load(model_data):
    args, v = model_data["args"], model_data["unet_v"]
    set_seed(model_data["seed"])
    model = UNet(c_in=args.image_channels, c_out=args.image_channels, image_size=args.image_size,
                 f_settings=model_data["f_settings"], device=args.device, variant=v).to(args.device)
    model.load_state_dict(torch.load(model_data["modelpath"], weights_only=True))
    return model, Diffusion(noise_steps=args.noise_steps, img_size=args.image_size, device=args.device), args


def rotation_results(model_data, thatas):
    """Config E sweep (ddpm_tasks.py:346-369): the same seed for every angle, so only the rotation differs.
    One batched trajectory for all angles (identical noise for every angle, as the reference's per-angle re-seeding gives);
    under torch.distributed the angles are partitioned over the ranks and rank 0 gets the gathered lists (the other ranks
    get (None, None)) -- BASELINE config 5."""
    model, diffusion, args = _load(model_data)
    set_seed(model_data["seed"])
    return diffusion.sample_rotation_sweep_sharded(model, 4, args.image_channels, list(thatas))


def shift_results(model_data, shift):
    model, diffusion, args = _load(model_data)
    x_all = []
    for sh in shift:
        set_seed(model_data["seed"])
        x_all.append(diffusion.sample_shift(model, n=4, image_channels=args.image_channels, shift=sh))
    return x_all
