"""Host-side data / image I/O of the reference's `modules/utils.py`, without torchvision
(SURVEY.md section 8f-1).  Not on the timed path."""
import math
import os
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader, Dataset, TensorDataset

IMG_EXT = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def get_data_MNIST(args):
    """CSV loader (utils.py:55-82): column 0 = label, 784 pixel columns; resize 28 -> 32 (bilinear,
    antialiased like torchvision's tensor Resize), normalise to [-1, 1]."""
    import pandas as pd
    data = pd.read_csv(args.dataset_path)
    labels = torch.tensor(data.iloc[:, 0].values, dtype=torch.long)
    feats = torch.tensor(data.iloc[:, 1:].values / 255.0, dtype=torch.float32).view(-1, 1, 28, 28)
    feats = F.interpolate(feats, size=(32, 32), mode="bilinear", antialias=True, align_corners=False)
    feats = (feats - 0.5) / 0.5
    dataset = TensorDataset(feats, labels)
    return DataLoader(dataset, batch_size=args.batch_size, shuffle=True), dataset


class ImageFolder(Dataset):
    """root/<class>/<image> like torchvision.datasets.ImageFolder (utils.py:43-52): RGB, shorter side
    resized to `size` (bilinear), scaled to [0,1], normalised with mean = std = 0.5."""

    def __init__(self, root, size):
        self.size = size
        classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
        if not classes:
            raise FileNotFoundError(f"Couldn't find any class folder in {root}.")
        self.class_to_idx = {c: i for i, c in enumerate(classes)}
        self.samples = []
        for c in classes:
            for dirpath, _, files in sorted(os.walk(os.path.join(root, c))):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXT):
                        self.samples.append((os.path.join(dirpath, f), self.class_to_idx[c]))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        from PIL import Image
        path, label = self.samples[i]
        img = Image.open(path).convert("RGB")
        w, h = img.size
        s = self.size
        if w <= h:
            nw, nh = s, max(1, int(s * h / w))
        else:
            nw, nh = max(1, int(s * w / h)), s
        img = img.resize((nw, nh), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        return (x - 0.5) / 0.5, label


def get_data(args):
    dataset = ImageFolder(args.dataset_path, args.image_size)
    return DataLoader(dataset, batch_size=args.batch_size, shuffle=True), dataset


def _to_pil(img):
    """uint8 / float (C,H,W) or (H,W) tensor -> PIL image (torchvision ToPILImage semantics)."""
    from PIL import Image
    t = img.detach().cpu()
    if t.dtype != torch.uint8:
        t = (t * 255).to(torch.uint8)          # ToPILImage: float in [0,1] -> mul(255).byte()
    if t.dim() == 3 and t.shape[0] == 1:
        t = t[0]
    if t.dim() == 2:
        return Image.fromarray(t.numpy(), mode="L")
    return Image.fromarray(t.permute(1, 2, 0).numpy())


def save_gen_images(path_str, data, fileno):
    """`<path>/image_<n>.png`, one file per generated image (utils.py:175-198)."""
    save_dir = Path(path_str)
    save_dir.mkdir(parents=True, exist_ok=True)
    for i in range(data.shape[0]):
        _to_pil(data[i]).save(save_dir / f"image_{fileno[i]}.png", format="PNG")


def save_dataset_MNIST(path_str, dataset):
    save_dir = Path(path_str)
    save_dir.mkdir(parents=True, exist_ok=True)
    for i, (image, _) in enumerate(dataset):
        _to_pil(image).save(save_dir / f"image_{i}.png", format="PNG")


def make_collage(filedir, savedir, images_per_collage, total_image, image_size):
    """sqrt(n) x sqrt(n) RGB collages `<savedir>_collage_<start>.png` (utils.py:208-234)."""
    from PIL import Image
    per = int(math.sqrt(images_per_collage))
    side = int(image_size * math.sqrt(images_per_collage))
    for start in np.arange(0, total_image, images_per_collage):
        files = [f"{filedir}/image_{i}.png" for i in np.arange(start, start + images_per_collage, 1)]
        images = [Image.open(f).resize((image_size, image_size)) for f in files]
        collage = Image.new("RGB", (side, side))
        for i in range(per):
            for j in range(per):
                collage.paste(images[i * per + j], (i * image_size, j * image_size))
        collage.save(savedir + f"_collage_{start}.png")
