"""Minimal host-side image I/O (modules/utils.py:36-40 `save_images`) without torchvision."""
import math

import torch


def make_grid(images, nrow=8, padding=2):
    """uint8 (N,C,H,W) -> (C, H', W') grid, same layout rule as torchvision.utils.make_grid."""
    images = images.detach().cpu()
    if images.shape[1] == 1:
        images = images.repeat(1, 3, 1, 1)
    n, c, h, w = images.shape
    xmaps = min(nrow, n)
    ymaps = int(math.ceil(n / xmaps))
    H, W = h + padding, w + padding
    grid = torch.zeros(c, H * ymaps + padding, W * xmaps + padding, dtype=images.dtype)
    k = 0
    for yy in range(ymaps):
        for xx in range(xmaps):
            if k >= n:
                break
            grid[:, yy * H + padding:yy * H + padding + h, xx * W + padding:xx * W + padding + w] = images[k]
            k += 1
    return grid


def save_images(images, path, **kwargs):
    from PIL import Image
    grid = make_grid(images, **kwargs)
    Image.fromarray(grid.permute(1, 2, 0).numpy()).save(path)
