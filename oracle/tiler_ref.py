"""ORACLE (test infrastructure only - never imported by the product path).

numpy restatement of the reference's tiled-patch inference loop
(src/utils.py:353-454) and its helpers (normalize :159-171, pad :174-181,
add_gaussian_noise :29-36, get_gaussian_weights :314-350), plus the PSNR the
reference gets from skimage (src/utils.py:134-156; PSNR = 10 log10(R^2/MSE) in
float64).  `model` is any callable NCHW float32 torch tensor -> tensor (the
oracle models in this directory, on CPU).
"""
from __future__ import annotations

import numpy as np
import torch


def to_unit_range(img: np.ndarray) -> np.ndarray:
    """utils.py:159-171."""
    if img.dtype == np.uint16:
        return (img.astype(np.float32) / 65535.0).astype(np.float32)
    if img.dtype == np.uint8:
        return (img.astype(np.float32) / 255.0).astype(np.float32)
    peak = np.max(img)
    if peak > 1.0:
        return (img.astype(np.float32) / peak).astype(np.float32)
    return img.astype(np.float32)


def tile_origins(extent: int, patch: int, overlap: int) -> list:
    """utils.py:385-388: stride = max(patch-overlap,1); last tile flush with the edge."""
    stride = max(patch - overlap, 1)
    return list(range(0, extent - patch, stride)) + [max(extent - patch, 0)]


def gaussian_window(h: int, w: int, channels: int, sigma_scale: float = 0.125) -> np.ndarray:
    """utils.py:314-350: float64 maths, centre at size/2, sigma = size/8, cast to f32."""
    yy = (np.arange(h, dtype=np.float64) - h / 2.0) ** 2 / (2.0 * (h * sigma_scale) ** 2)
    xx = (np.arange(w, dtype=np.float64) - w / 2.0) ** 2 / (2.0 * (w * sigma_scale) ** 2)
    g = np.exp(-(yy[:, None] + xx[None, :]))
    return np.repeat(g[:, :, None], channels, axis=2).astype(np.float32)


def degrade(tile: np.ndarray, sigma: float) -> np.ndarray:
    """utils.py:29-36: seed 0 for every tile; f64 noise added into the f32 tile."""
    np.random.seed(0)
    tile = tile.copy()
    tile += np.random.normal(0, sigma / 255.0, tile.shape)
    return np.clip(tile, 0, 1).astype(np.float32)


def reflect_pad8(x: torch.Tensor, factor: int = 8) -> torch.Tensor:
    """utils.py:174-181: reflect pad right/bottom to the next multiple of 8."""
    h, w = x.shape[-2:]
    ph = (h // factor + 1) * factor - h if h % factor else 0
    pw = (w // factor + 1) * factor - w if w % factor else 0
    return torch.nn.functional.pad(x, (0, pw, 0, ph), mode="reflect")


def tiled_inference(model, image: np.ndarray, patch_size=None, patch_overlap=32,
                    need_degradation=False, noise_level=None, pad=None, normalize=None,
                    postprocess=None, collect_tiles=None) -> np.ndarray:
    """utils.py:353-454 without the timer.  Returns an array of image.dtype."""
    x = (normalize or to_unit_range)(image)
    h, w = x.shape[:2]
    if patch_size:
        ps = min(patch_size, max(h, w))
        ys, xs = tile_origins(h, ps, patch_overlap), tile_origins(w, ps, patch_overlap)
    else:
        ps, ys, xs = max(h, w), [0], [0]
    c_out = min(3, x.shape[2])
    acc = np.zeros((h, w, c_out), np.float32)
    wsum = np.zeros((h, w, c_out), np.float32)
    win = gaussian_window(ps, ps, c_out)
    with torch.no_grad():
        for y0 in ys:
            for x0 in xs:
                tile = x[y0:y0 + ps, x0:x0 + ps, :].copy()
                if need_degradation and noise_level is not None:
                    tile = degrade(tile, noise_level)
                t = torch.from_numpy(np.ascontiguousarray(tile.transpose(2, 0, 1)))[None]
                if pad is not None:
                    th, tw = t.shape[-2:]
                    o = model(pad(t))[:, :, :th, :tw]
                else:
                    o = model(t)
                if postprocess is not None:
                    o = postprocess(o)
                pred = o[0].numpy().transpose(1, 2, 0)
                if collect_tiles is not None:
                    collect_tiles.append(pred.copy())
                ch, cw = pred.shape[:2]
                acc[y0:y0 + ch, x0:x0 + cw] += pred * win[:ch, :cw]
                wsum[y0:y0 + ch, x0:x0 + cw] += win[:ch, :cw]
    acc /= np.maximum(wsum, 1e-8)
    if image.dtype == np.uint16:
        return np.clip(acc * 65535.0, 0, 65535).round().astype(np.uint16)
    if image.dtype == np.uint8:
        return np.clip(acc * 255.0, 0, 255).round().astype(np.uint8)
    lo, hi = np.min(image), np.max(image)
    return np.clip(acc * hi, lo, hi).astype(image.dtype)


def psnr(target: np.ndarray, pred: np.ndarray, data_range=None) -> float:
    """skimage.metrics.peak_signal_noise_ratio as called at utils.py:146:
    float64 MSE over all elements, 10*log10(R^2/MSE)."""
    if data_range is None:
        data_range = 255 if pred.dtype == np.uint8 else 65535 if pred.dtype == np.uint16 else 1.0
    err = np.mean((target.astype(np.float64) - pred.astype(np.float64)) ** 2)
    return float("inf") if err == 0 else float(10.0 * np.log10(data_range ** 2 / err))
