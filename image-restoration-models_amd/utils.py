"""Glue API with the reference's call surface (src/utils.py): model factory,
patch-config lookup, the tiled-patch inference loop and the metrics.

The hot loop (reference: src/utils.py:353-454, one synchronous forward + D2H +
numpy blend per tile) runs here as ONE device pipeline per image:
``irm_tile_extract`` (normalise / seeded noise / reflect pad) -> batched model
forward over all tiles -> ``irm_window_blend`` (Gaussian blend, /weight,
requantise) -> one D2H of the uint8 result.  Results equal the reference loop's
given equal per-tile predictions (same float32 operation order).
"""
from __future__ import annotations

import os
import time
from typing import Callable, Literal

import numpy as np
import torch
from torch.nn import Module

from . import _hip, deblurganv2, dncnn, mair, ops, rednet, restormer
from .configs import PATCH_CONFIG, ROOT_RESULTS_DIR, ROOT_WEIGHTS_DIR
from .dncnn import DnCNN
from .rednet import REDNet
from .restormer import Restormer
from .mair import MaIR, MaIRUNet
from .deblurganv2 import FPNMobileNet

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))

#: model classes whose forward is the HIP path (isinstance dispatch as in utils.py:280/292)
_PAD8_MODELS = (Restormer, MaIR, MaIRUNet)


def get_model_total_parameters(model: Module) -> int:
    return sum(p.numel() for p in model.parameters())


def add_gaussian_noise(img: np.ndarray, sigma: int | float = 15):
    """src/utils.py:29-36 (seed 0 on every call, float64 noise added into a float32 array)."""
    if img.dtype != np.float32 and img.dtype != np.float64:
        img = img.astype(np.float32) / 255.
    np.random.seed(seed=0)
    img += np.random.normal(0, sigma / 255., img.shape)
    return np.clip(img, 0, 1).astype(np.float32)


def normalize(img: np.ndarray):
    """src/utils.py:159-171."""
    if img.dtype == np.uint16:
        out = img.astype(np.float32) / 65535.0
    elif img.dtype == np.uint8:
        out = img.astype(np.float32) / 255.0
    else:
        peak = np.max(img)
        out = img.astype(np.float32) / peak if peak > 1.0 else img.astype(np.float32)
    return out.astype(np.float32)


def pad(x: torch.Tensor, downscale_factor: int = 8):
    """src/utils.py:174-181: reflect-pad right/bottom to the next multiple of the factor."""
    h, w = x.shape[-2:]
    padh = (h // downscale_factor + 1) * downscale_factor - h if h % downscale_factor else 0
    padw = (w // downscale_factor + 1) * downscale_factor - w if w % downscale_factor else 0
    return torch.nn.functional.pad(x, (0, padw, 0, padh), 'reflect')


def get_gaussian_weights(height: int, width: int, n_channels=3, sigma_scale=0.125):
    """src/utils.py:314-350: Gaussian blending window, float64 maths, centre at size/2."""
    yy = (np.arange(height) - height / 2.0) ** 2 / (2 * (height * sigma_scale) ** 2)
    xx = (np.arange(width) - width / 2.0) ** 2 / (2 * (width * sigma_scale) ** 2)
    g = np.exp(-(yy[:, None] + xx[None, :]))
    return np.repeat(g[:, :, np.newaxis], n_channels, axis=2).astype(np.float32)


def get_patch_config(task, subtask, model_name) -> dict | None:
    """src/utils.py:184-213."""
    model_key = model_name.split(' ')[0]
    config = PATCH_CONFIG.get(model_key, None)
    if isinstance(config, list):
        if model_key == 'DeblurGANv2':
            config = config[0] if 'Inception' in model_name else config[1]
        elif model_key == 'MaIR':
            config = config[0] if subtask.lower() == 'gaussian' else config[1]
        elif model_key == 'Restormer':
            config = config[0] if task.lower() == 'denoising' else config[1]
        else:
            config = config[0]
    return config


def _restormer_opt(name: str) -> str:
    return os.path.join(_PKG_DIR, 'restormer', 'options', name + '.yml')


def get_model_instance(task, subtask, model_name, device: torch.device, gray=False,
                       sigma: int | float | None = None) -> torch.nn.Module:
    """src/utils.py:216-267: same (task, subtask, model, gray, sigma) -> weights table."""
    model_key = model_name.split(' ')[0]
    if model_key == 'REDNet':
        if task == 'denoising' and subtask == 'gaussian' and sigma is not None:
            return rednet.get_model(f'{ROOT_WEIGHTS_DIR}/REDNet/{sigma}.pt', device)
    elif model_key == 'DnCNN':
        if task == 'denoising' and subtask == 'gaussian':
            if gray:
                if sigma is not None:
                    return dncnn.get_model(f'{ROOT_WEIGHTS_DIR}/DnCNN/dncnn_{sigma}.pth', 1, 17, device)
                return dncnn.get_model(f'{ROOT_WEIGHTS_DIR}/DnCNN/dncnn_gray_blind.pth', 1, 20, device)
            if sigma is None:
                return dncnn.get_model(f'{ROOT_WEIGHTS_DIR}/DnCNN/dncnn_color_blind.pth', 3, 20, device)
    elif model_key == 'Restormer':
        kind = 'Gray' if gray else 'Color'
        if task == 'denoising':
            if subtask == 'gaussian':
                suffix = f'Sigma{sigma}' if sigma is not None else ''
                return restormer.get_model(_restormer_opt(f'Gaussian{kind}Denoising_Restormer{suffix}'), device)
            if subtask == 'real':
                return restormer.get_model(_restormer_opt('RealDenoising_Restormer'), device)
        if task == 'deblurring':
            if subtask == 'defocus':
                if 'Dual-pixel' in model_name:
                    return restormer.get_model(_restormer_opt('DefocusDeblur_DualPixel_16bit_Restormer'), device)
                return restormer.get_model(_restormer_opt('DefocusDeblur_Single_8bit_Restormer'), device)
            if subtask == 'motion':
                return restormer.get_model(_restormer_opt('Deblurring_Restormer'), device)
    elif model_key == 'MaIR':
        opt_dir = os.path.join(_PKG_DIR, 'mair', 'options')
        if task == 'denoising':
            if subtask == 'gaussian' and not gray and sigma is not None:
                return mair.get_model(os.path.join(opt_dir, f'test_MaIR_CDN_s{sigma}.yml'))
            if subtask == 'real':
                return mair.get_model(os.path.join(opt_dir, 'test_MaIR_RealDN.yml'))
        if task == 'deblurring' and subtask == 'motion':
            return mair.get_model(os.path.join(opt_dir, 'test_MaIR_MotionDeblur.yml'))
    elif model_key == 'DeblurGANv2':
        if task == 'deblurring' and subtask == 'motion':
            if 'Inception' in model_name:
                return deblurganv2.get_model(f'{ROOT_WEIGHTS_DIR}/DeblurGANv2/fpn_inception.h5', device)
            if 'MobileNet' in model_name:
                return deblurganv2.get_model(f'{ROOT_WEIGHTS_DIR}/DeblurGANv2/fpn_mobilenet.h5', device)
    raise ValueError('No model instance found for current configuration.')


# ---------------------------------------------------------------------------
# metrics (src/utils.py:134-156; skimage is restated)
# ---------------------------------------------------------------------------

def psnr(target: np.ndarray, pred: np.ndarray, data_range) -> float:
    err = np.mean((np.asarray(target, dtype=np.float64) - np.asarray(pred, dtype=np.float64)) ** 2)
    return float('inf') if err == 0 else float(10 * np.log10((data_range ** 2) / err))


def ssim(target: np.ndarray, pred: np.ndarray, data_range, channel_axis=None) -> float:
    """structural_similarity with skimage's defaults (7x7 uniform window, K1=.01,
    K2=.03, sample covariance, border crop).  Restated from the published
    algorithm: parity with skimage is unpinned (skimage is not installed here)."""
    from scipy.ndimage import uniform_filter
    if channel_axis is not None:
        vals = [ssim(np.take(target, i, axis=channel_axis), np.take(pred, i, axis=channel_axis), data_range)
                for i in range(target.shape[channel_axis])]
        return float(np.mean(vals))
    x, y = target.astype(np.float64), pred.astype(np.float64)
    win, npx = 7, 49
    cov_norm = npx / (npx - 1)
    ux, uy = uniform_filter(x, win), uniform_filter(y, win)
    uxx, uyy, uxy = uniform_filter(x * x, win), uniform_filter(y * y, win), uniform_filter(x * y, win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    p = (win - 1) // 2
    return float(s[p:-p, p:-p].mean())


def calculate_metrics(pred: np.ndarray, target: np.ndarray, data_range=None):
    """PSNR and SSIM between prediction and target (src/utils.py:134-156)."""
    if data_range is None:
        data_range = 255 if pred.dtype == np.uint8 else 65535 if pred.dtype == np.uint16 else 1.0
    psnr_value = psnr(target, pred, data_range)
    if pred.ndim == 3 and pred.shape[2] == 3:
        ssim_value = ssim(target, pred, data_range, channel_axis=2)
    elif pred.ndim == 3 and pred.shape[2] == 1:
        ssim_value = ssim(target[:, :, 0], pred[:, :, 0], data_range)
    else:
        ssim_value = ssim(target, pred, data_range)
    return psnr_value, ssim_value


# ---------------------------------------------------------------------------
# tiled-patch inference
# ---------------------------------------------------------------------------

def tile_origins(extent: int, patch: int, overlap: int) -> list:
    """src/utils.py:385-388."""
    stride = max(patch - overlap, 1)
    return list(range(0, extent - patch, stride)) + [max(extent - patch, 0)]


_WINDOW_CACHE: dict = {}
_SIDE_STREAMS: dict = {}


def _side_stream(device, index: int):
    key = (str(device), index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def _window_on(device, ps: int) -> torch.Tensor:
    key = (str(device), ps)
    if key not in _WINDOW_CACHE:
        _WINDOW_CACHE[key] = torch.from_numpy(get_gaussian_weights(ps, ps, 1)[:, :, 0].copy()).to(device)
    return _WINDOW_CACHE[key]


#: bound of the per-model graph cache: entries, and bytes of graph-owned memory (static buffers + workspaces)
_GRAPH_MAX_ENTRIES, _GRAPH_MAX_BYTES = 8, 96 << 30


def _weights_version(model: Module):
    return _hip.param_key(model)


def _release_workspace(model: Module):
    rel = getattr(model, "release_workspace", None)
    if callable(rel):
        rel()


def graphed_forward(model: Module, x: torch.Tensor) -> torch.Tensor:
    """model(x) replayed from a HIP graph where that pays: the conv stacks and MaIR at small images are bound by
    the host's launch rate (hundreds of kernels of 10-60 us per image), not by the GPU.

    Opt-in per model (`model.hip_graph = True`, the default of the built-in model classes); off while a KernelTimer
    is installed (per-launch events need eager launches) and with IRM_NO_GRAPH=1.  The first call for an (input
    shape, stream, weights version) runs eagerly once (weight packing, LDS attributes), then captures the forward on
    torch's capture stream - every kernel of libirm_hip.so is enqueued on torch's current stream and allocates
    nothing, so the capture holds plain kernel nodes; later calls copy the input into the graph's static buffer and
    replay.  The returned tensor is the graph's static output: consume it (on the same stream) before the next call,
    as the tiler does.

    Ownership: a graph's kernels hold raw pointers, so every buffer they touch must live exactly as long as the
    graph.  The model's workspace is therefore dropped before the capture (`release_workspace()`), re-allocated
    INSIDE it - from the graph's private memory pool - and dropped again afterwards: the pool keeps those blocks for
    the graph alone, and no later eager call, other input shape or other graph can be handed the same memory
    (DnCNN's eagerly allocated layer buffers were freed on a shape change while an older graph still wrote
    to them).  The graphs live on the model object (`model._irm_graphs`), so they die with it; entries of other
    weight versions are dropped, the rest is bounded (least recently used)."""
    if (ops.TIMER is not None or not getattr(model, "hip_graph", False) or os.environ.get("IRM_NO_GRAPH")
            or not x.is_cuda or torch.cuda.is_current_stream_capturing()):
        return model(x)
    graphs = model.__dict__.get("_irm_graphs")
    if graphs is None:
        graphs = model.__dict__["_irm_graphs"] = {}
    version = _weights_version(model)
    key = (tuple(x.shape), x.device.index, torch.cuda.current_stream().cuda_stream, version)
    ent = graphs.pop(key, None)
    if ent is None:
        for k in [k for k in graphs if k[3] != version]:
            del graphs[k]                            # stale weights
        model(x)                                     # eager warm-up (packs weights, sets kernel attributes)
        _release_workspace(model)
        static_in = x.clone()
        g = torch.cuda.CUDAGraph()
        before = torch.cuda.memory_reserved(x.device)
        with torch.cuda.graph(g):
            static_out = model(static_in)
        _release_workspace(model)                    # the buffers stay reserved in g's private pool
        ent = (g, static_in, static_out, max(torch.cuda.memory_reserved(x.device) - before, 0))
        while graphs and (len(graphs) >= _GRAPH_MAX_ENTRIES
                          or sum(e[3] for e in graphs.values()) + ent[3] > _GRAPH_MAX_BYTES):
            del graphs[next(iter(graphs))]           # least recently used first (dict order = recency, see below)
    graphs[key] = ent                                # (re-)insert at the end: most recently used
    g, static_in, static_out, _ = ent
    static_in.copy_(x)
    g.replay()
    return static_out


def tiled_forward_device(model: Module, img_dev: torch.Tensor, patch_size, patch_overlap, pad8: bool,
                         noise_sigma=None, target_dev: torch.Tensor | None = None, max_batch: int = 8,
                         keep_tiles: list | None = None, hooks: str | None = None):
    """Device pipeline for one uint8/uint16 HWC image already on the GPU.

    Returns (out uint8/uint16 HWC device tensor, sse device tensor or None).
    Nothing here synchronises with the host.  hooks="deblurganv2" selects that model's normalize / pad /
    postprocess (src/deblurganv2/__init__.py:11-28) instead of /255 and the reflect pad to 8.
    """
    outs = tiled_forward_device_batch(model, [img_dev], patch_size, patch_overlap, pad8, noise_sigma,
                                      None if target_dev is None else [target_dev], max_batch, keep_tiles, hooks)
    return outs[0]


def tiled_forward_device_batch(model: Module, imgs_dev: list, patch_size, patch_overlap, pad8: bool,
                               noise_sigma=None, targets_dev: list | None = None, max_batch: int = 8,
                               keep_tiles: list | None = None, hooks: str | None = None) -> list:
    """The device pipeline for SEVERAL images of one shape at once (throughput serving; the reference's loop,
    src/utils.py:353-454, is one image and one tile at a time): the tiles of all images form one batch axis, so the
    low-resolution levels of the network fill the GPU and every kernel's tail is paid once per batch instead of once per
    image (Restormer, 1280x720: 53.8 ms for one frame, 52.3 ms per frame for two, tools/bench_batch.py).  Each image is
    extracted, blended, requantised and scored exactly as in the single-image call - a tile's result does not depend on
    its batch (tests/test_gpu_fullsize.py) - and the list of (out, sse) pairs is returned in order."""
    norm_mean, norm_inv_std, post_scale, post_shift = 0.0, 1.0, 1.0, 0.0
    pad_mode = "reflect8" if pad8 else "none"
    if hooks == "deblurganv2":
        norm_mean = float(np.float32(0.5) * np.float32(255.0))
        norm_inv_std = float(np.float32(1.0) / (np.float32(0.5) * np.float32(255.0)))
        post_scale, post_shift, pad_mode = 0.5, 1.0, "zero32"
    img0 = imgs_dev[0]
    h, w, c = img0.shape
    if any(tuple(im.shape) != (h, w, c) or im.dtype != img0.dtype for im in imgs_dev):
        raise ValueError("tiled_forward_device_batch: the images of a batch must share shape and dtype")
    is_u16 = img0.dtype in (torch.uint16, torch.int16)
    dev = img0.device
    if patch_size:
        ps = min(patch_size, max(h, w))
        ys, xs = tile_origins(h, ps, patch_overlap), tile_origins(w, ps, patch_overlap)
    else:
        ps, ys, xs = max(h, w), [0], [0]
    th, tw = min(ps, h), min(ps, w)
    if pad_mode == "reflect8":
        ph = (th // 8 + 1) * 8 if th % 8 else th
        pw = (tw // 8 + 1) * 8 if tw % 8 else tw
    elif pad_mode == "zero32":
        ph, pw = (th // 32 + 1) * 32, (tw // 32 + 1) * 32
    else:
        ph, pw = th, tw
    origins = [(y0, x0) for y0 in ys for x0 in xs]
    T = len(origins)
    K = len(imgs_dev)
    org = torch.tensor(origins, dtype=torch.int32).to(dev, non_blocking=True)
    noise = None
    if noise_sigma is not None:
        np.random.seed(seed=0)                       # utils.py:33: same field for every tile
        noise = torch.from_numpy(np.random.normal(0, noise_sigma / 255., (th, tw, c))).to(dev)
    tiles = torch.empty(K * T, c, ph, pw, dtype=torch.float32, device=dev)
    for k, im in enumerate(imgs_dev):
        _hip.call("irm_tile_extract", _hip.ptr(im), int(is_u16), _hip.ptr(org), _hip.ptr(noise),
                  _hip.ptr(tiles[k * T:]), h, w, c, th, tw, ph, pw, T, float(norm_mean), float(norm_inv_std),
                  int(pad_mode == "zero32"))
    c_out = min(3, c)
    pred = None
    if ops.TIMER is not None:
        ops.TIMER.break_chain()                    # the tile extraction above is not a timed launch
    NT = K * T
    nstreams = min(int(getattr(model, "num_streams", 1)), NT)
    if nstreams > 1 and not os.environ.get("IRM_EXPERIMENTAL_STREAMS"):
        # EXPERIMENTAL, off in the product: on some GPUs of the pool overlapping forwards were not bit-reproducible
        # (rare stale read of an in-place updated buffer, cause not established: DESIGN.md section 6)
        raise ValueError("model.num_streams > 1 is experimental (not bit-reproducible on every GPU, DESIGN.md section 6); "
                         "set IRM_EXPERIMENTAL_STREAMS=1 to run it")
    if nstreams > 1:
        # independent tile groups on separate HIP streams: one group's HBM-bound kernels overlap the
        # other's MFMA-bound GEMMs; the groups join before the blend
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        per = -(-NT // nstreams)
        done, outs = [], []
        for gi, i in enumerate(range(0, NT, per)):
            st = _side_stream(dev, gi)
            st.wait_event(ready)
            with torch.cuda.stream(st):
                o = model(tiles[i:i + per])              # (side streams: eager)
                outs.append((i, o))
                ev = torch.cuda.Event()
                ev.record(st)
                done.append(ev)
        for ev in done:
            main.wait_event(ev)
        pred = torch.empty(NT, *outs[0][1].shape[1:], dtype=torch.float32, device=dev)
        for i, o in outs:
            pred[i:i + o.shape[0]] = o
            o.record_stream(main)
    else:
        for i in range(0, NT, max_batch):
            o = graphed_forward(model, tiles[i:i + max_batch]) if callable(getattr(model, "forward", None)) else model(tiles[i:i + max_batch])
            if pred is None:
                pred = o if o.shape[0] == NT else torch.empty(NT, *o.shape[1:], dtype=torch.float32, device=dev)
            if pred is not o:
                pred[i:i + o.shape[0]] = o
    if keep_tiles is not None:
        keep_tiles.append(pred[:, :c_out, :th, :tw].clone())
    results = []
    for k in range(K):
        out = torch.empty(h, w, c_out, dtype=img0.dtype, device=dev)
        sse, tgt = None, None
        if targets_dev is not None and targets_dev[k] is not None:
            sse, tgt = torch.zeros(1, dtype=torch.int64, device=dev), targets_dev[k]
        _hip.call("irm_window_blend", _hip.ptr(pred[k * T:]), _hip.ptr(org), _hip.ptr(_window_on(dev, ps)), _hip.ptr(out),
                  int(is_u16), _hip.ptr(tgt), _hip.ptr(sse), h, w, c_out, pred.shape[1], th, tw,
                  pred.shape[2], pred.shape[3], ps, T, post_scale, post_shift)
        results.append((out, sse))
    return results


def run_model_inference(model: Module, input_img: np.ndarray, device: torch.device,
                        normalize: Callable = normalize, patch_size: int | None = None, patch_overlap: int = 32,
                        need_degradation=False, noise_level=None, pad: Callable | None = None,
                        postprocess: Callable | None = None, progress_bar=None):
    """Run inference; returns (prediction, inference_time_ms) like src/utils.py:353-454.

    uint8/uint16 images with the stock normalize/pad hooks take the device
    pipeline; anything else (float images, custom hooks) takes a per-tile loop
    with the reference's host-side blend - the model forward is the HIP path
    in both."""
    start_time = time.time()
    dg = (normalize is deblurganv2.normalize and pad is deblurganv2.pad and postprocess is deblurganv2.postprocess
          and input_img.dtype == np.uint8)
    stock = dg or (normalize is globals()['normalize'] and (pad is None or pad is globals()['pad'])
                   and postprocess is None and input_img.dtype in (np.uint8, np.uint16))
    with torch.no_grad():
        if stock:
            dev = torch.device(device)
            src = input_img if input_img.dtype == np.uint8 else input_img.view(np.int16)
            img_dev = torch.from_numpy(np.ascontiguousarray(src)).to(dev)
            sigma = noise_level if (need_degradation and noise_level is not None) else None
            out, _ = tiled_forward_device(model, img_dev, patch_size, patch_overlap, pad is not None, sigma,
                                          max_batch=getattr(model, 'max_tiles_per_batch', 8),
                                          hooks="deblurganv2" if dg else None)
            output_img = out.cpu().numpy()
            if input_img.dtype == np.uint16:
                output_img = output_img.view(np.uint16)
        else:
            output_img = _run_tiles_on_host(model, input_img, device, normalize, patch_size, patch_overlap,
                                            need_degradation, noise_level, pad, postprocess)
    return output_img, (time.time() - start_time) * 1000


def _run_tiles_on_host(model, input_img, device, normalize_fn, patch_size, patch_overlap, need_degradation,
                       noise_level, pad_fn, postprocess):
    """Per-tile loop with arbitrary hooks (reference order of operations, utils.py:379-450)."""
    img = normalize_fn(input_img)
    h, w = img.shape[:2]
    if patch_size:
        ps = min(patch_size, max(h, w))
        ys, xs = tile_origins(h, ps, patch_overlap), tile_origins(w, ps, patch_overlap)
    else:
        ps, ys, xs = max(h, w), [0], [0]
    c_out = min(3, img.shape[2])
    acc = np.zeros((h, w, c_out), np.float32)
    wsum = np.zeros((h, w, c_out), np.float32)
    win = get_gaussian_weights(ps, ps, c_out)
    for y0 in ys:
        for x0 in xs:
            tile = img[y0:y0 + ps, x0:x0 + ps, :].copy()
            if need_degradation and noise_level is not None:
                tile = add_gaussian_noise(tile, noise_level)
            t = torch.from_numpy(tile.transpose(2, 0, 1)).unsqueeze(0).to(device)
            if pad_fn is not None:
                hp, wp = t.shape[-2:]
                o = model(pad_fn(t))[:, :, :hp, :wp]
            else:
                o = model(t)
            if postprocess is not None:
                o = postprocess(o)
            p = o.squeeze(0).cpu().numpy().transpose(1, 2, 0)
            ch, cw = p.shape[:2]
            acc[y0:y0 + ch, x0:x0 + cw, :] += p * win[:ch, :cw]
            wsum[y0:y0 + ch, x0:x0 + cw, :] += win[:ch, :cw]
    acc /= np.maximum(wsum, 1e-8)
    if input_img.dtype == np.uint16:
        return np.clip(acc * 65535.0, 0, 65535).round().astype(np.uint16)
    if input_img.dtype == np.uint8:
        return np.clip(acc * 255.0, 0, 255).round().astype(np.uint8)
    lo, hi = np.min(input_img), np.max(input_img)
    return np.clip(acc * hi, lo, hi).astype(input_img.dtype)


def get_model_prediction(model: Module, input_image: np.ndarray, device: torch.device, patch_size: int,
                         patch_overlap: int, need_degradation=False, noise_level=None, progress_bar=None):
    """src/utils.py:270-311: dispatch on the model class (reflect-pad-to-8 models vs plain)."""
    kw = dict(patch_size=patch_size, patch_overlap=patch_overlap, need_degradation=need_degradation,
              noise_level=noise_level, progress_bar=progress_bar)
    if isinstance(model, (FPNMobileNet,)):                      # utils.py:280-291
        return run_model_inference(model, input_image, device, normalize=deblurganv2.normalize, pad=deblurganv2.pad,
                                   postprocess=deblurganv2.postprocess, **kw)
    if isinstance(model, _PAD8_MODELS):
        return run_model_inference(model, input_image, device, pad=pad, **kw)
    return run_model_inference(model, input_image, device, **kw)
