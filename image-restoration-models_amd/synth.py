"""Deterministic synthetic weights.

There are no pretrained checkpoints in this environment (the reference downloads
them: scripts/download_weights.sh:125-191), so benches, tests and golden
fixtures use weights generated here.  The generator is counter based (numpy
Philox keyed by sha256(seed, parameter name)), not torch's RNG, so the GPU box
regenerates bit-identical tensors without any reference code and independent
of parameter iteration order.
"""
from __future__ import annotations

import hashlib
import re
from typing import Iterable

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    digest = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    key = int.from_bytes(digest[:8], "little")
    return np.random.Generator(np.random.Philox(key=key))


def uniform(seed: int, name: str, shape: Iterable[int], lo: float, hi: float) -> torch.Tensor:
    """float32 tensor, U[lo, hi), reproducible from (seed, name, shape)."""
    shape = tuple(int(s) for s in shape)
    u = _rng(seed, name).random(int(np.prod(shape)) if shape else 1, dtype=np.float64)
    v = (lo + (hi - lo) * u).astype(np.float32).reshape(shape)
    return torch.from_numpy(v)


#: default (regex, kind, args) rules, first match wins
_DEFAULT_RULES = (
    (r"temperature$", "range", (2.0, 6.0)),
    (r"skip_scale2?$", "range", (0.8, 1.2)),
    (r"(norm\w*|ln_\w+)\.(body\.)?weight$", "range", (0.9, 1.1)),
    (r"(norm\w*|ln_\w+)\.(body\.)?bias$", "range", (-0.1, 0.1)),
)


def synth_tensor(seed: int, name: str, shape, gain: float = 1.0, rules=()) -> torch.Tensor:
    """One synthetic parameter.

    Rules (``rules`` first, then the defaults): a regex on the parameter name
    selects either a fixed value range or a gain for the fan-in scaled uniform
    init.  Tensors with >= 2 dims get U(-a, a), a = gain*sqrt(3/fan_in)
    (variance gain^2/fan_in); 1-d tensors that match no rule are biases,
    U(-0.05, 0.05).
    """
    shape = tuple(shape)
    for pat, kind, args in tuple(rules) + _DEFAULT_RULES:
        if re.search(pat, name):
            if kind == "range":
                return uniform(seed, name, shape, *args)
            if kind == "gain":
                gain = args
            break
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        a = gain * (3.0 / max(fan_in, 1)) ** 0.5
        return uniform(seed, name, shape, -a, a)
    return uniform(seed, name, shape, -0.05, 0.05)


def synth_state_dict(shapes: dict, seed: int = 42, rules=()) -> dict:
    """name -> tensor for a dict name -> shape."""
    return {k: synth_tensor(seed, k, s, rules=rules) for k, s in shapes.items()}


# ---------------------------------------------------------------------------
# synthetic images: GoPro-shaped frames etc.
# ---------------------------------------------------------------------------

def _box_filter(img: np.ndarray, taps: int, axis: int) -> np.ndarray:
    """Separable box low-pass with edge replication, float64 in / out."""
    pad = taps // 2
    padw = [(0, 0)] * img.ndim
    padw[axis] = (pad, taps - 1 - pad)
    p = np.pad(img, padw, mode="edge")
    c = np.cumsum(p, axis=axis)
    zero = np.zeros_like(np.take(c, [0], axis=axis))
    c = np.concatenate([zero, c], axis=axis)
    n = img.shape[axis]
    hi = np.take(c, np.arange(taps, taps + n), axis=axis)
    lo = np.take(c, np.arange(0, n), axis=axis)
    return (hi - lo) / taps


def synth_image_pair(index: int, h: int = 720, w: int = 1280, c: int = 3, seed_base: int = 1000,
                     blur: int = 15):
    """(input_u8, target_u8) HWC.  target = low-passed uniform noise (natural
    image-like spectrum), input = target blurred by a horizontal `blur`-px box
    (motion-blur-like), both requantised to uint8."""
    rng = np.random.default_rng(seed_base + index)
    raw = rng.integers(0, 256, size=(h, w, c)).astype(np.float64)
    t = _box_filter(_box_filter(raw, 9, 0), 9, 1)
    # stretch contrast back to the full 8-bit range
    t = (t - t.min()) / max(t.max() - t.min(), 1e-9) * 255.0
    target = np.clip(np.rint(t), 0, 255).astype(np.uint8)
    if blur and blur > 1:
        b = _box_filter(target.astype(np.float64), blur, 1)
        inp = np.clip(np.rint(b), 0, 255).astype(np.uint8)
    else:
        inp = target.copy()
    return inp, target
