"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement (functional PyTorch fp32) of DeblurGANv2's FPN-MobileNet generator as the reference runs
it (src/deblurganv2/models/fpn_mobilenet.py:6-147, models/mobilenet_v2.py:5-110) - in TRAIN mode
(`model.train(True)`, src/deblurganv2/__init__.py:38): every BatchNorm2d normalises with the statistics of
the current (batch-1) input and every InstanceNorm2d(affine=False) with instance statistics - plus the
tiler hooks of src/deblurganv2/__init__.py:11-28 (normalize via albumentations Normalize(mean=.5, std=.5),
aug.py:31-39; zero pad to (h//32+1)*32; postprocess (x+1)/2).  Pinned against the imported reference
modules by oracle/gen_golden.py.  FPN-Inception needs timm's InceptionResNetV2 (absent): not restated.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

# (t, c, n, s) of MobileNetV2 (mobilenet_v2.py:68-77); features[0] is conv_bn(3, 32, 2)
_SETTING = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]


def block_specs():
    """[(index in features, inp, oup, stride, expand)] for features[1:]."""
    specs, inp, idx = [], 32, 1
    for t, c, n, s in _SETTING:
        for i in range(n):
            specs.append((idx, inp, c, s if i == 0 else 1, t))
            inp, idx = c, idx + 1
    return specs


def _bn(x, p, pre):
    """BatchNorm2d in training mode on a batch of independent tiles: per-sample statistics (the reference
    feeds one tile at a time), biased variance, eps 1e-5, affine."""
    return F.instance_norm(x, None, None, p[pre + "weight"], p[pre + "bias"], True, 0.1, 1e-5)


def _inorm(x):
    return F.instance_norm(x, None, None, None, None, True, 0.1, 1e-5)


def _inverted_residual(x, p, pre, inp, oup, stride, t):
    hid = round(inp * t)
    y = x
    i = 0
    if t != 1:
        y = F.relu6(_bn(F.conv2d(y, p[pre + "conv.0.weight"]), p, pre + "conv.1."))
        i = 3
    y = F.relu6(_bn(F.conv2d(y, p[pre + f"conv.{i}.weight"], None, stride, 1, groups=hid), p, pre + f"conv.{i + 1}."))
    y = _bn(F.conv2d(y, p[pre + f"conv.{i + 3}.weight"]), p, pre + f"conv.{i + 4}.")
    return x + y if (stride == 1 and inp == oup) else y


def fpn_mobilenet_forward(x, p):
    """FPNMobileNet.forward (fpn_mobilenet.py:53-70); x (B, 3, H, W), H and W multiples of 32."""
    f = "fpn.features."
    feats = {}
    y = F.relu6(_bn(F.conv2d(x, p[f + "0.0.weight"], None, 2, 1), p, f + "0.1."))
    for (idx, inp, oup, s, t) in block_specs():
        if idx > 15:
            break
        y = _inverted_residual(y, p, f + f"{idx}.", inp, oup, s, t)
        feats[idx] = y
    enc0, enc1, enc2, enc3, enc4 = feats[1], feats[3], feats[6], feats[10], feats[15]
    lat = [F.conv2d(e, p[f"fpn.lateral{i}.weight"]) for i, e in enumerate([enc0, enc1, enc2, enc3, enc4])]
    up = lambda t, s: F.interpolate(t, scale_factor=s, mode="nearest")                       # noqa: E731
    td = lambda t, name: F.relu(_inorm(F.conv2d(t, p[f"fpn.{name}.0.weight"], p[f"fpn.{name}.0.bias"], padding=1)))  # noqa
    map4 = lat[4]
    map3 = td(lat[3] + up(map4, 2), "td1")
    map2 = td(lat[2] + up(map3, 2), "td2")
    map1 = td(lat[1] + up(map2, 2), "td3")

    def head(t, name):
        t = F.relu(F.conv2d(t, p[name + ".block0.weight"], padding=1))
        return F.relu(F.conv2d(t, p[name + ".block1.weight"], padding=1))
    cat = torch.cat([up(head(map4, "head4"), 8), up(head(map3, "head3"), 4), up(head(map2, "head2"), 2),
                     head(map1, "head1")], dim=1)
    sm = F.relu(_inorm(F.conv2d(cat, p["smooth.0.weight"], p["smooth.0.bias"], padding=1)))
    sm = up(sm, 2)
    sm = F.relu(_inorm(F.conv2d(sm + lat[0], p["smooth2.0.weight"], p["smooth2.0.bias"], padding=1)))
    sm = up(sm, 2)
    final = F.conv2d(sm, p["final.weight"], p["final.bias"], padding=1)
    return torch.clamp(torch.tanh(final) + x, min=-1, max=1)


# tiler hooks (src/deblurganv2/__init__.py:11-28)
def normalize(img: np.ndarray) -> np.ndarray:
    """albumentations Normalize(mean=(.5,.5,.5), std=(.5,.5,.5)) on uint8: (x/255 - 0.5)/0.5 in float32
    (albumentations is absent here: restated from aug.py:31-39 - mean and std are multiplied by
    max_pixel_value=255 first: (x - 127.5) / 127.5)."""
    mean = np.float32(0.5) * np.float32(255.0)
    denom = np.float32(1.0) / (np.float32(0.5) * np.float32(255.0))
    return ((img.astype(np.float32) - mean) * denom).astype(np.float32)


def pad32(x: torch.Tensor) -> torch.Tensor:
    h, w = x.shape[-2:]
    return F.pad(x, (0, (w // 32 + 1) * 32 - w, 0, (h // 32 + 1) * 32 - h), "constant", 0)


def postprocess(x: torch.Tensor) -> torch.Tensor:
    return (x + 1) / 2.0


def fpn_inception_decoder(x, encs, p):
    """FPN-Inception WITHOUT its timm encoder: FPN.forward's lateral / reflect-pad / top-down path
    (src/deblurganv2/models/fpn_inception.py:153-170) and FPNInception.forward's heads, smoothing and output
    (:65-81) on given encoder maps encs = (enc0..enc4); InstanceNorm2d(affine=False) in train mode.  Pinned by
    oracle/gen_golden.py --only fpn_inception against the reference class with constant-output encoder stages."""
    e0, e1, e2, e3, e4 = encs
    lat = lambda i, e: F.conv2d(e, p[f"fpn.lateral{i}.weight"])            # noqa: E731
    l4 = F.pad(lat(4, e4), (1, 1, 1, 1), "reflect")
    l3 = F.pad(lat(3, e3), (1, 1, 1, 1), "reflect")
    l2 = F.pad(lat(2, e2), (1, 2, 1, 2), "reflect")
    l1 = F.pad(lat(1, e1), (1, 1, 1, 1), "reflect")
    map0 = F.pad(lat(0, e0), (0, 1, 0, 1), "reflect")
    up = lambda t, s: F.interpolate(t, scale_factor=s, mode="nearest")     # noqa: E731
    td = lambda n, t: F.relu(_inorm(F.conv2d(t, p[f"fpn.{n}.0.weight"], p[f"fpn.{n}.0.bias"], padding=1)))   # noqa: E731
    map4 = l4
    map3 = td("td1", l3 + up(map4, 2))
    map2 = td("td2", l2 + up(map3, 2))
    map1 = td("td3", l1 + up(map2, 2))

    def head(i, t):
        t = F.relu(F.conv2d(t, p[f"head{i}.block0.weight"], padding=1))
        return F.relu(F.conv2d(t, p[f"head{i}.block1.weight"], padding=1))
    cat = torch.cat([up(head(4, map4), 8), up(head(3, map3), 4), up(head(2, map2), 2), head(1, map1)], dim=1)
    sm = F.relu(_inorm(F.conv2d(cat, p["smooth.0.weight"], p["smooth.0.bias"], padding=1)))
    sm = up(sm, 2)
    sm = F.relu(_inorm(F.conv2d(sm + map0, p["smooth2.0.weight"], p["smooth2.0.bias"], padding=1)))
    sm = up(sm, 2)
    return torch.clamp(torch.tanh(F.conv2d(sm, p["final.weight"], p["final.bias"], padding=1)) + x, -1, 1)
