"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement, in plain functional PyTorch fp32, of the Restormer forward
pass of the reference (src/restormer/restormer.py).  It consumes a state dict
with the reference's parameter names, so a checkpoint for the reference module
drives it unchanged.  Pinned against the imported reference module by
oracle/gen_golden.py (max-abs difference recorded in tests/golden/MANIFEST.json).

Each function cites the reference lines it restates.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def layer_norm_c(x, weight, bias=None, eps=1e-5):
    """Per-pixel LayerNorm over channels of an NCHW tensor.

    restormer.py:25-70.  WithBias (bias given): (x-mu)/sqrt(var+eps)*w+b.
    BiasFree (bias None): x/sqrt(var+eps)*w - mu is used for the variance only
    (restormer.py:38-39).  var is the biased variance.
    """
    mu = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, keepdim=True, unbiased=False)
    w = weight.view(1, -1, 1, 1)
    if bias is None:
        return x / torch.sqrt(var + eps) * w
    return (x - mu) / torch.sqrt(var + eps) * w + bias.view(1, -1, 1, 1)


def mdta(x, p, prefix, heads):
    """Multi-DConv-head transposed attention, restormer.py:111-132."""
    b, c, h, w = x.shape
    qkv = F.conv2d(x, p[prefix + "qkv.weight"], p.get(prefix + "qkv.bias"))
    qkv = F.conv2d(qkv, p[prefix + "qkv_dwconv.weight"], p.get(prefix + "qkv_dwconv.bias"),
                   padding=1, groups=3 * c)
    q, k, v = qkv.reshape(b, 3, heads, c // heads, h * w).unbind(dim=1)
    q = F.normalize(q, dim=-1)           # L2 over HW, eps 1e-12 (restormer.py:122-123)
    k = F.normalize(k, dim=-1)
    logits = torch.matmul(q, k.transpose(-1, -2)) * p[prefix + "temperature"].view(1, heads, 1, 1)
    attn = torch.softmax(logits, dim=-1)
    out = torch.matmul(attn, v).reshape(b, c, h, w)
    return F.conv2d(out, p[prefix + "project_out.weight"], p.get(prefix + "project_out.bias"))


def gdfn(x, p, prefix):
    """Gated-dconv feed-forward, restormer.py:76-93 (erf GELU, gate = first half)."""
    y = F.conv2d(x, p[prefix + "project_in.weight"], p.get(prefix + "project_in.bias"))
    y = F.conv2d(y, p[prefix + "dwconv.weight"], p.get(prefix + "dwconv.bias"), padding=1,
                 groups=y.shape[1])
    hid = y.shape[1] // 2
    y = F.gelu(y[:, :hid]) * y[:, hid:]
    return F.conv2d(y, p[prefix + "project_out.weight"], p.get(prefix + "project_out.bias"))


def transformer_block(x, p, prefix, heads):
    """restormer.py:137-150."""
    n1 = layer_norm_c(x, p[prefix + "norm1.body.weight"], p.get(prefix + "norm1.body.bias"))
    x = x + mdta(n1, p, prefix + "attn.", heads)
    n2 = layer_norm_c(x, p[prefix + "norm2.body.weight"], p.get(prefix + "norm2.body.bias"))
    return x + gdfn(n2, p, prefix + "ffn.")


def _stage(x, p, name, heads):
    i = 0
    while f"{name}.{i}.norm1.body.weight" in p:
        x = transformer_block(x, p, f"{name}.{i}.", heads)
        i += 1
    return x


def _conv3(x, w, b=None):
    return F.conv2d(x, w, b, padding=1)


def restormer_forward(x, p, heads=(1, 2, 4, 8), dual_pixel_task=False):
    """restormer.py:245-284.  x: (B, C_in, H, W) float32, H and W multiples of 8."""
    e1_in = _conv3(x, p["patch_embed.proj.weight"], p.get("patch_embed.proj.bias"))
    e1 = _stage(e1_in, p, "encoder_level1", heads[0])
    e2 = _stage(F.pixel_unshuffle(_conv3(e1, p["down1_2.body.0.weight"]), 2), p, "encoder_level2", heads[1])
    e3 = _stage(F.pixel_unshuffle(_conv3(e2, p["down2_3.body.0.weight"]), 2), p, "encoder_level3", heads[2])
    lat = _stage(F.pixel_unshuffle(_conv3(e3, p["down3_4.body.0.weight"]), 2), p, "latent", heads[3])

    d3 = torch.cat([F.pixel_shuffle(_conv3(lat, p["up4_3.body.0.weight"]), 2), e3], dim=1)
    d3 = F.conv2d(d3, p["reduce_chan_level3.weight"], p.get("reduce_chan_level3.bias"))
    d3 = _stage(d3, p, "decoder_level3", heads[2])

    d2 = torch.cat([F.pixel_shuffle(_conv3(d3, p["up3_2.body.0.weight"]), 2), e2], dim=1)
    d2 = F.conv2d(d2, p["reduce_chan_level2.weight"], p.get("reduce_chan_level2.bias"))
    d2 = _stage(d2, p, "decoder_level2", heads[1])

    d1 = torch.cat([F.pixel_shuffle(_conv3(d2, p["up2_1.body.0.weight"]), 2), e1], dim=1)
    d1 = _stage(d1, p, "decoder_level1", heads[0])
    d1 = _stage(d1, p, "refinement", heads[0])

    if dual_pixel_task:
        d1 = d1 + F.conv2d(e1_in, p["skip_conv.weight"], p.get("skip_conv.bias"))
        return _conv3(d1, p["output.weight"], p.get("output.bias"))
    return _conv3(d1, p["output.weight"], p.get("output.bias")) + x
