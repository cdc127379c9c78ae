"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement (functional PyTorch fp32) of MaIRUNet's forward
(src/mair/realDenoising/basicsr/models/archs/mairunet_arch.py:21-739), its scan-index tables
(shift_scanf_util.py:67-244; MaIRUNet never passes shift_size, :476-581, the flat MaIR alternates) and of the
third-party selective scan it calls (mamba_ssm==2.2.5 `selective_scan_fn`, NOT in the reference tree).

Pinning: everything except `selective_scan` is checked against the imported reference module by
oracle/gen_golden.py (the reference file is imported with stubs for timm.layers / mamba_ssm / the registry,
with THIS file's `selective_scan` standing in for the absent CUDA kernel).  `selective_scan` itself restates
the published recurrence (call site mairunet_arch.py:252-258; einsum structure csms6s.py:168-215):
**parity unpinned** - no reference artefact pins its arithmetic; it is cross-checked only against an
independent float64 evaluation in tests/test_cpu.py.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- scan index tables
def _snake(idx: np.ndarray, scan_len: int, shift_len: int = 0) -> np.ndarray:
    """Visit order of `sscan` (shift_scanf_util.py:67-126) on an index image [H][W]: column stripes left to
    right - with shift_len > 0 a first stripe of width shift_len, then stripes of scan_len -, every odd
    stripe bottom-up; inside a stripe a boustrophedon over rows (odd visited rows right-to-left)."""
    H, W = idx.shape
    edges = ([0, shift_len] if shift_len else [0])
    while edges[-1] < W:
        edges.append(min(edges[-1] + scan_len, W))
    out = []
    for s in range(len(edges) - 1):
        cols = np.arange(edges[s], edges[s + 1])
        for hv in range(H):
            row = H - 1 - hv if s % 2 else hv
            out.append(idx[row, cols[::-1] if hv % 2 else cols])
    return np.concatenate(out)


def scan_ids(H: int, W: int, scan_len: int = 4, shift_len: int = 0):
    """(ids [4][L], inverse [4][L]) int64 as mair_ids_generate / mair_shift_ids_generate
    (shift_scanf_util.py:169-203) return them: direction 0 the image, 1 the image rotated by 180 degrees,
    2 its transpose, 3 the rotated transpose."""
    idx = np.arange(H * W).reshape(H, W)
    rot = idx[::-1, ::-1]
    ids = np.stack([_snake(m, scan_len, shift_len) for m in (idx, rot, idx.T, rot.T)])
    return torch.from_numpy(ids.copy()), torch.from_numpy(np.argsort(ids, axis=-1))


# --------------------------------------------------------------------------- selective scan (unpinned)
def selective_scan(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                   return_last_state=False):
    """u, delta: (b, kd, L); A: (kd, n); B, C: (b, k, n, L) grouped over kd; D, delta_bias: (kd).
    delta = softplus(delta + bias); h_t = exp(delta_t A) h_{t-1} + delta_t B_t u_t; y_t = <h_t, C_t> + D u_t."""
    assert z is None and not return_last_state
    b, kd, L = u.shape
    k, n = B.shape[1], A.shape[1]
    d = kd // k
    u, delta = u.float(), delta.float()
    if delta_bias is not None:
        delta = delta + delta_bias.float().view(1, -1, 1)
    if delta_softplus:
        delta = F.softplus(delta)
    Bx = B.float().repeat_interleave(d, dim=1)          # (b, kd, n, L)
    Cx = C.float().repeat_interleave(d, dim=1)
    h = torch.zeros(b, kd, n, dtype=torch.float32)
    ys = []
    for t in range(L):
        dt = delta[:, :, t].unsqueeze(-1)
        h = torch.exp(dt * A.float().unsqueeze(0)) * h + dt * Bx[:, :, :, t] * u[:, :, t].unsqueeze(-1)
        ys.append((h * Cx[:, :, :, t]).sum(-1))
    y = torch.stack(ys, dim=-1)
    if D is not None:
        y = y + D.float().view(1, -1, 1) * u
    return y


# --------------------------------------------------------------------------- blocks
def losh2d(x, p, pre, ids, inv):
    """LoSh2D.forward (mairunet_arch.py:263-282) on channel-last x (B, H, W, C)."""
    B, H, W, _ = x.shape
    L = H * W
    xz = F.linear(x, p[pre + "in_proj.weight"], p.get(pre + "in_proj.bias"))
    xc, z = xz.chunk(2, dim=-1)
    xc = xc.permute(0, 3, 1, 2)
    Dn = xc.shape[1]
    xc = F.silu(F.conv2d(xc, p[pre + "conv2d.weight"], p.get(pre + "conv2d.bias"), padding=1, groups=Dn))
    # forward_core (:226-261)
    Wx, Wdt = p[pre + "x_proj_weight"], p[pre + "dt_projs_weight"]       # (4, R+2N, D), (4, D, R)
    K, R = 4, Wdt.shape[2]
    N = (Wx.shape[1] - R) // 2
    flat = xc.reshape(B, 1, Dn, L)
    xs = torch.cat([flat.index_select(-1, ids[k]) for k in range(K)], dim=1)               # (B, 4, D, L)
    x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, Wx)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("bkrl,kdr->bkdl", dts, Wdt)
    y = selective_scan(xs.reshape(B, -1, L), dts.reshape(B, -1, L), -torch.exp(p[pre + "A_logs"].float()),
                       Bs, Cs, p[pre + "Ds"].float(), delta_bias=p[pre + "dt_projs_bias"].float().reshape(-1),
                       delta_softplus=True).view(B, K, Dn, L)
    y = torch.cat([y[:, k].index_select(-1, inv[k]) for k in range(K)], dim=1).reshape(B, K * Dn, H, W)
    # ShuffleAttn gate (:21-60): per d a 4x4 mix of the four directions' global means
    m = y.mean(dim=(2, 3))                                                                   # (B, 4D), index k*D+d
    gw = p[pre + "gating.gating.1.weight"].reshape(Dn, 4, 4)                                 # [d][k'][k]
    gb = p[pre + "gating.gating.1.bias"].reshape(Dn, 4)                                      # [d][k']
    g = torch.sigmoid(torch.einsum("dqk,bkd->bqd", gw, m.view(B, 4, Dn)) + gb.t().unsqueeze(0))   # (B, k', d)
    y = (y.view(B, 4, Dn, H, W) * g.view(B, 4, Dn, 1, 1)).sum(dim=1)                         # (B, D, H, W)
    y = y.permute(0, 2, 3, 1)
    y = F.layer_norm(y, (Dn,), p[pre + "out_norm.weight"], p[pre + "out_norm.bias"], 1e-5)
    y = y * F.silu(z)
    return F.linear(y, p[pre + "out_proj.weight"], p.get(pre + "out_proj.bias"))


def vss_block(x, p, pre, hw, ids, inv, mlp="mlp"):
    """VSSBlock.forward (mairunet_arch.py:362-380) / RMB.forward (mair_arch.py:376-390, MLP named conv_blk)
    on tokens x (B, L, C)."""
    B, L, C = x.shape
    xi = x.view(B, hw[0], hw[1], C)
    h = F.layer_norm(xi, (C,), p[pre + "ln_1.weight"], p[pre + "ln_1.bias"], 1e-5)
    xi = xi * p[pre + "skip_scale"] + losh2d(h, p, pre + "self_attention.", ids, inv)
    h = F.layer_norm(xi, (C,), p[pre + "ln_2.weight"], p[pre + "ln_2.bias"], 1e-5)
    h = F.linear(F.gelu(F.linear(h, p[pre + mlp + ".fc1.weight"], p[pre + mlp + ".fc1.bias"])),
                 p[pre + mlp + ".fc2.weight"], p[pre + mlp + ".fc2.bias"])
    return (xi * p[pre + "skip_scale2"] + h).view(B, L, C)


def _stage(x, p, name, hw, tables):
    i = 0
    while f"{name}.{i}.ln_1.weight" in p:
        x = vss_block(x, p, f"{name}.{i}.", hw, *tables)
        i += 1
    return x


def _tok(x):
    return x.flatten(2).transpose(1, 2)


def _img(x, hw):
    return x.transpose(1, 2).reshape(x.shape[0], -1, hw[0], hw[1])


def mairunet_forward(x, p, scan_len=4, dual_pixel_task=False):
    """MaIRUNet.forward (mairunet_arch.py:644-739); x (B, C, H, W), H and W multiples of 8."""
    B, _, H, W = x.shape
    sizes = [(H, W), (H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8)]
    tabs = [scan_ids(h, w, scan_len) for h, w in sizes]
    conv3 = lambda t, w: F.conv2d(t, w, None, padding=1)                                     # noqa: E731
    e1_in = _tok(conv3(x, p["patch_embed.proj.weight"]))
    e1 = _stage(e1_in, p, "encoder_level1", sizes[0], tabs[0])
    e2 = _stage(_tok(F.pixel_unshuffle(conv3(_img(e1, sizes[0]), p["down1_2.body.0.weight"]), 2)), p,
                "encoder_level2", sizes[1], tabs[1])
    e3 = _stage(_tok(F.pixel_unshuffle(conv3(_img(e2, sizes[1]), p["down2_3.body.0.weight"]), 2)), p,
                "encoder_level3", sizes[2], tabs[2])
    lat = _stage(_tok(F.pixel_unshuffle(conv3(_img(e3, sizes[2]), p["down3_4.body.0.weight"]), 2)), p,
                 "latent", sizes[3], tabs[3])
    d3 = torch.cat([_tok(F.pixel_shuffle(conv3(_img(lat, sizes[3]), p["up4_3.body.0.weight"]), 2)), e3], 2)
    d3 = _tok(F.conv2d(_img(d3, sizes[2]), p["reduce_chan_level3.weight"], p.get("reduce_chan_level3.bias")))
    d3 = _stage(d3, p, "decoder_level3", sizes[2], tabs[2])
    d2 = torch.cat([_tok(F.pixel_shuffle(conv3(_img(d3, sizes[2]), p["up3_2.body.0.weight"]), 2)), e2], 2)
    d2 = _tok(F.conv2d(_img(d2, sizes[1]), p["reduce_chan_level2.weight"], p.get("reduce_chan_level2.bias")))
    d2 = _stage(d2, p, "decoder_level2", sizes[1], tabs[1])
    d1 = torch.cat([_tok(F.pixel_shuffle(conv3(_img(d2, sizes[1]), p["up2_1.body.0.weight"]), 2)), e1], 2)
    d1 = _stage(d1, p, "decoder_level1", sizes[0], tabs[0])
    d1 = _img(_stage(d1, p, "refinement", sizes[0], tabs[0]), sizes[0])
    if dual_pixel_task:
        d1 = d1 + F.conv2d(_img(e1_in, sizes[0]), p["skip_conv.weight"], p.get("skip_conv.bias"))
        return F.conv2d(d1, p["output.weight"], p.get("output.bias"), padding=1)
    return F.conv2d(d1, p["output.weight"], p.get("output.bias"), padding=1) + x


RGB_MEAN = (0.4488, 0.4371, 0.4040)


def mair_forward(x, p, scan_len=4, img_range=1.0):
    """MaIR.forward, denoising branch (upsampler=None) (src/mair/basicsr/archs/mair_arch.py:682-730):
    mean shift, conv_first, patch-norm, residual Mamba groups (blocks alternate the unshifted / shifted
    scan tables, mair_arch.py:455, 379-382; each group ends in conv3x3 + residual, :863-864), final norm,
    conv_after_body + residual, conv_last + input residual, un-shift."""
    B, C, H, W = x.shape
    hw = (H, W)
    mean = (torch.tensor(RGB_MEAN) if C == 3 else torch.zeros(1)).view(1, -1, 1, 1).to(x)
    tabs = [scan_ids(H, W, scan_len), scan_ids(H, W, scan_len, scan_len // 2)]
    x = (x - mean) * img_range
    first = F.conv2d(x, p["conv_first.weight"], p["conv_first.bias"], padding=1)
    E = first.shape[1]
    t = F.layer_norm(_tok(first), (E,), p["patch_embed.norm.weight"], p["patch_embed.norm.bias"], 1e-5)
    li = 0
    while f"layers.{li}.conv.weight" in p:
        g_in, bi = t, 0
        while f"layers.{li}.residual_group.blocks.{bi}.ln_1.weight" in p:
            t = vss_block(t, p, f"layers.{li}.residual_group.blocks.{bi}.", hw, *tabs[bi % 2], mlp="conv_blk")
            bi += 1
        t = _tok(F.conv2d(_img(t, hw), p[f"layers.{li}.conv.weight"], p[f"layers.{li}.conv.bias"], padding=1)) + g_in
        li += 1
    t = F.layer_norm(t, (E,), p["norm.weight"], p["norm.bias"], 1e-5)
    res = F.conv2d(_img(t, hw), p["conv_after_body.weight"], p["conv_after_body.bias"], padding=1) + first
    x = x + F.conv2d(res, p["conv_last.weight"], p["conv_last.bias"], padding=1)
    return x / img_range + mean
