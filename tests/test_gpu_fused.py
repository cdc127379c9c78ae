"""GPU parity tests of the whole-branch fused kernels (fused_block.hip) through the C ABI against float64
references of the same chain of reference ops (restormer.py:25-70, 76-93, 148).

Bar: <= 2e-5 x max(1, |ref|max) vs float64 (fp32-level: a plain fp32 torch chain measures 1e-6..6e-6 here, the fused kernels 1e-6..8e-6, tools/err_fused.py) (the path's budget is 1e-3 max-abs, BASELINE.json north_star),
and not worse than 2x the un-fused exact-f32 chain on the same inputs where that is compared."""
import pytest
import torch
import torch.nn.functional as F

from irm_amd import _hip, ops, synth

pytestmark = pytest.mark.gpu
TOL = 2e-5


def rnd(name, shape, lo=-1.0, hi=1.0):
    return synth.uniform(321, name, shape, lo, hi)


def gdfn_ref(x, lnw, lnb, ln_mode, pin_w, pin_b, dw_w, dw_b, pout_w, pout_b, eps=1e-5):
    """float64: x + project_out(gelu(dw(h)[:hid]) * dw(h)[hid:]), h = project_in(LN(x))."""
    xd = x.double()
    mu = xd.mean(1, keepdim=True)
    var = xd.var(1, unbiased=False, keepdim=True)
    if ln_mode == _hip.LN_WITHBIAS:
        xn = (xd - mu) / torch.sqrt(var + eps) * lnw.double()[None, :, None, None] + lnb.double()[None, :, None, None]
    else:
        xn = xd / torch.sqrt(var + eps) * lnw.double()[None, :, None, None]
    hid = pout_w.shape[1]
    h = F.conv2d(xn, pin_w.double()[:, :, None, None], None if pin_b is None else pin_b.double())
    h = F.conv2d(h, dw_w.double().view(2 * hid, 1, 3, 3), None if dw_b is None else dw_b.double(), padding=1,
                 groups=2 * hid)
    g = F.gelu(h[:, :hid]) * h[:, hid:]
    return xd + F.conv2d(g, pout_w.double()[:, :, None, None], None if pout_b is None else pout_b.double())


GDFN_CASES = [
    # C, hid, H, W, B, ln_mode, bias, scale of the weights
    (96, 255, 16, 64, 1, 1, False, 0.3),
    (96, 255, 24, 40, 2, 2, True, 0.3),
    (48, 127, 16, 32, 2, 1, True, 0.3),
    (48, 127, 20, 36, 1, 2, False, 0.3),
    (96, 255, 8, 8, 1, 1, False, 0.3),
    (32, 85, 12, 20, 1, 1, True, 0.3),
    (64, 170, 8, 44, 2, 2, False, 0.3),
    (96, 255, 40, 72, 1, 1, True, 3.0),          # large weights: the power-of-two operand scales adapt
    (96, 250, 16, 32, 1, 1, False, 1e-3),        # tiny weights
]


@pytest.mark.parametrize("C,hid,H,W,B,ln,bias,ws", GDFN_CASES)
def test_gdfn_fused(dev, C, hid, H, W, B, ln, bias, ws):
    tag = f"f{C}_{hid}_{H}_{W}_{B}_{ln}"
    big = rnd(tag + "x", (B, C + 3, H, W), -1.5, 2.0)
    lnw = rnd(tag + "lw", (C,), 0.5, 1.5)
    lnb = rnd(tag + "lb", (C,), -0.2, 0.2) if ln == 1 else None
    pin_w = rnd(tag + "pi", (2 * hid, C), -ws, ws)
    pout_w = rnd(tag + "po", (C, hid), -ws, ws)
    dw_w = rnd(tag + "dw", (2 * hid, 9), -0.4, 0.4)
    pin_b = rnd(tag + "pib", (2 * hid,), -0.3, 0.3) if bias else None
    dw_b = rnd(tag + "dwb", (2 * hid,), -0.3, 0.3) if bias else None
    pout_b = rnd(tag + "pob", (C,), -0.3, 0.3) if bias else None
    x = big[:, 1:1 + C]
    ref = gdfn_ref(x, lnw, lnb, ln, pin_w, pin_b, dw_w, dw_b, pout_w, pout_b)
    pk = _hip.pack_gdfn_fused(pin_w.to(dev), pin_b, dw_w, dw_b, pout_w, lnw, lnb)
    xb = big.to(dev)
    yb = torch.full((B, C + 2, H, W), 7.0, device=dev)
    ops.gdfn_fused(pk, xb[:, 1:1 + C], yb[:, 2:2 + C], C, hid, ln_mode=ln,
                   bias=None if pout_b is None else pout_b.to(dev))
    y = yb.cpu()
    scale = max(1.0, float(ref.abs().max()))
    err = float((y[:, 2:2 + C].double() - ref).abs().max())
    assert err <= TOL * scale, (err, scale)
    assert torch.all(y[:, :2] == 7.0), "wrote outside its channel slice"
    assert torch.equal(xb.cpu(), big), "input modified"


APPLY_CASES = [
    # C, hid, H, W, B, ln_mode, bias
    (96, 255, 16, 64, 1, 1, False),
    (96, 255, 24, 40, 2, 2, True),
    (48, 127, 16, 32, 2, 1, True),
    (48, 127, 20, 36, 3, 2, False),
    (96, 255, 8, 8, 1, 1, True),
    (32, 85, 12, 20, 1, 1, True),
    (64, 170, 8, 44, 2, 2, False),
    (96, 255, 40, 72, 2, 1, True),
]


@pytest.mark.parametrize("C,hid,H,W,B,ln,bias", APPLY_CASES)
def test_attn_gdfn_fused(dev, C, hid, H, W, B, ln, bias):
    """x' = x + bias_o + Mfold[b] v, y = x' + GDFN(x') in one kernel (restormer.py:131, 147-148) vs float64."""
    tag = f"a{C}_{hid}_{H}_{W}_{B}_{ln}"
    ws = 0.3
    big = rnd(tag + "x", (B, C + 3, H, W), -1.5, 2.0)
    vbig = rnd(tag + "v", (B, 3 * C, H, W), -2.0, 2.0)          # v = the last C channels of a qkv buffer
    mf = rnd(tag + "m", (B, C, C), -0.2, 0.2)                     # per-image folded matrix
    lnw = rnd(tag + "lw", (C,), 0.5, 1.5)
    lnb = rnd(tag + "lb", (C,), -0.2, 0.2) if ln == 1 else None
    pin_w = rnd(tag + "pi", (2 * hid, C), -ws, ws)
    pout_w = rnd(tag + "po", (C, hid), -ws, ws)
    dw_w = rnd(tag + "dw", (2 * hid, 9), -0.4, 0.4)
    pin_b = rnd(tag + "pib", (2 * hid,), -0.3, 0.3) if bias else None
    dw_b = rnd(tag + "dwb", (2 * hid,), -0.3, 0.3) if bias else None
    pout_b = rnd(tag + "pob", (C,), -0.3, 0.3) if bias else None
    bo = rnd(tag + "bo", (C,), -0.3, 0.3) if bias else None
    x, v = big[:, 1:1 + C], vbig[:, 2 * C:]
    x1 = x.double() + torch.einsum("bij,bjhw->bihw", mf.double(), v.double())
    if bo is not None:
        x1 = x1 + bo.double()[None, :, None, None]
    ref = gdfn_ref(x1, lnw, lnb, ln, pin_w, pin_b, dw_w, dw_b, pout_w, pout_b)
    pk = _hip.pack_gdfn_fused(pin_w.to(dev), pin_b, dw_w, dw_b, pout_w, lnw, lnb, kperm=True)
    frag = _hip.pack_mfold_frag(mf).to(dev)
    assert torch.equal(_hip.unpack_mfold_frag(frag, B, C), (mf.half().float() + (mf - mf.half().float()).half().float()))
    xb, vb = big.to(dev), vbig.to(dev)
    yb = torch.full((B, C + 2, H, W), 7.0, device=dev)
    ops.attn_gdfn_fused(pk, xb[:, 1:1 + C], vb[:, 2 * C:], frag, yb[:, 2:2 + C], C, hid, ln_mode=ln,
                        bias_o=None if bo is None else bo.to(dev), bias=None if pout_b is None else pout_b.to(dev))
    y = yb.cpu()
    scale = max(1.0, float(ref.abs().max()))
    err = float((y[:, 2:2 + C].double() - ref).abs().max())
    assert err <= TOL * scale, (err, scale)
    assert torch.all(y[:, :2] == 7.0), "wrote outside its channel slice"
    assert torch.equal(xb.cpu(), big) and torch.equal(vb.cpu(), vbig), "input modified"
    y2 = torch.empty(B, C, H, W, device=dev)
    ops.attn_gdfn_fused(pk, xb[:, 1:1 + C], vb[:, 2 * C:], frag, y2, C, hid, ln_mode=ln,
                        bias_o=None if bo is None else bo.to(dev), bias=None if pout_b is None else pout_b.to(dev))
    assert torch.equal(y2.cpu(), y[:, 2:2 + C]), "not deterministic"


def to_tm(t):
    """[B][C][H][W] planar -> the same container holding the tile-major channel-last order of include/irm_hip.h
    ([tile][8 x 32 pixels][C])."""
    B, C, H, W = t.shape
    return t.reshape(B, C, H // 8, 8, W // 32, 32).permute(0, 2, 4, 3, 5, 1).reshape(B, C, H, W).contiguous()


def from_tm(t):
    B, C, H, W = t.shape
    return t.reshape(B, H // 8, W // 32, 8, 32, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, H, W).contiguous()


@pytest.mark.parametrize("C,hid,H,W,B,lay", [(96, 255, 16, 64, 2, 7), (96, 255, 24, 32, 1, 1), (48, 127, 8, 96, 3, 2),
                                             (96, 255, 40, 64, 1, 4), (48, 127, 16, 32, 2, 5), (64, 170, 8, 64, 1, 3)])
def test_attn_gdfn_fused_tile_major_layouts(dev, C, hid, H, W, B, lay):
    """Every combination of tile-major x (bit 0) / v (bit 1) / y (bit 2) gives the bytes of the planar call."""
    tag = f"al{C}_{H}_{W}_{lay}"
    big = rnd(tag + "x", (B, C + 3, H, W), -1.5, 2.0)
    vbig = rnd(tag + "v", (B, 3 * C, H, W), -2.0, 2.0)
    mf = rnd(tag + "m", (B, C, C), -0.2, 0.2)
    lnw, lnb = rnd(tag + "lw", (C,), 0.5, 1.5), rnd(tag + "lb", (C,), -0.2, 0.2)
    pk = _hip.pack_gdfn_fused(rnd(tag + "pi", (2 * hid, C), -.3, .3).to(dev), None, rnd(tag + "dw", (2 * hid, 9), -.4, .4),
                              None, rnd(tag + "po", (C, hid), -.3, .3), lnw, lnb, kperm=True)
    frag = _hip.pack_mfold_frag(mf).to(dev)
    bo = rnd(tag + "bo", (C,), -0.3, 0.3).to(dev)
    y0 = torch.empty(B, C, H, W, device=dev)
    ops.attn_gdfn_fused(pk, big.to(dev)[:, 1:1 + C], vbig.to(dev)[:, 2 * C:], frag, y0, C, hid, ln_mode=1, bias_o=bo)
    xt, vt = big.clone(), vbig.clone()
    if lay & 1:
        xt[:, 1:1 + C] = to_tm(big[:, 1:1 + C])
    if lay & 2:
        vt[:, 2 * C:] = to_tm(vbig[:, 2 * C:])
    yb = torch.full((B, C + 2, H, W), 7.0, device=dev)
    ops.attn_gdfn_fused(pk, xt.to(dev)[:, 1:1 + C], vt.to(dev)[:, 2 * C:], frag, yb[:, 2:2 + C], C, hid, ln_mode=1, bias_o=bo,
                        x_tm=bool(lay & 1), v_tm=bool(lay & 2), y_tm=bool(lay & 4))
    y = yb[:, 2:2 + C].cpu()
    assert torch.equal(from_tm(y) if lay & 4 else y, y0.cpu())
    assert torch.all(yb[:, :2] == 7.0)


@pytest.mark.parametrize("H,W,B,ln,bias", [(16, 64, 2, 1, False), (24, 32, 1, 2, True), (8, 96, 3, 1, True), (40, 64, 1, 1, True)])
def test_gdfn_tail_c192(dev, H, W, B, ln, bias):
    """C = 192: LayerNorm + project_in with h written tile-major channel-last (irm_ln_gemm_presplit_cl_f16x3_f32), then the
    depth-wise conv, the gate, project_out and the residual in ONE kernel (irm_gdfn_tail_f16x3_f32), vs float64."""
    C, hid, hp = 192, 510, 512
    tag = f"gt{H}_{W}_{B}_{ln}"
    ws = 0.2
    big = rnd(tag + "x", (B, C + 3, H, W), -1.5, 2.0)
    lnw = rnd(tag + "lw", (C,), 0.5, 1.5)
    lnb = rnd(tag + "lb", (C,), -0.2, 0.2) if ln == 1 else None
    pin_w, pout_w = rnd(tag + "pi", (2 * hid, C), -ws, ws), rnd(tag + "po", (C, hid), -ws, ws)
    dw_w = rnd(tag + "dw", (2 * hid, 9), -0.4, 0.4)
    pin_b = rnd(tag + "pib", (2 * hid,), -0.3, 0.3) if bias else None
    dw_b = rnd(tag + "dwb", (2 * hid,), -0.3, 0.3) if bias else None
    pout_b = rnd(tag + "pob", (C,), -0.3, 0.3) if bias else None
    x = big[:, 1:1 + C]
    ref = gdfn_ref(x, lnw, lnb, ln, pin_w, pin_b, dw_w, dw_b, pout_w, pout_b)
    frag, s_w, bp = _hip.pack_pin_padded(pin_w.to(dev), None if pin_b is None else pin_b.to(dev), hp)
    s_x = _hip.ln_split_scale(lnw, lnb, C, ln == 1)
    pk = _hip.pack_gdfn_tail(dw_w, dw_b, pout_w.to(dev))
    xb = big.to(dev)
    h_cl = torch.full((B * 2 * hp * H * W + 64,), 7.0, device=dev)
    ops.ln_gemm_presplit_cl(frag, xb[:, 1:1 + C], h_cl, 2 * hp, C, lnw.to(dev), None if lnb is None else lnb.to(dev), ln, s_x,
                            out_scale=1.0 / (s_w * s_x), bias=bp)
    assert torch.all(h_cl[-64:] == 7.0), "wrote past h"
    # h itself: un-permute [B][tile][chunk of 64 channels][256][64] and compare with float64 project_in(LN(x))
    xd = x.double()
    mu, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    xn = ((xd - mu) if ln == 1 else xd) / torch.sqrt(var + 1e-5) * lnw.double()[None, :, None, None]
    if ln == 1:
        xn = xn + lnb.double()[None, :, None, None]
    href = F.conv2d(xn, pin_w.double()[:, :, None, None], None if pin_b is None else pin_b.double())
    hc = (h_cl[:-64].cpu().view(B, H // 8, W // 32, 2 * hp // 64, 8, 32, 64).permute(0, 3, 6, 1, 4, 2, 5)
          .reshape(B, 2 * hp, H, W))                     # [B][ty][tx][chunk][8][32][64] -> planar
    assert float((hc[:, :hid].double() - href[:, :hid]).abs().max()) <= TOL * max(1.0, float(href.abs().max()))
    assert float((hc[:, hp:hp + hid].double() - href[:, hid:]).abs().max()) <= TOL * max(1.0, float(href.abs().max()))
    assert float(hc[:, hid:hp].abs().max()) == 0.0 and float(hc[:, hp + hid:].abs().max()) == 0.0
    ops.gdfn_tail(pk, h_cl, xb[:, 1:1 + C], C, hid, hp, bias=None if pout_b is None else pout_b.to(dev))
    y = xb.cpu()
    scale = max(1.0, float(ref.abs().max()))
    err = float((y[:, 1:1 + C].double() - ref).abs().max())
    assert err <= TOL * scale, (err, scale)
    assert torch.equal(y[:, 0], big[:, 0]) and torch.equal(y[:, 1 + C:], big[:, 1 + C:]), "wrote outside its channel slice"


def test_gdfn_fused_deterministic(dev):
    C, hid, H, W = 96, 255, 32, 64
    x = rnd("detx", (2, C, H, W), -2, 2).to(dev)
    pk = _hip.pack_gdfn_fused(rnd("d1", (2 * hid, C), -.3, .3).to(dev), None, rnd("d2", (2 * hid, 9), -.4, .4), None,
                              rnd("d3", (C, hid), -.3, .3), rnd("d4", (C,), .5, 1.5), rnd("d5", (C,), -.2, .2))
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    ops.gdfn_fused(pk, x, y1, C, hid, ln_mode=1)
    ops.gdfn_fused(pk, x, y2, C, hid, ln_mode=1)
    assert torch.equal(y1, y2)


def qkv_ref(x, lnw, lnb, ln_mode, w, b, dw_w, dw_b, eps=1e-5):
    xd = x.double()
    mu = xd.mean(1, keepdim=True)
    var = xd.var(1, unbiased=False, keepdim=True)
    if ln_mode == _hip.LN_WITHBIAS:
        xn = (xd - mu) / torch.sqrt(var + eps) * lnw.double()[None, :, None, None] + lnb.double()[None, :, None, None]
    else:
        xn = xd / torch.sqrt(var + eps) * lnw.double()[None, :, None, None]
    M = w.shape[0]
    h = F.conv2d(xn, w.double()[:, :, None, None], None if b is None else b.double())
    return F.conv2d(h, dw_w.double().view(M, 1, 3, 3), None if dw_b is None else dw_b.double(), padding=1, groups=M)


QKV_CASES = [
    # C, H, W, B, ln_mode, bias
    (96, 16, 64, 1, 1, False),
    (96, 24, 40, 2, 2, True),
    (48, 16, 32, 2, 1, True),          # M = 144: the last stage is half empty
    (48, 20, 36, 1, 2, False),
    (96, 8, 8, 1, 1, False),
    (32, 12, 20, 1, 1, True),
    (64, 40, 72, 1, 2, False),
]


@pytest.mark.parametrize("C,H,W,B,ln,bias", QKV_CASES)
def test_qkv_dw_fused(dev, C, H, W, B, ln, bias):
    tag = f"q{C}_{H}_{W}_{B}_{ln}"
    M = 3 * C
    big = rnd(tag + "x", (B, C + 3, H, W), -1.5, 2.0)
    lnw = rnd(tag + "lw", (C,), 0.5, 1.5)
    lnb = rnd(tag + "lb", (C,), -0.2, 0.2) if ln == 1 else None
    w = rnd(tag + "w", (M, C), -0.3, 0.3)
    dw_w = rnd(tag + "dw", (M, 9), -0.4, 0.4)
    wb = rnd(tag + "wb", (M,), -0.3, 0.3) if bias else None
    dw_b = rnd(tag + "dwb", (M,), -0.3, 0.3) if bias else None
    x = big[:, 1:1 + C]
    ref = qkv_ref(x, lnw, lnb, ln, w, wb, dw_w, dw_b)
    pk = _hip.pack_qkv_fused(w.to(dev), wb, dw_w, dw_b, lnw, lnb)
    xb = big.to(dev)
    yb = torch.full((B, M + 2, H, W), 7.0, device=dev)
    ops.qkv_dw_fused(pk, xb[:, 1:1 + C], yb[:, 1:1 + M], C, M, ln_mode=ln)
    y = yb.cpu()
    scale = max(1.0, float(ref.abs().max()))
    err = float((y[:, 1:1 + M].double() - ref).abs().max())
    assert err <= TOL * scale, (err, scale)
    assert torch.all(y[:, 0] == 7.0) and torch.all(y[:, M + 1] == 7.0), "wrote outside its channel slice"


@pytest.mark.parametrize("C,heads,H,W,B,f16", [(96, 1, 16, 64, 2, True), (96, 2, 24, 32, 1, True), (48, 1, 8, 96, 3, True),
                                               (96, 1, 40, 64, 1, True), (96, 2, 16, 64, 2, False), (48, 1, 16, 32, 1, False)])
def test_qk_tile_major_chain(dev, C, heads, H, W, B, f16):
    """qkv_dw_fused(tm) + Gram pass (tm): q, k tile-major [tile][2C][256] - v, the attention matrix and the folded matrix
    as with planar q, k (the Gram sum runs over the same pixels in another order: a few ulps)."""
    tag = f"tm{C}_{heads}_{H}_{W}"
    M, N = 3 * C, H * W
    x = rnd(tag + "x", (B, C, H, W), -1.5, 2.0).to(dev)
    lnw, lnb = rnd(tag + "lw", (C,), 0.5, 1.5), rnd(tag + "lb", (C,), -0.2, 0.2)
    w, dw_w = rnd(tag + "w", (M, C), -0.3, 0.3), rnd(tag + "dw", (M, 9), -0.4, 0.4)
    pk = _hip.pack_qkv_fused(w.to(dev), None, dw_w, None, lnw, lnb)
    # (f16: the Gram pass emulated on the fp16 matrix cores; else the f32-input ring pass, as under a BiasFree LayerNorm)
    gs = _hip.gram_scales(w.view(M, C, 1, 1), None, dw_w.view(M, 1, 3, 3), None, lnw, lnb, True).to(dev) if f16 else None
    temp, wout = rnd(tag + "t", (heads,), 2.0, 6.0).to(dev), rnd(tag + "wo", (C, C), -0.3, 0.3).to(dev)
    res = []
    for tm in (False, True, 2):
        qkv = torch.full((B, M, H, W), 7.0, device=dev)
        if tm == 2:        # x read tile-major, v written tile-major as well
            ops.qkv_dw_fused(pk, to_tm(x.cpu()).to(dev), qkv, C, M, ln_mode=1, tm=True, x_tm=True, v_tm=True)
            v_back = from_tm(qkv[:, 2 * C:].cpu())
            assert torch.equal(v_back, res[0][0][:, 2 * C:]) and torch.equal(qkv[:, :2 * C].cpu(), res[1][0][:, :2 * C])
            continue
        ops.qkv_dw_fused(pk, x, qkv, C, M, ln_mode=1, tm=tm)
        _, nchunk, rec = ops.mdta_plan(B, C, heads, N)
        part = torch.full((B * heads * nchunk * rec,), float("nan"), device=dev)
        gsum = torch.empty(B * heads * rec, device=dev)
        mfold = torch.zeros(B * ops.mfold_numel(C), device=dev)
        attn = torch.empty(B, heads, C // heads, C // heads, device=dev)
        ops.mdta_fold(qkv, part, gsum, temp, wout, mfold, C, heads, attn=attn, gram_scale=gs, tm=tm)
        res.append((qkv.cpu(), attn.cpu(), mfold.cpu()))
    (q0, a0, m0), (q1, a1, m1) = res
    assert torch.equal(q0[:, 2 * C:], q1[:, 2 * C:]), "v must not depend on the q, k layout"
    # the tile-major block of tile (ty, tx) holds channel ch of pixels (8 ty + r, 32 tx + c) at [ch][32 r + c]
    back = q1[:, :2 * C].reshape(B, H // 8, W // 32, 2 * C, 8, 32).permute(0, 3, 1, 4, 2, 5).reshape(B, 2 * C, H, W)
    assert torch.equal(back, q0[:, :2 * C]), "tile-major q, k are not a permutation of the planar ones"
    assert float((a0 - a1).abs().max()) <= 2e-6 and float((m0 - m1).abs().max()) <= 2e-6


@pytest.mark.parametrize("H,W,B,x_tm", [(32, 32, 2, False), (16, 128, 1, True), (64, 64, 3, True)])
def test_qkv_gram_cm_c48(dev, H, W, B, x_tm):
    """qkv_gram_cm (C = 48, one head): v bit-identical to qkv_dw_fused, the attention matrix and the folded matrix as from
    the tile-major q, k + Gram pass (other summation order: a few ulps); batch-independent (image 0 alone == image 0 of B)."""
    C, heads = 48, 1
    tag = f"qg{H}_{W}"
    M, N = 3 * C, H * W
    x = rnd(tag + "x", (B, C, H, W), -1.5, 2.0).to(dev)
    lnw, lnb = rnd(tag + "lw", (C,), 0.5, 1.5), rnd(tag + "lb", (C,), -0.2, 0.2)
    w, dw_w = rnd(tag + "w", (M, C), -0.3, 0.3), rnd(tag + "dw", (M, 9), -0.4, 0.4)
    pk = _hip.pack_qkv_fused(w.to(dev), None, dw_w, None, lnw, lnb)
    gs = _hip.gram_scales(w.view(M, C, 1, 1), None, dw_w.view(M, 1, 3, 3), None, lnw, lnb, True).to(dev)
    temp, wout = rnd(tag + "t", (heads,), 2.0, 6.0).to(dev), rnd(tag + "wo", (C, C), -0.3, 0.3).to(dev)
    assert ops.can_qkv_gram(C, heads, H, W)
    xin = to_tm(x.cpu()).to(dev) if x_tm else x

    def fold(fused, xi, nb):
        qkv = torch.full((nb, M, H, W), 7.0, device=dev)
        _, nchunk, rec = ops.mdta_plan(nb, C, heads, N)
        nready = None
        if fused:
            nchunk = (H // 8) * (W // 32) // ops.QKV_GRAM_NCH
            part = torch.full((nb * nchunk * rec,), float("nan"), device=dev)
            nready = ops.qkv_gram_cm(pk, xi, qkv, gs, part, C, ln_mode=1, x_tm=x_tm, v_tm=True)
            assert nready == nchunk
            assert torch.all(qkv[:, :2 * C] == 7.0), "q, k must not be written"
        else:
            part = torch.full((nb * nchunk * rec,), float("nan"), device=dev)
            ops.qkv_dw_fused(pk, xi, qkv, C, M, ln_mode=1, tm=True, x_tm=x_tm, v_tm=True)
        gsum = torch.empty(nb * rec, device=dev)
        mfold = torch.zeros(nb * ops.mfold_numel(C), device=dev)
        attn = torch.empty(nb, heads, C, C, device=dev)
        ops.mdta_fold(qkv, part, gsum, temp, wout, mfold, C, heads, attn=attn, gram_scale=gs, tm=True, nchunk_ready=nready)
        return qkv[:, 2 * C:].cpu(), attn.cpu(), mfold.cpu()

    v0, a0, m0 = fold(False, xin, B)
    v1, a1, m1 = fold(True, xin, B)
    assert torch.equal(v0, v1), "v differs from qkv_dw_fused"
    assert float((a0 - a1).abs().max()) <= 2e-6 and float((m0 - m1).abs().max()) <= 2e-6, \
        (float((a0 - a1).abs().max()), float((m0 - m1).abs().max()))
    if B > 1:
        _, a2, m2 = fold(True, xin[:1].contiguous(), 1)
        assert torch.equal(a2[0], a1[0]) and torch.equal(m2.view(-1), m1.view(B, -1)[0]), "depends on the batch"


def _logu(name, shape, lo, hi):
    """log-uniform magnitudes in [lo, hi] with random signs (trained checkpoints: weights over many decades)."""
    u = rnd(name, shape, 0.0, 1.0)
    s = torch.where(rnd(name + "s", shape) < 0, -1.0, 1.0)
    return s * torch.exp(torch.log(torch.tensor(lo)) + u * (torch.log(torch.tensor(hi)) - torch.log(torch.tensor(lo))))


@pytest.mark.parametrize("act_scale", [1e-6, 1.0, 1e4])
@pytest.mark.parametrize("ln", [1, 2])
def test_fused_trained_like_statistics(dev, act_scale, ln):
    """LayerNorm gains up to 30, weights spanning 1e-5 ... 10, activations of 1e-6 and 1e4.
    The fused kernels scale their fp16 operands by powers of two chosen at pack time (any weight magnitude) and
    normalise in fp32 before the split, so the result must stay within 2x of a plain fp32 chain's error vs float64
    (plus the fp32 rounding of the output itself).  One documented exception (precision): the gated
    activations are split after a FIXED 2^-4 scaling, so activations below ~1e-4 reach the fp16 subnormals and the
    branch output carries an ABSOLUTE error floor of ~1e-6 (seen here with BiasFree LayerNorm on 1e-6 inputs)."""
    C, hid, H, W, B = 96, 255, 24, 40, 1
    tag = f"tl{ln}_{act_scale}"
    x = rnd(tag + "x", (B, C, H, W), -1.5, 2.0) * act_scale
    lnw = _logu(tag + "lw", (C,), 0.1, 30.0).abs()
    lnb = rnd(tag + "lb", (C,), -2.0, 2.0) if ln == 1 else None
    pin_w, pout_w = _logu(tag + "pi", (2 * hid, C), 1e-5, 10.0), _logu(tag + "po", (C, hid), 1e-5, 3.0) / hid ** 0.5
    dw_w = _logu(tag + "dw", (2 * hid, 9), 1e-4, 1.0)
    ref = gdfn_ref(x, lnw, lnb, ln, pin_w, None, dw_w, None, pout_w, None)
    xd = x.to(dev)
    # plain fp32 chain on the same device (torch kernels) as the yardstick
    mu = xd.mean(1, keepdim=True)
    var = xd.var(1, unbiased=False, keepdim=True)
    xn = ((xd - mu) if ln == 1 else xd) / torch.sqrt(var + 1e-5) * lnw.to(dev)[None, :, None, None]
    if ln == 1:
        xn = xn + lnb.to(dev)[None, :, None, None]
    h = F.conv2d(F.conv2d(xn, pin_w.to(dev)[:, :, None, None]), dw_w.to(dev).view(-1, 1, 3, 3), padding=1, groups=2 * hid)
    y32 = xd + F.conv2d(F.gelu(h[:, :hid]) * h[:, hid:], pout_w.to(dev)[:, :, None, None])
    y = torch.empty_like(xd)
    ops.gdfn_fused(_hip.pack_gdfn_fused(pin_w.to(dev), None, dw_w, None, pout_w, lnw, lnb), xd, y, C, hid, ln_mode=ln)
    scale = float(ref.abs().max())
    e16, e32 = float((y.cpu().double() - ref).abs().max()), float((y32.cpu().double() - ref).abs().max())
    print(f"gdfn act x{act_scale:g} ln{ln}: |ref|max {scale:.3e}  fused {e16:.3e}  fp32 chain {e32:.3e}")
    assert e16 <= 1e-3 * max(1.0, scale) and e16 <= 2.0 * e32 + 4e-7 * scale + 2e-6
    # the qkv branch with the same statistics
    M = 3 * C
    w, dq = _logu(tag + "qw", (M, C), 1e-5, 10.0), _logu(tag + "qd", (M, 9), 1e-4, 1.0)
    refq = qkv_ref(x, lnw, lnb, ln, w, None, dq, None)
    q32 = F.conv2d(F.conv2d(xn, w.to(dev)[:, :, None, None]), dq.to(dev).view(-1, 1, 3, 3), padding=1, groups=M)
    yq = torch.empty(B, M, H, W, device=dev)
    ops.qkv_dw_fused(_hip.pack_qkv_fused(w.to(dev), None, dq, None, lnw, lnb), xd, yq, C, M, ln_mode=ln)
    scale = float(refq.abs().max())
    e16, e32 = float((yq.cpu().double() - refq).abs().max()), float((q32.cpu().double() - refq).abs().max())
    print(f"qkv  act x{act_scale:g} ln{ln}: |ref|max {scale:.3e}  fused {e16:.3e}  fp32 chain {e32:.3e}")
    assert e16 <= 1e-3 * max(1.0, scale) and e16 <= 2.0 * e32 + 4e-7 * scale


def test_gate_out_of_range_saturates_instead_of_nan(dev):
    """the gated activations are split to fp16 behind a fixed 2^-4 scale.  |gelu(h1) h2| beyond ~1e6 used to
    become fp16 infinities and a NaN output tile; the scaled value is now saturated at +-65000 before the split
    (irm_sat_h), in the fused branch kernel and in the streaming emulated GEMM alike: the result is finite (and, being
    out of the emulation's documented range, not accurate)."""
    C, hid, H, W = 96, 255, 16, 32
    x = rnd("satx", (1, C, H, W), -1.5, 2.0).to(dev)
    pin_w = rnd("satpi", (2 * hid, C), -600.0, 600.0)
    dw_w, pout_w = rnd("satdw", (2 * hid, 9), -0.4, 0.4), rnd("satpo", (C, hid), -0.1, 0.1)
    lnw, lnb = rnd("satlw", (C,), 0.5, 1.5), rnd("satlb", (C,), -0.2, 0.2)
    ref = gdfn_ref(x.cpu(), lnw, lnb, 1, pin_w, None, dw_w, None, pout_w, None)
    gate_max = float((ref - x.cpu().double()).abs().max())
    y = torch.empty_like(x)
    ops.gdfn_fused(_hip.pack_gdfn_fused(pin_w.to(dev), None, dw_w, None, pout_w, lnw, lnb), x, y, C, hid, ln_mode=1)
    assert gate_max > 1e6 and bool(torch.isfinite(y).all())
    # streaming emulated GEMM without the LayerNorm prologue (2^-4 split of the raw input)
    g = (rnd("satg", (1, hid, H, W), -1.0, 1.0) * 3e7).to(dev)
    out = torch.empty_like(x)
    ops.gemm1x1(_hip.pack_gemm_weight_split(pout_w.to(dev)), g, out, C, hid, res=x, split=True)
    assert bool(torch.isfinite(out).all())
