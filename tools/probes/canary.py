"""Out-of-bounds write check: every op writes into the middle of a sentinel-filled buffer (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from irm_amd import ops, _hip, synth
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
PAD = 1 << 16
def guarded(shape):
    n = 1
    for s in shape: n *= s
    buf = torch.full((n + 2 * PAD,), 12345.0, device=dev)
    return buf, buf[PAD:PAD + n].view(*shape)
def check(name, buf, n):
    lo, hi = buf[:PAD], buf[PAD + n:]
    bad = int((lo != 12345.0).sum()) + int((hi != 12345.0).sum())
    print(f"{name:60s} {'OK' if bad == 0 else f'*** {bad} sentinel floats overwritten ***'}", flush=True)
B = 3
r = lambda n, s, lo=-1., hi=1.: synth.uniform(5, n, s, lo, hi)
# dense 3x3 convs: the shapes of Restormer (store modes 0 / 1 / 2)
for ci, co, H, mode in ((3, 48, 512, 0), (48, 24, 512, 1), (96, 48, 256, 1), (192, 96, 128, 1), (384, 768, 64, 2), (192, 384, 128, 2), (96, 192, 256, 2), (96, 3, 512, 0)):
    X = torch.randn(B, ci, H, H, generator=g).to(dev)
    cw = _hip.pack_conv3x3((torch.randn(co, ci, 3, 3, generator=g) * 0.05).to(dev))
    oc, oh = (co * 4, H // 2) if mode == 1 else (co // 4, H * 2) if mode == 2 else (co, H)
    buf, y = guarded((B, oc, oh, oh))
    ops.conv3x3(cw, X, y, ci, co, store_mode=mode); torch.cuda.synchronize()
    check(f"conv3x3_f16x3 ci{ci} co{co} {H} mode{mode}", buf, y.numel())
    cwe = _hip.pack_conv3x3_weight((torch.randn(co, ci, 3, 3, generator=g) * 0.05)).to(dev)
    buf, y = guarded((B, oc, oh, oh))
    ops.conv3x3(cwe, X, y, ci, co, store_mode=mode); torch.cuda.synchronize()
    check(f"conv3x3 (exact) ci{ci} co{co} {H} mode{mode}", buf, y.numel())
for C, H in ((48, 512), (96, 256), (192, 128), (384, 64)):
    X = torch.randn(B, C, H, H, generator=g).to(dev)
    buf, st = guarded((B, 2, H, H)); ops.ln_stats(X, st); torch.cuda.synchronize(); check(f"ln_stats C{C} {H}", buf, st.numel())
    hid = int(C * 2.66)
    lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    stt = torch.empty(B, 2, H, H, device=dev); ops.ln_stats(X, stt)
    for M in (3 * C, 2 * hid):
        for split in (True, False):
            w = torch.randn(M, C, generator=g) * 0.1
            wp = (_hip.pack_gemm_weight_split(w) if split else _hip.pack_gemm_weight(w)).to(dev)
            buf, y = guarded((B, M, H, H))
            ops.gemm1x1(wp, X, y, M, C, stats=stt, lnw=lnw, lnb=lnb, ln_mode=1, split=split); torch.cuda.synchronize()
            check(f"gemm1x1 LN M{M} K{C} {H} split={int(split)}", buf, y.numel())
    G = torch.randn(B, hid, H, H, generator=g).to(dev)
    for split in (True, False):
        w = torch.randn(C, hid, generator=g) * 0.1
        wp = (_hip.pack_gemm_weight_split(w) if split else _hip.pack_gemm_weight(w)).to(dev)
        buf, y = guarded((B, C, H, H)); y.copy_(X)
        ops.gemm1x1(wp, G, y, C, hid, res=y, split=split); torch.cuda.synchronize()
        check(f"gemm1x1 res in place M{C} K{hid} {H} split={int(split)}", buf, y.numel())
    Hh = torch.randn(B, 2 * hid, H, H, generator=g).to(dev)
    dwg = (torch.randn(2 * hid, 9, generator=g) * 0.3).to(dev)
    buf, y = guarded((B, hid, H, H)); ops.dwconv3x3_gate(Hh, dwg, y); torch.cuda.synchronize(); check(f"dwconv3x3_gate hid{hid} {H}", buf, y.numel())
    QKV = torch.randn(B, 3 * C, H, H, generator=g).to(dev)
    dw = (torch.randn(3 * C, 9, generator=g) * 0.3).to(dev)
    buf, y = guarded((B, 3 * C, H, H)); ops.dwconv3x3(QKV, dw, y); torch.cuda.synchronize(); check(f"dwconv3x3 C{3*C} {H}", buf, y.numel())
    heads = max(1, C // 48)
    temp = torch.ones(heads, device=dev); wout = (torch.randn(C, C, generator=g) * 0.2).to(dev)
    chunk, nchunk, rec = ops.mdta_plan(B, C, heads, H * H)
    sc = torch.full((2 * C,), 1024.0, device=dev)
    for gs, split in ((None, False), (sc, True)):
        bp, part = guarded((B * heads * nchunk * rec,)); bg, gsum = guarded((B * heads * rec,)); bm, mf = guarded((B * ops.mfold_numel(C),))
        mf.zero_()
        ops.mdta_fold(QKV, part, gsum, temp, wout, mf, C, heads, split=split, gram_scale=gs); torch.cuda.synchronize()
        check(f"mdta part C{C} h{heads} {H} f16x3={gs is not None}", bp, part.numel()); check("   gsum", bg, gsum.numel()); check("   mfold", bm, mf.numel())
    if C <= 96:
        pk = _hip.pack_gdfn_fused(r("a", (2 * hid, C), -.3, .3).to(dev), None, r("b", (2 * hid, 9), -.4, .4), None,
                                  r("c", (C, hid), -.3, .3), r("d", (C,), .5, 1.5), r("e", (C,), -.2, .2))
        buf, y = guarded((B, C, H, H)); ops.gdfn_fused(pk, X, y, C, hid, ln_mode=1); torch.cuda.synchronize(); check(f"gdfn_fused C{C} {H}", buf, y.numel())
        pkq = _hip.pack_qkv_fused(r("a2", (3 * C, C), -.3, .3).to(dev), None, r("b2", (3 * C, 9), -.4, .4), None, r("d", (C,), .5, 1.5), r("e", (C,), -.2, .2))
        buf, y = guarded((B, 3 * C, H, H)); ops.qkv_dw_fused(pkq, X, y, C, 3 * C, ln_mode=1); torch.cuda.synchronize(); check(f"qkv_dw_fused C{C} {H}", buf, y.numel())
