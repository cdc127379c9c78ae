"""Phase shares of the fused kernels from the FB_STAMP diagnostic build (tools/variants/libirm_stamp.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import irm_amd  # noqa
from irm_amd import _hip, ops, synth
_hip.LIB_PATH = os.path.abspath(os.environ.get("IRM_STAMP_LIB", "tools/variants/libirm_stamp.so"))
dev = torch.device("cuda:0")
dbg = torch.zeros(256 * 10, dtype=torch.int64, device=dev)
os.environ["FB_DBG_PTR"] = str(dbg.data_ptr())
names = ["tail->top", "LN+split", "residual issue", "wait DMA+barrier", "GEMM(0)", "iterations (more)", "last iteration", "epilogue", "APPLY: v wait + barrier", "APPLY: MFMAs + barrier"]
for C, hid, H, W, B in [(96, 255, 512, 512, 6), (48, 127, 512, 512, 6)]:
    r = lambda n, s, lo=-1., hi=1.: synth.uniform(5, n, s, lo, hi)
    x = torch.randn(B, C, H, W, device=dev); y = torch.empty_like(x); yq = torch.empty(B, 3 * C, H, W, device=dev)
    pk = _hip.pack_gdfn_fused(r("a", (2 * hid, C), -.3, .3).to(dev), None, r("b", (2 * hid, 9), -.4, .4), None,
                              r("c", (C, hid), -.3, .3), r("d", (C,), .5, 1.5), r("e", (C,), -.2, .2))
    pkq = _hip.pack_qkv_fused(r("a2", (3 * C, C), -.3, .3).to(dev), None, r("b2", (3 * C, 9), -.4, .4), None, r("d", (C,), .5, 1.5), r("e", (C,), -.2, .2))
    pka = _hip.pack_gdfn_fused(r("a", (2 * hid, C), -.3, .3).to(dev), None, r("b", (2 * hid, 9), -.4, .4), None,
                               r("c", (C, hid), -.3, .3), r("d", (C,), .5, 1.5), r("e", (C,), -.2, .2), kperm=True)
    frag = _hip.pack_mfold_frag(r("m", (B, C, C), -.2, .2)).to(dev)
    for name, fn in (("gdfn", lambda: ops.gdfn_fused(pk, x, y, C, hid, ln_mode=1)), ("qkv", lambda: ops.qkv_dw_fused(pkq, x, yq, C, 3 * C, ln_mode=1)),
                     ("attn_gdfn", lambda: ops.attn_gdfn_fused(pka, x, yq[:, 2 * C:], frag, y, C, hid, ln_mode=1)),
                     ("qkv tile-major (q, k; x, v channel-last)", lambda: ops.qkv_dw_fused(pkq, x, yq, C, 3 * C, ln_mode=1, tm=True, x_tm=True, v_tm=True)),
                     ("attn_gdfn channel-last x, v, y", lambda: ops.attn_gdfn_fused(pka, x, yq[:, 2 * C:], frag, y, C, hid, ln_mode=1, x_tm=True, v_tm=True, y_tm=True))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); dbg.zero_(); fn(); torch.cuda.synchronize()
        d = dbg.view(256, 10).double().cpu()
        tot = d.sum(1).mean()
        print(f"{name} C{C}: {tot:.0f} cycles per workgroup (24 items);  " + "  ".join(f"{n} {100 * d[:, i].mean() / tot:.1f}%" for i, n in enumerate(names)))
