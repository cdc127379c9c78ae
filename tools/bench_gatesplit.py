"""GDFN tail of the C >= 192 levels: dwconv3x3_gate + streaming emulated GEMM (residual) against
dwconv3x3_gate_split + K-streamed pre-split GEMM (gemm_ps.hip), headline shapes (6 tiles)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irm_amd  # noqa
from irm_amd import _hip, ops
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


B = 6
for (M, hid, H, W) in [(192, 510, 128, 128), (384, 1021, 64, 64)]:
    N = H * W
    KS = -(-hid // 32)
    h = torch.randn(B, 2 * hid, H, W, device=dev)
    w9 = torch.randn(2 * hid, 9, device=dev) * 0.3
    w2 = torch.randn(M, hid) * 0.05
    g = torch.empty(B, hid, H, W, device=dev)
    gs = torch.empty(B * 32 * KS * N, device=dev)
    y = torch.randn(B, M, H, W, device=dev)
    y2 = y.clone()
    ws = _hip.pack_gemm_weight_split(w2).to(dev)
    frag, s_w = _hip.pack_gemm_weight_presplit(w2.to(dev), k_pad=32 * KS)
    sc = 1.0 / (s_w * ops.GATE_SPLIT_SCALE)
    t_g = timeit(lambda: ops.dwconv3x3_gate(h, w9, g))
    t_p = timeit(lambda: ops.gemm1x1(ws, g, y, M, hid, res=y, split=True))
    tt = {ch: timeit(lambda: ops.dwconv3x3_gate_split(h, w9, gs, ch=ch)) for ch in (8, 16, 32)}
    t_gs = timeit(lambda: ops.dwconv3x3_gate_split(h, w9, gs))
    line = f"M{M} hid{hid} {H}x{W}: gate {t_g:6.1f} + gemm {t_p:6.1f} = {t_g + t_p:6.1f} us | gate_split ch8/16/32 {tt[8]:.1f}/{tt[16]:.1f}/{tt[32]:.1f} auto {t_gs:6.1f}"
    for shp in ((42, 82) if M <= 192 else (41, 81)):
        t = timeit(lambda: ops.gemm_presplit_res(frag, gs, y2, M, KS, out_scale=sc, res=y2, wg_shape=shp))
        line += f" + ps_res[{shp}] {t:6.1f} = {t_gs + t:6.1f} us"
    print(line, flush=True)
