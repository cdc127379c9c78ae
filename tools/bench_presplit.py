"""LayerNorm + 1x1 conv of the C >= 192 levels: ln_stats + emulated GEMM (gemm_xres / gemm_ring) against
ln_split + pre-split GEMM (gemm_ps.hip) on the headline shapes (6 tiles).  --lib path: A/B of library variants;
--plans: sweep (ct, mgroups) of the pre-split GEMM."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irm_amd  # noqa
from irm_amd import _hip, ops
if "--lib" in sys.argv:
    _hip.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
dev = torch.device("cuda:0")


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


B = 6
for (M, K, H, W) in [(1020, 192, 128, 128), (576, 192, 128, 128), (2042, 384, 64, 64), (1152, 384, 64, 64)]:
    N = H * W
    x = torch.randn(B, K, H, W, device=dev)
    y = torch.empty(B, M, H, W, device=dev)
    y2 = torch.empty(B, M, H, W, device=dev)
    w = torch.randn(M, K) * 0.1
    st = torch.empty(B, 2, N, device=dev)
    lnw, lnb = torch.rand(K, device=dev) + 0.5, torch.rand(K, device=dev) - 0.5
    ws = _hip.pack_gemm_weight_split(w).to(dev)
    frag, s_w = _hip.pack_gemm_weight_presplit(w.to(dev))
    s_x = _hip.ln_split_scale(lnw, lnb, K, True)
    xs = torch.empty(B * K * N, device=dev)
    t_stats = timeit(lambda: ops.ln_stats(x, st))
    t_old = timeit(lambda: ops.gemm1x1(ws, x, y, M, K, stats=st, lnw=lnw, lnb=lnb, ln_mode=1, split=True))
    t_split = timeit(lambda: ops.ln_split(x, xs, lnw, lnb, 1, s_x))
    kw = {}
    t_new = timeit(lambda: ops.gemm_presplit(frag, xs, y2, M, K, out_scale=1.0 / (s_w * s_x), **kw))
    t_fused = None
    if K == 192:
        t_fused = timeit(lambda: _hip.call("irm_ln_gemm_presplit_f16x3_f32", _hip.ptr(frag), _hip.ptr(x), x.stride(0), _hip.ptr(lnw),
                                           _hip.ptr(lnb), 1, float(s_x), 1e-5, _hip.ptr(y2), y2.stride(0), None,
                                           float(1.0 / (s_w * s_x)), B, M, K, N, 1))
        print(f"   LN fused into the GEMM: {t_fused:6.1f} us")
    err = float((y - y2).abs().max())
    gb = 4.0 * B * N * (K + M) / 1e3
    print(f"M{M} K{K} {H}x{W}: ln_stats {t_stats:6.1f} + gemm {t_old:6.1f} = {t_stats + t_old:6.1f} us | ln_split {t_split:6.1f} "
          f"+ gemm_ps {t_new:6.1f} = {t_split + t_new:6.1f} us ({gb / t_new:5.0f} GB/s, {6e-6 * B * M * K * N / t_new:5.0f} TF f16) "
          f"| old vs new max-abs {err:.1e}", flush=True)
    if "--plans" in sys.argv:
        mt = (M + 15) // 16
        shapes = [(42, 8), (42, 6), (32, 8), (32, 6), (43, 4)] if K == 192 else [(81, 8), (81, 6), (41, 8), (41, 6)]
        for shp, ct in shapes:
            chunks = -(-mt // ct)
            for mg in (1, 2, 3, 4):
                if mg > chunks or (mg - 1) * -(-chunks // mg) >= chunks:
                    continue
                t = timeit(lambda: ops.gemm_presplit(frag, xs, y2, M, K, out_scale=1.0 / (s_w * s_x), ct=ct, mgroups=mg,
                                                     wg_shape=shp), 10)
                print(f"    shape {shp} ct {ct} mgroups {mg}: {t:6.1f} us", flush=True)
