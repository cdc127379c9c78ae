"""Emulated LN-GEMMs of the C = 192 level (input-resident kernel) - A/B of library variants: --lib path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irm_amd  # noqa
from irm_amd import _hip, ops
if "--lib" in sys.argv:
    _hip.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
dev = torch.device("cuda:0")
for (M, K, H, W) in [(1020, 192, 128, 128), (576, 192, 128, 128), (768, 192, 64, 64), (288, 192, 64, 64)]:
    B = 6
    x = torch.randn(B, K, H, W, device=dev); y = torch.empty(B, M, H, W, device=dev)
    w = torch.randn(M, K) * 0.1
    st = torch.empty(B, 2, H * W, device=dev); ops.ln_stats(x, st)
    lnw, lnb = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    ws = _hip.pack_gemm_weight_split(w).to(dev)
    fn = lambda: ops.gemm1x1(ws, x, y, M, K, stats=st, lnw=lnw, lnb=lnb, ln_mode=1, split=True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    ref = (w.double().to(dev) @ torch.nn.functional.layer_norm(x.double().permute(0, 2, 3, 1), (K,)).permute(0, 3, 1, 2).reshape(B, K, -1)).reshape(B, M, H, W)
    err = float((y.double() - ref).abs().max())
    print(f"M{M} K{K} {H}x{W}: {t*1e6:8.1f} us  {4.0*B*H*W*(K+M)/t/1e9:6.0f} GB/s  err {err:.1e}", flush=True)
