"""Residual 1x1 convs (attention apply / project_out shapes of the headline) - A/B of library variants: --lib path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irm_amd  # noqa
from irm_amd import _hip, ops
if "--lib" in sys.argv:
    _hip.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
dev = torch.device("cuda:0")
B = 6
for (M, K, H, split) in [(96, 96, 512, True), (48, 48, 512, True), (96, 96, 256, True), (192, 510, 128, True), (384, 1021, 64, True),
                         (192, 192, 128, True), (96, 192, 256, False), (96, 96, 512, False)]:
    x = torch.randn(B, K, H, H, device=dev); r = torch.randn(B, M, H, H, device=dev); y = torch.empty(B, M, H, H, device=dev)
    w = torch.randn(M, K) * 0.1
    wp = (_hip.pack_gemm_weight_split(w) if split else _hip.pack_gemm_weight(w)).to(dev)
    fn = lambda: ops.gemm1x1(wp, x, y, M, K, res=r, split=split)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    ref = torch.einsum("mk,bkn->bmn", w.double().to(dev), x.double().reshape(B, K, -1)).reshape(B, M, H, H) + r.double()
    err = float((y.double() - ref).abs().max())
    print(f"M{M} K{K} {H}x{H} split={int(split)}: {t*1e6:8.1f} us  {4.0*B*H*H*(K+2*M)/t/1e9:6.0f} GB/s  err {err:.1e}", flush=True)
