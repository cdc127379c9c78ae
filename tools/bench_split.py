"""Split (fp16x3) vs exact GEMM at the big pin / qkv shapes; IRM_GEMM_DBG bits apply to both (1 no DMA, 2 no stores)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import irm_amd
from irm_amd import _hip, ops
dev = torch.device("cuda:0")
for (M, K, H, W) in [(510, 96, 512, 512), (288, 96, 512, 512), (510, 96, 256, 256)]:
    B = 6
    x = torch.randn(B, K, H, W, device=dev); y = torch.empty(B, M, H, W, device=dev)
    w = torch.randn(M, K) * 0.1
    st = torch.empty(B, 2, H * W, device=dev); ops.ln_stats(x, st)
    lnw, lnb = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    we, ws = _hip.pack_gemm_weight(w).to(dev), _hip.pack_gemm_weight_split(w).to(dev)
    for name, wp, split in (("exact", we, False), ("split", ws, True)):
        fn = lambda: ops.gemm1x1(wp, x, y, M, K, stats=st, lnw=lnw, lnb=lnb, ln_mode=1, split=split)
        for _ in range(2): fn()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[2] * 1e-3
        print(f"M{M} K{K} {H}x{W} {name}: {t*1e6:8.1f} us  {2.0*B*M*K*H*W/t/1e12:6.1f} TF-equiv  {4.0*B*H*W*(K+M)/t/1e9:6.0f} GB/s", flush=True)
