import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import irm_amd
from irm_amd import _hip, ops
dev = torch.device("cuda:0")
for (M, K, H, W) in [(510, 96, 512, 512), (510, 96, 512, 520), (288, 96, 512, 512), (288, 96, 512, 520), (96, 255, 512, 512), (96, 255, 512, 520), (96, 96, 512, 512), (96, 96, 512, 520)]:
    B = 6
    N = H * W
    x = torch.randn(B, K, H, W, device=dev); y = torch.empty(B, M, H, W, device=dev)
    res = M == 96
    r = torch.randn(B, M, H, W, device=dev) if res else None
    w = _hip.pack_gemm_weight(torch.randn(M, K) * 0.1).to(dev)
    for _ in range(2): ops.gemm1x1(w, x, y, M, K, res=r)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm1x1(w, x, y, M, K, res=r); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[2] * 1e-3
    print(f"M{M} K{K} {H}x{W}: {t*1e6:8.1f} us {2.0*B*M*K*N/t/1e12:6.1f} TF {4.0*B*N*(K+M+(M if res else 0))/t/1e9:6.0f} GB/s", flush=True)
