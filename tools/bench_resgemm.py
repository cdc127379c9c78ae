"""Residual 1x1 convs of the C >= 192 levels (project_out, attention apply) on the emulated streaming kernel: sweep of
the launch plan (ct output tiles per pass, ygroups) against the planner's choice."""
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
for (M, K, H, W) in [(192, 510, 128, 128), (384, 1021, 64, 64), (192, 192, 128, 128), (384, 384, 64, 64)]:
    x = torch.randn(B, K, H, W, device=dev)
    y = torch.randn(B, M, H, W, device=dev)
    ws = _hip.pack_gemm_weight_split(torch.randn(M, K) * 0.05).to(dev)
    bias = torch.zeros(M, device=dev)
    t0 = timeit(lambda: ops.gemm1x1(ws, x, y, M, K, res=y, bias=bias, split=True))
    print(f"M{M} K{K} {H}x{W}: planner {t0:6.1f} us ({4.0 * B * H * W * (K + 2 * M) / t0 / 1e3:5.0f} GB/s)", flush=True)
    mt = (M + 15) // 16
    for ct in (3, 4, 6, 8):
        chunks = -(-mt // ct)
        for yg in range(1, chunks + 1):
            if (yg - 1) * -(-chunks // yg) >= chunks:
                continue
            try:
                t = timeit(lambda: ops.gemm1x1(ws, x, y, M, K, res=y, bias=bias, split=True, ct=ct, ygroups=yg), 10)
            except Exception as e:
                print(f"    ct {ct} yg {yg}: {type(e).__name__}")
                continue
            print(f"    ct {ct} yg {yg}: {t:6.1f} us", flush=True)
