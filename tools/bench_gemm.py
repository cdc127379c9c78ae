#!/usr/bin/env python3
"""Isolated timing of irm_gemm1x1_f32 on the Restormer shapes (HIP events, median of reps).
Usage: python tools/bench_gemm.py [reps]  (env IRM_GEMM_GENERIC=1 selects the non-ring kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irm_amd  # noqa
from irm_amd import _hip, ops

SHAPES = [  # M, K, N, B, ln, res
    (510, 96, 262144, 6, 1, 0), (288, 96, 262144, 6, 1, 0), (96, 255, 262144, 6, 0, 1), (96, 96, 262144, 6, 0, 1),
    (254, 48, 262144, 6, 1, 0), (1020, 192, 16384, 6, 1, 0), (2042, 384, 4096, 6, 1, 0), (384, 1021, 4096, 6, 0, 1),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    only = int(sys.argv[2]) if len(sys.argv) > 2 else None
    dev = torch.device("cuda:0")
    for idx, (M, K, N, B, ln, res) in enumerate(SHAPES):
        if only is not None and idx != only:
            continue
        H = 512 if N >= 262144 else int(N ** 0.5)
        W = N // H
        x = torch.randn(B, K, H, W, device=dev)
        y = torch.empty(B, M, H, W, device=dev)
        r = torch.randn(B, M, H, W, device=dev) if res else None
        w = _hip.pack_gemm_weight(torch.randn(M, K) * 0.1).to(dev)
        stats = torch.empty(B, 2, N, device=dev)
        ops.ln_stats(x, stats)
        lnw, lnb = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        kw = dict(res=r, stats=stats if ln else None, lnw=lnw if ln else None, lnb=lnb if ln else None, ln_mode=ln)
        for _ in range(2):
            ops.gemm1x1(w, x, y, M, K, **kw)
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm1x1(w, x, y, M, K, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2] * 1e-3
        fl = 2.0 * B * M * K * N
        by = 4.0 * B * N * (K + M + (M if res else 0))
        print(f"M{M:5d} K{K:5d} N{N:7d} B{B} ln{ln} res{res}: {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TF  {by/t/1e9:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
