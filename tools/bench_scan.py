"""Selective-scan micro-benchmark: the five MaIRUNet shapes of a 256x256 image (C5) over chunk lengths.
usage: python tools/bench_scan.py  (on the GPU box)"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from irm_amd import ops
from oracle import mair_ref

dev = torch.device("cuda:0")
SHAPES = [(65536, 192, 8, 6), (16384, 192, 8, 6), (4096, 384, 16, 12), (1024, 768, 32, 24), (65536, 96, 4, 3)]
for (L, D, N, R) in SHAPES:
    H = W = int(math.isqrt(L))
    J = R + 2 * N
    ids, _ = mair_ref.scan_ids(H, W, 4)
    g = torch.Generator().manual_seed(1)
    xT = torch.rand(1, L, D, generator=g).to(dev)
    pT = (torch.rand(1, L, 4 * J, generator=g) - 0.5).to(dev)
    dtw = ((torch.rand(4, D, R, generator=g) - 0.5)).to(dev)
    dtb = (torch.rand(4, D, generator=g) * 2 - 4).to(dev)
    A = (-torch.exp(torch.rand(4 * D, N, generator=g) * 1.5)).to(dev)
    Ds = torch.rand(4 * D, generator=g).to(dev)
    idd = ids.int().to(dev)
    yT = torch.empty(1, 4, L, D, device=dev)
    DB = -(-D // 64)
    plan = ops.scan_plan(1, L, D)[0]
    line = []
    for chunk in sorted({32, 48, 64, 96, 128, 192, 256, 384, 512, plan}):
        if chunk > L:
            continue
        nchunk = -(-L // chunk)
        state = torch.empty(2 * 4 * DB * nchunk * N * 64, device=dev)
        sdt = torch.empty(4 * DB * nchunk * 64, device=dev)
        ysum = torch.empty(4 * DB * nchunk * 64, device=dev)
        def run():
            ops.selective_scan(xT, pT, idd, dtw, dtb, A, Ds, yT, state, sdt, ysum, 1, L, D, N, R, chunk)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        line.append(f"{chunk}{'*' if chunk == plan else ''}:{e0.elapsed_time(e1) / 20 * 1e3:.0f}")
    print(f"L{L} D{D} N{N}: " + "  ".join(line), flush=True)
