#!/usr/bin/env python3
"""Isolated timing of irm_conv3x3_f32 (HIP events). Usage: python tools/bench_conv.py [reps] [index]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irm_amd  # noqa
from irm_amd import _hip, ops

SHAPES = [  # ci, co, H, W, B, store_mode
    (96, 192, 256, 256, 6, 2), (192, 384, 128, 128, 6, 2), (64, 64, 256, 256, 16, 0), (128, 128, 128, 128, 16, 0),
    (48, 24, 512, 512, 6, 1), (96, 3, 512, 512, 6, 0), (3, 48, 512, 512, 6, 0),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    only = int(sys.argv[2]) if len(sys.argv) > 2 else None
    dev = torch.device("cuda:0")
    for idx, (ci, co, H, W, B, st) in enumerate(SHAPES):
        if only is not None and idx != only:
            continue
        x = torch.randn(B, ci, H, W, device=dev)
        oc, oh, ow = (co, H, W) if st == 0 else (co * 4, H // 2, W // 2) if st == 1 else (co // 4, 2 * H, 2 * W)
        y = torch.empty(B, oc, oh, ow, device=dev)
        w = _hip.pack_conv3x3_weight(torch.randn(co, ci, 3, 3) * 0.05).to(dev)
        bias = torch.zeros(co, device=dev) if st == 0 and ci == co else None
        for _ in range(2):
            ops.conv3x3(w, x, y, ci, co, bias=bias, relu1=bias is not None, store_mode=st)
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv3x3(w, x, y, ci, co, bias=bias, relu1=bias is not None, store_mode=st)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2] * 1e-3
        fl = 18.0 * B * ci * co * H * W
        by = 4.0 * B * H * W * (ci + co)
        print(f"ci{ci:4d} co{co:4d} {H}x{W} B{B} st{st}: {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TF  {by/t/1e9:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
