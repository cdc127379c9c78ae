#!/usr/bin/env python3
"""Throughput of the non-headline BASELINE.json configurations (parity-test cases, not bench lines):
C2 DnCNN-blind gray 256x256, C3 Restormer colour blind-denoise 512x512 sigma 25, C5 MaIRUNet 256x256,
plus REDNet 128-tiles.  uint8 image resident in HBM -> uint8 result on the device, per image."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irm_amd  # noqa
from irm_amd import dncnn, mair, ops, rednet, restormer, synth, utils
from irm_amd.configs import PATCH_CONFIG


def run(name, model, img, ps, ov, pad8, sigma, steps=6, warm=2, detail=None):
    dev = torch.device("cuda:0")
    img_dev = torch.from_numpy(img).to(dev)
    for _ in range(warm):
        utils.tiled_forward_device(model, img_dev, ps, ov, pad8, sigma, max_batch=getattr(model, "max_tiles_per_batch", 8))
    torch.cuda.synchronize()
    timer = ops.KernelTimer(detail=bool(detail))
    ops.TIMER = timer
    t0 = time.perf_counter()
    for _ in range(steps):
        utils.tiled_forward_device(model, img_dev, ps, ov, pad8, sigma, max_batch=getattr(model, "max_tiles_per_batch", 8))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ops.TIMER = None
    ks = timer.summary()
    agg = {}
    for k, v in ks.items():
        d = agg.setdefault(k.split(" ")[0], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        d["ms"] += v["ms"] / steps; d["flops"] += v["flops"] / steps; d["bytes"] += v["bytes"] / steps
        d["launches"] += v["launches"] // steps
    print(f"{name}: {1.0/dt:8.2f} img/s  {dt*1e3:8.2f} ms/img")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        print(f"    {k:16s} {v['ms']:8.3f} ms x{v['launches']:4d}  {v['flops']/max(v['ms'],1e-9)/1e9:7.1f} TF  {v['bytes']/max(v['ms'],1e-9)/1e6:7.0f} GB/s")
    if detail:
        rows = {k: {"launches": v["launches"], "us_per_launch": v["ms"] * 1e3 / v["launches"], "ms_per_step": v["ms"] / steps,
                    "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12, "gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9} for k, v in ks.items()}
        json.dump(dict(sorted(rows.items(), key=lambda kv: -kv[1]["ms_per_step"])), open(detail, "w"), indent=1)


def main():
    which = sys.argv[1:] or ["c2", "c3", "c5", "rednet"]
    dev = torch.device("cuda:0")
    if "c2" in which:
        m = dncnn.DnCNN(1, 1, 64, 20, "R").load_synthetic(42).eval().to(dev)
        img, _ = synth.synth_image_pair(0, 256, 256, 1, seed_base=3000, blur=0)
        run("C2 DnCNN-blind gray 256x256 sigma25", m, img, 256, 48, False, 25)
    if "c3" in which:
        m = restormer.Restormer(LayerNorm_type="BiasFree").load_synthetic(42).eval().to(dev)
        img, _ = synth.synth_image_pair(0, 512, 512, 3, seed_base=2000, blur=0)
        c = PATCH_CONFIG["Restormer"][0]
        run("C3 Restormer colour blind-denoise 512x512 sigma25 (9 tiles 256^2)", m, img, c["patch_size"], c["patch_overlap"], True, 25)
    if "c5" in which:
        m = mair.MaIRUNet(dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4, ssm_ratio=2.0, flp_ratio=4.0,
                          mlp_ratio=1.5, scan_len=4).load_synthetic(42).eval().to(dev)
        img, _ = synth.synth_image_pair(0, 256, 256, 3, seed_base=4000, blur=0)
        c = PATCH_CONFIG["MaIR"][1]
        run("C5 MaIRUNet real-denoise 256x256 (1 tile)", m, img, c["patch_size"], c["patch_overlap"], True, None,
            detail="gpurun_out/detail_mair.json")
    if "rednet" in which:
        m = rednet.REDNet().load_synthetic(42).eval().to(dev)
        img, _ = synth.synth_image_pair(0, 256, 256, 1, seed_base=3000, blur=0)
        c = PATCH_CONFIG["REDNet"]
        run("REDNet gray 256x256 (9 tiles 128^2)", m, img, c["patch_size"], c["patch_overlap"], False, 25)


if __name__ == "__main__":
    main()
