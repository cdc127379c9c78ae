#!/usr/bin/env python3
"""Measurement of the non-headline workloads (BASELINE.json configs[1], [2], [4] and the other model rows of
SURVEY section 8) to the same bar as bench.py: one JSON line per workload with throughput, the roofline of its
dominant kernel (HIP events on the launch stream inside the timed steps) and a CPU baseline = the oracle on a
bounded sample, whose output is also compared with the GPU result on the same sample.

usage: python tools/bench_configs.py [c2 c3 c5 rednet deblurgan mair_cdn] [--steps K] [--out file.jsonl]
A step = one uint8 image resident in HBM -> tiles -> model -> blended uint8 image on the device."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irm_amd  # noqa: E402,F401
from irm_amd import deblurganv2, dncnn, mair, ops, rednet, restormer, synth, utils  # noqa: E402
from irm_amd.configs import PATCH_CONFIG  # noqa: E402

PEAK_TF, PEAK_GBS, PEAK_F16_TF = 157.3, 8000.0, 2500.0
EMULATED_KERNELS = ("conv3x3_f16x3", "attn_gdfn_fused", "gdfn_fused", "qkv_dw_fused", "gemm1x1_f16x3", "dwgemm_f16x3", "gemm_ps_f16x3", "gemm_ps_res_f16x3")
MFMA_KERNELS = {"gemm1x1", "conv3x3", "dwgemm", "mdta_gram"}


def sd_cpu(model):
    return {k: v.detach().cpu() for k, v in model.state_dict().items()}


def workloads(dev):
    from oracle import convnets_ref, deblurgan_ref, mair_ref, restormer_ref
    w = {}

    def c2():
        m = dncnn.DnCNN(1, 1, 64, 20, "R").load_synthetic(42).eval().to(dev)
        img, _ = synth.synth_image_pair(0, 256, 256, 1, seed_base=3000, blur=0)
        # BASELINE configs[1] is a BATCH of 256x256 images: 8 per step through utils.tiled_forward_device_batch
        return dict(name="DnCNN-blind gray sigma 25, batch of 8 images 256x256 (BASELINE configs[1])", model=m, img=img, batch=8,
                    tiler=dict(ps=256, ov=48, pad8=False, sigma=25), sample=(256, 256),
                    oracle=lambda t, sd: convnets_ref.dncnn_forward(t, sd))
    w["c2"] = c2

    def c3():
        m = restormer.Restormer(LayerNorm_type="BiasFree").load_synthetic(42).eval().to(dev)
        img, _ = synth.synth_image_pair(0, 512, 512, 3, seed_base=2000, blur=0)
        c = PATCH_CONFIG["Restormer"][0]
        return dict(name="Restormer colour blind-denoise sigma 25, 512x512 = 9 tiles of 256x256 (BASELINE configs[2])",
                    model=m, img=img, tiler=dict(ps=c["patch_size"], ov=c["patch_overlap"], pad8=True, sigma=25),
                    sample=(128, 128), oracle=lambda t, sd: restormer_ref.restormer_forward(t, sd))
    w["c3"] = c3

    def c5():
        m = mair.MaIRUNet(dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4, ssm_ratio=2.0, flp_ratio=4.0,
                          mlp_ratio=1.5, scan_len=4).load_synthetic(42).eval().to(dev)
        img, _ = synth.synth_image_pair(0, 256, 256, 3, seed_base=4000, blur=0)
        c = PATCH_CONFIG["MaIR"][1]
        return dict(name="MaIRUNet real-denoise, SIDD-shaped 256x256 crop (BASELINE configs[4])", model=m, img=img,
                    tiler=dict(ps=c["patch_size"], ov=c["patch_overlap"], pad8=True, sigma=None), sample=(64, 64),
                    oracle=lambda t, sd: mair_ref.mairunet_forward(t, sd, scan_len=4))
    w["c5"] = c5

    def red():
        m = rednet.REDNet().load_synthetic(42).eval().to(dev)
        img, _ = synth.synth_image_pair(0, 256, 256, 1, seed_base=3000, blur=0)
        c = PATCH_CONFIG["REDNet"]
        return dict(name="REDNet gray sigma 25, 256x256 = 9 tiles of 128x128", model=m, img=img,
                    tiler=dict(ps=c["patch_size"], ov=c["patch_overlap"], pad8=False, sigma=25), sample=(128, 128),
                    oracle=lambda t, sd: convnets_ref.rednet_forward(t, sd))
    w["rednet"] = red

    def dg():
        m = deblurganv2.FPNMobileNet().load_synthetic(42).to(dev)
        img, _ = synth.synth_image_pair(0, 720, 1280, 3, seed_base=1000, blur=15)
        c = PATCH_CONFIG["DeblurGANv2"][1]
        return dict(name="DeblurGANv2 FPN-MobileNet motion deblur, 1280x720 (one 2048-patch = whole frame)", model=m,
                    img=img, tiler=dict(ps=c["patch_size"], ov=c["patch_overlap"], pad8=False, sigma=None,
                                        hooks="deblurganv2"), sample=(256, 256), sample_norm="deblurgan",
                    oracle=lambda t, sd: deblurgan_ref.fpn_mobilenet_forward(t, sd))
    w["deblurgan"] = dg

    def cdn():
        import yaml
        with open(os.path.join(os.path.dirname(mair.__file__), "options", "test_MaIR_CDN_s25.yml")) as f:
            net = dict(yaml.safe_load(f)["network_g"])
        net.pop("type")
        m = mair.MaIR(**net).load_synthetic(42).eval().to(dev)
        img, _ = synth.synth_image_pair(0, 256, 256, 3, seed_base=4000, blur=0)
        c = PATCH_CONFIG["MaIR"][0]
        return dict(name="MaIR colour Gaussian denoise sigma 25 (flat, 6x6 blocks, C=180), 256x256 = 9 tiles of 128x128",
                    model=m, img=img, tiler=dict(ps=c["patch_size"], ov=c["patch_overlap"], pad8=True, sigma=25),
                    sample=(32, 32), oracle=lambda t, sd: mair_ref.mair_forward(t, sd, scan_len=4))
    w["mair_cdn"] = cdn
    return w


def run(key, spec, steps, warm, dev, cpu, detail=None):
    model, img, tk = spec["model"], spec["img"], spec["tiler"]
    K = int(spec.get("batch", 1))                       # images per step (one batched forward over all their tiles)
    imgs_dev = [torch.from_numpy(img).to(dev)]
    for k in range(1, K):                               # further images of the same generator
        imgs_dev.append(torch.from_numpy(synth.synth_image_pair(k, img.shape[0], img.shape[1], img.shape[2], seed_base=3000,
                                                                blur=0)[0]).to(dev))
    mb = max(getattr(model, "max_tiles_per_batch", 8), K)

    def step():
        return utils.tiled_forward_device_batch(model, imgs_dev, tk["ps"], tk["ov"], tk["pad8"], tk["sigma"], max_batch=mb,
                                                hooks=tk.get("hooks"))
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    # throughput: the product path as users run it (HIP-graph replay of the per-batch forward, utils.graphed_forward)
    nrep = max(steps, 20)
    rounds = []
    for _ in range(3):                      # median of three rounds (a 1 ms step is at the mercy of one host hiccup)
        t0 = time.perf_counter()
        for _ in range(nrep):
            step()
        torch.cuda.synchronize()
        rounds.append((time.perf_counter() - t0) / nrep)
    dt = sorted(rounds)[1]
    # roofline: a second pass with per-launch HIP events (eager launches; its own step time is not `value`)
    timer = ops.KernelTimer(detail=detail is not None)
    ops.TIMER = timer
    t1 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt_eager = (time.perf_counter() - t1) / steps
    ops.TIMER = None
    if detail:
        rows = {k: {"launches": v["launches"], "us_per_launch": v["ms"] * 1e3 / v["launches"], "ms_per_step": v["ms"] / steps,
                    "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12, "gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9}
                for k, v in timer.summary().items()}
        with open(detail, "w") as f:
            json.dump(dict(sorted(rows.items(), key=lambda kv: -kv[1]["ms_per_step"])), f, indent=1)
    agg = {}
    for k, v in timer.summary().items():
        d = agg.setdefault(k.split(" ")[0], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        for f in d:
            d[f] += v[f]
    tot = sum(v["ms"] for v in agg.values())
    dom, dv = max(agg.items(), key=lambda kv: kv[1]["ms"])
    if dom in EMULATED_KERNELS:
        # fp32 emulated on the fp16 matrix cores: three MFMA passes per product against the dense fp16 peak (as bench.py)
        ach = 3.0 * dv["flops"] / (dv["ms"] * 1e-3) / 1e12
        roof = dict(kernel=dom, bound="mfma", achieved=ach, peak=PEAK_F16_TF, unit="TFLOP/s", frac=ach / PEAK_F16_TF,
                    fp32_equivalent_tflops=ach / 3.0)
    elif dom in MFMA_KERNELS:
        ach = dv["flops"] / (dv["ms"] * 1e-3) / 1e12
        roof = dict(kernel=dom, bound="mfma", achieved=ach, peak=PEAK_TF, unit="TFLOP/s", frac=ach / PEAK_TF)
    else:
        ach = dv["bytes"] / (dv["ms"] * 1e-3) / 1e9
        roof = dict(kernel=dom, bound="hbm", achieved=ach, peak=PEAK_GBS, unit="GB/s", frac=ach / PEAK_GBS)
    roof.update(traffic=None, launches=dv["launches"], avg_launch_us=dv["ms"] * 1e3 / dv["launches"],
                share_of_kernel_time=dv["ms"] / tot)
    out = {"metric": "images/sec", "workload_id": key, "value": K / dt, "unit": "images/s", "images_per_step": K, "n_gpus": 1,
           "steps": nrep, "warmup": warm, "ms_per_step": dt * 1e3, "ms_per_step_eager_with_events": dt_eager * 1e3, "higher_is_better": True, "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "config": {"workload": spec["name"], "image": list(img.shape),
                                                           "tile": tk["ps"], "overlap": tk["ov"]},
           "roofline": roof,
           "kernels": {k: {"launches": v["launches"] // steps, "ms_per_step": v["ms"] / steps,
                           "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12, "gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9}
                       for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}}
    if cpu:
        # bounded CPU sample: the oracle forward on a crop of the image (float input as the tiler would produce),
        # scaled to the image by pixel count; the same crop goes through the GPU model for the parity figure
        sh, sw = spec["sample"]
        crop = img[:sh, :sw]
        if spec.get("sample_norm") == "deblurgan":
            x = deblurganv2.normalize(crop)
        else:
            x = crop.astype(np.float32) / 255.0
        t = torch.from_numpy(np.ascontiguousarray(x.reshape(sh, sw, -1).transpose(2, 0, 1)))[None].float()
        sd = sd_cpu(model)
        threads = torch.get_num_threads()
        t0 = time.time()
        with torch.no_grad():
            y = spec["oracle"](t, sd)
        ct = time.time() - t0
        yg = model(t.to(dev)).cpu()
        scale = (img.shape[0] * img.shape[1]) / float(sh * sw)
        out["cpu_baseline"] = dict(value=1.0 / (ct * scale), unit="images/s", cores=threads, kind="port",
                                   sample=f"oracle forward on a {sh}x{sw} crop ({ct:.1f} s on {threads} torch threads), "
                                          f"scaled by pixel count x{scale:g}; tiler overlap excluded",
                                   max_abs_vs_gpu=float((y - yg[:, :y.shape[1]]).abs().max()))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=[])
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--detail", default=None, help="per-shape kernel table (json) of the LAST workload run")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    table = workloads(dev)
    lines = []
    for key in (args.which or list(table)):
        spec = table[key]()
        res = run(key, spec, args.steps, args.warmup, dev, not args.no_cpu_baseline, args.detail)
        line = json.dumps(res)
        print(line, flush=True)
        lines.append(line)
        del spec
        torch.cuda.empty_cache()
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
