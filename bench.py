#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): images/s + PSNR, Restormer motion-deblur
on 1280x720 GoPro-shaped synthetic uint8 frames, one image per GPU per step.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = the whole tiled-patch hot path for ONE frame per rank (reference:
src/utils.py:353-454 + src/restormer/restormer.py): tile extraction (6 tiles of
512x512, overlap 96), the batched Restormer forward in libirm_hip.so, the
Gaussian-window blend + requantisation, and the squared error vs the target.
Frames are resident in HBM before the timed region; the uint8 result stays on
the device.  Images are independent units: rank r processes its own frames, no
data-path collective; PSNR rows are gathered once at the end (RCCL all_gather).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline`
(the dominant kernel: the f32-MFMA 1x1 GEMM, timed with HIP events on the launch
stream during the timed steps) and `cpu_baseline` (the CPU oracle on the host
cores over a bounded sample, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import irm_amd  # noqa: E402,F401
from irm_amd import ops, parallel, restormer, synth, utils  # noqa: E402
from irm_amd.configs import PATCH_CONFIG  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
PEAK_HBM_GBS = 8000.0             # HBM3E spec peak
H, W, C = 720, 1280, 3
N_FRAMES = 4                      # distinct synthetic frames per rank, cycled over the steps


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (profiling runs)")
    ap.add_argument("--no-kernel-timer", action="store_true", help="no per-launch events in the timed steps")
    ap.add_argument("--streams", type=int, default=1, help="tile groups run on this many HIP streams")
    ap.add_argument("--detail", default=None, help="write a per-shape kernel table (json) to this path")
    ap.add_argument("--cpu-tile", type=int, default=512, help="tile edge of the CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline(model, frame_u8, gpu_tile, tile_edge):
    """Oracle (PyTorch-CPU restatement of the reference forward, pinned to the reference by
    oracle/gen_golden.py) on a bounded sample: ONE tile of the first frame, all host threads."""
    from oracle import restormer_ref, tiler_ref
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    x = tiler_ref.to_unit_range(frame_u8)[:tile_edge, :tile_edge]
    t = torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))[None]
    threads = torch.get_num_threads()
    t0 = time.time()
    with torch.no_grad():
        y = restormer_ref.restormer_forward(t, sd)
    dt = time.time() - t0
    tiles_per_image = 6 * (512.0 / tile_edge) ** 2
    out = dict(value=1.0 / (dt * tiles_per_image), unit="images/s", cores=threads, kind="port",
               sample=f"1 tile {tile_edge}x{tile_edge} of frame 0 through the oracle Restormer forward "
                      f"({dt:.1f} s on {threads} torch threads, os.cpu_count={os.cpu_count()}); "
                      f"image rate = 1/({tiles_per_image:g} x tile time), tiler cost excluded")
    if gpu_tile is not None and tile_edge == 512:
        out["max_abs_vs_gpu_tile0"] = float((y[0] - gpu_tile.cpu()).abs().max())
    return out


#: timer group -> (kernel label, roofline that bounds it, regex of its instantiations in the PMC table)
ROOFLINE_KERNELS = {
    "gemm1x1_f16x3": ("gemm_xres_kernel / gemm_ring_kernel<F16> (irm_gemm1x1_f16x3_f32: LayerNorm + 1x1 conv, fp32 emulated "
                      "by three fp16 MFMAs, fp32 accumulate)", "hbm", r"^(gemm_ring_kernel<.*, true>|gemm_xres_kernel)"),
    "gemm1x1": ("gemm_ring_kernel (irm_gemm1x1_f32, exact f32 MFMA)", "mfma", r"^gemm_(ring_kernel<.*, false>|pw_kernel.*)$"),
    "dwgemm": ("dwgemm_kernel (irm_dwgemm_f32)", "mfma", r"^dwgemm_kernel"),
    "conv3x3": ("conv3x3_ring_kernel (irm_conv3x3_f32)", "mfma", r"^conv3x3_"),
    "dwconv3x3": ("dwconv3x3_kernel (irm_dwconv3x3_f32)", "hbm", r"^dwconv3x3_kernel<false"),
    "mdta_gram": ("mdta_gram_ring_kernel (irm_mdta_gram_f32)", "hbm", r"^mdta_gram"),
}


def pmc_traffic(pattern):
    """HBM bytes per launch of the kernels whose name matches pattern, from the committed PMC passes of
    this same command (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950; tools/pmc_traffic.py).  None when the file is absent: the
    counters cannot be read from inside an unprofiled run."""
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "g_final_pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except OSError:
        return None
    rows = [v for k, v in d.items() if re.search(pattern, k) and v.get("hbm_bytes_per_launch") is not None]
    n = sum(v["launches"] for v in rows)
    return sum(v["launches"] * v["hbm_bytes_per_launch"] for v in rows) / n if n else None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:         # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints its version banner on stdout when the communicator is created: keep stdout for the
        # ONE JSON line by pointing fd 1 at stderr while the communicator comes up
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    model = restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
    model.num_streams = args.streams
    cfg = PATCH_CONFIG["Restormer"][1]                      # deblurring: 512 / 96
    frames, targets, host_frames = [], [], []
    for i in range(N_FRAMES):
        inp, tgt = synth.synth_image_pair(rank * 1000 + i, H, W, C, seed_base=1000, blur=15)
        host_frames.append((inp, tgt))
        frames.append(torch.from_numpy(inp).to(dev))
        targets.append(torch.from_numpy(tgt).to(dev))

    def step(i, keep=None):
        return utils.tiled_forward_device(model, frames[i % N_FRAMES], cfg["patch_size"], cfg["patch_overlap"],
                                          pad8=True, target_dev=targets[i % N_FRAMES],
                                          max_batch=model.max_tiles_per_batch, keep_tiles=keep)

    keep = []
    for i in range(args.warmup):
        step(i, keep if i == 0 else None)
    torch.cuda.synchronize()

    timer = None if args.no_kernel_timer else ops.KernelTimer(detail=args.detail is not None)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ops.TIMER = timer
    t0 = time.perf_counter()
    results = [step(i) for i in range(args.steps)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None

    # max over ranks, PSNR rows gathered once (tens of bytes per image)
    rows = [(rank * args.steps + i, float(10 * np.log10(255.0 ** 2 / max(float(s.item()) / (H * W * C), 1e-12))))
            for i, (_, s) in enumerate(results)]
    elapsed, table = parallel.gather_results(elapsed, rows, dev)
    psnr = table[:, 1].numpy()

    if rank == 0:
        out = {
            "metric": "images/sec + PSNR, Restormer motion-deblur 1280x720",
            "value": world * args.steps / elapsed, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if os.environ.get("IRM_GEMM_EXACT") else
                     "f32 (1x1 convs: fp32 emulated by 3 fp16 MFMAs on hi/lo operand splits with fp32 accumulation, "
                     "error vs float64 <= the exact-f32 kernel's; everything else exact f32; IRM_GEMM_EXACT=1: "
                     "f32 MFMA everywhere)",
            "data": "synthetic",
            "config": {"workload": "Restormer motion-deblur (WithBias LN, 26.13M params, synthetic weights seed 42) on "
                                   "1280x720x3 uint8 GoPro-shaped synthetic frames; 6 tiles 512x512 (overlap 96) per "
                                   "frame, batched; one frame per GPU per step",
                       "global_batch": world, "tile": cfg["patch_size"], "overlap": cfg["patch_overlap"],
                       "parallelism": f"per-image shard x{world}, no data-path collective"},
            "psnr_db_mean": float(psnr.mean()), "psnr_db_std": float(psnr.std()),
        }
        if timer is not None:
            ks = timer.summary()
            if args.detail:
                rows = {k: {"launches": v["launches"], "us_per_launch": v["ms"] * 1e3 / v["launches"],
                            "ms_per_step": v["ms"] / args.steps, "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                            "gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9} for k, v in ks.items()}
                with open(args.detail, "w") as f:
                    json.dump(dict(sorted(rows.items(), key=lambda kv: -kv[1]["ms_per_step"])), f, indent=1)
                agg = {}
                for k, v in ks.items():
                    d = agg.setdefault(k.split(" ")[0], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
                    for f_ in d:
                        d[f_] += v[f_]
                ks = agg
            tot_ms = sum(k["ms"] for k in ks.values())
            # roofline of the dominant kernel (largest share of the kernel time in the timed steps)
            dom = max(ks, key=lambda k: ks[k]["ms"])
            g = ks[dom]
            label, bound, pmc_re = ROOFLINE_KERNELS.get(dom, (dom, "hbm", None))
            if bound == "mfma":
                ach, peak, unit = g["flops"] / (g["ms"] * 1e-3) / 1e12, PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
            else:
                ach, peak, unit = g["bytes"] / (g["ms"] * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
            out["roofline"] = {"kernel": label, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                               "frac": ach / peak, "traffic": pmc_traffic(pmc_re) if pmc_re else None,
                               "traffic_unit": "bytes per launch (HBM, PMC)",
                               "algorithmic_bytes_per_launch": g["bytes"] / g["launches"],
                               "fp32_equivalent_tflops": g["flops"] / (g["ms"] * 1e-3) / 1e12,
                               "launches": g["launches"],
                               "avg_launch_us": g["ms"] * 1e3 / g["launches"],
                               "share_of_kernel_time": g["ms"] / tot_ms}
            out["kernels"] = {
                k: {"launches": v["launches"], "ms_per_step": v["ms"] / args.steps,
                    "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12, "gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                    "hbm_frac": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS}
                for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms"])}
            # whole-step bound: max(F/peakF, B/peakB) / t  
            F = sum(v["flops"] for v in ks.values()) / args.steps
            Bt = sum(v["bytes"] for v in ks.values()) / args.steps
            bound_s = max(F / (PEAK_F32_MFMA_TFLOPS * 1e12), Bt / (PEAK_HBM_GBS * 1e9))
            out["step_model"] = {"gflop": F / 1e9, "gbytes": Bt / 1e9, "bound_ms": bound_s * 1e3,
                                 "frac_of_bound": bound_s / (elapsed / args.steps)}
        if world == 1 and not args.no_cpu_baseline:
            gpu_tile = keep[0][0] if keep else None
            out["cpu_baseline"] = cpu_baseline(model, host_frames[0][0], gpu_tile, args.cpu_tile)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
