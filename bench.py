#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): images/s + PSNR, Restormer motion-deblur
on 1280x720 GoPro-shaped synthetic uint8 frames.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch per rank: FOUR frames (--frames-per-step), each through tile
extraction (6 tiles of 512x512, overlap 96; reference: src/utils.py:353-454), their 24 tiles through ONE batched
Restormer forward in libirm_hip.so (src/restormer/restormer.py), each frame through the Gaussian-window blend +
requantisation and the squared error vs its target.  Frames are resident in HBM before the timed region; the
uint8 results stay on the device.  Images are independent units: rank r processes its own frames, no data-path
collective; PSNR rows and the ids of failed frames are gathered once at the end (RCCL all_gather).

`python bench.py --gpus N` without a launcher starts its own N ranks (a child `python -m torch.distributed.run`,
before anything touches a GPU) and exits with their status.

Rank 0 prints ONE JSON line (contract in the task statement): `value` = frames per second over all ranks,
`ms_per_step` (the four-frame step) and `ms_per_frame`; `roofline` (the kernel group with the largest share of the
kernel time, HIP events on the launch stream during the timed steps; PMC traffic from the committed passes of this
command); `step_model` per frame: the reference-decomposition bound, this build's OWN bound (`own_frac`) and
`hbm_util`; `psnr_cpu / psnr_gpu / abs_dpsnr` of frame 0 against the reference's own CPU run of that frame
(tests/golden fixture); `failed_image_ids`; `cpu_baseline` (the CPU oracle on the host cores over a bounded sample,
N=1 only) and, at N=1, two reference legs measured in child processes: `value_exact_f32` (every GEMM on the
f32-input MFMA, IRM_GEMM_EXACT=1) and `value_pcie_inclusive` (the reference-shaped get_model_prediction call:
numpy uint8 in host memory -> numpy uint8).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0             # HBM3E spec peak
# SURVEY 8(d) / BASELINE.md section 3: the reference's op-boundary decomposition of one 1280x720 frame (every
# op's inputs read once, outputs written once, fp32) and its FLOPs - the fixed yardstick of the whole step,
# independent of how many of those boundaries this build has fused away
REF_STEP_GBYTES, REF_STEP_TFLOP = 271.1, 7.54
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
    ap.add_argument("--frames-per-step", type=int, default=4,
                    help="frames whose tiles form one batch (one pass of the hot path = one step)")
    ap.add_argument("--detail", default=None, help="write a per-shape kernel table (json) to this path")
    ap.add_argument("--all-launch-events", action="store_true",
                    help="diagnostic: HIP events around EVERY launch of the timed steps (the behaviour before the end of round 3)")
    ap.add_argument("--cpu-tile", type=int, default=512, help="tile edge of the CPU-baseline sample")
    ap.add_argument("--no-legs", action="store_true", help="skip the exact-f32 / PCIe-inclusive child legs")
    ap.add_argument("--leg", choices=["pcie"], default=None, help=argparse.SUPPRESS)     # child process mode
    return ap.parse_args()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: run the N ranks as a child torch.distributed.run
    and return its exit code.  Nothing in this process has touched a GPU (no exec after GPU initialisation)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def child_leg(extra_args, env_extra, timeout=600):
    """One measurement in a child process (fresh library state / environment); returns its JSON line or None."""
    cmd = [sys.executable, os.path.abspath(__file__)] + extra_args
    try:
        r = subprocess.run(cmd, env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=timeout)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
    except (subprocess.SubprocessError, ValueError, OSError):
        pass
    return None


def pcie_leg(dev, steps, warmup):
    """The reference-shaped synchronous call (src/utils.py:353-454: numpy uint8 frame in host memory ->
    get_model_prediction -> numpy uint8 frame; one H2D, one D2H and a host synchronisation per frame)."""
    model = restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
    cfg = utils.get_patch_config("deblurring", "motion", "Restormer")
    frames = [synth.synth_image_pair(i, H, W, C, seed_base=1000, blur=15)[0] for i in range(N_FRAMES)]
    for i in range(warmup):
        utils.get_model_prediction(model, frames[i % N_FRAMES], dev, **cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        utils.get_model_prediction(model, frames[i % N_FRAMES], dev, **cfg)
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"value": 1.0 / dt, "ms_per_step": dt * 1e3, "steps": steps}))


def cpu_baseline(model, frame_u8, gpu_tile, tile_edge):
    """Oracle (PyTorch-CPU restatement of the reference forward, pinned to the reference by
    oracle/gen_golden.py) on a bounded sample: ONE tile of the first frame, all host threads."""
    from oracle import restormer_ref, tiler_ref
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    x = tiler_ref.to_unit_range(frame_u8)[:tile_edge, :tile_edge]
    t = torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))[None]
    # the GPU box gives a 1-GPU job a share of the host (16 CPUs), not the 256 it reports: 128 torch threads
    # oversubscribe that share and under-state the CPU
    threads = max(1, min(int(os.environ.get("IRM_CPU_THREADS", "16")), os.cpu_count() or 1))
    torch.set_num_threads(threads)
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


def psnr_parity(frame0, target_u8):
    """PSNR parity of frame 0 against the REFERENCE's CPU path (north_star: within 0.01 dB): psnr_cpu is the PSNR of
    the reference's own run_model_inference output for this frame (src/utils.py:353-454 with the reference Restormer
    on the CPU, 134 s on 8 cores, generated once by oracle/gen_golden.py --only fullsize_frame and committed as
    tests/golden/restormer_fullsize_frame.npz - data, not code); psnr_gpu is this run's frame."""
    path = os.path.join(ROOT, "tests", "golden", "restormer_fullsize_frame.npz")
    if frame0 is None or not os.path.exists(path):
        return {"psnr_cpu": None, "psnr_gpu": None, "abs_dpsnr": None}
    g = np.load(path)
    out_u8 = frame0[0].cpu().numpy()
    mse = float(np.mean((out_u8.astype(np.float64) - target_u8.astype(np.float64)) ** 2))
    psnr_gpu = float(10 * np.log10(255.0 ** 2 / max(mse, 1e-12)))
    diff = np.abs(out_u8.astype(np.int32) - g["pred_u8"].astype(np.int32))
    return {"psnr_cpu": float(g["psnr"]), "psnr_gpu": psnr_gpu, "abs_dpsnr": abs(psnr_gpu - float(g["psnr"])),
            "frame0_u8_bytes_differing_from_cpu_reference": int((diff > 0).sum()), "frame0_u8_max_diff": int(diff.max()),
            "psnr_parity_source": "frame 0 (rank 0); CPU side = the reference's run_model_inference + reference Restormer "
                                  "(tests/golden/restormer_fullsize_frame.npz, oracle/gen_golden.py --only fullsize_frame)"}


def pmc_step_totals():
    """(HBM bytes per frame, frames) over the irm kernels of the committed PMC passes, or (None, None)."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
    except OSError:
        return None, None
    frames = d.get("_frames") or sum(v["launches"] for k, v in d.items() if k.startswith("blend_kernel"))
    if not frames:
        return None, None
    # (runtime copy / fill kernels are the set-up - weight uploads, buffer clears - not the frames)
    tot = sum(v["launches"] * v["hbm_bytes_per_launch"] for k, v in d.items()
              if not k.startswith(("_", "__amd_rocclr")) and v.get("hbm_bytes_per_launch") is not None)
    return tot / frames, frames


#: timer group -> (kernel label, roofline that bounds it, regex of its instantiations in the PMC table)
ROOFLINE_KERNELS = {
    "attn_gdfn_fused": ("lnpw_dw_fused_kernel<GATE, APPLY> (irm_attn_gdfn_fused_f16x3_f32: x' = x + project_out(attn @ v) by the folded "
                        "per-image matrix, then LayerNorm + project_in + depth-wise 3x3 + GELU gate + project_out + residual, in one "
                        "kernel; 1x1 convs = fp32 emulated by three fp16 MFMAs)", "mfma",
                        r"^lnpw_dw_fused_kernel<\d+, \d+, true, true"),
    "gdfn_fused": ("lnpw_dw_fused_kernel<GATE> (irm_gdfn_fused_f16x3_f32: LayerNorm + project_in + depth-wise 3x3 + GELU gate + "
                   "project_out + residual in one kernel; 1x1 convs = fp32 emulated by three fp16 MFMAs)", "mfma",
                   r"^lnpw_dw_fused_kernel<\d+, \d+, true, false"),
    "qkv_dw_fused": ("lnpw_dw_fused_kernel<!GATE> (irm_qkv_dw_fused[_tm]_f16x3_f32: LayerNorm + qkv 1x1 + depth-wise 3x3)", "hbm",
                     r"^lnpw_dw_fused_kernel<\d+, \d+, false"),
    "gdfn_tail": ("gdfn_tail_kernel (irm_gdfn_tail_f16x3_f32: depth-wise 3x3 + GELU gate + project_out + residual of the C = 192 level in "
                  "one kernel on a tile-major channel-last h)", "hbm", r"^gdfn_tail_kernel"),
    "gemm1x1_f16x3": ("gemm_xres_kernel / gemm_ring_kernel<F16> (irm_gemm1x1_f16x3_f32: LayerNorm + 1x1 conv, fp32 emulated "
                      "by three fp16 MFMAs, fp32 accumulate)", "hbm", r"^(gemm_ring_kernel<.*, true>|gemm_xres_kernel)"),
    "gemm_ps_f16x3": ("gemm_ps_kernel (irm_gemm_presplit_f16x3_f32: 1x1 conv on pre-split fp16 hi/lo fragments of LayerNorm(x), "
                      "C >= 192 levels; three fp16 MFMAs per product, fp32 accumulate)", "hbm", r"^gemm_ps_kernel"),
    "ln_split": ("ln_split_kernel (irm_ln_split_f16: LayerNorm + fp16 hi/lo split in MFMA fragment order)", "hbm", r"^ln_split_kernel"),
    "conv3x3_thin": ("conv3x3_thin_{in,out}_kernel (irm_conv3x3_thin_f32: 3x3 convs with <= 4 channels on one side, exact fp32 "
                     "on the vector pipe)", "hbm", r"^conv3x3_thin"),
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
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
    except OSError:
        return None
    rows = [v for k, v in d.items() if not k.startswith("_") and re.search(pattern, k)
            and v.get("hbm_bytes_per_launch") is not None]
    n = sum(v["launches"] for v in rows)
    return sum(v["launches"] * v["hbm_bytes_per_launch"] for v in rows) / n if n else None


PMC_FILE = next((p for p in (os.path.join(ROOT, "profiles", r, "final_pmc_traffic.json") for r in ("r03", "r02"))
                 if os.path.exists(p)), os.path.join(ROOT, "profiles", "r03", "final_pmc_traffic.json"))


def pmc_provenance():
    """Where the PMC traffic figure comes from: the committed file and the git revision it was collected at
    (written into the file by tools/pmc_traffic.py as "_git_sha")."""
    try:
        with open(PMC_FILE) as f:
            sha = json.load(f).get("_git_sha")
    except OSError:
        return None
    return {"file": os.path.relpath(PMC_FILE, ROOT), "collected_at_git_sha": sha}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))              # no GPU call has happened in this process
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.leg == "pcie":
        return pcie_leg(dev, args.steps, args.warmup)
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
    model.num_streams = args.streams                        # (> 1: experimental, needs IRM_EXPERIMENTAL_STREAMS=1)
    cfg = PATCH_CONFIG["Restormer"][1]                      # deblurring: 512 / 96
    frames, targets, host_frames = [], [], []
    for i in range(N_FRAMES):
        inp, tgt = synth.synth_image_pair(rank * 1000 + i, H, W, C, seed_base=1000, blur=15)
        host_frames.append((inp, tgt))
        frames.append(torch.from_numpy(inp).to(dev))
        targets.append(torch.from_numpy(tgt).to(dev))

    FPS = max(1, args.frames_per_step)
    model.max_tiles_per_batch = max(model.max_tiles_per_batch, 6 * FPS)     # all tiles of a step in ONE batched forward

    def step(i, keep=None):
        """One pass of the hot path over one batch: the tiles of FPS frames (6 each) through ONE batched forward,
        every frame extracted / blended / requantised / scored on its own; returns FPS (uint8 frame, SSE) pairs."""
        ids = [(i * FPS + k) % N_FRAMES for k in range(FPS)]
        return utils.tiled_forward_device_batch(model, [frames[j] for j in ids], cfg["patch_size"], cfg["patch_overlap"],
                                                pad8=True, targets_dev=[targets[j] for j in ids],
                                                max_batch=model.max_tiles_per_batch, keep_tiles=keep)

    keep = []
    frame0 = None                                   # (uint8 frame, SSE) of frame 0: the PSNR-parity record below
    for i in range(args.warmup):
        r = step(i, keep if i == 0 else None)
        if i == 0:
            frame0 = r[0]
    torch.cuda.synchronize()

    # Per-launch events on EVERY launch (~600 per step) belong to an un-timed profile step: on some hosts of the pool they slow the
    # step itself by up to 60 % (same box, alternating processes - 71 ... 75 ms per frame with them, 46.3 ... 46.5
    # without, host enqueue time 125 instead of 9 ms per step).  The timed steps carry events around the launches of the ROOFLINE
    # kernel only (the group with the largest share in the profile step: 24 launches per step).
    ks_all, timer = None, None
    if not args.no_kernel_timer:
        prof = ops.KernelTimer(detail=args.detail is not None)
        ops.TIMER = prof
        step(0)
        ks_all = prof.summary()
        ops.TIMER = None
        agg = {}
        for k, v in ks_all.items():
            agg[k.split(" ")[0]] = agg.get(k.split(" ")[0], 0.0) + v["ms"]
        timer = ops.KernelTimer(only=None if args.all_launch_events else {max(agg, key=agg.get)})
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ops.TIMER = timer
    # no cyclic-GC pass inside the timed region (a full collection over the timer's records stops the launching thread for
    # tens of milliseconds); `host_enqueue_ms_per_step` (the host's own time to enqueue a step, ~8 ms) in the JSON line shows
    # whether the host, not the GPU, paced a run (the outlier runs of round 3: 70 ... 125 ms with events on every launch).
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    results, failed = [], []
    for i in range(args.steps):
        try:                                        # a frame that raises is reported, not fatal
            results.append((i, step(i)))
        except Exception as e:                      # noqa: BLE001
            failed.extend((rank * args.steps + i) * FPS + k for k in range(FPS))
            print(f"[bench] rank {rank}: step {i} failed: {type(e).__name__}: {e}", file=sys.stderr)
    t_enq = time.perf_counter() - t0               # host time to enqueue every step (close to `elapsed` <=> the host, not the GPU, paced the run)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    ops.TIMER = None

    # max over ranks, PSNR rows gathered once (tens of bytes per image)
    rows = [((rank * args.steps + i) * FPS + k, float(10 * np.log10(255.0 ** 2 / max(float(s.item()) / (H * W * C), 1e-12))))
            for i, pairs in results for k, (_, s) in enumerate(pairs)]
    elapsed, table, failed_all = parallel.gather_results(elapsed, rows, dev, failed_ids=failed)
    psnr = table[:, 1].numpy() if table.shape[0] else np.array([float("nan")])
    n_done = int(table.shape[0])
    if frame0 is None and rank == 0:                # --warmup 0: frame 0 for the parity record, outside the timed region
        frame0 = step(0, keep)[0]

    if rank == 0:
        out = {
            "metric": "images/sec + PSNR, Restormer motion-deblur 1280x720",
            "value": n_done / elapsed, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "host_enqueue_ms_per_step": t_enq / args.steps * 1e3, "frames_per_step": FPS, "ms_per_frame": elapsed / max(n_done, 1) * world * 1e3,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if os.environ.get("IRM_GEMM_EXACT") else
                     "f32 (1x1 convs: fp32 emulated by 3 fp16 MFMAs on hi/lo operand splits with fp32 accumulation, "
                     "error vs float64 at the level of an fp32 chain; everything else exact f32; value_exact_f32: "
                     "f32 MFMA everywhere)",
            "data": "synthetic",
            "config": {"workload": "Restormer motion-deblur (WithBias LN, 26.13M params, synthetic weights seed 42) on "
                                   "1280x720x3 uint8 GoPro-shaped synthetic frames; 6 tiles 512x512 (overlap 96) per "
                                   f"frame; a step = {FPS} frame(s) per GPU, their {6 * FPS} tiles in one batched forward",
                       "global_batch": world * FPS, "tile": cfg["patch_size"], "overlap": cfg["patch_overlap"],
                       "parallelism": f"per-image shard x{world}, no data-path collective"},
            "psnr_db_mean": float(psnr.mean()), "psnr_db_std": float(psnr.std()),
            "failed_image_ids": failed_all,
        }
        out.update(psnr_parity(frame0, host_frames[0][1]))
        if timer is not None:
            kd = timer.summary()                           # the roofline kernel in the timed steps
            ks = ks_all                                    # every kernel in the profile step
            if args.detail:
                rows = {k: {"launches": v["launches"], "us_per_launch": v["ms"] * 1e3 / v["launches"],
                            "ms_per_step": v["ms"], "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
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
            g = kd.get(dom, ks[dom])                       # its launches inside the timed steps (HIP events on the launch stream)
            label, bound, pmc_re = ROOFLINE_KERNELS.get(dom, (dom, "hbm", None))
            emulated = dom in ("attn_gdfn_fused", "gdfn_tail", "gdfn_fused", "qkv_dw_fused", "gemm1x1_f16x3", "dwgemm_f16x3", "gemm_ps_f16x3")
            if bound == "mfma" and emulated:
                # the unit that executes the arithmetic is the fp16 matrix core: three MFMA passes per fp32 product
                ach, peak, unit = 3.0 * g["flops"] / (g["ms"] * 1e-3) / 1e12, PEAK_F16_MFMA_TFLOPS, "TFLOP/s"
            elif bound == "mfma":
                ach, peak, unit = g["flops"] / (g["ms"] * 1e-3) / 1e12, PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
            else:
                ach, peak, unit = g["bytes"] / (g["ms"] * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
            out["roofline"] = {"kernel": label, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                               "frac": ach / peak, "traffic": pmc_traffic(pmc_re) if pmc_re else None,
                               "traffic_unit": "bytes per launch (HBM, PMC)", "traffic_source": pmc_provenance(),
                               "note": ("achieved = 3 x algorithmic fp32 FLOPs (2 M K N per 1x1 conv + 18 per depth-wise output; "
                                        "3 = the fp16 MFMA passes of the fp32 emulation; halo recompute not counted) / "
                                        "HIP-event time, against the dense fp16 MFMA peak; fp32_equivalent_tflops is the "
                                        "same without the factor 3 (f32-input MFMA peak: 157.3)") if bound == "mfma" and emulated else
                                       "achieved = algorithmic fp32 FLOPs / HIP-event time vs the f32-input MFMA peak" if bound == "mfma" else
                                       "achieved = algorithmic bytes (inputs once, outputs once, fp32) / HIP-event time",
                               "algorithmic_bytes_per_launch": g["bytes"] / g["launches"],
                               "fp32_equivalent_tflops": g["flops"] / (g["ms"] * 1e-3) / 1e12,
                               "launches": g["launches"],
                               "avg_launch_us": g["ms"] * 1e3 / g["launches"],
                               "share_of_kernel_time": ks[dom]["ms"] / tot_ms,
                               "timing": "HIP events around every launch of this kernel inside the timed steps; share_of_kernel_time "
                                         "and the `kernels` table: one un-timed profile step with events on every launch"}
            if dom in ("gdfn_fused", "attn_gdfn_fused"):
                # second view, clearly separate from `achieved`: SURVEY 8(d) counts the bytes at the REFERENCE's op
                # boundaries; the rows this one kernel replaces are LN + project_in (1 + 2r), dwconv + gate (3r) and
                # project_out + residual (r + 2) tensors of C N floats, r = hid / C = 255 / 96 (127 / 48 at C = 48): (3 + 6r) C N
                # (+ 3 C N for the attention branch's project_out + residual when that runs in this kernel as well).  The
                # kernel's own boundary traffic is `achieved`'s basis under bound "hbm" only.
                r_ = 255.0 / 96.0
                own_cn = 3.0 if dom == "attn_gdfn_fused" else 2.0      # own boundary (timer): x (+ v) read, y written, in C N floats
                ref_bytes = g["bytes"] / own_cn * (3.0 + 6.0 * r_ + (3.0 if dom == "attn_gdfn_fused" else 0.0))
                out["roofline"]["reference_rows"] = {
                    "rows": ("R2 attention project_out + residual + " if dom == "attn_gdfn_fused" else "") +
                            "R1 LayerNorm + R2 project_in + R3 depth-wise 3x3 + R5 gate + R2 project_out + residual (SURVEY 8a)",
                    "survey_bytes_per_launch": ref_bytes / g["launches"],
                    "gbs": ref_bytes / (g["ms"] * 1e-3) / 1e9,
                    "frac_of_hbm_peak": ref_bytes / (g["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                    "note": "bytes the reference's op-by-op decomposition moves for the rows this kernel fuses (SURVEY 8d: "
                            "18.9 C N floats per GDFN branch) / this kernel's time: what the fusion is worth in the survey's "
                            "own unit; the kernel itself moves 2 C N floats (+ halo) and is bound by its matrix + vector + LDS pipes"}
            out["kernels"] = {
                k: {"launches": v["launches"], "ms_per_step": v["ms"], "ms_per_frame": v["ms"] / FPS,
                    "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12, "gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                    "hbm_frac": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS}
                for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms"])}
            # whole-step bound: max(F/peakF, B/peakB) / t  
            # (the model is per FRAME: SURVEY 8d's unit; a step is FPS frames)
            F = sum(v["flops"] for v in ks.values()) / FPS
            Bt = sum(v["bytes"] for v in ks.values()) / FPS
            exact = bool(os.environ.get("IRM_GEMM_EXACT"))
            t_step = elapsed / (args.steps * FPS)
            hbm_ms, mfma_ms = REF_STEP_GBYTES / PEAK_HBM_GBS * 1e3, REF_STEP_TFLOP / PEAK_F32_MFMA_TFLOPS * 1e3
            # the bound that applies to the arithmetic actually run: with the 1x1 convs emulated on the fp16 cores the
            # f32-MFMA bound no longer binds, the reference's kernel-boundary HBM traffic does
            bound_ms = max(hbm_ms, mfma_ms) if exact else hbm_ms
            # this build's OWN bound: the bytes its kernel boundaries still move at 8 TB/s against the
            # matrix work it issues (three fp16 MFMA passes per fp32 product on the emulated GEMMs) at the dense fp16
            # peak - the fraction that cannot flatter, because fused-away traffic is not counted
            own_hbm_ms = Bt / (PEAK_HBM_GBS * 1e9) * 1e3
            own_mfma_ms = (F / (PEAK_F32_MFMA_TFLOPS * 1e12) if exact else 3.0 * F / (PEAK_F16_MFMA_TFLOPS * 1e12)) * 1e3
            own_bound_ms = max(own_hbm_ms, own_mfma_ms)
            pmc_bytes, pmc_frames = pmc_step_totals()
            out["step_model"] = {
                "unit": "one 1280x720 frame", "ms_per_frame": t_step * 1e3,
                "reference_gbytes": REF_STEP_GBYTES, "reference_tflop": REF_STEP_TFLOP,
                "hbm_bound_ms": hbm_ms, "f32_mfma_bound_ms": mfma_ms, "bound": "f32 mfma" if exact and mfma_ms > hbm_ms else "hbm",
                "bound_ms": bound_ms, "frac_of_bound": bound_ms / (t_step * 1e3),
                "this_build_kernel_boundary_gbytes": Bt / 1e9, "this_build_algorithmic_gflop": F / 1e9,
                "own_hbm_bound_ms": own_hbm_ms, "own_mfma_bound_ms": own_mfma_ms, "own_bound_ms": own_bound_ms,
                "own_frac": own_bound_ms / (t_step * 1e3),
                "hbm_util": None if pmc_bytes is None else pmc_bytes / (t_step * PEAK_HBM_GBS * 1e9),
                "hbm_gbytes_per_frame_pmc": None if pmc_bytes is None else pmc_bytes / 1e9,
                "note": "reference_* = the reference's op-boundary decomposition (SURVEY 8d); this_build_* = what is left "
                        "at this build's kernel boundaries after fusion (timer rows); own_* = this build's own boundary bytes / "
                        "8 TB/s vs its issued matrix work (3 fp16 MFMA passes per emulated fp32 product) / 2.5 PFLOP/s; "
                        "hbm_util = HBM bytes per frame measured by the committed PMC passes / (step time x 8 TB/s)"}
        if world == 1 and not args.no_legs and not os.environ.get("IRM_GEMM_EXACT"):
            # reference legs, each in its own process (the emulation switch is read when the weights are packed)
            torch.cuda.synchronize()
            ex = child_leg(["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-kernel-timer", "--no-legs"],
                           {"IRM_GEMM_EXACT": "1"})
            out["value_exact_f32"] = None if ex is None else ex["value"]
            out["value_exact_f32_note"] = "IRM_GEMM_EXACT=1: every GEMM on the f32-input MFMA, no fused branch kernels; 5 steps"
            pc = child_leg(["--leg", "pcie", "--steps", "8", "--warmup", "2"], {})
            out["value_pcie_inclusive"] = None if pc is None else pc["value"]
            out["value_pcie_inclusive_note"] = ("get_model_prediction(model, numpy_frame, device, **patch_config): uint8 frame in "
                                                "host memory -> uint8 frame in host memory, synchronous, 8 frames")
        if world == 1 and not args.no_cpu_baseline:
            gpu_tile = keep[0][0] if keep else None
            out["cpu_baseline"] = cpu_baseline(model, host_frames[0][0], gpu_tile, args.cpu_tile)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
