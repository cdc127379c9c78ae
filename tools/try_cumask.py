#!/usr/bin/env python3
"""Experiment: two tile-group streams restricted to complementary CU masks (hipExtStreamCreateWithCUMask)."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irm_amd  # noqa
from irm_amd import restormer, synth, utils
from irm_amd.configs import PATCH_CONFIG

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(mask_words):
    arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(mask_words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def run(name, streams, steps=6):
    dev = torch.device("cuda:0")
    model = restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
    model.num_streams = len(streams) if streams else 1
    cfg = PATCH_CONFIG["Restormer"][1]
    inp, tgt = synth.synth_image_pair(0, 720, 1280, 3)
    img = torch.from_numpy(inp).to(dev)
    if streams:
        utils._SIDE_STREAMS.clear()
        for i, s in enumerate(streams):
            utils._SIDE_STREAMS[(str(dev), i)] = s
    for _ in range(2):
        utils.tiled_forward_device(model, img, cfg["patch_size"], cfg["patch_overlap"], True, max_batch=9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        utils.tiled_forward_device(model, img, cfg["patch_size"], cfg["patch_overlap"], True, max_batch=9)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step", flush=True)


if __name__ == "__main__":
    full = [0xFFFFFFFF] * 8
    run("1 stream", None)
    run("2 streams unmasked", [torch.cuda.Stream(), torch.cuda.Stream()])
    run("2 streams lower/upper 128 CUs", [masked_stream([0xFFFFFFFF] * 4 + [0] * 4), masked_stream([0] * 4 + [0xFFFFFFFF] * 4)])
    run("2 streams even/odd CUs", [masked_stream([0x55555555] * 8), masked_stream([0xAAAAAAAA] * 8)])
    run("2 streams 192/64 CUs", [masked_stream([0xFFFFFFFF] * 6 + [0] * 2), masked_stream([0] * 6 + [0xFFFFFFFF] * 2)])
    run("3 streams unmasked", [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()])
    run("3 streams thirds", [masked_stream([0x49249249, 0x92492492, 0x24924924, 0x49249249, 0x92492492, 0x24924924, 0x49249249, 0x92492492]),
                             masked_stream([0x92492492, 0x24924924, 0x49249249, 0x92492492, 0x24924924, 0x49249249, 0x92492492, 0x24924924]),
                             masked_stream([0x24924924, 0x49249249, 0x92492492, 0x24924924, 0x49249249, 0x92492492, 0x24924924, 0x49249249])])
