"""Micro-benchmark of the fused branch kernels at the C4 shapes (HIP events, interleaved rounds)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import irm_amd  # noqa
from irm_amd import _hip, ops, synth

if "--lib" in sys.argv:
    _hip.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
dev = torch.device("cuda:0")
SHAPES = [(96, 255, 512, 512, 6), (96, 255, 256, 256, 6), (48, 127, 512, 512, 6)]
if "--locality" in sys.argv:       # same tiles, per-plane contiguous tile interiors (8 rows x 32 px = 1 KiB): what DRAM page locality is worth
    SHAPES = [(96, 255, 512, 512, 6), (96, 255, 8192, 32, 6), (96, 255, 2048, 128, 6), (48, 127, 512, 512, 6), (48, 127, 8192, 32, 6)]
res = {}
for C, hid, H, W, B in SHAPES:
    r = lambda n, s, lo=-1., hi=1.: synth.uniform(5, n, s, lo, hi)
    x = torch.randn(B, C, H, W, device=dev)
    y = torch.empty_like(x)
    pk = _hip.pack_gdfn_fused(r("a", (2 * hid, C), -.3, .3).to(dev), None, r("b", (2 * hid, 9), -.4, .4), None,
                              r("c", (C, hid), -.3, .3), r("d", (C,), .5, 1.5), r("e", (C,), -.2, .2))
    for _ in range(3):
        ops.gdfn_fused(pk, x, y, C, hid, ln_mode=1)
    torch.cuda.synchronize()
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gdfn_fused(pk, x, y, C, hid, ln_mode=1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    px = B * H * W
    res[f"gdfn C{C} hid{hid} {H}x{W} B{B}"] = dict(ms=ms, gbs=px * 2 * C * 4 / ms / 1e6,
                                                  tflops_f32eq=px * (6.0 * hid * C + 36 * hid) / ms / 1e9)
for C, hid, H, W, B in SHAPES:
    M = 3 * C
    r = lambda n, s, lo=-1., hi=1.: synth.uniform(6, n, s, lo, hi)
    x = torch.randn(B, C, H, W, device=dev)
    y = torch.empty(B, M, H, W, device=dev)
    pk = _hip.pack_qkv_fused(r("a", (M, C), -.3, .3).to(dev), None, r("b", (M, 9), -.4, .4), None, r("d", (C,), .5, 1.5),
                             r("e", (C,), -.2, .2))
    for _ in range(3):
        ops.qkv_dw_fused(pk, x, y, C, M, ln_mode=1)
    torch.cuda.synchronize()
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.qkv_dw_fused(pk, x, y, C, M, ln_mode=1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    res[f"qkv C{C} {H}x{W} B{B}"] = dict(ms=ms, gbs=B * H * W * (C + M) * 4 / ms / 1e6)
print(json.dumps(res, indent=1))
