"""A/B of the f16x3 Gram pass between two builds of the library in one process (interleaved rounds):
  python tools/ab_gram.py tools/ab/libirm_base.so [new.so]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import irm_amd  # noqa
from irm_amd import _hip, ops

dev = torch.device("cuda:0")


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in _hip.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, ctypes.c_int
    return lib


libs = {"base": load(sys.argv[1]), "new": load(sys.argv[2] if len(sys.argv) > 2 else _hip.LIB_PATH)}
B = 6
res = {}
for (C, heads, H, W) in [(96, 1, 512, 512), (96, 2, 256, 256), (48, 1, 512, 512), (192, 4, 128, 128), (384, 8, 64, 64)]:
    N = H * W
    qkv = torch.randn(B, 3 * C, H, W, device=dev)
    chunk, nchunk, rec = ops.mdta_plan(B, C, heads, N)
    part = {k: torch.empty(B * heads * nchunk * rec, device=dev) for k in libs}
    scale = torch.ones(2 * C, device=dev)
    times = {k: [] for k in libs}
    for rnd in range(8):
        for k, lib in libs.items():
            _hip._lib = lib
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _hip.call("irm_mdta_gram_f16x3_f32", _hip.ptr(qkv), qkv.stride(0), _hip.ptr(scale), _hip.ptr(part[k]), B, C, heads, N, chunk)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[k].append(e0.elapsed_time(e1) * 100)
    tb, tn = sorted(times["base"]), sorted(times["new"])
    res[f"C{C} h{heads} {H}x{W}"] = dict(base_us=tb[len(tb) // 2], new_us=tn[len(tn) // 2], base_min=tb[0], new_min=tn[0],
                                          maxabs=float((part["base"] - part["new"]).abs().max()))
print(json.dumps(res, indent=1))
