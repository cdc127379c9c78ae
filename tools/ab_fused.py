"""A/B of the fused branch kernels between two builds of libirm_hip.so in ONE process (interleaved rounds, HIP events):
  python tools/ab_fused.py tools/ab/libirm_base.so [image-restoration-models_amd/libirm_hip.so]
The base library is a build from before round 3 (gate taps not pre-scaled): its operands are packed accordingly."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import irm_amd  # noqa
from irm_amd import _hip, ops, synth

dev = torch.device("cuda:0")


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in _hip.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, ctypes.c_int
    return lib


# first argument: the pre-round-3 build (old operand layout); every further library takes the current layout
libs = {"base": load(sys.argv[1])}
for pth in (sys.argv[2:] or [_hip.LIB_PATH]):
    libs["new" if pth == (sys.argv[2:] or [_hip.LIB_PATH])[0] else os.path.basename(pth)] = load(pth)
SHAPES = [(96, 255, 512, 512, 6), (96, 255, 256, 256, 6), (48, 127, 512, 512, 6)]
res = {}
for C, hid, H, W, B in SHAPES:
    r = lambda n, s, lo=-1., hi=1.: synth.uniform(5, n, s, lo, hi)
    x = torch.randn(B, C, H, W, device=dev)
    y = {k: torch.empty_like(x) for k in libs}
    args = (r("a", (2 * hid, C), -.3, .3).to(dev), None, r("b", (2 * hid, 9), -.4, .4), None, r("c", (C, hid), -.3, .3),
            r("d", (C,), .5, 1.5), r("e", (C,), -.2, .2))
    pk = {k: _hip.pack_gdfn_fused(*args, gate_prescale=(k != "base")) for k in libs}
    M = 3 * C
    yq = {k: torch.empty(B, M, H, W, device=dev) for k in libs}
    pq = _hip.pack_qkv_fused(r("qa", (M, C), -.3, .3).to(dev), None, r("qb", (M, 9), -.4, .4), None, r("d", (C,), .5, 1.5),
                             r("e", (C,), -.2, .2))
    times = {(k, op): [] for k in libs for op in ("gdfn", "qkv")}
    for rnd in range(6):
        for k, lib in libs.items():
            _hip._lib = lib
            for op in ("gdfn", "qkv"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 8
                e0.record()
                for _ in range(n):
                    if op == "gdfn":
                        ops.gdfn_fused(pk[k], x, y[k], C, hid, ln_mode=1)
                    else:
                        ops.qkv_dw_fused(pq, x, yq[k], C, M, ln_mode=1)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[(k, op)].append(e0.elapsed_time(e1) / n * 1e3)
    d = float((y["base"] - y["new"]).abs().max())
    dq = float((yq["base"] - yq["new"]).abs().max())
    for op in ("gdfn", "qkv"):
        tb, tn = sorted(times[("base", op)]), sorted(times[("new", op)])
        res[f"{op} C{C} {H}x{W} B{B}"] = dict(base_us_med=tb[len(tb) // 2], new_us_med=tn[len(tn) // 2], base_us_min=tb[0],
                                              new_us_min=tn[0], ratio=tn[len(tn) // 2] / tb[len(tb) // 2],
                                              **{k + "_us_med": sorted(times[(k, op)])[len(tb) // 2] for k in libs if k not in ("base", "new")})
    res[f"maxabs base-vs-new C{C} {H}x{W}"] = dict(gdfn=d, qkv=dq)
print(json.dumps(res, indent=1))
