"""Round-3 experiment on the two-stream stale read: the 1280x720 frame on
one stream vs three runs with the tiles on two HIP streams, for the product library and for the diagnostic build
whose every kernel starts with an agent-scope acquire (-DIRM_ACQUIRE_ENTRY: buffer_inv sc1 on every wave).
  python tools/probes/two_stream_acq.py [lib.so ...]        (one JSON line per library)"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import irm_amd  # noqa
from irm_amd import _hip, restormer, synth, utils

dev = torch.device("cuda:0")
libs = sys.argv[1:] or [_hip.LIB_PATH]
inp, _ = synth.synth_image_pair(0, 720, 1280, 3, seed_base=1000, blur=15)
img = torch.from_numpy(inp).to(dev)
os.environ["IRM_NO_GRAPH"] = "1"
os.environ["IRM_EXPERIMENTAL_STREAMS"] = "1"
for path in libs:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in _hip.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, ctypes.c_int
    _hip._lib = lib
    model = restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
    keep = []
    base, _ = utils.tiled_forward_device(model, img, 512, 96, pad8=True, max_batch=8, keep_tiles=keep)
    base = base.clone()
    torch.cuda.synchronize()
    model._allow_two_streams = True
    model.num_streams = 2
    rows = []
    for _ in range(4):
        k2 = []
        two, _ = utils.tiled_forward_device(model, img, 512, 96, pad8=True, max_batch=8, keep_tiles=k2)
        torch.cuda.synchronize()
        d = (two.int() - base.int()).abs()
        rows.append(dict(u8_bytes_differing=int((d > 0).sum()), u8_max=int(d.max()),
                         float_tiles_maxabs=float((k2[0] - keep[0]).abs().max()),
                         float_values_differing=int((k2[0] != keep[0]).sum())))
    print(json.dumps({"lib": os.path.basename(path), "two_stream_runs_vs_one_stream": rows}), flush=True)
    del model
