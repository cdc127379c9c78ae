"""Forward time of the headline Restormer for 6 / 12 / 18 tiles of 512x512 per launch sequence (= 1 / 2 / 3 frames of
1280x720): what batching the tiles of several frames buys (kernel tails, under-filled low-resolution levels)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import irm_amd  # noqa
from irm_amd import restormer

dev = torch.device("cuda:0")
model = restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
res = {}
for T in (6, 12, 18):
    x = torch.rand(T, 3, 512, 512, device=dev)
    for _ in range(2):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 6
    for _ in range(n):
        model(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res[f"{T} tiles"] = dict(ms=dt * 1e3, ms_per_frame=dt * 1e3 * 6 / T)
    model.release_workspace()
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
