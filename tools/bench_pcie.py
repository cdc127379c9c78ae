"""PCIe-inclusive rate of the headline workload: numpy uint8 frame in host memory -> get_model_prediction ->
numpy uint8 frame (one H2D and one D2H copy per frame plus a host synchronisation), as the reference-shaped
call is used (scripts/tests.py:391).  Reported next to bench.py's value, never as it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import irm_amd
from irm_amd import restormer, synth, utils
dev = torch.device("cuda:0")
model = restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
cfg = utils.get_patch_config("deblurring", "motion", "Restormer")
frames = [synth.synth_image_pair(i, 720, 1280, 3, seed_base=1000, blur=15)[0] for i in range(4)]
for i in range(2):
    utils.get_model_prediction(model, frames[i], dev, **cfg)
torch.cuda.synchronize()
n = 12
t0 = time.perf_counter()
for i in range(n):
    pred, ms = utils.get_model_prediction(model, frames[i % 4], dev, **cfg)
dt = (time.perf_counter() - t0) / n
print(f"get_model_prediction (host numpy in/out, synchronous): {dt*1e3:.2f} ms per frame = {1/dt:.2f} images/s")
