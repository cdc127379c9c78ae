import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import irm_amd  # noqa
from irm_amd import _hip, ops, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_fused import qkv_ref, rnd
dev = torch.device("cuda:0")
C, H, W = 96, 64, 64
M = 3 * C
x = rnd("ex", (1, C, H, W), -1.5, 2.0)
def run(tag, w, dq, lnw, lnb):
    refq = qkv_ref(x, lnw, lnb, 1, w, None, dq, None)
    pkq = _hip.pack_qkv_fused(w.to(dev), None, dq, None, lnw, lnb)
    yq = torch.empty(1, M, H, W, device=dev)
    ops.qkv_dw_fused(pkq, x.to(dev), yq, C, M, ln_mode=1)
    d = (yq.cpu().double() - refq).abs()
    rel = d / refq.abs().clamp_min(1e-3)
    i = d.argmax(); idx = [int(v) for v in torch.unravel_index(i, d.shape)]
    print(f"{tag}: max {d.max():.3e} mean {d.mean():.3e} ref max {refq.abs().max():.2f} worst {idx} got {float(yq.cpu().flatten()[i]):.6f} want {float(refq.flatten()[i]):.6f}")
    # error by channel mod 32 and by position in tile
    dm = d[0].amax(dim=(1, 2)); print("   per-channel max (first 32):", " ".join(f"{v:.0e}" for v in dm[:32].tolist()))
    pm = d[0].amax(dim=0); print("   rows max:", " ".join(f"{v:.0e}" for v in pm.amax(dim=1)[:16].tolist()), "| cols max:", " ".join(f"{v:.0e}" for v in pm.amax(dim=0)[:34].tolist()))
w = rnd("e4", (M, C), -.3, .3); dq = rnd("e5", (M, 9), -.4, .4)
ident = torch.zeros(M, 9); ident[:, 4] = 1.0
ones, zeros = torch.ones(C), torch.zeros(C)
run("full", w, dq, rnd("elw", (C,), .5, 1.5), rnd("elb", (C,), -.2, .2))
run("dw=identity", w, ident, rnd("elw", (C,), .5, 1.5), rnd("elb", (C,), -.2, .2))
run("dw=identity, ln affine off", w, ident, ones, zeros)
weye = torch.zeros(M, C); weye[:C] = torch.eye(C); weye[C:2*C] = torch.eye(C) * 0.25
run("W=eye, dw=identity, affine off", weye, ident, ones, zeros)
