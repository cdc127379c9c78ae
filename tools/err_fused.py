"""Error of the fused branch kernels vs float64, next to a plain fp32 torch chain on the same inputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import irm_amd  # noqa
from irm_amd import _hip, ops, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_fused import gdfn_ref, qkv_ref, rnd
dev = torch.device("cuda:0")
for C, hid, H, W in [(96, 255, 64, 64), (48, 127, 64, 64)]:
    x = rnd("ex", (1, C, H, W), -1.5, 2.0)
    lnw, lnb = rnd("elw", (C,), .5, 1.5), rnd("elb", (C,), -.2, .2)
    pin, pout, dw = rnd("e1", (2 * hid, C), -.3, .3), rnd("e2", (C, hid), -.3, .3), rnd("e3", (2 * hid, 9), -.4, .4)
    ref = gdfn_ref(x, lnw, lnb, 1, pin, None, dw, None, pout, None)
    pk = _hip.pack_gdfn_fused(pin.to(dev), None, dw, None, pout, lnw, lnb)
    y = torch.empty(1, C, H, W, device=dev)
    ops.gdfn_fused(pk, x.to(dev), y, C, hid, ln_mode=1)
    # fp32 chain
    xf = x.to(dev)
    mu = xf.mean(1, keepdim=True); var = xf.var(1, unbiased=False, keepdim=True)
    xn = (xf - mu) / torch.sqrt(var + 1e-5) * lnw.to(dev)[None, :, None, None] + lnb.to(dev)[None, :, None, None]
    h = F.conv2d(xn, pin.to(dev)[:, :, None, None]); h = F.conv2d(h, dw.to(dev).view(-1, 1, 3, 3), padding=1, groups=2 * hid)
    y32 = xf + F.conv2d(F.gelu(h[:, :hid]) * h[:, hid:], pout.to(dev)[:, :, None, None])
    d = (y.cpu().double() - ref).abs(); d32 = (y32.cpu().double() - ref).abs()
    print(f"gdfn C{C}: fused max {d.max():.3e} mean {d.mean():.3e} | torch fp32 max {d32.max():.3e} mean {d32.mean():.3e} | ref max {ref.abs().max():.2f}")
    i = d.argmax(); print("   worst at", [int(v) for v in torch.unravel_index(i, d.shape)])
    M = 3 * C
    w, dq = rnd("e4", (M, C), -.3, .3), rnd("e5", (M, 9), -.4, .4)
    refq = qkv_ref(x, lnw, lnb, 1, w, None, dq, None)
    pkq = _hip.pack_qkv_fused(w.to(dev), None, dq, None, lnw, lnb)
    yq = torch.empty(1, M, H, W, device=dev)
    ops.qkv_dw_fused(pkq, x.to(dev), yq, C, M, ln_mode=1)
    hq = F.conv2d(F.conv2d(xn, w.to(dev)[:, :, None, None]), dq.to(dev).view(-1, 1, 3, 3), padding=1, groups=M)
    d = (yq.cpu().double() - refq).abs(); d32 = (hq.cpu().double() - refq).abs()
    print(f"qkv  C{C}: fused max {d.max():.3e} mean {d.mean():.3e} | torch fp32 max {d32.max():.3e} mean {d32.mean():.3e} | ref max {refq.abs().max():.2f}")
