"""Consumer kernels reading a buffer that a torch copy has just rewritten (changing data), noisy neighbour (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from irm_amd import restormer, ops, _hip
dev = torch.device("cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.Generator().manual_seed(0)
noise_model = restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
xn = torch.rand(3, 3, 512, 512, generator=g).to(dev)
noise_model(xn); torch.cuda.synchronize()
B = 3
def test(name, C, H, consumer, out_shape, producer="copy"):
    xs = [torch.randn(B, C, H, H, generator=g).to(dev) for _ in range(10)]
    Wk = torch.empty(B, C, H, H, device=dev)
    Y = torch.empty(*out_shape, device=dev)
    dwp = (torch.randn(C, 9, generator=g) * 0.3).to(dev)
    def chain():
        outs = []
        for x in xs:
            if producer == "copy":
                Wk.copy_(x)
            else:
                ops.dwconv3x3(x, dwp, Wk)
            consumer(Wk, Y)
            outs.append(Y.clone())
        return outs
    ref = chain(); torch.cuda.synchronize()
    bad, worst = 0, 0.0
    for trial in range(8):
        with torch.cuda.stream(s2):
            noise_model(xn)
        with torch.cuda.stream(s1):
            got = chain()
        torch.cuda.synchronize()
        for a, b in zip(ref, got):
            d = float((a - b).abs().max()); bad += d > 0; worst = max(worst, d)
    print(f"{name:40s} producer {producer:7s}: {bad:3d} of 80 outputs differ, worst {worst:.3e}", flush=True)
for prod in ("copy", "dwconv"):
    test("ln_stats C192 128^2 (PQ 16)", 192, 128, lambda w, y: ops.ln_stats(w, y), (B, 2, 128, 128), prod)
    test("ln_stats C384 64^2 (PQ 16)", 384, 64, lambda w, y: ops.ln_stats(w, y), (B, 2, 64, 64), prod)
    test("ln_stats C96 256^2 (PQ 64)", 96, 256, lambda w, y: ops.ln_stats(w, y), (B, 2, 256, 256), prod)
    cw = _hip.pack_conv3x3((torch.randn(96, 192, 3, 3, generator=g) * 0.05).to(dev))
    test("conv3x3_f16x3 ci192 co96 128^2", 192, 128, lambda w, y: ops.conv3x3(cw, w, y, 192, 96), (B, 96, 128, 128), prod)
    dw = (torch.randn(192, 9, generator=g) * 0.3).to(dev)
    test("dwconv3x3 C192 128^2", 192, 128, lambda w, y: ops.dwconv3x3(w, dw, y), (B, 192, 128, 128), prod)
