"""Each op against a noisy neighbour: stream 2 runs the full model, stream 1 repeats one op (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from irm_amd import restormer, ops, _hip, synth
dev = torch.device("cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.Generator().manual_seed(0)
noise_model = restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
xn = torch.rand(3, 3, 512, 512, generator=g).to(dev)
noise_model(xn); torch.cuda.synchronize()

def noisy(name, args, fn, reps=30):
    fn(args); torch.cuda.synchronize(); ref = args["y"].clone()
    bad, worst = 0, 0.0
    for r in range(2):
        with torch.cuda.stream(s2):
            noise_model(xn)
        outs = []
        with torch.cuda.stream(s1):
            for _ in range(reps):
                args["y"].zero_()
                fn(args)
                outs.append(args["y"].clone())
        torch.cuda.synchronize()
        for o in outs:
            d = float((o - ref).abs().max())
            bad += d > 0; worst = max(worst, d)
    print(f"{name:46s} {bad:3d} of {2*reps} runs differ, worst {worst:.3e}", flush=True)

B = 3
r = lambda n, s, lo=-1., hi=1.: synth.uniform(5, n, s, lo, hi)
for ci, co, H, mode in ((3, 48, 512, 0), (48, 24, 512, 1), (96, 48, 256, 1), (192, 96, 128, 1), (384, 768, 64, 2), (192, 384, 128, 2), (96, 192, 256, 2), (96, 3, 512, 0)):
    X = torch.randn(B, ci, H, H, generator=g).to(dev)
    cw = _hip.pack_conv3x3((torch.randn(co, ci, 3, 3, generator=g) * 0.05).to(dev))
    oc, oh = (co * 4, H // 2) if mode == 1 else (co // 4, H * 2) if mode == 2 else (co, H)
    noisy(f"conv3x3_f16x3 ci{ci} co{co} {H} mode{mode}", dict(x=X, y=torch.zeros(B, oc, oh, oh, device=dev)),
          lambda a, cw=cw, ci=ci, co=co, mode=mode: ops.conv3x3(cw, a["x"], a["y"], ci, co, store_mode=mode))
for C, H, hid in ((96, 512, 255), (48, 512, 127), (96, 256, 255)):
    X = torch.randn(B, C, H, H, generator=g).to(dev)
    pk = _hip.pack_gdfn_fused(r("a", (2 * hid, C), -.3, .3).to(dev), None, r("b", (2 * hid, 9), -.4, .4), None,
                              r("c", (C, hid), -.3, .3), r("d", (C,), .5, 1.5), r("e", (C,), -.2, .2))
    noisy(f"gdfn_fused C{C} {H}", dict(x=X, y=torch.zeros_like(X)), lambda a, pk=pk, C=C, hid=hid: ops.gdfn_fused(pk, a["x"], a["y"], C, hid, ln_mode=1))
    pkq = _hip.pack_qkv_fused(r("a2", (3 * C, C), -.3, .3).to(dev), None, r("b2", (3 * C, 9), -.4, .4), None, r("d", (C,), .5, 1.5), r("e", (C,), -.2, .2))
    noisy(f"qkv_dw_fused C{C} {H}", dict(x=X, y=torch.zeros(B, 3 * C, H, H, device=dev)), lambda a, pkq=pkq, C=C: ops.qkv_dw_fused(pkq, a["x"], a["y"], C, 3 * C, ln_mode=1))
    ws = _hip.pack_gemm_weight_split((torch.randn(C, C, generator=g) * 0.1)).to(dev)
    noisy(f"gemm1x1_f16x3 res C{C} {H}", dict(x=X, y=torch.zeros_like(X), r=X.clone()), lambda a, ws=ws, C=C: ops.gemm1x1(ws, a["x"], a["y"], C, C, res=a["r"], split=True))
    QKV = torch.randn(B, 3 * C, H, H, generator=g).to(dev)
    temp = torch.ones(1, device=dev); wout = (torch.randn(C, C, generator=g) * 0.2).to(dev)
    chunk, nchunk, rec = ops.mdta_plan(B, C, 1, H * H)
    sc = torch.full((2 * C,), 1024.0, device=dev)
    noisy(f"mdta_fold f16x3 C{C} {H}", dict(q=QKV, part=torch.zeros(B * nchunk * rec, device=dev), gs=torch.zeros(B * rec, device=dev), y=torch.zeros(B * ops.mfold_numel(C), device=dev)),
          lambda a, C=C: ops.mdta_fold(a["q"], a["part"], a["gs"], temp, wout, a["y"], C, 1, split=True, gram_scale=sc))
