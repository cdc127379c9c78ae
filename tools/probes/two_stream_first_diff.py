"""First op whose output differs under concurrency, without in-stream probes: the forward is cut after op k (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from irm_amd import restormer, ops
dev = torch.device("cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
STATE = {"limit": 10 ** 9}
COUNT, LAST = {}, {}
def wrap(name, out_index):
    orig = getattr(ops, name)
    def f(*a, **k):
        sid = torch.cuda.current_stream().cuda_stream
        c = COUNT.get(sid, 0)
        COUNT[sid] = c + 1
        if c >= STATE["limit"]:
            return None
        r = orig(*a, **k)
        LAST[sid] = (c, name, a[out_index])
        return r
    setattr(ops, name, f)
for n, i in (("gemm1x1", 2), ("dwconv3x3", 2), ("dwconv3x3_gate", 2), ("ln_stats", 1), ("mdta_fold", 5), ("conv3x3", 2)):
    wrap(n, i)
m = restormer.Restormer(num_blocks=[0, 0, 1, 0], num_refinement_blocks=0, LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
g = torch.Generator().manual_seed(1)
xa, xb = torch.rand(3, 3, 512, 512, generator=g).to(dev), torch.rand(3, 3, 512, 512, generator=g).to(dev)
def run(x, st):
    COUNT.clear(); LAST.clear()
    with torch.cuda.stream(st):
        m(x)
for k in range(1, 29):
    STATE["limit"] = k
    # serial references on the same streams (workspaces of s1 / s2)
    run(xa, s1); torch.cuda.synchronize(); ia, na, ta = LAST[s1.cuda_stream]; ra = ta.clone()
    run(xb, s2); torch.cuda.synchronize(); ib, nb_, tb = LAST[s2.cuda_stream]; rb = tb.clone()
    worst = 0.0
    for trial in range(4):
        COUNT.clear(); LAST.clear()
        with torch.cuda.stream(s1):
            m(xa)
        with torch.cuda.stream(s2):
            m(xb)
        torch.cuda.synchronize()
        worst = max(worst, float((LAST[s1.cuda_stream][2] - ra).abs().max()), float((LAST[s2.cuda_stream][2] - rb).abs().max()))
    print(f"cut after op #{k-1:2d} {na:16s} {tuple(ta.shape)}: max-abs concurrent vs serial {worst:.3e}", flush=True)
