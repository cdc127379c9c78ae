"""Gram pass (irm_mdta_gram_f16x3_f32) on the headline shapes; --lib path for A/B of library variants."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irm_amd  # noqa
from irm_amd import _hip, ops
if "--lib" in sys.argv:
    _hip.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
dev = torch.device("cuda:0")
B = 6
for (C, heads, H, W) in [(96, 1, 512, 512), (96, 2, 256, 256), (48, 1, 512, 512), (192, 4, 128, 128), (384, 8, 64, 64)]:
    N = H * W
    qkv = torch.randn(B, 3 * C, H, W, device=dev)
    chunk, nchunk, rec = ops.mdta_plan(B, C, heads, N)
    part = torch.empty(B * heads * nchunk * rec, device=dev)
    scale = torch.ones(2 * C, device=dev)
    fn = lambda: _hip.call("irm_mdta_gram_f16x3_f32", _hip.ptr(qkv), qkv.stride(0), _hip.ptr(scale), _hip.ptr(part), B, C, heads, N, chunk)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30 * 1e3
    print(f"C{C} h{heads} {H}x{W}: {t:7.1f} us  {8.0 * B * C * N / t / 1e3:6.0f} GB/s  (chunk {chunk}, nchunk {nchunk})", flush=True)
