"""Fused dw3x3(+gate)+1x1 kernel against the two-kernel path it replaces, at the headline shapes."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import irm_amd
from irm_amd import _hip, ops
dev = torch.device("cuda:0")

def timeit(fn, n=5):
    for _ in range(2): fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2] * 1e3

B = 6
for (M, K, H, W, gate) in ([(96, 255, 512, 512, True)] if os.environ.get("IRM_BENCH_ONE") else [(96, 255, 512, 512, True), (48, 127, 512, 512, True), (96, 255, 256, 256, True), (96, 96, 512, 512, False), (48, 48, 512, 512, False), (96, 96, 256, 256, False)]):
    kin = 2 * K if gate else K
    x = torch.randn(B, kin, H, W, device=dev)
    w9 = torch.randn(kin, 9, device=dev) * 0.3
    wp = _hip.pack_gemm_weight(torch.randn(M, K) * 0.1).to(dev)
    dwp = _hip.pack_dw_table(w9, None, K, gate)
    y = torch.randn(B, M, H, W, device=dev)
    g = torch.empty(B, K, H, W, device=dev)
    st = torch.empty(B, 2, H * W, device=dev)
    def unfused():
        if gate: ops.dwconv3x3_gate(x, w9, g)
        else: ops.dwconv3x3(x, w9, g)
        ops.gemm1x1(wp, g, y, M, K, res=y, stats_out=st)
    def fused():
        ops.dwgemm(wp, dwp, x, y, M, K, gate=gate, res=y, stats_out=st)
    tu, tf = timeit(unfused), timeit(fused)
    print(f"M{M} K{K} {H}x{W} gate{int(gate)}: unfused {tu:8.1f} us  fused {tf:8.1f} us  ({tu / tf:.2f}x)  "
          f"fused reads {4.0 * B * kin * H * W / tf / 1e3:6.0f} GB/s, {2.0 * B * M * K * H * W / tf / 1e6:5.1f} TF", flush=True)
