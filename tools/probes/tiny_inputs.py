import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import irm_amd
from irm_amd import _hip, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for scale in (1.0, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
    M, K, H, W, B = 96, 255, 16, 16, 1
    x = (torch.rand(B, K, H, W) * 2 - 1) * scale
    w = (torch.rand(M, K) * 2 - 1) * 0.2
    ref = torch.einsum("mk,bkhw->bmhw", w.double(), x.double())
    y16 = torch.zeros(B, M, H, W, device=dev); y32 = torch.zeros(B, M, H, W, device=dev)
    ops.gemm1x1(_hip.pack_gemm_weight_split(w).to(dev), x.to(dev), y16, M, K, res=y16.clone(), split=True)
    ops.gemm1x1(_hip.pack_gemm_weight(w).to(dev), x.to(dev), y32, M, K, res=y32.clone())
    e16 = (y16.cpu().double() - ref).abs().max().item(); e32 = (y32.cpu().double() - ref).abs().max().item()
    print(f"|x| <= {scale:g}: |y| max {ref.abs().max().item():.3e}  err f16x3 {e16:.3e}  err f32 {e32:.3e}", flush=True)
