"""ctypes binding of libirm_hip.so (C ABI: include/irm_hip.h).

The product path has no CPU fallback: if the library is missing, or a call is
rejected, this module raises.  Tensors are passed as raw device pointers, work
is enqueued on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libirm_hip.so")

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SILU, ACT_RELU6 = 0, 1, 2, 3, 4
LN_NONE, LN_WITHBIAS, LN_BIASFREE = 0, 1, 2

_P, _L, _I, _F = C.c_void_p, C.c_long, C.c_int, C.c_float

#: symbol -> argtypes, mirrors include/irm_hip.h one to one
SIGNATURES = {
    "irm_version": [],
    "irm_ln_stats_f32": [_P, _L, _P, _I, _I, _I, _F, _P],
    "irm_gemm1x1_f32": [_P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _F, _P, _P],
    "irm_gemm1x1_f16x3_f32": [_P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _F, _P, _P],
    "irm_dwconv3x3_f32": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P],
    "irm_dwconv3x3_gate_f32": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "irm_dwgemm_f32": [_P, _L, _P, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _P, _F, _P],
    "irm_dwgemm_f16x3_f32": [_P, _L, _P, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _P, _F, _P],
    "irm_mdta_gram_f32": [_P, _L, _P, _I, _I, _I, _I, _I, _P],
    "irm_mdta_finalize_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "irm_mdta_finalize_f16x3_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "irm_conv3x3_f32": [_P, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "irm_tile_extract": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _I, _P],
    "irm_chan_stats_f32": [_P, _L, _P, _I, _I, _I, _F, _P],
    "irm_chan_stats_ws_f32": [_P, _L, _P, _P, _L, _I, _I, _I, _F, _P],
    "irm_chan_norm_act_f32": [_P, _L, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P],
    "irm_conv3x3_s2_f32": [_P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _P],
    "irm_dwconv3x3_s2_f32": [_P, _L, _P, _P, _L, _I, _I, _I, _I, _P],
    "irm_upsample_add_f32": [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P],
    "irm_transpose_f32": [_P, _L, _P, _L, _I, _I, _I, _P],
    "irm_selective_scan_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "irm_losh_combine_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _F, _P],
    "irm_window_blend": [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P],
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load libirm_hip.so once; raise HipLibraryError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C image-restoration-models_amd/csrc).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = _I
        _lib = lib
    return _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    assert t.is_cuda, "HIP kernels need device tensors"
    return t.data_ptr()


def call(name: str, *args):
    """Invoke a C-ABI entry point on the current stream; raise on a non-zero status."""
    rc = getattr(load(), name)(*args, _stream())
    if rc != 0:
        raise HipLibraryError(f"{name} failed with status {rc} "
                              f"({'invalid arguments' if rc == -1 else 'HIP launch error'})")


# ---------------------------------------------------------------------------
# host-side weight packing (MFMA B-operand order, see include/irm_hip.h)
# ---------------------------------------------------------------------------

def pack_gemm_weight(w: torch.Tensor) -> torch.Tensor:
    """W [M][K] -> wp[ceil(M/16)][4*ceil(K/16)][64], wp[mt][ks][l] = W[16mt+(l&15)][4ks+(l>>4)]."""
    w = w.detach().reshape(w.shape[0], -1).float()
    m, k = w.shape
    mt, ks = (m + 15) // 16, 4 * ((k + 15) // 16)
    wpad = torch.zeros(mt * 16, ks * 4, dtype=torch.float32, device=w.device)
    wpad[:m, :k] = w
    # [mt][16 r][ks][4 g] -> [mt][ks][g][r]  (lane = g*16 + r)
    return wpad.view(mt, 16, ks, 4).permute(0, 2, 3, 1).contiguous().view(-1)


def pack_gemm_weight_split(w: torch.Tensor) -> torch.Tensor:
    """W [M][K] -> fp16 hi/lo parts in irm_gemm1x1_f16x3_f32's order (same number of floats as pack_gemm_weight):
    [mtile][stage][hi|lo][lane = g*16 + m][j] = part[16 mtile + m][16 stage + 4 j + g]."""
    w = w.detach().reshape(w.shape[0], -1).float()
    m, k = w.shape
    mt, st = (m + 15) // 16, (k + 15) // 16
    wpad = torch.zeros(mt * 16, st * 16, dtype=torch.float32, device=w.device)
    wpad[:m, :k] = w
    hi = wpad.half()
    lo = (wpad - hi.float()).half()

    def arrange(t):                                   # [mt][16 m][st][4 j][4 g] -> [mt][st][g][m][j]
        return t.view(mt, 16, st, 4, 4).permute(0, 2, 4, 1, 3)
    packed = torch.stack([arrange(hi), arrange(lo)], dim=2).contiguous()      # [mt][st][2][g][m][j]
    return packed.view(-1).view(torch.float32)


def pack_conv3x3_weight(w: torch.Tensor) -> torch.Tensor:
    """W [Co][Ci][3][3] -> wp[9][ceil(Co/16)][2*ceil(Ci/8)][64]."""
    w = w.detach().float()
    co, ci = w.shape[:2]
    mt, ks = (co + 15) // 16, 2 * ((ci + 7) // 8)
    wpad = torch.zeros(9, mt * 16, ks * 4, dtype=torch.float32, device=w.device)
    wpad[:, :co, :ci] = w.reshape(co, ci, 9).permute(2, 0, 1)
    return wpad.view(9, mt, 16, ks, 4).permute(0, 1, 3, 4, 2).contiguous().view(-1)


def pack_dw_table(w9: torch.Tensor, bias, K: int, gate: bool) -> torch.Tensor:
    """Depth-wise coefficients in irm_dwgemm_f32's per-stage order: w9 [K or 2K][9] (+ bias) ->
    [4*ceil(K/4)][40 | 20] floats, every value twice (include/irm_hip.h)."""
    w9 = w9.detach().reshape(-1, 9).float()
    n = 20 if gate else 10
    rows = 4 * ((K + 3) // 4)
    t = torch.zeros(rows, n, dtype=torch.float32, device=w9.device)
    t[:K, 0:9] = w9[:K]
    if bias is not None:
        t[:K, 9] = bias[:K].float()
    if gate:
        t[:K, 10:19] = w9[K:2 * K]
        if bias is not None:
            t[:K, 19] = bias[K:2 * K].float()
    return t.repeat_interleave(2, dim=1).contiguous().view(-1)


def deconv_as_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d(k3,s1,p1) weight [Ci][Co][3][3] -> equivalent Conv2d weight [Co][Ci][3][3]
    (rednet.py:46-60): transpose the channel axes and flip both spatial axes."""
    return w.detach().transpose(0, 1).flip(-1, -2).contiguous()


def choose_ct(mtiles: int, options=(9, 8, 6, 4, 3)) -> int:
    """Output-channel tiles per pass: least padding, then the largest tile."""
    best = None
    for ct in options:
        waste = -(-mtiles // ct) * ct - mtiles
        if best is None or waste < best[0]:
            best = (waste, ct)
    return best[1]
