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
    "irm_ln_split_f16": [_P, _L, _P, _P, _I, _F, _F, _P, _I, _I, _I, _P],
    "irm_ln_gemm_presplit_f16x3_f32": [_P, _P, _L, _P, _P, _I, _F, _F, _P, _L, _P, _F, _I, _I, _I, _I, _I, _P],
    "irm_ln_gemm_presplit_cl_f16x3_f32": [_P, _P, _L, _P, _P, _I, _F, _F, _P, _L, _P, _F, _I, _I, _I, _I, _I, _I, _P],
    "irm_gdfn_tail_f16x3_f32": [_P, _L, _P, _P, _P, _P, _L, _F, _I, _I, _I, _I, _I, _I, _P],
    "irm_gemm_presplit_f16x3_f32": [_P, _P, _P, _L, _P, _F, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "irm_dwconv3x3_gate_split_f16": [_P, _L, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P],
    "irm_gemm_presplit_res_f16x3_f32": [_P, _P, _P, _L, _P, _L, _P, _F, _I, _I, _I, _I, _I, _P],
    "irm_dwconv3x3_f32": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P],
    "irm_dwconv3x3_gate_f32": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "irm_dwgemm_f32": [_P, _L, _P, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _P, _F, _P],
    "irm_dwgemm_f16x3_f32": [_P, _L, _P, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _P, _F, _P],
    "irm_gdfn_fused_f16x3_f32": [_P, _P, _P, _P, _L, _P, _L, _I, _F, _F, _F, _I, _I, _I, _I, _I, _P],
    "irm_qkv_dw_fused_f16x3_f32": [_P, _P, _L, _P, _L, _I, _F, _F, _I, _I, _I, _I, _I, _P],
    "irm_mdta_gram_f32": [_P, _L, _P, _I, _I, _I, _I, _I, _P],
    "irm_mdta_gram_f16x3_f32": [_P, _L, _P, _P, _I, _I, _I, _I, _I, _P],
    "irm_mdta_finalize_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "irm_mdta_finalize_f16x3_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "irm_qkv_dw_fused_tm_f16x3_f32": [_P, _P, _L, _P, _L, _I, _F, _F, _I, _I, _I, _I, _I, _I, _P],
    "irm_qkv_gram_cm_f16x3_f32": [_P, _P, _L, _P, _L, _P, _P, _I, _F, _F, _I, _I, _I, _I, _I, _I, _P],
    "irm_qkv_dw_cm_f16x3_f32": [_P, _P, _L, _P, _L, _I, _F, _F, _I, _I, _I, _I, _I, _I, _P],
    "irm_mdta_gram_tm_f16x3_f32": [_P, _L, _P, _P, _I, _I, _I, _I, _I, _P],
    "irm_mdta_gram_tm_f32": [_P, _L, _P, _I, _I, _I, _I, _I, _P],
    "irm_mdta_finalize_frag_f16x3_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "irm_attn_gdfn_fused_f16x3_f32": [_P, _P, _P, _P, _L, _P, _L, _P, _P, _P, _L, _I, _F, _F, _F, _I, _I, _I, _I, _I, _I, _P],
    "irm_conv3x3_f32": [_P, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "irm_conv3x3_thin_f32": [_P, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "irm_conv3x3_f16x3_f32": [_P, _F, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
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
    global _lib, LIB_PATH
    if _lib is None:
        if os.environ.get("IRM_HIP_LIB"):                # diagnostic: an experiment build of the same ABI (tools/build_variant.sh)
            LIB_PATH = os.path.abspath(os.environ["IRM_HIP_LIB"])
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


def split_is_safe(w: torch.Tensor, lnw=None, lnb=None) -> bool:
    """Range guard of the UNSCALED fp16 hi/lo split used by irm_gemm1x1_f16x3_f32 / irm_dwgemm_f16x3_f32 (the fused
    branch kernels scale their operands by powers of two instead and need no guard).  False -> the caller keeps
    that layer on the exact f32-input MFMA entry point.
      * hi must stay finite with headroom: max|W| < 2^14;
      * small weights have SUBNORMAL lo parts (the matrix cores do not flush them): a weight is then known to
        2^-25 absolute instead of 2^-24 relative; max|W| >= 2^-6 keeps the largest weights at 2^-19 relative;
      * with a LayerNorm prologue the normalised activation must fit fp16: max|ln.weight| sqrt(K) + max|ln.bias| < 2^15."""
    w = w.detach().reshape(w.shape[0], -1).float()
    m = float(w.abs().max()) if w.numel() else 0.0
    if not (m == m) or m >= 2.0 ** 14 or m < 2.0 ** -6:
        return False
    if lnw is not None:
        bound = float(lnw.detach().abs().max()) * (w.shape[1] ** 0.5) + (float(lnb.detach().abs().max()) if lnb is not None else 0.0)
        if not (bound < 2.0 ** 15):
            return False
    return True


def _pow2_floor(v: float) -> float:
    import math
    return 2.0 ** math.floor(math.log2(v))


def pack_gemm_weight_presplit(w: torch.Tensor, k_pad: int | None = None):
    """W [M][K] (K % 32 == 0) -> (fragments, s_w) for irm_gemm_presplit_f16x3_f32: W s_w split into fp16 hi + lo with the
    power of two s_w chosen so that max|W| s_w lies in [2^13, 2^14) (the lo parts stay normal fp16 numbers for weights
    of any magnitude), in MFMA fragment order [mtile][k-step][hi|lo][lane = 16 g + m][e] = part[16 mtile + m][32 ks + 8 g + e]
    (rows beyond M zero; k_pad: zero columns up to that K, a multiple of 32).  Returned as a float32 view (two halves
    per element)."""
    w = w.detach().reshape(w.shape[0], -1).float()
    if k_pad is not None and k_pad > w.shape[1]:
        w = torch.cat([w, torch.zeros(w.shape[0], k_pad - w.shape[1], dtype=w.dtype, device=w.device)], dim=1)
    m, k = w.shape
    assert k % 32 == 0
    mt, ks = (m + 15) // 16, k // 32
    amax = float(w.abs().max()) if w.numel() else 0.0
    s_w = _pow2_floor(2.0 ** 14 / amax) if amax > 0 and amax == amax else 1.0
    if amax * s_w >= 2.0 ** 14:
        s_w *= 0.5
    wpad = torch.zeros(mt * 16, k, dtype=torch.float32, device=w.device)
    wpad[:m] = w * s_w
    hi = wpad.half()
    lo = (wpad - hi.float()).half()

    def arrange(t):                                   # [mt][16 m][ks][4 g][8 e] -> [mt][ks][g][m][e]
        return t.view(mt, 16, ks, 4, 8).permute(0, 2, 3, 1, 4)
    packed = torch.stack([arrange(hi), arrange(lo)], dim=2).contiguous()      # [mt][ks][2][g][m][e]
    return packed.view(-1).view(torch.float32), s_w


def ln_split_scale(lnw: torch.Tensor, lnb, K: int, with_bias: bool) -> float:
    """Power-of-two operand scale s_x of irm_ln_split_f16.  A WithBias LayerNorm output obeys |y_j| <= sqrt(K - 1) |w_j|
    + |b_j| for ANY input: s_x puts that bound just below 2^15, overflow is impossible.  A BiasFree LayerNorm
    (x / sqrt(var + eps) * w: the mean is not removed) has no such bound; s_x leaves 16x headroom over sqrt(K) max|w|
    and the kernel clamps at +-65000 (a lower scale costs nothing for ordinary values: lo parts stay normal fp16 down
    to |y| s_x = 2^-3)."""
    wmax = float(lnw.detach().abs().max())
    bmax = float(lnb.detach().abs().max()) if (with_bias and lnb is not None) else 0.0
    bound = wmax * (max(K - 1, 1) ** 0.5) + bmax
    if not (bound > 0.0) or bound != bound:
        return 1.0
    s = _pow2_floor(2.0 ** 15 / bound)
    if bound * s >= 2.0 ** 15:
        s *= 0.5
    return s if with_bias else s / 16.0


def plan_presplit(mtiles: int, npt: int, K: int, c0: float | None = None):
    """(ct, mgroups, wg_shape) of an irm_gemm_presplit_f16x3_f32 launch: ct output tiles per chunk, the chunks of a pixel
    block split over `mgroups` workgroups.  K 192: four waves x three pixel tiles, ct 4, two workgroups per CU (512
    slots; measured best of 42 / 32 / 43 on 6 x 128^2: 128 vs 145 us for M 1020); K 384: eight waves x one tile, one
    workgroup per CU.  cost = rounds of the slots x (chunks per workgroup x ct + c0), c0 = the part that does not shrink
    with the tile range (loading the resident operands)."""
    if K == 192:
        shape, per_wg, slots, cts = 43, 12, 512, (4,)
    else:
        shape, per_wg, slots, cts = 81, 8, 256, (8, 6)
    if c0 is None:
        c0 = 8.0 if K == 192 else 6.0      # (a workgroup's 24 KiB of resident operands per wave cost about that many tile-stages)
    nblk = -(-npt // per_wg)
    best = None
    for ct in cts:
        chunks = -(-mtiles // ct)
        for mg in range(1, chunks + 1):
            cpg = -(-chunks // mg)
            if (mg - 1) * cpg >= chunks:
                continue                                   # an empty group
            rounds = -(-(nblk * mg) // slots)
            cost = rounds * (cpg * ct + c0)
            key = (cost, mg, -ct)
            if best is None or key < best[0]:
                best = (key, ct, mg)
    return best[1], best[2], shape


def plan_gemm(mtiles: int, blocks: int, options=(9, 8, 6, 4, 3), slots: int = 512, c0: float = 4.0, res: bool = False):
    """(ct, ygroups) of a 1x1-conv launch: ct output-channel tiles per pass, the passes of a pixel tile split over
    `ygroups` workgroups.  Cost model in tile-pass units: rounds x passes per workgroup x (ct + c0), with
    rounds = ceil(blocks x ygroups / slots) (two ring-kernel workgroups of 57-73 KiB LDS are resident per CU: 512 slots),
    passes = ceil(ceil(mtiles / ct) / ygroups) and c0 the part of a pass that does not shrink with ct (every pass streams
    all K input rows of its pixel tile through the LDS ring).  What it fixes: 576 workgroups on 512 slots (M 2042 / 1152,
    K 384 at Restormer's 64 x 64 level) ran as one round plus a straggler round of 64 - nearly twice the time of 480."""
    best = None
    for ct in options:
        chunks = -(-mtiles // ct)
        # residual GEMMs with <= 6 tiles per pass run three workgroups per CU (gemm_pw.hip, round 3): 768 slots, each
        # a third slower while all three are resident
        sl, slow = (slots * 3 // 2, 1.5) if (res and ct <= 6) else (slots, 1.0)
        for yg in range(1, chunks + 1):
            passes = -(-chunks // yg)
            if (yg - 1) * passes >= chunks:
                continue                                   # an empty group
            rounds = -(-(blocks * yg) // sl)
            cost = rounds * passes * (ct + c0) * slow
            key = (cost, yg, -ct)
            if best is None or key < best[0]:
                best = (key, ct, yg)
    return best[1], best[2]


def param_key(module):
    """Per-forward fingerprint of a module's weights (keys the packed-weight caches and the HIP graphs).

    Cached on the module: the list of (owner._parameters dict, name, Parameter).  Every call checks that each slot
    still holds the SAME Parameter object (an assignment `m.conv.weight = nn.Parameter(..)` anywhere in the tree, or
    load_state_dict(assign=True), is seen and the list rebuilt) and folds every parameter's storage address and
    `_version` (in-place updates, load_state_dict, optimiser steps, `.to()` / `.float()`) into the key: ~0.1 ms for
    Restormer's 700 tensors.  NOT seen: writes through `p.data` (`.data` carries its own version counter) - after such
    a write call `_hip.invalidate(module)`."""
    d = module.__dict__
    slots = d.get("_irm_pslots")
    if slots is not None:
        for owner, name, p in slots:
            if owner.get(name) is not p:
                slots = None
                break
    if slots is None:
        slots = [(m._parameters, n, p) for m in module.modules() for n, p in m._parameters.items() if p is not None]
        d["_irm_pslots"] = slots
        d["_irm_pgen"] = d.get("_irm_pgen", 0) + 1
    if not slots:
        return (None, 0, 0, d.get("_irm_pgen", 0))
    ver, addr = 0, 0
    for i, (_, _, p) in enumerate(slots):
        ver += p._version
        addr ^= p.data_ptr() * (2 * i + 1)
    return (str(slots[0][2].device), addr, ver, d["_irm_pgen"])


def invalidate(module):
    """Drop every cache derived from the module's weights (packed operands, HIP graphs): required after writes that
    param_key cannot see (`p.data.copy_()`, raw pointer writes)."""
    d = module.__dict__
    d.pop("_irm_pslots", None)
    d["_irm_pgen"] = d.get("_irm_pgen", 0) + 1
    d.pop("_irm_graphs", None)


def gram_scales(qkv_w, qkv_b, dw_w, dw_b, lnw, lnb, ln_with_bias: bool):
    """Per-channel power-of-two operand scales [2C] of irm_mdta_gram_f16x3_f32 (q channels, then k channels), or None
    when no static bound exists.  The scale must make overflow IMPOSSIBLE for any input, so it comes from a bound:
    a WithBias LayerNorm output obeys |y_j| <= sqrt(C - 1) |w_j| + |b_j| (a normalised deviation of C samples cannot
    exceed sqrt(C - 1)), hence |qkv_c| <= sum_j |W_cj| (...) + |bias_c| and |dw(qkv)_c| <= sum_taps |t| * that + |dw bias_c|.
    scale_c = 2^floor(log2(2^14.8 / bound_c)).  Typical activations sit ~2^-8 below the bound: their hi part keeps 11
    bits and the lo part stays far above the fp16 subnormals.  A BiasFree LayerNorm (x / sigma, mean not removed) has no
    such bound: the caller keeps the f32-input MFMA Gram there."""
    if not ln_with_bias or lnw is None:
        return None
    w = qkv_w.detach().reshape(qkv_w.shape[0], -1).double()
    C = w.shape[1]
    a = (C - 1) ** 0.5 * lnw.detach().double().abs() + (lnb.detach().double().abs() if lnb is not None else 0.0)
    pre = w.abs() @ a + (qkv_b.detach().double().abs() if qkv_b is not None else 0.0)
    taps = dw_w.detach().reshape(dw_w.shape[0], -1).double().abs().sum(1)
    bound = (taps * pre + (dw_b.detach().double().abs() if dw_b is not None else 0.0))[:2 * C]
    if not bool(torch.isfinite(bound).all()):
        return None
    e = torch.floor(torch.log2(torch.tensor(2.0 ** 14.8, dtype=torch.float64) / bound.clamp_min(1e-300)))
    e = torch.where(bound > 0, e, torch.zeros_like(e)).clamp(-100, 100)
    return torch.pow(torch.tensor(2.0, dtype=torch.float64), e).float().contiguous()


def pack_conv3x3_weight(w: torch.Tensor) -> torch.Tensor:
    """W [Co][Ci][3][3] -> wp[9][ceil(Co/16)][2*ceil(Ci/8)][64]."""
    w = w.detach().float()
    co, ci = w.shape[:2]
    mt, ks = (co + 15) // 16, 2 * ((ci + 7) // 8)
    wpad = torch.zeros(9, mt * 16, ks * 4, dtype=torch.float32, device=w.device)
    wpad[:, :co, :ci] = w.reshape(co, ci, 9).permute(2, 0, 1)
    return wpad.view(9, mt, 16, ks, 4).permute(0, 1, 3, 4, 2).contiguous().view(-1)


def pack_conv3x3_weight_split(w: torch.Tensor):
    """W [Co][Ci][3][3] -> operands of irm_conv3x3_f16x3_f32: (wp, inv_scale) with
    wp[mtile][stage][tap][hi|lo][lane = 16 g + m][8 halves j] = part(W[16 mtile + m][32 stage + 8 g + j][tap] * s),
    s a power of two with max|W| s in [2^13, 2^14) (lo parts stay normal fp16 numbers), inv_scale = 16 / s (the kernel
    scales the activations by 2^-4 before their split)."""
    w = w.detach().float().cpu()
    co, ci = w.shape[:2]
    mt, st = (co + 15) // 16, (ci + 31) // 32
    s = _pow2_scale(w)
    wpad = torch.zeros(mt * 16, st * 32, 9, dtype=torch.float32)
    wpad[:co, :ci] = w.reshape(co, ci, 9) * s
    hi, lo = _split_h(wpad)

    def arr(t):       # [mt][16 m][st][4 g][8 j][9 tap] -> [mt][st][tap][g][m][j]
        return t.view(mt, 16, st, 4, 8, 9).permute(0, 2, 5, 3, 1, 4)
    packed = torch.stack([arr(hi), arr(lo)], dim=3).contiguous()          # [mt][st][tap][2][g][m][8]
    return packed.view(-1).view(torch.float32), 16.0 / s


class ConvWeight:
    """A 3x3 conv weight packed for both kernels: `exact` (irm_conv3x3_f32) always, `split` + `inv_scale`
    (irm_conv3x3_f16x3_f32) unless IRM_GEMM_EXACT is set; ops.conv3x3 picks the emulated kernel where its alignment
    requirements hold."""
    __slots__ = ("exact", "split", "inv_scale", "raw")

    def __init__(self, exact, split=None, inv_scale=1.0, raw=None):
        self.exact, self.split, self.inv_scale, self.raw = exact, split, inv_scale, raw


def pack_conv3x3(w: torch.Tensor) -> ConvWeight:
    exact = pack_conv3x3_weight(w)
    # convs with <= 4 channels on one side (image <-> features) run on the vector pipe, exact fp32, from the plain
    # weight (irm_conv3x3_thin_f32): they are memory streams, not matrix work
    raw = w.detach().float().contiguous() if min(w.shape[0], w.shape[1]) <= 4 else None
    if os.environ.get("IRM_GEMM_EXACT"):
        return ConvWeight(exact, raw=raw)
    split, inv = pack_conv3x3_weight_split(w)
    return ConvWeight(exact, split.to(w.device), inv, raw=raw)


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


def _pow2_scale(t: torch.Tensor, target_exp: int = 13) -> float:
    """Power of two s with max|t| * s in [2^target_exp, 2^(target_exp+1)): the fp16 hi part then keeps 11 bits and
    the lo part stays a NORMAL fp16 number for every element down to 2^-17 of the largest (range guard of the
    hi/lo split, any weight magnitude)."""
    m = float(t.abs().max())
    if not (m > 0.0) or m != m or m == float("inf"):
        return 1.0
    import math
    return 2.0 ** (target_exp - math.floor(math.log2(m)))


def _split_h(t32: torch.Tensor):
    hi = t32.half()
    lo = (t32 - hi.float()).half()
    return hi, lo


def pack_gdfn_fused(pin_w, pin_b, dw_w, dw_b, pout_w, lnw, lnb, gate_prescale: bool = True, kperm: bool = False):
    """Operands of irm_gdfn_fused_f16x3_f32 (include/irm_hip.h) from the reference's parameters
    (FeedForward.project_in / dwconv / project_out and norm2, restormer.py:76-93, 141):
    returns (rec, w2, inv_s1, inv_s2).  The LayerNorm weight is folded into project_in (W diag(w)) and the
    LayerNorm bias into its bias (W b) in float64, so the kernel normalises only.
    kperm: project_in's input channels in the order of irm_attn_gdfn_fused_f16x3_f32 (k-slot 32 ks + 8 g + j <-> channel
    16 (2 ks + (j >> 2)) + 4 g + (j & 3))."""
    dev = pin_w.device
    pin = pin_w.detach().reshape(pin_w.shape[0], -1).double().cpu()
    pout = pout_w.detach().reshape(pout_w.shape[0], -1).double().cpu()
    C, hid = pin.shape[1], pout.shape[1]
    assert pin.shape[0] == 2 * hid and pout.shape[0] == C
    S, KS, CT = (hid + 15) // 16, (C + 31) // 32, (C + 15) // 16
    if CT == 5:
        CT = 6
    SS = (S + 1) // 2
    w1 = pin * lnw.detach().double().cpu()[None, :]
    b1 = torch.zeros(2 * hid, dtype=torch.float64)
    if pin_b is not None:
        b1 += pin_b.detach().double().cpu()
    if lnb is not None:
        b1 += pin @ lnb.detach().double().cpu()
    # ---- project_in: [half][16 S][32 KS] zero padded, scaled, split
    s1 = _pow2_scale(w1)
    w1f = torch.zeros(2, 16 * S, 32 * KS, dtype=torch.float32)
    w1f[:, :hid, :C] = (w1.float() * s1).view(2, hid, C)
    if kperm:
        slot = torch.arange(32 * KS)
        ks_, g_, j_ = slot // 32, (slot % 32) // 8, slot % 8
        w1f = w1f[:, :, 16 * (2 * ks_ + (j_ >> 2)) + 4 * g_ + (j_ & 3)].contiguous()
    hi, lo = _split_h(w1f)

    def arr1(t):      # [hct][S][16 m][KS][4 g][8 j] -> [S][hct][KS][g][m][j]
        return t.view(2, S, 16, KS, 4, 8).permute(1, 0, 3, 4, 2, 5)
    w1pk = torch.stack([arr1(hi), arr1(lo)], dim=3).contiguous()          # [S][hct][KS][2][g][m][8]
    w1pk = w1pk.view(S, -1).view(torch.float32)                             # [S][KS*1024]
    # ---- taps + bias of the depth-wise conv, bias of project_in
    dwf = torch.zeros(2, 16 * S, 10, dtype=torch.float32)
    dwf[:, :hid, :9] = dw_w.detach().reshape(2 * hid, 9).float().cpu().view(2, hid, 9)
    if dw_b is not None:
        dwf[:, :hid, 9] = dw_b.detach().float().cpu().view(2, hid)
    # the gate multiplies gelu(dw(h1)) by dw(h2) / 16 (the fp16 split range of the gated activations): a power of
    # two on the taps and the bias of the second half scales every partial sum of its stencil exactly
    if gate_prescale:                       # (False: the operand layout of builds before round 3, tools/ab_fused.py)
        dwf[1] *= 0.0625
    coef = dwf.view(2, S, 16, 10).permute(1, 3, 0, 2).contiguous().view(S, 320)     # [S][t][hct*16+m]
    b1f = torch.zeros(2, 16 * S, dtype=torch.float32)
    b1f[:, :hid] = b1.float().view(2, hid)
    bias1 = b1f.view(2, S, 16).permute(1, 0, 2).contiguous().view(S, 32)
    rec = torch.zeros(S + 1, KS * 1024 + 512, dtype=torch.float32)
    rec[:S, :KS * 1024] = w1pk
    rec[1:, KS * 1024:KS * 1024 + 320] = coef
    rec[:S, KS * 1024 + 320:KS * 1024 + 352] = bias1
    # ---- project_out: k-slot (g, j) of super-stage T <-> gate channel 32 T + 16 (j >> 2) + 4 g + (j & 3)
    s2 = _pow2_scale(pout)
    w2f = torch.zeros(16 * CT, 32 * SS, dtype=torch.float32)
    w2f[:C, :hid] = pout.float() * s2
    h2, l2 = _split_h(w2f)

    def arr2(t):      # [CT][16 m][SS][2 jh][4 g][4 jl] -> [SS][CT][g][m][jh][jl]
        return t.view(CT, 16, SS, 2, 4, 4).permute(2, 0, 4, 1, 3, 5)
    w2pk = torch.stack([arr2(h2), arr2(l2)], dim=2).contiguous()           # [SS][CT][2][g][m][8]
    w2pk = w2pk.view(-1).view(torch.float32)
    return rec.view(-1).to(dev), w2pk.to(dev), 1.0 / (16.0 * s1), 16.0 / s2


def pack_pin_padded(pin_w, pin_b, hid_pad: int):
    """FeedForward.project_in for the tail path of the C = 192 level: the two halves of its 2 hid output channels moved to
    [0, hid) and [hid_pad, hid_pad + hid) of 2 hid_pad rows (the rest zero), so that both halves start 16-byte aligned in the
    channel-last h of irm_ln_gemm_presplit_cl_f16x3_f32.  Returns (fragments, s_w, bias [2 hid_pad] or None)."""
    w = pin_w.detach().reshape(pin_w.shape[0], -1).float()
    hid = w.shape[0] // 2
    assert hid_pad >= hid and hid_pad % 16 == 0
    wp = torch.zeros(2 * hid_pad, w.shape[1], dtype=torch.float32, device=w.device)
    wp[:hid] = w[:hid]
    wp[hid_pad:hid_pad + hid] = w[hid:]
    frag, s_w = pack_gemm_weight_presplit(wp)
    bp = None
    if pin_b is not None:
        bp = torch.zeros(2 * hid_pad, dtype=torch.float32, device=w.device)
        bp[:hid] = pin_b.detach().float()[:hid]
        bp[hid_pad:hid_pad + hid] = pin_b.detach().float()[hid:]
    return frag, s_w, bp


def pack_gdfn_tail(dw_w, dw_b, pout_w):
    """Operands of irm_gdfn_tail_f16x3_f32 (include/irm_hip.h) from FeedForward.dwconv / project_out (restormer.py:84-93):
    (rec [S][512], w2, inv_s2); the taps / bias of the multiplier half x 2^-4, project_out as in pack_gdfn_fused."""
    dev = pout_w.device
    pout = pout_w.detach().reshape(pout_w.shape[0], -1).double().cpu()
    C, hid = pout.shape
    S, CT = (hid + 15) // 16, (C + 15) // 16
    SS = (S + 1) // 2
    dwf = torch.zeros(2, 16 * S, 10, dtype=torch.float32)
    dwf[:, :hid, :9] = dw_w.detach().reshape(2 * hid, 9).float().cpu().view(2, hid, 9)
    if dw_b is not None:
        dwf[:, :hid, 9] = dw_b.detach().float().cpu().view(2, hid)
    dwf[1] *= 0.0625
    coef = dwf.view(2, S, 16, 10).permute(1, 3, 0, 2).contiguous().view(S, 320)     # [S][t][half * 16 + m]
    rec = torch.zeros(S, 512, dtype=torch.float32)
    rec[:, :320] = coef
    s2 = _pow2_scale(pout)
    w2f = torch.zeros(16 * CT, 32 * SS, dtype=torch.float32)
    w2f[:C, :hid] = pout.float() * s2
    h2, l2 = _split_h(w2f)

    def arr2(t):      # [CT][16 m][SS][2 jh][4 g][4 jl] -> [SS][CT][g][m][jh][jl]
        return t.view(CT, 16, SS, 2, 4, 4).permute(2, 0, 4, 1, 3, 5)
    w2pk = torch.stack([arr2(h2), arr2(l2)], dim=2).contiguous().view(-1).view(torch.float32)
    return rec.view(-1).to(dev), w2pk.to(dev), 16.0 / s2


def pack_mfold_frag(m: torch.Tensor) -> torch.Tensor:
    """[B][C][C] matrices -> the fragment order irm_mdta_finalize_frag_f16x3_f32 writes (tests, host-side callers):
    [B][2 KS][KS][hi|lo][lane = 16 g + r][8 halves], lane half e of fragment (t, ks) = M[16 t + r][32 ks + 8 g + e]."""
    B, C, _ = m.shape
    KS = (C + 31) // 32
    mp = torch.zeros(B, 32 * KS, 32 * KS, dtype=torch.float32)
    mp[:, :C, :C] = m.detach().float().cpu()
    hi, lo = _split_h(mp)

    def arr(t):       # [B][2KS t][16 r][KS][4 g][8 e] -> [B][t][KS][g][r][e]
        return t.view(B, 2 * KS, 16, KS, 4, 8).permute(0, 1, 3, 4, 2, 5)
    return torch.stack([arr(hi), arr(lo)], dim=3).contiguous().view(-1).view(torch.float32).to(m.device)


def unpack_mfold_frag(frag: torch.Tensor, B: int, C: int) -> torch.Tensor:
    """Inverse of pack_mfold_frag: hi + lo as float32 [B][C][C] (tests)."""
    KS = (C + 31) // 32
    h = frag.detach().cpu().view(torch.float16).view(B, 2 * KS, KS, 2, 4, 16, 8).float()
    full = (h[:, :, :, 0] + h[:, :, :, 1]).permute(0, 1, 4, 2, 3, 5).reshape(B, 32 * KS, 32 * KS)
    return full[:, :C, :C].contiguous()


def pack_qkv_fused(qkv_w, qkv_b, dw_w, dw_b, lnw, lnb):
    """Operands of irm_qkv_dw_fused_f16x3_f32 from Attention.qkv / qkv_dwconv and norm1 (restormer.py:105-106, 140):
    returns (rec, inv_s1).  Stage s covers output channels 32 s .. 32 s + 31; LayerNorm weight / bias folded as in
    pack_gdfn_fused."""
    dev = qkv_w.device
    w = qkv_w.detach().reshape(qkv_w.shape[0], -1).double().cpu()
    M, C = w.shape
    S, KS = (M + 31) // 32, (C + 31) // 32
    w1 = w * lnw.detach().double().cpu()[None, :]
    b1 = torch.zeros(M, dtype=torch.float64)
    if qkv_b is not None:
        b1 += qkv_b.detach().double().cpu()
    if lnb is not None:
        b1 += w @ lnb.detach().double().cpu()
    s1 = _pow2_scale(w1)
    w1f = torch.zeros(32 * S, 32 * KS, dtype=torch.float32)
    w1f[:M, :C] = w1.float() * s1
    hi, lo = _split_h(w1f)

    def arr1(t):      # [S][hct][16 m][KS][4 g][8 j] -> [S][hct][KS][g][m][j]
        return t.view(S, 2, 16, KS, 4, 8).permute(0, 1, 3, 4, 2, 5)
    w1pk = torch.stack([arr1(hi), arr1(lo)], dim=3).contiguous().view(S, -1).view(torch.float32)
    dwf = torch.zeros(32 * S, 10, dtype=torch.float32)
    dwf[:M, :9] = dw_w.detach().reshape(M, 9).float().cpu()
    if dw_b is not None:
        dwf[:M, 9] = dw_b.detach().float().cpu()
    coef = dwf.view(S, 32, 10).permute(0, 2, 1).contiguous().view(S, 320)
    b1f = torch.zeros(32 * S, dtype=torch.float32)
    b1f[:M] = b1.float()
    rec = torch.zeros(S + 1, KS * 1024 + 512, dtype=torch.float32)
    rec[:S, :KS * 1024] = w1pk
    rec[1:, KS * 1024:KS * 1024 + 320] = coef
    rec[:S, KS * 1024 + 320:KS * 1024 + 352] = b1f.view(S, 32)
    return rec.view(-1).to(dev), 1.0 / (16.0 * s1)


def deconv_as_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d(k3,s1,p1) weight [Ci][Co][3][3] -> equivalent Conv2d weight [Co][Ci][3][3]
    (rednet.py:46-60): transpose the channel axes and flip both spatial axes."""
    return w.detach().transpose(0, 1).flip(-1, -2).contiguous()


def choose_ct(mtiles: int, options=(9, 8, 6, 4, 3), blocks: int = 1 << 30, want: int = 256) -> int:
    """Output-channel tiles per pass: least padding, then the largest tile.  `blocks` = pixel tiles x images of the
    launch: when even one workgroup per (pixel tile, pass) would leave most of the 256 CUs idle (single small images:
    MaIRUNet's 32x32 / 64x64 levels), narrower passes that reach `want` workgroups win over padding."""
    best = None
    for ct in options:
        chunks = -(-mtiles // ct)
        waste = chunks * ct - mtiles
        short = max(0, want - blocks * chunks)
        key = (short, waste)
        if best is None or key < best[0]:
            best = (key, ct)
    return best[1]
