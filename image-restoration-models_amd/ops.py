"""Tensor-level wrappers of the C ABI (include/irm_hip.h).

Activations are float32 CUDA tensors shaped [B, C, H, W]; the channel and pixel
axes must be dense (stride(1) == H*W, stride(3) == 1) while the batch stride is
free, so channel slices of a larger buffer (``buf[:, a:b]``) are valid inputs and
outputs.  All calls are asynchronous on torch's current stream.
"""
from __future__ import annotations

import os

import torch

from . import _hip
from ._hip import ACT_GELU, ACT_NONE, ACT_RELU, ACT_RELU6, ACT_SILU, LN_BIASFREE, LN_NONE, LN_WITHBIAS  # noqa: F401


class KernelTimer:
    """Optional per-launch HIP-event timing (bench.py's roofline leg).

    While an instance is installed as ``ops.TIMER`` every wrapper below brackets
    its launch with two events on the launch stream (torch's current stream) and
    logs the call's algorithmic FLOPs and compulsory bytes (inputs read once,
    outputs written once, fp32).  ``summary()`` synchronises and aggregates."""

    def __init__(self, detail=False, only=None):
        self.records = []          # (kernel, start_event, end_event, flops, bytes)
        self.detail = detail       # key the summary by kernel + shape tag
        self.only = None if only is None else frozenset(only)   # time these kernel groups only (the others launch plainly)
        self._chain = {}           # stream -> end event of the previous timed launch on it
        self._done = {}            # totals of the records already folded (their events released)
        self._since = 0

    def _fold(self, k, e0, e1, fl, by):
        d = self._done.setdefault(k, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        d["launches"] += 1
        d["ms"] += e0.elapsed_time(e1)
        d["flops"] += fl
        d["bytes"] += by

    def drain(self):
        """Fold the records whose end event has completed (non-blocking query) and release their events: thousands of
        live timing events per step otherwise pile up in the HIP runtime until the run ends."""
        n = 0
        while n < len(self.records) and self.records[n][2].query():
            self._fold(*self.records[n])
            n += 1
        if n:
            del self.records[:n]

    def break_chain(self):
        """Call when work that is not timed here has been enqueued: the next launch records its own start."""
        self._chain.clear()

    def launch(self, kernel, fn, flops=0.0, nbytes=0.0, tag=""):
        # back-to-back launches on one stream share an event (end of one = start of the next): half the
        # event packets between the kernels, i.e. half of the timer's own cost in the timed region
        sid = torch.cuda.current_stream().cuda_stream
        if self.only is not None and kernel not in self.only:
            fn()
            self._chain.pop(sid, None)             # un-timed work in between: the next timed launch records its own start
            return
        e0 = self._chain.get(sid)
        if e0 is None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        fn()
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self._chain[sid] = e1
        self.records.append((kernel + (" " + tag if self.detail and tag else ""), e0, e1, float(flops), float(nbytes)))
        self._since += 1
        if self._since >= 256 and not os.environ.get("IRM_TIMER_KEEP_EVENTS"):
            self._since = 0
            self.drain()

    def summary(self):
        torch.cuda.synchronize()
        for rec in self.records:
            self._fold(*rec)
        self.records = []
        return {k: dict(v) for k, v in self._done.items()}


#: set to a KernelTimer to time launches; None = plain launches
TIMER: KernelTimer | None = None


def _launch(kernel, flops, nbytes, name, *args, tag=""):
    if TIMER is None:
        _hip.call(name, *args)
    else:
        TIMER.launch(kernel, lambda: _hip.call(name, *args), flops, nbytes, tag)


def _chk(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4):
        raise ValueError(f"{name}: expected a float32 CUDA tensor [B,C,H,W]")
    _, c, h, w = t.shape
    if t.stride(3) != 1 or t.stride(2) != w or (c > 1 and t.stride(1) != h * w):
        raise ValueError(f"{name}: channel/pixel axes must be dense (got strides {t.stride()})")
    return t


def _bs(t):
    return t.stride(0) if t is not None else 0


def target_blocks() -> int:
    """Workgroups wanted per launch: 4 per CU on the 256-CU MI355X."""
    return 1024


def ln_stats(x: torch.Tensor, stats: torch.Tensor, eps: float = 1e-5):
    """stats[b,0,n] = mean over channels, stats[b,1,n] = rstd (restormer.py:25-70)."""
    _chk(x, "x")
    B, C, H, W = x.shape
    assert stats.numel() >= B * 2 * H * W and stats.is_contiguous()
    N = H * W
    _launch("ln_stats", 5.0 * B * C * N, 4.0 * B * N * (C + 2), "irm_ln_stats_f32", _hip.ptr(x), _bs(x),
            _hip.ptr(stats), B, C, N, float(eps), tag=f"C{C} N{N} B{B}")


def gemm1x1(wp: torch.Tensor, x: torch.Tensor, y: torch.Tensor, M: int, K: int, *, res=None, bias=None,
            stats=None, lnw=None, lnb=None, ln_mode=LN_NONE, act=ACT_NONE, w_bs: int = 0, ct: int | None = None,
            ygroups: int | None = None, stats_out=None, eps: float = 1e-5, res_scale=None, split: bool = False):
    """y = act(W @ LN(x) + bias) (+ res); wp from _hip.pack_gemm_weight.
    stats_out: optional [B,2,N] buffer receiving the LayerNorm statistics of y (needs M <= 16*ct)."""
    _chk(x, "x"), _chk(y, "y")
    B, _, H, W = x.shape
    N = H * W
    assert x.shape[1] >= K and y.shape[1] >= M
    if res is not None:
        _chk(res, "res")
    mt = (M + 15) // 16
    if ct is None and ygroups is None and stats_out is None:
        # pixels per workgroup of the kernel that will run: 256 (emulated, no residual) or 128
        bn = 256 if (split and res is None) else 128
        ct, ygroups = _hip.plan_gemm(mt, -(-N // bn) * B, res=res is not None and K <= 512)
    if ct is None:
        ct = _hip.choose_ct(mt)
    if ygroups is None:
        nchunks = -(-mt // ct)
        blocks = -(-N // 128) * B
        ygroups = max(1, min(nchunks, -(-target_blocks() // blocks)))
    if stats_out is not None:
        assert mt <= ct, "fused output statistics need all output channels in one pass"
        ygroups = 1
    nbytes = 4.0 * B * N * (K + M + (M if res is not None else 0) + (2 if stats is not None else 0)
                            + (2 if stats_out is not None else 0))
    if split:
        # wp from _hip.pack_gemm_weight_split: fp32 emulation on the fp16 matrix cores (no residual)
        _launch("gemm1x1_f16x3", 2.0 * B * M * K * N, nbytes, "irm_gemm1x1_f16x3_f32", _hip.ptr(wp), int(w_bs), _hip.ptr(x), _bs(x),
                _hip.ptr(y), _bs(y), _hip.ptr(res), _bs(res), _hip.ptr(bias), _hip.ptr(stats), _hip.ptr(lnw),
                _hip.ptr(lnb), int(ln_mode), int(act), B, M, K, N, ct, ygroups, _hip.ptr(stats_out), float(eps),
                _hip.ptr(res_scale),
                tag=f"M{M} K{K} N{N} B{B} ln{int(ln_mode)} res{int(res is not None)} ct{ct} yg{ygroups}")
        return
    _launch("gemm1x1", 2.0 * B * M * K * N, nbytes, "irm_gemm1x1_f32", _hip.ptr(wp), int(w_bs), _hip.ptr(x), _bs(x),
            _hip.ptr(y), _bs(y), _hip.ptr(res), _bs(res), _hip.ptr(bias), _hip.ptr(stats), _hip.ptr(lnw),
            _hip.ptr(lnb), int(ln_mode), int(act), B, M, K, N, ct, ygroups, _hip.ptr(stats_out), float(eps),
            _hip.ptr(res_scale), tag=f"M{M} K{K} N{N} B{B} ln{int(ln_mode)} res{int(res is not None)} ct{ct} yg{ygroups}")


def can_presplit(K: int, N: int) -> bool:
    """Shapes of the pre-split LayerNorm + 1x1 conv path (gemm_ps.hip): the C >= 192 levels of Restormer."""
    return K in (192, 384) and N % 16 == 0


def ln_split(x: torch.Tensor, xs: torch.Tensor, lnw, lnb, ln_mode: int, scale: float, eps: float = 1e-5):
    """xs = fp16 hi/lo fragments of LN(x) * scale (irm_ln_split_f16); xs: flat float32 buffer of B*K*N elements."""
    _chk(x, "x")
    B, K, H, W = x.shape
    N = H * W
    assert xs.numel() >= B * K * N and xs.is_contiguous() and xs.dtype == torch.float32
    _launch("ln_split", 8.0 * B * K * N, 8.0 * B * K * N, "irm_ln_split_f16", _hip.ptr(x), _bs(x), _hip.ptr(lnw),
            _hip.ptr(lnb), int(ln_mode), float(scale), float(eps), _hip.ptr(xs), B, K, N, tag=f"K{K} N{N} B{B}")


def gemm_presplit(wps: torch.Tensor, xs: torch.Tensor, y: torch.Tensor, M: int, K: int, *, out_scale: float, bias=None,
                  ct: int | None = None, mgroups: int | None = None, wg_shape: int = 0):
    """y = (W xs) * out_scale + bias with xs from ln_split and wps from _hip.pack_gemm_weight_presplit
    (out_scale = 1 / (s_w s_x))."""
    _chk(y, "y")
    B, _, H, W = y.shape
    N = H * W
    assert y.shape[1] >= M and can_presplit(K, N) and xs.numel() >= B * K * N
    if ct is None or mgroups is None:
        ct, mgroups, wg_shape = _hip.plan_presplit((M + 15) // 16, B * N // 16, K)
    _launch("gemm_ps_f16x3", 2.0 * B * M * K * N, 4.0 * B * N * (K + M), "irm_gemm_presplit_f16x3_f32", _hip.ptr(wps),
            _hip.ptr(xs), _hip.ptr(y), _bs(y), _hip.ptr(bias), float(out_scale), int(ACT_NONE), B, M, K, N, int(ct),
            int(mgroups), int(wg_shape), tag=f"M{M} K{K} N{N} B{B} ct{ct} mg{mgroups}")


def ln_gemm_presplit(wps, x, y, M: int, K: int, lnw, lnb, ln_mode: int, x_scale: float, *, out_scale: float, bias=None,
                     xs=None, eps: float = 1e-5):
    """y = W LN(x) + bias on the pre-split path.  K = 192 and a plan with ONE workgroup per pixel block: one launch
    (irm_ln_gemm_presplit_f16x3_f32, LayerNorm + split inside the GEMM's operand load - the pair's arithmetic);
    otherwise ln_split into `xs` + gemm_presplit."""
    _chk(x, "x"), _chk(y, "y")
    B, _, H, W = x.shape
    N = H * W
    ct, mgroups, shape = _hip.plan_presplit((M + 15) // 16, B * N // 16, K)
    if K == 192 and mgroups == 1 and not os.environ.get("IRM_NO_LN_FUSE"):
        _launch("gemm_ps_f16x3", 2.0 * B * M * K * N, 4.0 * B * N * (K + M), "irm_ln_gemm_presplit_f16x3_f32", _hip.ptr(wps),
                _hip.ptr(x), _bs(x), _hip.ptr(lnw), _hip.ptr(lnb), int(ln_mode), float(x_scale), float(eps), _hip.ptr(y), _bs(y),
                _hip.ptr(bias), float(out_scale), B, M, K, N, 1, tag=f"M{M} K{K} N{N} B{B} ln-fused")
        return
    ln_split(x, xs, lnw, lnb, ln_mode, x_scale, eps)
    gemm_presplit(wps, xs, y, M, K, out_scale=out_scale, bias=bias, ct=ct, mgroups=mgroups, wg_shape=shape)


def can_gdfn_tail(C: int, H: int, W: int) -> bool:
    """GDFN tail in one kernel (irm_gdfn_tail_f16x3_f32): the C = 192 level on whole 8 x 32 tiles."""
    return C == 192 and H % 8 == 0 and W % 32 == 0


def ln_gemm_presplit_cl(wps, x, h_cl, M: int, K: int, lnw, lnb, ln_mode: int, x_scale: float, *, out_scale: float, bias=None,
                        eps: float = 1e-5):
    """h_cl = W LN(x) + bias written tile-major channel-last in 64-channel chunks [tile][M / 64][256][64] (flat float32 buffer
    of B * M * H * W elements): irm_ln_gemm_presplit_cl_f16x3_f32, K = 192, M % 64 == 0."""
    _chk(x, "x")
    B, _, H, W = x.shape
    N = H * W
    assert K == 192 and M % 64 == 0 and H % 8 == 0 and W % 32 == 0 and h_cl.numel() >= B * M * N and h_cl.is_contiguous()
    _launch("gemm_ps_f16x3", 2.0 * B * M * K * N, 4.0 * B * N * (K + M), "irm_ln_gemm_presplit_cl_f16x3_f32", _hip.ptr(wps),
            _hip.ptr(x), _bs(x), _hip.ptr(lnw), _hip.ptr(lnb), int(ln_mode), float(x_scale), float(eps), _hip.ptr(h_cl), M * N,
            _hip.ptr(bias), float(out_scale), B, M, K, H, W, 1, tag=f"M{M} K{K} N{N} B{B} ln-fused cl")


def gdfn_tail(pk, h_cl, x, C: int, hid: int, hid_pad: int, *, bias=None):
    """x += project_out(gelu(dw(h)[:hid]) * dw(h)[hid:]) + bias in one kernel, h_cl from ln_gemm_presplit_cl (2 hid_pad channels
    per pixel); pk = _hip.pack_gdfn_tail(...)."""
    _chk(x, "x")
    B, _, H, W = x.shape
    N = H * W
    rec, w2, inv_s2 = pk
    assert can_gdfn_tail(C, H, W) and x.shape[1] >= C and h_cl.numel() >= B * 2 * hid_pad * N
    _launch("gdfn_tail", B * N * (36.0 * hid + 2.0 * hid * C), 4.0 * B * N * (2 * hid_pad + 2 * C), "irm_gdfn_tail_f16x3_f32",
            _hip.ptr(h_cl), 2 * hid_pad * N, _hip.ptr(rec), _hip.ptr(w2), _hip.ptr(bias), _hip.ptr(x), _bs(x), float(inv_s2),
            B, C, hid, hid_pad, H, W, tag=f"C{C} hid{hid} {H}x{W} B{B}")


GATE_SPLIT_SCALE = 0.0625      # 2^-4: gated activations up to ~1e6 stay inside fp16 (as irm_gemm1x1_f16x3_f32 without LN)


def can_gate_split(M: int, hid: int, W: int, N: int) -> bool:
    """GDFN tail on pre-split operands (gemm_ps.hip): gate -> fragments -> K-streamed GEMM; the C >= 192 levels."""
    return 96 < M <= 384 and W % 16 == 0 and N % 16 == 0 and (-(-hid // 32)) % 4 == 0


def dwconv3x3_gate_split(x, w9, gs, *, bias=None, scale: float = GATE_SPLIT_SCALE, ch: int = 0):
    """gs = fp16 hi/lo fragments of gelu(dw(x[:, :hid])) * dw(x[:, hid:]) * scale (irm_dwconv3x3_gate_split_f16);
    gs: flat float32 buffer of B * 32 ceil(hid/32) * H * W elements."""
    _chk(x, "x")
    B, C2, H, W = x.shape
    hid = C2 // 2
    kp = 32 * -(-hid // 32)
    assert gs.numel() >= B * kp * H * W and gs.is_contiguous() and gs.dtype == torch.float32
    _launch("dwconv3x3_gate_split", 18.0 * B * C2 * H * W, 4.0 * B * (C2 + kp) * H * W, "irm_dwconv3x3_gate_split_f16",
            _hip.ptr(x), _bs(x), _hip.ptr(w9), _hip.ptr(bias), _hip.ptr(gs), float(scale), B, hid, H, W, int(ch),
            tag=f"hid{hid} {H}x{W} B{B}")


def gemm_presplit_res(wps, xs, y, M: int, KS: int, *, out_scale: float, res=None, bias=None, wg_shape: int = 0):
    """y = res + bias + (W xs) * out_scale, K = 32 KS streamed (irm_gemm_presplit_res_f16x3_f32); y may be res."""
    _chk(y, "y")
    B, _, H, W = y.shape
    N = H * W
    assert y.shape[1] >= M and xs.numel() >= B * 32 * KS * N
    if res is not None:
        _chk(res, "res")
    _launch("gemm_ps_res_f16x3", 2.0 * B * M * 32 * KS * N, 4.0 * B * N * (32 * KS + M + (M if res is not None else 0)),
            "irm_gemm_presplit_res_f16x3_f32", _hip.ptr(wps), _hip.ptr(xs), _hip.ptr(y), _bs(y), _hip.ptr(res), _bs(res),
            _hip.ptr(bias), float(out_scale), B, M, KS, N, int(wg_shape), tag=f"M{M} K{32 * KS} N{N} B{B}")


def dwconv3x3(x, w9, y, *, bias=None, act=ACT_NONE):
    """Depth-wise 3x3 (+bias, +activation); w9: [C, 9]."""
    _chk(x, "x"), _chk(y, "y")
    B, C, H, W = x.shape
    _launch("dwconv3x3", 18.0 * B * C * H * W, 8.0 * B * C * H * W, "irm_dwconv3x3_f32", _hip.ptr(x), _bs(x),
            _hip.ptr(w9), _hip.ptr(bias), _hip.ptr(y), _bs(y), B, C, H, W, int(act), tag=f"C{C} {H}x{W} B{B}")


def dwconv3x3_gate(x, w9, y, *, bias=None):
    """GDFN: y[:, c] = gelu(dw(x[:, c])) * dw(x[:, c + hid]); x has 2*hid channels."""
    _chk(x, "x"), _chk(y, "y")
    B, C2, H, W = x.shape
    _launch("dwconv3x3_gate", 18.0 * B * C2 * H * W, 4.0 * B * (C2 + C2 // 2) * H * W, "irm_dwconv3x3_gate_f32",
            _hip.ptr(x), _bs(x), _hip.ptr(w9), _hip.ptr(bias), _hip.ptr(y), _bs(y), B, C2 // 2, H, W,
            tag=f"hid{C2 // 2} {H}x{W} B{B}")


def can_fuse_dw(M: int, W: int) -> bool:
    """irm_dwgemm_f32 keeps every output channel of a pixel tile in one workgroup."""
    return M <= 96 and W % 4 == 0


def dwgemm(wp, dwp, x, y, M: int, K: int, *, gate: bool, res=None, bias=None, w_bs: int = 0, stats_out=None,
           eps: float = 1e-5, split: bool = False):
    """y = W @ g + bias (+ res), g = gelu(dw(x[:, :K])) * dw(x[:, K:2K]) (gate) or dw(x[:, :K]);
    dwp from _hip.pack_dw_table."""
    _chk(x, "x"), _chk(y, "y")
    B, Cx, H, W = x.shape
    assert Cx >= (2 * K if gate else K) and y.shape[1] >= M
    if res is not None:
        _chk(res, "res")
    N = H * W
    nbytes = 4.0 * B * N * ((2 * K if gate else K) + M + (M if res is not None else 0)
                            + (2 if stats_out is not None else 0))
    flops = B * N * (2.0 * M * K + (36.0 if gate else 18.0) * K)
    if split:
        # wp from _hip.pack_gemm_weight_split: the 1x1 part as an fp32 emulation on the fp16 matrix cores
        _launch("dwgemm_f16x3", flops, nbytes, "irm_dwgemm_f16x3_f32", _hip.ptr(wp), int(w_bs), _hip.ptr(dwp), _hip.ptr(x), _bs(x),
                _hip.ptr(y), _bs(y), _hip.ptr(res), _bs(res), _hip.ptr(bias), int(bool(gate)), B, M, K, H, W,
                _hip.ptr(stats_out), float(eps), tag=f"M{M} K{K} {H}x{W} B{B} gate{int(bool(gate))}")
        return
    _launch("dwgemm", flops, nbytes, "irm_dwgemm_f32", _hip.ptr(wp), int(w_bs), _hip.ptr(dwp), _hip.ptr(x), _bs(x),
            _hip.ptr(y), _bs(y), _hip.ptr(res), _bs(res), _hip.ptr(bias), int(bool(gate)), B, M, K, H, W,
            _hip.ptr(stats_out), float(eps), tag=f"M{M} K{K} {H}x{W} B{B} gate{int(bool(gate))}")


def can_fuse_gdfn(C: int, W: int) -> bool:
    """irm_gdfn_fused_f16x3_f32 keeps the input tile of all C channels in registers."""
    return C <= 96 and C % 16 == 0 and W % 4 == 0


def gdfn_fused(pk, x, y, C: int, hid: int, *, ln_mode, bias=None, eps: float = 1e-5):
    """y = x + project_out(gelu(dw(h)[:hid]) * dw(h)[hid:]), h = project_in(LN(x)) in one kernel (y is not x);
    pk = _hip.pack_gdfn_fused(...)."""
    _chk(x, "x"), _chk(y, "y")
    B, _, H, W = x.shape
    assert x.data_ptr() != y.data_ptr() and x.shape[1] >= C and y.shape[1] >= C
    rec, w2, inv_s1, inv_s2 = pk
    N = H * W
    flops = B * N * (2.0 * 2 * hid * C + 36.0 * hid + 2.0 * hid * C)
    _launch("gdfn_fused", flops, 4.0 * B * N * 2 * C, "irm_gdfn_fused_f16x3_f32", _hip.ptr(rec), _hip.ptr(w2),
            _hip.ptr(bias), _hip.ptr(x), _bs(x), _hip.ptr(y), _bs(y), int(ln_mode), float(eps), float(inv_s1),
            float(inv_s2), B, C, hid, H, W, tag=f"C{C} hid{hid} {H}x{W} B{B}")


def mfold_frag_numel(C: int) -> int:
    """Floats per image of the folded attention matrix in MFMA fragment order (irm_mdta_finalize_frag_f16x3_f32)."""
    ks = (C + 31) // 32
    return 2 * ks * ks * 512


def attn_gdfn_fused(pk, x, v, mfold_frag, y, C: int, hid: int, *, ln_mode, bias_o=None, bias=None, eps: float = 1e-5,
                    x_tm: bool = False, v_tm: bool = False, y_tm: bool = False):
    """y = x' + GDFN(x'), x' = x + bias_o + Mfold[b] v in one kernel (restormer.py:131, 147-148; y is neither x nor v);
    pk = _hip.pack_gdfn_fused(..., kperm=True), mfold_frag from mdta_fold(..., frag=True).
    x_tm / v_tm / y_tm: that tensor in the tile-major layout of include/irm_hip.h (whole 8 x 32 tiles)."""
    _chk(x, "x"), _chk(y, "y"), _chk(v, "v")
    B, _, H, W = x.shape
    assert x.data_ptr() != y.data_ptr() and x.shape[1] >= C and y.shape[1] >= C and v.shape[1] >= C and C % 16 == 0
    assert v.shape[0] == B and v.shape[2:] == x.shape[2:] and mfold_frag.numel() >= B * mfold_frag_numel(C)
    rec, w2, inv_s1, inv_s2 = pk
    N = H * W
    flops = B * N * (2.0 * C * C + 2.0 * 2 * hid * C + 36.0 * hid + 2.0 * hid * C)
    _launch("attn_gdfn_fused", flops, 4.0 * B * N * 3 * C, "irm_attn_gdfn_fused_f16x3_f32", _hip.ptr(rec), _hip.ptr(w2),
            _hip.ptr(bias), _hip.ptr(x), _bs(x), _hip.ptr(v), _bs(v), _hip.ptr(mfold_frag), _hip.ptr(bias_o),
            _hip.ptr(y), _bs(y), int(ln_mode), float(eps), float(inv_s1), float(inv_s2), B, C, hid, H, W,
            int(x_tm) | 2 * int(v_tm) | 4 * int(y_tm), tag=f"C{C} hid{hid} {H}x{W} B{B}")


def can_qk_tile_major(C: int, heads: int, H: int, W: int) -> bool:
    """q, k tile-major (qkv_dw_fused(tm=True) -> mdta_fold(tm=True)): whole 8 x 32 tiles and an LDS-DMA ring Gram pass."""
    return C % 16 == 0 and C % heads == 0 and C // heads in (48, 96) and H % 8 == 0 and W % 32 == 0


def _use_qkv_cm(C: int) -> bool:
    """The channel-major qkv kernel (fused_qkv_cm.hip, bit-identical results): default where it is faster in the model - C <= 64
    (C = 48 at 512^2: 1.71 vs 1.85 ms per 24 tiles; its registers leave room to request the next input three iterations
    early); at C = 96 it measures 3.12 vs 2.96 ms in the model (1.58 vs 1.66 ms in isolation).  IRM_QKV_CM=1 / 0 forces / forbids it."""
    e = os.environ.get("IRM_QKV_CM")
    return e == "1" if e in ("0", "1") else C <= 64


def qkv_dw_fused(pk, x, y, C: int, M: int, *, ln_mode, eps: float = 1e-5, tm: bool = False, x_tm: bool = False,
                 v_tm: bool = False):
    """y[:, :M] = dw3x3(W @ LN(x) + b) in one kernel (y is not x); pk = _hip.pack_qkv_fused(...).
    tm (M = 3C): q, k tile-major inside y[:, :2C] (include/irm_hip.h), for mdta_fold(tm=True) only."""
    _chk(x, "x"), _chk(y, "y")
    B, _, H, W = x.shape
    assert x.data_ptr() != y.data_ptr() and x.shape[1] >= C and y.shape[1] >= M
    rec, inv_s1 = pk
    N = H * W
    if tm:
        assert M == 3 * C and y.shape[1] == M and H % 8 == 0 and W % 32 == 0
        _launch("qkv_dw_fused", B * N * (2.0 * M * C + 18.0 * M), 4.0 * B * N * (C + M),
                "irm_qkv_dw_cm_f16x3_f32" if _use_qkv_cm(C) else "irm_qkv_dw_fused_tm_f16x3_f32",
                _hip.ptr(rec), _hip.ptr(x), _bs(x), _hip.ptr(y), _bs(y), int(ln_mode), float(eps), float(inv_s1), B, C,
                H, W, int(x_tm), int(v_tm), tag=f"C{C} M{M} {H}x{W} B{B} tm")
        return
    assert not (x_tm or v_tm), "tile-major x / v need tm=True"
    _launch("qkv_dw_fused", B * N * (2.0 * M * C + 18.0 * M), 4.0 * B * N * (C + M), "irm_qkv_dw_fused_f16x3_f32",
            _hip.ptr(rec), _hip.ptr(x), _bs(x), _hip.ptr(y), _bs(y), int(ln_mode), float(eps), float(inv_s1), B, C, M,
            H, W, tag=f"C{C} M{M} {H}x{W} B{B}")


def mdta_plan(B: int, C: int, heads: int, N: int):
    """(chunk, nchunk, record size) of the Gram pass for this problem size."""
    c = C // heads
    sb = 3 if c % 48 == 0 else 2 if c % 32 == 0 else 1
    rec = c * c + 2 * c
    if c in (48, 96) and N % 64 == 0 and not os.environ.get("IRM_GRAM_BLOCKS"):
        # the LDS-DMA ring kernels keep the whole c x c Gram in one workgroup of 72 KiB LDS: 2 per CU = 512 resident.
        # Whole rounds matter: 1026 workgroups (the old ceil(N / 1536) = 171 chunks x 6 images) are two full rounds
        # plus a third with 2 workgroups - half a round of the chip idle.  Cost model: rounds x (chunk + a fixed
        # per-workgroup part worth ~192 pixels: ring fill, record write-out, the reduction over one more record).
        slots, best = 2 * 256, None
        for chunk in range(128, 4096 + 1, 64):
            nchunk = -(-N // chunk)
            cost = -(-(B * heads * nchunk) // slots) * (chunk + 192)
            if best is None or cost < best[0]:
                best = (cost, chunk, nchunk)
        return best[1], best[2], rec
    nsub = (c // (16 * sb)) ** 2
    tb = int(os.environ.get("IRM_GRAM_BLOCKS", 0)) or target_blocks()
    per_wg = 1 if (c in (48, 96) and N % 64 == 0) else 4                        # units per workgroup
    chunk = -(-(N * B * heads * (1 if per_wg == 1 else nsub)) // (per_wg * tb))
    chunk = min(max(-(-chunk // 64) * 64, 256), 4096)
    return chunk, -(-N // chunk), rec


QKV_GRAM_NCH = 4          # tiles per partial Gram record of the Gram-fused qkv kernel (QC_NCH in fused_qkv_cm.hip)


def can_qkv_gram(C: int, heads: int, H: int, W: int) -> bool:
    """qkv_gram_cm: the Gram partials inside the qkv kernel (q, k never written): C = 48 with one head, whole tiles in
    chunks of QKV_GRAM_NCH."""
    return C == 48 and heads == 1 and H % 8 == 0 and W % 32 == 0 and ((H // 8) * (W // 32)) % QKV_GRAM_NCH == 0


def qkv_gram_cm(pk, x, y, gram_scale, part, C: int, *, ln_mode, eps: float = 1e-5, x_tm: bool = False, v_tm: bool = False) -> int:
    """v = y[:, 2C:] (planar or, v_tm, tile-major channel-last) and the Gram partial records of q, k into `part`
    (irm_qkv_gram_cm_f16x3_f32); returns nchunk, the records per image, for mdta_fold(..., nchunk_ready=nchunk)."""
    _chk(x, "x"), _chk(y, "y")
    B, _, H, W = x.shape
    assert can_qkv_gram(C, 1, H, W) and y.shape[1] == 3 * C and gram_scale.numel() == 2 * C
    rec, inv_s1 = pk
    N = H * W
    nchunk = (H // 8) * (W // 32) // QKV_GRAM_NCH
    assert part.numel() >= B * nchunk * (C * C + 2 * C)
    _launch("qkv_dw_fused", B * N * (2.0 * 3 * C * C + 18.0 * 3 * C + 2.0 * C * C), 4.0 * B * N * 2 * C, "irm_qkv_gram_cm_f16x3_f32",
            _hip.ptr(rec), _hip.ptr(x), _bs(x), _hip.ptr(y), _bs(y), _hip.ptr(gram_scale), _hip.ptr(part), int(ln_mode),
            float(eps), float(inv_s1), B, C, H, W, int(x_tm), int(v_tm), tag=f"C{C} {H}x{W} B{B} gram")
    return nchunk


def mdta_fold(qkv, part, gsum, temperature, wout, mfold, C: int, heads: int, attn=None, split: bool = False,
              gram_scale=None, frag: bool = False, tm: bool = False, nchunk_ready: int | None = None):
    """Gram pass + finalize: mfold[b] <- packed(W_out @ blockdiag(softmax(...))) (restormer.py:115-131);
    split: in the fp16 hi/lo order of the emulated GEMM kernels.  gram_scale (_hip.gram_scales): the Gram pass runs
    as an fp32 emulation on the fp16 matrix cores (c = 48 / 96 channels per head, N % 64 == 0).  frag: mfold as fp16
    hi/lo MFMA fragments for attn_gdfn_fused (mfold_frag_numel(C) floats per image, zero-initialised once)."""
    _chk(qkv, "qkv")
    B, _, H, W = qkv.shape
    N = H * W
    chunk, nchunk, rec = mdta_plan(B, C, heads, N)
    if nchunk_ready is not None:
        nchunk = nchunk_ready                      # the partial records are there already (qkv_gram_cm)
    assert part.numel() >= B * heads * nchunk * rec and gsum.numel() >= B * heads * rec
    c = C // heads
    assert not tm or (c in (48, 96) and N % 256 == 0), "tile-major q, k: the LDS-DMA ring passes only"
    if nchunk_ready is not None:
        pass
    elif gram_scale is not None and c in (48, 96) and N % 64 == 0 and not os.environ.get("IRM_GRAM_EXACT"):
        assert gram_scale.numel() == 2 * C and gram_scale.is_contiguous()
        _launch("mdta_gram_f16x3", 2.0 * B * heads * c * c * N, 8.0 * B * C * N,
                "irm_mdta_gram_tm_f16x3_f32" if tm else "irm_mdta_gram_f16x3_f32", _hip.ptr(qkv),
                _bs(qkv), _hip.ptr(gram_scale), _hip.ptr(part), B, C, heads, N, chunk,
                tag=f"C{C} h{heads} N{N} B{B} chunk{chunk}")
    else:
        _launch("mdta_gram", 2.0 * B * heads * c * c * N, 8.0 * B * C * N, "irm_mdta_gram_tm_f32" if tm else "irm_mdta_gram_f32",
                _hip.ptr(qkv), _bs(qkv), _hip.ptr(part), B, C, heads, N, chunk, tag=f"C{C} h{heads} N{N} B{B} chunk{chunk}")
    _launch("mdta_finalize", 2.0 * B * C * C * c, 4.0 * B * (heads * nchunk * rec + C * C),
            "irm_mdta_finalize_frag_f16x3_f32" if frag else "irm_mdta_finalize_f16x3_f32" if split else "irm_mdta_finalize_f32",
            _hip.ptr(part), _hip.ptr(gsum), _hip.ptr(temperature), _hip.ptr(wout), _hip.ptr(mfold), _hip.ptr(attn),
            B, C, heads, nchunk, tag=f"C{C} h{heads} nchunk{nchunk} B{B}")


def can_fuse_stats(M: int) -> bool:
    """True if a GEMM with M output channels can emit the next LayerNorm's statistics (single pass)."""
    mt = (M + 15) // 16
    return mt <= _hip.choose_ct(mt)


def mfold_numel(C: int) -> int:
    mt = (C + 15) // 16
    return mt * 4 * mt * 64


def conv3x3(wp, x, y, ci: int, co: int, *, bias=None, relu1=False, res=None, res_mode=0, relu2=False,
            store_mode=0, ct: int | None = None, ygroups: int | None = None):
    """Dense 3x3 conv with fused epilogue; store_mode 1 = PixelUnshuffle(2), 2 = PixelShuffle(2).
    wp: _hip.pack_conv3x3_weight(w) (exact f32 MFMA) or the pair _hip.pack_conv3x3_weight_split(w) (fp32 emulated on
    the fp16 matrix cores, irm_conv3x3_f16x3_f32; needs W % 4 == 0 and 16-byte aligned rows)."""
    _chk(x, "x"), _chk(y, "y")
    B, _, H, W = x.shape
    mt = (co + 15) // 16
    blocks = -(-W // 32) * -(-H // 8) * B
    nbytes = 4.0 * B * H * W * (ci + co + (co if res is not None else 0))
    if isinstance(wp, _hip.ConvWeight):
        aligned = (W % 4 == 0 and _bs(x) % 4 == 0 and _bs(y) % 4 == 0 and _bs(res) % 4 == 0 and x.data_ptr() % 16 == 0
                   and y.data_ptr() % 16 == 0 and (res is None or res.data_ptr() % 16 == 0))
        if wp.raw is not None and aligned and store_mode == 0 and not os.environ.get("IRM_NO_THIN_CONV"):
            _launch("conv3x3_thin", 18.0 * B * ci * co * H * W, nbytes, "irm_conv3x3_thin_f32", _hip.ptr(wp.raw), _hip.ptr(x),
                    _bs(x), _hip.ptr(y), _bs(y), _hip.ptr(res), _bs(res), _hip.ptr(bias), B, ci, co, H, W, int(relu1),
                    int(res_mode), int(relu2), tag=f"ci{ci} co{co} {H}x{W} B{B}")
            return
        wp = (wp.split, wp.inv_scale) if (wp.split is not None and aligned) else wp.exact
    if isinstance(wp, tuple):
        wps, inv_scale = wp
        if ct is None:
            # output tiles per pass: 12 / 8 = 3 / 2 weight chunks of 4 per fetched + converted input tile (Co >= 128: the
            # up-sampling convs), else one chunk of <= 4
            ct = 12 if mt >= 12 else 8 if mt >= 8 else 4 if mt % 4 == 0 else 3 if mt % 3 == 0 or mt > 4 else min(mt, 4)
            if os.environ.get("IRM_CONV_ONE_CHUNK"):
                ct = 4 if mt % 4 == 0 or mt > 9 else 3 if mt % 3 == 0 or mt > 4 else min(mt, 4)
            while ct > 1 and blocks * -(-mt // ct) < 256:       # small images: more passes, more workgroups
                ct = {12: 8, 8: 4}.get(ct, ct - 1)
        if ygroups is None:
            nchunks = -(-mt // ct)
            ygroups = max(1, min(nchunks, -(-512 // blocks)))
            while nchunks % ygroups:                            # equal numbers of passes per workgroup
                ygroups += 1
        _launch("conv3x3_f16x3", 18.0 * B * ci * co * H * W, nbytes, "irm_conv3x3_f16x3_f32", _hip.ptr(wps), float(inv_scale),
                _hip.ptr(x), _bs(x), _hip.ptr(y), _bs(y), _hip.ptr(res), _bs(res), _hip.ptr(bias), B, ci, co, H, W,
                int(relu1), int(res_mode), int(relu2), int(store_mode), ct, ygroups,
                tag=f"ci{ci} co{co} {H}x{W} B{B} ct{ct} yg{ygroups} st{store_mode}")
        return
    if ct is None:
        ct = _hip.choose_ct(mt, (6, 4, 3, 2, 1))
        # small images (FPN levels, level-4 tiles): fewer output tiles per workgroup so that the chip is filled
        for smaller in (4, 3, 2, 1):
            if blocks * -(-mt // ct) >= 256 or smaller >= ct:
                continue
            ct = smaller
    if ygroups is None:
        nchunks = -(-mt // ct)
        ygroups = max(1, min(nchunks, -(-target_blocks() // blocks)))
    _launch("conv3x3", 18.0 * B * ci * co * H * W, nbytes, "irm_conv3x3_f32", _hip.ptr(wp), _hip.ptr(x), _bs(x),
            _hip.ptr(y), _bs(y), _hip.ptr(res), _bs(res), _hip.ptr(bias), B, ci, co, H, W, int(relu1), int(res_mode),
            int(relu2), int(store_mode), ct, ygroups, tag=f"ci{ci} co{co} {H}x{W} B{B} ct{ct} yg{ygroups} st{store_mode}")


# --------------------------------------------------------------------------- MaIR / LoSh2D
def transpose(src: torch.Tensor, dst: torch.Tensor, R: int, C: int):
    """src [B][R][C] (batch stride free) -> dst [B][C][R]."""
    B = src.shape[0]
    _launch("transpose", 0.0, 8.0 * B * R * C, "irm_transpose_f32", _hip.ptr(src), src.stride(0), _hip.ptr(dst),
            dst.stride(0), B, R, C, tag=f"R{R} C{C} B{B}")


def scan_plan(B: int, L: int, D: int):
    """(chunk, nchunk, DB): time steps per wave.  A wave is one sequential recurrence; measured on the five MaIRUNet
    shapes of a 256x256 image (tools/bench_scan.py) the scan is fastest with ~3 waves per SIMD (3k waves: all
    resident at once, none queued behind a full SIMD) and chunks of at least 32 steps (whole batches of 8)."""
    DB = (D + 63) // 64
    nchunk = max(1, min(-(-L // 32), -(-3072 // (B * 4 * DB))))
    chunk = max(32, -(-(-(-L // nchunk)) // 8) * 8)
    return chunk, -(-L // chunk), DB


def selective_scan(xT, pT, ids, dtw, dtb, A, Dskip, yT, state, sdt, ysum, B, L, D, N, R, chunk):
    """irm_selective_scan_f32 (include/irm_hip.h): all four scan directions, gather/scatter fused."""
    flops = B * 4.0 * L * D * (2 * R + 9 * N + 12)
    nbytes = 4.0 * B * L * (D + 4 * (R + 2 * N) + 4 * D)
    _launch("selective_scan", flops, nbytes, "irm_selective_scan_f32", _hip.ptr(xT), _hip.ptr(pT), _hip.ptr(ids),
            _hip.ptr(dtw), _hip.ptr(dtb), _hip.ptr(A), _hip.ptr(Dskip), _hip.ptr(yT), _hip.ptr(state), _hip.ptr(sdt),
            _hip.ptr(ysum), B, L, D, N, R, chunk, tag=f"L{L} D{D} N{N} B{B} chunk{chunk}")


def losh_combine(ysum, gw, gb, gate, yT, nw, nb, z, out, B, L, D, nchunk, eps=1e-5):
    """gate + direction sum + out_norm + silu(z) gate -> planar out [B, D, H, W]."""
    _chk(z, "z"), _chk(out, "out")
    _launch("losh_combine", 12.0 * B * L * D, 4.0 * B * L * D * 6, "irm_losh_combine_f32", _hip.ptr(ysum), _hip.ptr(gw),
            _hip.ptr(gb), _hip.ptr(gate), _hip.ptr(yT), _hip.ptr(nw), _hip.ptr(nb), _hip.ptr(z), _bs(z), _hip.ptr(out),
            _bs(out), B, L, D, nchunk, float(eps), tag=f"L{L} D{D} B{B}")


# --------------------------------------------------------------------------- DeblurGANv2 FPN-MobileNet
_STATS_WS = {}


def chan_stats(x, stats, eps=1e-5):
    """stats[b, c] = (mean, rstd) over H*W (train-mode BatchNorm on one tile / InstanceNorm).  Large planes
    are split over several workgroups; the partials live in a per-stream workspace."""
    _chk(x, "x")
    B, C, H, W = x.shape
    need = 3 * B * C * (-(-1024 // (B * C)))
    key = (x.device, torch.cuda.current_stream().cuda_stream)
    ws = _STATS_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 8192), dtype=torch.float32, device=x.device)
        _STATS_WS[key] = ws
    _launch("chan_stats", 4.0 * B * C * H * W, 4.0 * B * C * H * W, "irm_chan_stats_ws_f32", _hip.ptr(x), _bs(x),
            _hip.ptr(stats), _hip.ptr(ws), ws.numel(), B, C, H * W, float(eps), tag=f"C{C} {H}x{W} B{B}")


def chan_norm_act(x, stats, y, *, weight=None, bias=None, res=None, act=ACT_NONE):
    _chk(x, "x"), _chk(y, "y")
    B, C, H, W = x.shape
    _launch("chan_norm_act", 4.0 * B * C * H * W, 4.0 * B * C * H * W * (3 if res is not None else 2),
            "irm_chan_norm_act_f32", _hip.ptr(x), _bs(x), _hip.ptr(stats), _hip.ptr(weight), _hip.ptr(bias),
            _hip.ptr(res), _bs(res), _hip.ptr(y), _bs(y), B, C, H * W, int(act), tag=f"C{C} {H}x{W} B{B}")


def conv3x3_s2(x, w, y, ci, co):
    _chk(x, "x"), _chk(y, "y")
    B, _, H, W = x.shape
    _launch("conv3x3_s2", 18.0 * B * ci * co * (H // 2) * (W // 2), 4.0 * B * (ci * H * W + co * H * W / 4),
            "irm_conv3x3_s2_f32", _hip.ptr(x), _bs(x), _hip.ptr(w), _hip.ptr(y), _bs(y), B, ci, co, H, W,
            tag=f"ci{ci} co{co} {H}x{W} B{B}")


def dwconv3x3_s2(x, w9, y):
    _chk(x, "x"), _chk(y, "y")
    B, C, H, W = x.shape
    _launch("dwconv3x3_s2", 18.0 * B * C * H * W / 4, 5.0 * B * C * H * W, "irm_dwconv3x3_s2_f32", _hip.ptr(x), _bs(x),
            _hip.ptr(w9), _hip.ptr(y), _bs(y), B, C, H, W, tag=f"C{C} {H}x{W} B{B}")


def upsample_add(src, out, scale, add=None):
    """out = (add or 0) + nearest-upsample(src, scale)."""
    _chk(src, "src"), _chk(out, "out")
    B, C, Hs, Ws = src.shape
    _launch("upsample_add", 0.0, 4.0 * B * C * Hs * Ws * (1 + scale * scale * (2 if add is not None else 1)),
            "irm_upsample_add_f32", _hip.ptr(src), _bs(src), _hip.ptr(add), _bs(add), _hip.ptr(out), _bs(out), B, C, Hs, Ws,
            int(scale), tag=f"C{C} {Hs}x{Ws} x{scale} B{B}")
