"""Restormer forward on MI355X through libirm_hip.so.

Drop-in for the reference's ``src/restormer/restormer.py`` (class ``Restormer``,
same constructor keywords, same ``state_dict`` keys and shapes, ``model(x)``
with x float32 NCHW on the GPU).  The sub-modules below only *hold* the
parameters under the reference's names - they are never called; ``forward``
drives the HIP kernels:

  per TransformerBlock (restormer.py:137-150), activations planar NCHW, batch =
  all tiles of one image:
    ln_stats -> gemm1x1(qkv, LN prologue) -> dwconv3x3 -> mdta_gram ->
    mdta_finalize (softmax + fold with project_out) -> gemm1x1(Mfold, v, +x)
    ln_stats -> gemm1x1(project_in, LN prologue) -> dwconv3x3_gate ->
    gemm1x1(project_out, +x)
  and conv3x3 (patch embed, down/up-sampling with the pixel (un)shuffle folded
  into the store, output conv + input residual).  The U-Net concatenations
  (restormer.py:264-273) are not copies: producers write straight into channel
  slices of the concat buffers.

There is no PyTorch fallback: without the HIP library ``forward`` raises.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import _hip, ops

#: synthetic-weight rules (see synth.py): keep the restored image in range
SYNTH_RULES = (
    (r"^output\.weight$", "gain", 0.02),
    (r"^skip_conv\.weight$", "gain", 0.5),
)


def _hidden(dim, factor):
    return int(dim * factor)          # restormer.py:80


class _Norm(nn.Module):
    """Parameter holder with the reference's LayerNorm key layout (norm.body.weight/.bias)."""

    def __init__(self, dim, kind):
        super().__init__()
        self.body = nn.Module()
        self.body.weight = nn.Parameter(torch.ones(dim))
        if kind != "BiasFree":
            self.body.bias = nn.Parameter(torch.zeros(dim))
        self.mode = _hip.LN_BIASFREE if kind == "BiasFree" else _hip.LN_WITHBIAS

    @property
    def w(self):
        return self.body.weight

    @property
    def b(self):
        return getattr(self.body, "bias", None)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, bias):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = nn.Conv2d(dim, dim * 3, 1, bias=bias)
        self.qkv_dwconv = nn.Conv2d(dim * 3, dim * 3, 3, padding=1, groups=dim * 3, bias=bias)
        self.project_out = nn.Conv2d(dim, dim, 1, bias=bias)


class FeedForward(nn.Module):
    def __init__(self, dim, ffn_expansion_factor, bias):
        super().__init__()
        hid = _hidden(dim, ffn_expansion_factor)
        self.hidden = hid
        self.project_in = nn.Conv2d(dim, hid * 2, 1, bias=bias)
        self.dwconv = nn.Conv2d(hid * 2, hid * 2, 3, padding=1, groups=hid * 2, bias=bias)
        self.project_out = nn.Conv2d(hid, dim, 1, bias=bias)


class TransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type):
        super().__init__()
        self.dim = dim
        self.norm1 = _Norm(dim, LayerNorm_type)
        self.attn = Attention(dim, num_heads, bias)
        self.norm2 = _Norm(dim, LayerNorm_type)
        self.ffn = FeedForward(dim, ffn_expansion_factor, bias)


class _Proj(nn.Module):
    def __init__(self, cin, cout, bias):
        super().__init__()
        self.proj = nn.Conv2d(cin, cout, 3, padding=1, bias=bias)


class _Resample(nn.Module):
    """Holder for Downsample / Upsample: key 'body.0.weight' (restormer.py:171-189)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False))


def _stage(dim, heads, n, f, bias, ln):
    return nn.Sequential(*[TransformerBlock(dim, heads, f, bias, ln) for _ in range(n)])


class Restormer(nn.Module):
    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=(4, 6, 6, 8),
                 num_refinement_blocks=4, heads=(1, 2, 4, 8), ffn_expansion_factor=2.66, bias=False,
                 LayerNorm_type="WithBias", dual_pixel_task=False):
        super().__init__()
        f, ln = ffn_expansion_factor, LayerNorm_type
        d1, d2, d3, d4 = dim, dim * 2, dim * 4, dim * 8
        self.inp_channels, self.out_channels, self.dim = inp_channels, out_channels, dim
        self.patch_embed = _Proj(inp_channels, d1, bias=False)
        self.encoder_level1 = _stage(d1, heads[0], num_blocks[0], f, bias, ln)
        self.down1_2 = _Resample(d1, d1 // 2)
        self.encoder_level2 = _stage(d2, heads[1], num_blocks[1], f, bias, ln)
        self.down2_3 = _Resample(d2, d2 // 2)
        self.encoder_level3 = _stage(d3, heads[2], num_blocks[2], f, bias, ln)
        self.down3_4 = _Resample(d3, d3 // 2)
        self.latent = _stage(d4, heads[3], num_blocks[3], f, bias, ln)
        self.up4_3 = _Resample(d4, d4 * 2)
        self.reduce_chan_level3 = nn.Conv2d(d4, d3, 1, bias=bias)
        self.decoder_level3 = _stage(d3, heads[2], num_blocks[2], f, bias, ln)
        self.up3_2 = _Resample(d3, d3 * 2)
        self.reduce_chan_level2 = nn.Conv2d(d3, d2, 1, bias=bias)
        self.decoder_level2 = _stage(d2, heads[1], num_blocks[1], f, bias, ln)
        self.up2_1 = _Resample(d2, d2 * 2)
        self.decoder_level1 = _stage(d2, heads[0], num_blocks[0], f, bias, ln)
        self.refinement = _stage(d2, heads[0], num_refinement_blocks, f, bias, ln)
        self.dual_pixel_task = dual_pixel_task
        if dual_pixel_task:
            self.skip_conv = nn.Conv2d(d1, d2, 1, bias=bias)
        self.output = nn.Conv2d(d2, out_channels, 3, padding=1, bias=bias)
        self._packed = None
        self._packed_key = None
        self._split = False
        self._ws_by_stream = {}
        #: tiles of one image processed per forward by the device tiler (utils.tiled_forward_device)
        self.max_tiles_per_batch = 12     # (two 1280x720 frames: utils.tiled_forward_device_batch)
        self.hip_graph = True      # the tiler replays the per-batch forward from a HIP graph (utils.graphed_forward)
        self._tap = None           # tests: a dict that receives a copy of the `refinement` output (restormer.py:274)

    # ------------------------------------------------------------------ weights
    def load_synthetic(self, seed=42):
        """Fill every parameter from the deterministic generator (synth.py)."""
        from .. import synth
        shapes = {k: tuple(v.shape) for k, v in self.state_dict().items()}
        self.load_state_dict(synth.synth_state_dict(shapes, seed=seed, rules=SYNTH_RULES), strict=True)
        return self

    def _param_key(self):
        return _hip.param_key(self)

    def _pack(self):
        """Packed / flattened device copies of the weights, rebuilt when parameters change."""
        key = self._param_key()
        if self._packed is not None and self._packed_key == key:
            return self._packed
        pk = {}

        def f32(t):
            return None if t is None else t.detach().float().contiguous()

        # the two LayerNorm-prologue GEMMs of a block (qkv, project_in: 2/3 of the GEMM FLOPs) run as an fp32
        # emulation on the fp16 matrix cores (irm_gemm1x1_f16x3_f32; its error against float64 is not larger
        # than the exact-fp32 kernel's, and a LayerNorm output cannot leave the fp16 range): weights split into
        # fp16 hi/lo here.  IRM_GEMM_EXACT=1 keeps every GEMM on the f32-input MFMA.
        self._split = not os.environ.get("IRM_GEMM_EXACT")

        for name, m in self.named_modules():
            if isinstance(m, TransformerBlock):
                a, ff = m.attn, m.ffn
                pk[name] = dict(
                    qkv=_hip.pack_gemm_weight(a.qkv.weight), qkv_b=f32(a.qkv.bias),
                    qkv_dw=f32(a.qkv_dwconv.weight.reshape(-1, 9)), qkv_dw_b=f32(a.qkv_dwconv.bias),
                    wout=f32(a.project_out.weight.reshape(m.dim, m.dim)), wout_b=f32(a.project_out.bias),
                    temp=f32(a.temperature.reshape(-1)),
                    pin=_hip.pack_gemm_weight(ff.project_in.weight), pin_b=f32(ff.project_in.bias),
                    ffn_dw=f32(ff.dwconv.weight.reshape(-1, 9)), ffn_dw_b=f32(ff.dwconv.bias),
                    pout=_hip.pack_gemm_weight(ff.project_out.weight), pout_b=f32(ff.project_out.bias),
                    n1w=f32(m.norm1.w), n1b=f32(m.norm1.b), n2w=f32(m.norm2.w), n2b=f32(m.norm2.b))
                if self._split:
                    # per layer: outside the safe range of the unscaled split (trained checkpoints with large LayerNorm
                    # gains or tiny / huge weights) the layer stays on the exact f32 MFMA (_hip.split_is_safe)
                    if _hip.split_is_safe(a.qkv.weight, m.norm1.w, m.norm1.b):
                        pk[name]["qkv_s"] = _hip.pack_gemm_weight_split(a.qkv.weight)
                    if _hip.split_is_safe(ff.project_in.weight, m.norm2.w, m.norm2.b):
                        pk[name]["pin_s"] = _hip.pack_gemm_weight_split(ff.project_in.weight)
                    if _hip.split_is_safe(ff.project_out.weight):
                        pk[name]["pout_s"] = _hip.pack_gemm_weight_split(ff.project_out.weight)
                    pk[name]["mfold_split"] = _hip.split_is_safe(a.project_out.weight)
                    if ops.can_gate_split(m.dim, ff.hidden, 16, 16):
                        # GDFN tail on pre-split operands (gemm_ps.hip): project_out fragments, K padded to 32 ceil(hid / 32)
                        frag, s_w = _hip.pack_gemm_weight_presplit(ff.project_out.weight, k_pad=32 * -(-ff.hidden // 32))
                        pk[name]["pout_ps"] = (frag, 1.0 / (s_w * ops.GATE_SPLIT_SCALE))
                    if m.dim == 192:
                        # GDFN tail in one kernel (fused_tail.hip): project_in with its halves padded to a multiple of 16
                        # channels, written tile-major channel-last; taps / project_out packed for irm_gdfn_tail_f16x3_f32
                        hp = 64 * -(-ff.hidden // 64)
                        frag, s_w, bp = _hip.pack_pin_padded(ff.project_in.weight, ff.project_in.bias, hp)
                        s_x = _hip.ln_split_scale(m.norm2.w, m.norm2.b, m.dim, m.norm2.mode == ops.LN_WITHBIAS)
                        pk[name]["pin_cl"] = (frag, 1.0 / (s_w * s_x), s_x, bp, hp)
                        pk[name]["tail"] = _hip.pack_gdfn_tail(ff.dwconv.weight, ff.dwconv.bias, ff.project_out.weight)
                    if m.dim in (192, 384):
                        # LayerNorm + qkv / project_in with pre-split operands (gemm_ps.hip): (fragments, 1 / (s_w s_x), s_x);
                        # power-of-two scales on both operands - no range guard needed
                        for slot, conv, nrm in (("qkv_ps", a.qkv, m.norm1), ("pin_ps", ff.project_in, m.norm2)):
                            frag, s_w = _hip.pack_gemm_weight_presplit(conv.weight)
                            s_x = _hip.ln_split_scale(nrm.w, nrm.b, m.dim, nrm.mode == ops.LN_WITHBIAS)
                            pk[name][slot] = (frag, 1.0 / (s_w * s_x), s_x)
                    # Gram pass on the fp16 matrix cores where a static bound of |q|, |k| exists (WithBias LayerNorm)
                    gs = _hip.gram_scales(a.qkv.weight, a.qkv.bias, a.qkv_dwconv.weight, a.qkv_dwconv.bias, m.norm1.w,
                                          m.norm1.b, m.norm1.mode == ops.LN_WITHBIAS)
                    if gs is not None:
                        pk[name]["gram_s"] = gs.to(a.qkv.weight.device)
                if self._split and ops.can_fuse_gdfn(m.dim, 4):
                    # whole-branch kernels (fused_block.hip): LN + qkv + dwconv, and the complete GDFN
                    pk[name].update(
                        qkv_f=_hip.pack_qkv_fused(a.qkv.weight, a.qkv.bias, a.qkv_dwconv.weight, a.qkv_dwconv.bias,
                                                  m.norm1.w, m.norm1.b),
                        gdfn_f=_hip.pack_gdfn_fused(ff.project_in.weight, ff.project_in.bias, ff.dwconv.weight,
                                                    ff.dwconv.bias, ff.project_out.weight, m.norm2.w, m.norm2.b))
                    if pk[name]["mfold_split"] and m.dim % 16 == 0:
                        # attention apply inside the GDFN kernel (irm_attn_gdfn_fused_f16x3_f32): project_in's
                        # input channels in the order its first MFMA leaves x' in the registers
                        pk[name]["gdfn_fa"] = _hip.pack_gdfn_fused(
                            ff.project_in.weight, ff.project_in.bias, ff.dwconv.weight, ff.dwconv.bias,
                            ff.project_out.weight, m.norm2.w, m.norm2.b, kperm=True)
                if ops.can_fuse_dw(m.dim, 4):
                    # depth-wise coefficient tables of the fused dw + 1x1 kernel (irm_dwgemm_f32)
                    c, dw, dwb = m.dim, a.qkv_dwconv.weight.reshape(-1, 9), a.qkv_dwconv.bias
                    pk[name].update(
                        v_dwp=_hip.pack_dw_table(dw[2 * c:], None if dwb is None else dwb[2 * c:], c, False),
                        ffn_dwp=_hip.pack_dw_table(ff.dwconv.weight, ff.dwconv.bias, ff.hidden, True))
        for name in ("down1_2", "down2_3", "down3_4", "up4_3", "up3_2", "up2_1"):
            pk[name] = _hip.pack_conv3x3(getattr(self, name).body[0].weight)
        pk["patch_embed"] = _hip.pack_conv3x3(self.patch_embed.proj.weight)
        pk["patch_embed_b"] = f32(self.patch_embed.proj.bias)
        pk["output"] = _hip.pack_conv3x3(self.output.weight)
        pk["output_b"] = f32(self.output.bias)
        for name in ("reduce_chan_level3", "reduce_chan_level2") + (("skip_conv",) if self.dual_pixel_task else ()):
            pk[name] = _hip.pack_gemm_weight(getattr(self, name).weight)
            pk[name + "_b"] = f32(getattr(self, name).bias)
            # (un-normalised input: the emulated GEMM scales it by 2^-4 and saturates, as for project_out)
            if self._split and name != "skip_conv" and _hip.split_is_safe(getattr(self, name).weight):
                pk[name + "_s"] = _hip.pack_gemm_weight_split(getattr(self, name).weight)
        self._packed, self._packed_key = pk, key
        return pk

    # ------------------------------------------------------------------ workspace
    @property
    def _ws(self):
        """Workspace of the current stream (concurrent forwards on different streams must not share)."""
        return self._ws_by_stream.setdefault(torch.cuda.current_stream().cuda_stream if torch.cuda.is_available()
                                             else 0, {})

    def _buf(self, name, numel, device):
        ws = self._ws
        t = ws.get(name)
        if t is None or t.numel() < numel or t.device != device:
            t = torch.empty(int(numel), dtype=torch.float32, device=device)
            ws[name] = t
        return t[:numel]

    def release_workspace(self):
        self._ws_by_stream.clear()

    # ------------------------------------------------------------------ kernels
    def _block(self, blk: TransformerBlock, w: dict, x: torch.Tensor, have_stats: bool = False,
               want_stats: bool = True) -> bool:
        """One TransformerBlock, in place on x (a [B][C][H][W] view, channel/pixel axes dense).

        have_stats: the "stats" workspace already holds the LayerNorm statistics of x (written by the
        epilogue of the GEMM that produced x).  Returns True if it left the statistics of its output
        there for the next block (fused into project_out's epilogue; needs C <= 144)."""
        B, C, H, W = x.shape
        N = H * W
        dev = x.device
        heads = blk.attn.num_heads
        hid = blk.ffn.hidden
        fuse = ops.can_fuse_stats(C)
        stats = self._buf("stats", B * 2 * N, dev)
        big_a = self._buf("scratch_a", B * max(3 * C, 2 * hid) * N, dev)
        big_b = self._buf("scratch_b", B * max(3 * C, hid) * N, dev)
        qkv = big_a[:B * 3 * C * N].view(B, 3 * C, H, W)
        qkv2 = big_b[:B * 3 * C * N].view(B, 3 * C, H, W)
        # --- attention branch: x += project_out(softmax(q k^T) v)   (restormer.py:111-132, 147)
        split = self._split and N % 4 == 0            # the emulation kernel needs the 16-byte fast path
        s_qkv, s_pin, s_pout = split and "qkv_s" in w, split and "pin_s" in w, split and "pout_s" in w
        s_fold = split and w.get("mfold_split", False)
        presplit = split and "qkv_ps" in w and ops.can_presplit(C, N) and not os.environ.get("IRM_NO_PRESPLIT")
        if presplit:
            # LayerNorm + fp16 hi/lo split once (statistics in the kernel), then a pure matrix-core GEMM
            xs = self._buf("xsplit", B * C * N, dev)
            frag, out_scale, s_x = w["qkv_ps"]
            ops.ln_gemm_presplit(frag, x, qkv, 3 * C, C, w["n1w"], w["n1b"], blk.norm1.mode, s_x, out_scale=out_scale,
                                 bias=w["qkv_b"], xs=xs)
        else:
            if not have_stats:
                ops.ln_stats(x, stats)
            ops.gemm1x1(w["qkv_s" if s_qkv else "qkv"], x, qkv, 3 * C, C, bias=w["qkv_b"], stats=stats, lnw=w["n1w"],
                        lnb=w["n1b"], ln_mode=blk.norm1.mode, split=s_qkv)
        fuse_dw = "v_dwp" in w and ops.can_fuse_dw(C, W) and not os.environ.get("IRM_NO_FUSE_DW")
        if fuse_dw:
            # q, k only: the depth-wise conv of v happens inside the apply GEMM below
            ops.dwconv3x3(qkv[:, :2 * C], w["qkv_dw"], qkv2[:, :2 * C],
                          bias=None if w["qkv_dw_b"] is None else w["qkv_dw_b"][:2 * C])
        else:
            ops.dwconv3x3(qkv, w["qkv_dw"], qkv2, bias=w["qkv_dw_b"])
        _, nchunk, rec = ops.mdta_plan(B, C, heads, N)
        part = self._buf("gram_part", B * heads * nchunk * rec, dev)
        gsum = self._buf("gram_sum", B * heads * rec, dev)
        mfold_n = ops.mfold_numel(C)
        ws = self._ws
        mfold = ws.get(("mfold", C, B))
        if mfold is None or mfold.device != dev:
            mfold = torch.zeros(B * mfold_n, dtype=torch.float32, device=dev)
            ws[("mfold", C, B)] = mfold
        # the folded per-image matrix in the order of the kernel that applies it (fp16 hi/lo when emulated)
        ops.mdta_fold(qkv2, part, gsum, w["temp"], w["wout"], mfold, C, heads, split=s_fold,
                      gram_scale=w.get("gram_s") if split else None)
        if fuse_dw:
            ops.dwgemm(mfold, w["v_dwp"], qkv[:, 2 * C:], x, C, C, gate=False, res=x, bias=w["wout_b"],
                       w_bs=mfold_n, stats_out=stats if fuse else None, split=s_fold)
        else:
            ops.gemm1x1(mfold, qkv2[:, 2 * C:], x, C, C, res=x, bias=w["wout_b"], w_bs=mfold_n,
                        stats_out=stats if fuse else None, split=s_fold)
        # --- feed-forward branch: x += project_out(gelu(dw(h1)) * dw(h2))   (restormer.py:88-93, 148)
        if (split and "tail" in w and ops.can_gdfn_tail(C, H, W) and not os.environ.get("IRM_NO_GDFN_TAIL")):
            # LayerNorm + project_in -> h tile-major channel-last; depth-wise + gate + project_out + residual in ONE kernel
            frag, out_scale, s_x, bp, hp = w["pin_cl"]
            h_cl = self._buf("h_cl", B * 2 * hp * N, dev)
            ops.ln_gemm_presplit_cl(frag, x, h_cl, 2 * hp, C, w["n2w"], w["n2b"], blk.norm2.mode, s_x, out_scale=out_scale, bias=bp)
            ops.gdfn_tail(w["tail"], h_cl, x, C, hid, hp, bias=w["pout_b"])
            return False
        h = big_a[:B * 2 * hid * N].view(B, 2 * hid, H, W)
        g = big_b[:B * hid * N].view(B, hid, H, W)
        if presplit:
            frag, out_scale, s_x = w["pin_ps"]
            ops.ln_gemm_presplit(frag, x, h, 2 * hid, C, w["n2w"], w["n2b"], blk.norm2.mode, s_x, out_scale=out_scale,
                                 bias=w["pin_b"], xs=xs)
        else:
            if not fuse:
                ops.ln_stats(x, stats)
            ops.gemm1x1(w["pin_s" if s_pin else "pin"], x, h, 2 * hid, C, bias=w["pin_b"], stats=stats, lnw=w["n2w"],
                        lnb=w["n2b"], ln_mode=blk.norm2.mode, split=s_pin)
        emit = fuse and want_stats
        if (split and "pout_ps" in w and not emit and ops.can_gate_split(C, hid, W, N) and os.environ.get("IRM_GATE_SPLIT")):
            # gate -> fp16 hi/lo fragments (the bytes of g), then a K-streamed matrix-core GEMM in place on x.  OPT-IN
            # (IRM_GATE_SPLIT=1): the GEMM gains (C = 384: 85 vs 126 us, C = 192: 107 vs 116 us in the model) but the
            # fragment-writing gate kernel loses more (87 vs 65 us, 174 vs 138 us); same-box A/B of the whole step:
            # 54.37 ms without, 54.65 ms with it at C = 384
            frag, out_scale = w["pout_ps"]
            ks = -(-hid // 32)
            gs = self._buf("gsplit", B * 32 * ks * N, dev)
            ops.dwconv3x3_gate_split(h, w["ffn_dw"], gs, bias=w["ffn_dw_b"])
            ops.gemm_presplit_res(frag, gs, x, C, ks, out_scale=out_scale, res=x, bias=w["pout_b"])
        elif fuse_dw:
            ops.dwgemm(w["pout_s" if s_pout else "pout"], w["ffn_dwp"], h, x, C, hid, gate=True, res=x,
                       bias=w["pout_b"], stats_out=stats if emit else None, split=s_pout)
        else:
            ops.dwconv3x3_gate(h, w["ffn_dw"], g, bias=w["ffn_dw_b"])
            ops.gemm1x1(w["pout_s" if s_pout else "pout"], g, x, C, hid, res=x, bias=w["pout_b"],
                        stats_out=stats if emit else None, split=s_pout)
        return emit

    def _block_fused(self, blk: TransformerBlock, w: dict, x: torch.Tensor, alt: torch.Tensor, x_tm: bool = False,
                     y_tm: bool = False) -> torch.Tensor:
        """One TransformerBlock on the whole-branch kernels (C <= 96): x -> alt, returns alt.

        qkv_dw_fused (LN1 + qkv + depth-wise, restormer.py:105-106) -> Gram + softmax + fold (:118-129) ->
        1x1 with the folded per-image matrix on v, + x in place (:131, 147) -> gdfn_fused (LN2 + GDFN + x, :148)
        into alt: neighbouring tiles read each other's halo of x, so the last step cannot run in place."""
        B, C, H, W = x.shape
        N = H * W
        dev = x.device
        heads, hid = blk.attn.num_heads, blk.ffn.hidden
        qkv = self._buf("scratch_a", B * 3 * C * N, dev).view(B, 3 * C, H, W)
        # q, k tile-major for the Gram pass (its only reader) where the f16x3 ring pass runs on whole tiles
        tm = ops.can_qk_tile_major(C, heads, H, W) and not os.environ.get("IRM_NO_QK_TM")
        fa = "gdfn_fa" in w and not os.environ.get("IRM_NO_APPLY_FUSE")
        assert not (x_tm or y_tm) or (tm and fa), "tile-major x / y: only between the kernels that understand them"
        v_tm = tm and fa and not os.environ.get("IRM_NO_ACT_TM")
        gram_in_qkv = (tm and "gram_s" in w and ops.can_qkv_gram(C, heads, H, W) and not os.environ.get("IRM_GRAM_EXACT")
                       and not os.environ.get("IRM_NO_QKV_GRAM"))
        _, nchunk, rec = ops.mdta_plan(B, C, heads, N)
        nready = None
        if gram_in_qkv:
            nchunk = (H // 8) * (W // 32) // ops.QKV_GRAM_NCH
        part = self._buf("gram_part", B * heads * nchunk * rec, dev)
        if gram_in_qkv:
            # C = 48, one head: the Gram partials come out of the qkv kernel, q and k are never written
            nready = ops.qkv_gram_cm(w["qkv_f"], x, qkv, w["gram_s"], part, C, ln_mode=blk.norm1.mode, x_tm=x_tm, v_tm=v_tm)
        else:
            ops.qkv_dw_fused(w["qkv_f"], x, qkv, C, 3 * C, ln_mode=blk.norm1.mode, tm=tm, x_tm=x_tm, v_tm=v_tm)
        gsum = self._buf("gram_sum", B * heads * rec, dev)
        mfold_n = ops.mfold_numel(C)
        ws = self._ws
        mfold = ws.get(("mfold", C, B))
        if mfold is None or mfold.device != dev:
            mfold = torch.zeros(B * mfold_n, dtype=torch.float32, device=dev)
            ws[("mfold", C, B)] = mfold
        s_fold = w.get("mfold_split", False)
        if fa:
            # x' = x + project_out(attn @ v) is formed in the GDFN kernel's prologue and never written (:131, 147-148)
            mfrag = ws.get(("mfold_frag", C, B))
            if mfrag is None or mfrag.device != dev:
                mfrag = torch.zeros(B * ops.mfold_frag_numel(C), dtype=torch.float32, device=dev)
                ws[("mfold_frag", C, B)] = mfrag
            ops.mdta_fold(qkv, part, gsum, w["temp"], w["wout"], mfrag, C, heads, gram_scale=w.get("gram_s"), frag=True, tm=tm,
                          nchunk_ready=nready)
            ops.attn_gdfn_fused(w["gdfn_fa"], x, qkv[:, 2 * C:], mfrag, alt, C, hid, ln_mode=blk.norm2.mode,
                                bias_o=w["wout_b"], bias=w["pout_b"], x_tm=x_tm, v_tm=v_tm, y_tm=y_tm)
            return alt
        ops.mdta_fold(qkv, part, gsum, w["temp"], w["wout"], mfold, C, heads, split=s_fold, gram_scale=w.get("gram_s"), tm=tm,
                      nchunk_ready=nready)
        ops.gemm1x1(mfold, qkv[:, 2 * C:], x, C, C, res=x, bias=w["wout_b"], w_bs=mfold_n, split=s_fold)
        ops.gdfn_fused(w["gdfn_f"], x, alt, C, hid, ln_mode=blk.norm2.mode, bias=w["pout_b"])
        return alt

    def _run_stage(self, name, pk, x, have_stats=False):
        """The TransformerBlocks of one stage (or of several consecutive stages on the same tensor: a tuple of names), in
        place on x."""
        names = (name,) if isinstance(name, str) else tuple(name)
        blocks = [(f"{n}.{i}", blk) for n in names for i, blk in enumerate(getattr(self, n))]
        B, C, H, W = x.shape
        if (len(blocks) and all("gdfn_f" in pk[k] for k, _ in blocks) and ops.can_fuse_gdfn(C, W) and (H * W) % 4 == 0
                and len({blk.attn.num_heads for _, blk in blocks}) == 1 and not os.environ.get("IRM_NO_FUSE_BLOCK")):
            cur, alt = x, self._buf(f"alt_{C}", B * C * H * W, x.device).view(B, C, H, W)
            # between the blocks x travels tile-major channel-last (include/irm_hip.h): the first block reads the planar
            # input, the last one writes the planar output
            heads = blocks[0][1].attn.num_heads
            act_tm = (all("gdfn_fa" in pk[k] for k, _ in blocks) and ops.can_qk_tile_major(C, heads, H, W)
                      and not os.environ.get("IRM_NO_APPLY_FUSE") and not os.environ.get("IRM_NO_QK_TM")
                      and not os.environ.get("IRM_NO_ACT_TM"))
            for i, (k, blk) in enumerate(blocks):
                out = self._block_fused(blk, pk[k], cur, alt, x_tm=act_tm and i > 0, y_tm=act_tm and i + 1 < len(blocks))
                cur, alt = out, cur
            if cur is not x:                           # odd number of blocks: the result belongs in x
                x.copy_(cur)
                if ops.TIMER is not None:
                    ops.TIMER.break_chain()
            return False
        for n in names:
            stage = getattr(self, n)
            for i, blk in enumerate(stage):
                have_stats = self._block(blk, pk[f"{n}.{i}"], x, have_stats, want_stats=i + 1 < len(stage))
        return have_stats

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, inp_img: torch.Tensor) -> torch.Tensor:
        if not inp_img.is_cuda:
            raise _hip.HipLibraryError("irm_amd Restormer runs on the GPU only (no CPU fallback); "
                                       "move the model and input to 'cuda'")
        x = inp_img.float().contiguous()
        B, Cin, H, W = x.shape
        if H % 8 or W % 8:
            raise ValueError("Restormer needs H and W to be multiples of 8 (the tiler pads, utils.pad)")
        dev = x.device
        pk = self._pack()
        d1, d2, d3, d4 = self.dim, self.dim * 2, self.dim * 4, self.dim * 8
        H2, W2, H3, W3, H4, W4 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8

        def buf(name, ch, h, w):
            return self._buf(name, B * ch * h * w, dev).view(B, ch, h, w)

        cat1 = buf("cat1", 2 * d1, H, W)       # [up2_1 | enc1]   (restormer.py:272)
        cat2 = buf("cat2", 2 * d2, H2, W2)     # [up3_2 | enc2]   (restormer.py:267)
        cat3 = buf("cat3", 2 * d3, H3, W3)     # [up4_3 | enc3]   (restormer.py:262)
        lat = buf("latent", d4, H4, W4)
        dec3 = buf("dec3", d3, H3, W3)
        dec2 = buf("dec2", d2, H2, W2)

        e1 = cat1[:, d1:]
        ops.conv3x3(pk["patch_embed"], x, e1, Cin, d1, bias=pk["patch_embed_b"])
        if self.dual_pixel_task:
            e1_in = buf("enc1_in", d1, H, W)
            e1_in.copy_(e1)
            if ops.TIMER is not None:
                ops.TIMER.break_chain()            # a torch kernel sits between two timed launches
        self._run_stage("encoder_level1", pk, e1)
        e2 = cat2[:, d2:]
        ops.conv3x3(pk["down1_2"], e1, e2, d1, d1 // 2, store_mode=1)
        self._run_stage("encoder_level2", pk, e2)
        e3 = cat3[:, d3:]
        ops.conv3x3(pk["down2_3"], e2, e3, d2, d2 // 2, store_mode=1)
        self._run_stage("encoder_level3", pk, e3)
        ops.conv3x3(pk["down3_4"], e3, lat, d3, d3 // 2, store_mode=1)
        self._run_stage("latent", pk, lat)

        ops.conv3x3(pk["up4_3"], lat, cat3[:, :d3], d4, d4 * 2, store_mode=2)
        rs3 = "reduce_chan_level3_s" in pk and (H3 * W3) % 4 == 0 and not os.environ.get("IRM_NO_RC_SPLIT")
        ops.gemm1x1(pk["reduce_chan_level3" + ("_s" if rs3 else "")], cat3, dec3, d3, 2 * d3, bias=pk["reduce_chan_level3_b"],
                    split=rs3)
        self._run_stage("decoder_level3", pk, dec3)
        ops.conv3x3(pk["up3_2"], dec3, cat2[:, :d2], d3, d3 * 2, store_mode=2)
        rs2 = "reduce_chan_level2_s" in pk and (H2 * W2) % 4 == 0 and not os.environ.get("IRM_NO_RC_SPLIT")
        ops.gemm1x1(pk["reduce_chan_level2" + ("_s" if rs2 else "")], cat2, dec2, d2, 2 * d2, bias=pk["reduce_chan_level2_b"],
                    split=rs2)
        self._run_stage("decoder_level2", pk, dec2)
        ops.conv3x3(pk["up2_1"], dec2, cat1[:, :d1], d2, d2 * 2, store_mode=2)
        # (decoder_level1 and refinement work on the same tensor: one chain, no planar round trip between them)
        self._run_stage(("decoder_level1", "refinement"), pk, cat1)
        tap = self.__dict__.get("_tap")
        if tap is not None:                              # test tap: the trunk output before the `output` conv
            tap["refinement"] = cat1.clone()

        out = torch.empty(B, self.out_channels, H, W, dtype=torch.float32, device=dev)
        if self.dual_pixel_task:
            ops.gemm1x1(pk["skip_conv"], e1_in, cat1, d2, d1, res=cat1, bias=pk["skip_conv_b"])
            ops.conv3x3(pk["output"], cat1, out, d2, self.out_channels, bias=pk["output_b"])
        else:
            ops.conv3x3(pk["output"], cat1, out, d2, self.out_channels, bias=pk["output_b"],
                        res=x, res_mode=1)
        return out
