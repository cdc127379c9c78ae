"""MaIRUNet forward on MI355X (drop-in for
src/mair/realDenoising/basicsr/models/archs/mairunet_arch.py:445-739).

Same constructor keywords and state_dict keys as the reference; the modules below only hold the
parameters.  Activations are planar NCHW (the reference's token layout (B, HW, C) is an internal detail of
its PyTorch graph); per VSSBlock (mairunet_arch.py:362-380):

  ln_stats -> gemm1x1(in_proj, LN prologue) -> dwconv3x3(+bias, SiLU) -> gemm1x1(x_proj of the 4 directions,
  stacked) -> transpose to channel-last -> irm_selective_scan_f32 (gather by scan order, dt_proj, SSM
  recurrence, inverse scatter; replaces mamba_ssm) -> irm_losh_combine_f32 (ShuffleAttn gate, direction sum,
  out_norm, * silu(z)) -> gemm1x1(out_proj, + x * skip_scale) -> ln_stats -> gemm1x1(fc1, LN, bias, GELU) ->
  gemm1x1(fc2, bias, + x * skip_scale2)

The scan-index tables (shift_scanf_util.py:67-244) are built once per (H, W, scan_len) with vectorised
numpy and cached - the reference rebuilds 8 tables with Python loops on every eval forward
(mairunet_arch.py:657-666); MaIRUNet never uses the shifted variant (shift_size is never passed, :476-581).
"""
from __future__ import annotations

import math

import numpy as np
import os

import torch
import torch.nn as nn

from .. import _hip, ops
from . import SYNTH_RULES

_IDS_CACHE: dict = {}


def _snake(idx: np.ndarray, scan_len: int, shift_len: int = 0) -> np.ndarray:
    """Visit order of the nested S-shaped scan on an index image [H][W]: column stripes left to right (with
    shift_len > 0 a first stripe of that width, then stripes of `scan_len`), odd stripes bottom-up, rows inside
    a stripe alternately left-to-right / right-to-left."""
    H, W = idx.shape
    edges = [0, shift_len] if shift_len else [0]
    while edges[-1] < W:
        edges.append(min(edges[-1] + scan_len, W))
    order = []
    hv = np.arange(H)
    for s in range(len(edges) - 1):
        cols = np.arange(edges[s], edges[s + 1])
        rows = (H - 1 - hv) if s % 2 else hv
        block = idx[rows][:, cols]                    # [H][w] in visit order of the rows
        block[1::2] = block[1::2, ::-1]               # odd visited rows run right to left
        order.append(block.reshape(-1))
    return np.concatenate(order)


def scan_ids(H: int, W: int, scan_len: int, device, shift_len: int = 0) -> torch.Tensor:
    """[4][H*W] int32 on `device`: direction 0 the image, 1 rotated by 180 degrees, 2 transposed, 3 both
    (mair_ids_generate / mair_shift_ids_generate, shift_scanf_util.py:169-203)."""
    key = (H, W, scan_len, shift_len, str(device))
    if key not in _IDS_CACHE:
        idx = np.arange(H * W).reshape(H, W)
        rot = idx[::-1, ::-1]
        ids = np.stack([_snake(m, scan_len, shift_len) for m in (idx, rot, idx.T, rot.T)])
        _IDS_CACHE[key] = torch.from_numpy(ids.astype(np.int32)).to(device)
    return _IDS_CACHE[key]


class _P(nn.Module):
    """bare parameter container"""


class LoSh2D(nn.Module):
    def __init__(self, d_model, d_state, ssm_ratio, bias=False, conv_bias=True):
        super().__init__()
        self.d_model, self.d_state = d_model, d_state
        self.d_inner = int(ssm_ratio * d_model)
        self.dt_rank = math.ceil(d_model / 16)
        D, N, R = self.d_inner, d_state, self.dt_rank
        self.in_proj = nn.Linear(d_model, 2 * D, bias=bias)
        self.conv2d = nn.Conv2d(D, D, 3, padding=1, groups=D, bias=conv_bias)
        self.x_proj_weight = nn.Parameter(torch.zeros(4, R + 2 * N, D))
        self.dt_projs_weight = nn.Parameter(torch.zeros(4, D, R))
        self.dt_projs_bias = nn.Parameter(torch.zeros(4, D))
        self.A_logs = nn.Parameter(torch.zeros(4 * D, N))
        self.Ds = nn.Parameter(torch.ones(4 * D))
        self.out_norm = nn.LayerNorm(D)
        self.out_proj = nn.Linear(D, d_model, bias=bias)
        self.gating = _P()
        self.gating.gating = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(4 * D, 4 * D, 1, groups=D), nn.Sigmoid())


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class VSSBlock(nn.Module):
    """mairunet_arch.py:332-380; `mlp_name='conv_blk'` gives the flat MaIR's RMB (mair_arch.py:346-390)."""

    def __init__(self, hidden_dim, d_state, ssm_ratio, mlp_ratio, bias=False, mlp_name="mlp"):
        super().__init__()
        self.hidden_dim, self.mlp_name = hidden_dim, mlp_name
        self.ln_1 = nn.LayerNorm(hidden_dim)
        self.self_attention = LoSh2D(hidden_dim, d_state, ssm_ratio, bias=bias)
        self.skip_scale = nn.Parameter(torch.ones(hidden_dim))
        setattr(self, mlp_name, Mlp(hidden_dim, int(hidden_dim * mlp_ratio)))
        self.ln_2 = nn.LayerNorm(hidden_dim)
        self.skip_scale2 = nn.Parameter(torch.ones(hidden_dim))

    @property
    def ffn(self):
        return getattr(self, self.mlp_name)


class _Proj(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.proj = nn.Conv2d(cin, cout, 3, padding=1, bias=False)


class _Resample(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False))


def pack_block(m: VSSBlock) -> dict:
    """Packed / flattened device copies of one block's weights."""
    def f32(t):
        return None if t is None else t.detach().float().contiguous()
    a = m.self_attention
    D = a.d_inner
    # the two LayerNorm-prologue GEMMs (in_proj, fc1) also in the hi/lo fp16 order of irm_gemm1x1_f16x3_f32
    split = {} if os.environ.get("IRM_GEMM_EXACT") else dict(
        inp_s=_hip.pack_gemm_weight_split(a.in_proj.weight), fc1_s=_hip.pack_gemm_weight_split(m.ffn.fc1.weight))
    if split:
        # round 3: the GEMMs without a LayerNorm prologue too (x_proj, out_proj, fc2: 3.6 of the 15.2 ms of a 256x256
        # image on the exact f32 MFMA), per layer inside the safe range of the unscaled weight split; their inputs
        # (silu / out_norm / gelu outputs) take the 2^-4 scaled, saturating split of irm_gemm1x1_f16x3_f32
        for key, wt in (("xproj_s", a.x_proj_weight.reshape(-1, D)), ("outp_s", a.out_proj.weight), ("fc2_s", m.ffn.fc2.weight)):
            if _hip.split_is_safe(wt):
                split[key] = _hip.pack_gemm_weight_split(wt)
    return dict(
        **split,
        inp=_hip.pack_gemm_weight(a.in_proj.weight), inp_b=f32(a.in_proj.bias),
        dw=f32(a.conv2d.weight.reshape(D, 9)), dw_b=f32(a.conv2d.bias),
        xproj=_hip.pack_gemm_weight(a.x_proj_weight.reshape(-1, D)),
        dtw=f32(a.dt_projs_weight), dtb=f32(a.dt_projs_bias),
        A=f32(-torch.exp(a.A_logs.detach().float())), Ds=f32(a.Ds),
        gw=f32(a.gating.gating[1].weight.reshape(4 * D, 4)), gb=f32(a.gating.gating[1].bias),
        onw=f32(a.out_norm.weight), onb=f32(a.out_norm.bias),
        outp=_hip.pack_gemm_weight(a.out_proj.weight), outp_b=f32(a.out_proj.bias),
        ln1w=f32(m.ln_1.weight), ln1b=f32(m.ln_1.bias), ln2w=f32(m.ln_2.weight), ln2b=f32(m.ln_2.bias),
        s1=f32(m.skip_scale), s2=f32(m.skip_scale2),
        fc1=_hip.pack_gemm_weight(m.ffn.fc1.weight), fc1_b=f32(m.ffn.fc1.bias),
        fc2=_hip.pack_gemm_weight(m.ffn.fc2.weight), fc2_b=f32(m.ffn.fc2.bias))


class MambaHost(nn.Module):
    """Workspace + the VSSBlock driver shared by MaIRUNet and the flat MaIR."""

    def _init_host(self):
        self._packed, self._packed_key, self._ws_by_stream = None, None, {}

    def _param_key(self):
        return _hip.param_key(self)

    def _buf(self, name, numel, device):
        ws = self._ws_by_stream.setdefault(torch.cuda.current_stream().cuda_stream, {})
        t = ws.get(name)
        if t is None or t.numel() < numel or t.device != device:
            t = torch.empty(int(numel), dtype=torch.float32, device=device)
            ws[name] = t
        return t[:numel]

    def release_workspace(self):
        self._ws_by_stream.clear()

    # ------------------------------------------------------------------ one VSSBlock, in place on x
    def _block(self, blk: VSSBlock, w: dict, x: torch.Tensor, ids: torch.Tensor, have_stats=False, want_stats=True):
        B, C, H, W = x.shape
        L = H * W
        dev = x.device
        a = blk.self_attention
        D, N, R = a.d_inner, a.d_state, a.dt_rank
        J = R + 2 * N
        hid = blk.ffn.fc1.out_features
        fuse = ops.can_fuse_stats(C)
        stats = self._buf("stats", B * 2 * L, dev)
        xz = self._buf("xz", B * 2 * D * L, dev).view(B, 2 * D, H, W)
        xc = self._buf("xc", B * D * L, dev).view(B, D, H, W)
        proj = self._buf("proj", B * 4 * J * L, dev).view(B, 4 * J, H, W)
        xT = self._buf("xT", B * L * D, dev).view(B, L, D)
        pT = self._buf("pT", B * L * 4 * J, dev).view(B, L, 4 * J)
        yT = self._buf("yT", B * 4 * L * D, dev)
        chunk, nchunk, DB = ops.scan_plan(B, L, D)
        state = self._buf("scan_state", 2 * B * 4 * DB * nchunk * N * 64, dev)
        sdt = self._buf("scan_sdt", B * 4 * DB * nchunk * 64, dev)
        ysum = self._buf("scan_ysum", B * 4 * DB * nchunk * 64, dev)
        gate = self._buf("gate", B * 4 * D, dev)
        # --- x = x * skip_scale + LoSh2D(LN(x))     (mairunet_arch.py:263-282, 374-375)
        if not have_stats:
            ops.ln_stats(x, stats)
        split = "inp_s" in w and L % 4 == 0           # fp32 emulation on the fp16 matrix cores (needs the 16-byte path)
        ops.gemm1x1(w["inp_s" if split else "inp"], x, xz, 2 * D, C, bias=w["inp_b"], stats=stats, lnw=w["ln1w"],
                    lnb=w["ln1b"], ln_mode=ops.LN_WITHBIAS, split=split)
        ops.dwconv3x3(xz[:, :D], w["dw"], xc, bias=w["dw_b"], act=ops.ACT_SILU)
        s_xp, s_out, s_fc2 = split and "xproj_s" in w, split and "outp_s" in w, split and "fc2_s" in w
        ops.gemm1x1(w["xproj_s" if s_xp else "xproj"], xc, proj, 4 * J, D, split=s_xp)
        ops.transpose(xc.view(B, D, L), xT, D, L)
        ops.transpose(proj.view(B, 4 * J, L), pT, 4 * J, L)
        ops.selective_scan(xT, pT, ids, w["dtw"], w["dtb"], w["A"], w["Ds"], yT, state, sdt, ysum, B, L, D, N, R, chunk)
        yn = xc                                          # xc is dead after the transposes: reuse for out_norm output
        ops.losh_combine(ysum, w["gw"], w["gb"], gate, yT, w["onw"], w["onb"], xz[:, D:], yn, B, L, D, nchunk)
        ops.gemm1x1(w["outp_s" if s_out else "outp"], yn, x, C, D, res=x, bias=w["outp_b"], res_scale=w["s1"],
                    stats_out=stats if fuse else None, split=s_out)
        # --- x = x * skip_scale2 + fc2(gelu(fc1(LN(x))))     (mairunet_arch.py:62-78, 377)
        h = self._buf("mlp_h", B * hid * L, dev).view(B, hid, H, W)
        if not fuse:
            ops.ln_stats(x, stats)
        ops.gemm1x1(w["fc1_s" if split else "fc1"], x, h, hid, C, bias=w["fc1_b"], stats=stats, lnw=w["ln2w"],
                    lnb=w["ln2b"], ln_mode=ops.LN_WITHBIAS, act=ops.ACT_GELU, split=split)
        emit = fuse and want_stats
        ops.gemm1x1(w["fc2_s" if s_fc2 else "fc2"], h, x, C, hid, res=x, bias=w["fc2_b"], res_scale=w["s2"],
                    stats_out=stats if emit else None, split=s_fc2)
        return emit



class MaIRUNet(MambaHost):
    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=(4, 6, 6, 8), ssm_ratio=1.5,
                 num_refinement_blocks=4, drop_path_rate=0., bias=False, dual_pixel_task=False, flp_ratio=2,
                 mlp_ratio=2, dynamic_ids=False, img_size=64, scan_len=8, batch_size=1):
        super().__init__()
        self.inp_channels, self.out_channels, self.dim, self.scan_len = inp_channels, out_channels, dim, scan_len
        d1, d2, d3, d4 = dim, dim * 2, dim * 4, dim * 8

        def stage(c, n, d_state, ratio):
            return nn.ModuleList([VSSBlock(c, d_state, ssm_ratio, ratio, bias=False) for _ in range(n)])

        self.patch_embed = _Proj(inp_channels, d1)
        self.encoder_level1 = stage(d1, num_blocks[0], 4, flp_ratio)
        self.down1_2 = _Resample(d1, d1 // 2)
        self.encoder_level2 = stage(d2, num_blocks[1], 8, mlp_ratio)
        self.down2_3 = _Resample(d2, d2 // 2)
        self.encoder_level3 = stage(d3, num_blocks[2], 16, mlp_ratio)
        self.down3_4 = _Resample(d3, d3 // 2)
        self.latent = stage(d4, num_blocks[3], 32, mlp_ratio)
        self.up4_3 = _Resample(d4, d4 * 2)
        self.reduce_chan_level3 = nn.Conv2d(d4, d3, 1, bias=bias)
        self.decoder_level3 = stage(d3, num_blocks[2], 16, mlp_ratio)
        self.up3_2 = _Resample(d3, d3 * 2)
        self.reduce_chan_level2 = nn.Conv2d(d3, d2, 1, bias=bias)
        self.decoder_level2 = stage(d2, num_blocks[1], 8, mlp_ratio)
        self.up2_1 = _Resample(d2, d2 * 2)
        self.decoder_level1 = stage(d2, num_blocks[0], 8, mlp_ratio)
        self.refinement = stage(d2, num_refinement_blocks, 8, mlp_ratio)
        self.dual_pixel_task = dual_pixel_task
        if dual_pixel_task:
            self.skip_conv = nn.Conv2d(d1, d2, 1, bias=bias)
        self.output = nn.Conv2d(d2, out_channels, 3, padding=1, bias=bias)
        self._init_host()
        self.max_tiles_per_batch = 4
        self.hip_graph = True      # the tiler replays the per-batch forward from a HIP graph (utils.graphed_forward)

    # ------------------------------------------------------------------ weights
    def load_synthetic(self, seed=42):
        from .. import synth
        shapes = {k: tuple(v.shape) for k, v in self.state_dict().items()}
        self.load_state_dict(synth.synth_state_dict(shapes, seed=seed, rules=SYNTH_RULES), strict=True)
        return self

    def _pack(self):
        key = self._param_key()
        if self._packed is not None and self._packed_key == key:
            return self._packed

        def f32(t):
            return None if t is None else t.detach().float().contiguous()

        pk = {name: pack_block(m) for name, m in self.named_modules() if isinstance(m, VSSBlock)}
        for name in ("down1_2", "down2_3", "down3_4", "up4_3", "up3_2", "up2_1"):
            pk[name] = _hip.pack_conv3x3(getattr(self, name).body[0].weight)
        pk["patch_embed"] = _hip.pack_conv3x3(self.patch_embed.proj.weight)
        pk["output"] = _hip.pack_conv3x3(self.output.weight)
        pk["output_b"] = f32(self.output.bias)
        for name in ("reduce_chan_level3", "reduce_chan_level2") + (("skip_conv",) if self.dual_pixel_task else ()):
            pk[name] = _hip.pack_gemm_weight(getattr(self, name).weight)
            pk[name + "_b"] = f32(getattr(self, name).bias)
        self._packed, self._packed_key = pk, key
        return pk

    def _run_stage(self, name, pk, x, ids):
        blocks = getattr(self, name)
        have = False
        for i, blk in enumerate(blocks):
            have = self._block(blk, pk[f"{name}.{i}"], x, ids, have, want_stats=i + 1 < len(blocks))

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, inp_img: torch.Tensor) -> torch.Tensor:
        if not inp_img.is_cuda:
            raise _hip.HipLibraryError("irm_amd MaIRUNet runs on the GPU only (no CPU fallback); "
                                       "move the model and input to 'cuda'")
        x = inp_img.float().contiguous()
        B, Cin, H, W = x.shape
        if H % 8 or W % 8:
            raise ValueError("MaIRUNet needs H and W to be multiples of 8 (the tiler pads, utils.pad)")
        dev = x.device
        pk = self._pack()
        d1, d2, d3, d4 = self.dim, self.dim * 2, self.dim * 4, self.dim * 8
        sz = [(H, W), (H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8)]
        ids = [scan_ids(h, w, self.scan_len, dev) for h, w in sz]

        def buf(name, ch, hw):
            return self._buf(name, B * ch * hw[0] * hw[1], dev).view(B, ch, hw[0], hw[1])

        cat1, cat2, cat3 = buf("cat1", 2 * d1, sz[0]), buf("cat2", 2 * d2, sz[1]), buf("cat3", 2 * d3, sz[2])
        lat, dec3, dec2 = buf("latent", d4, sz[3]), buf("dec3", d3, sz[2]), buf("dec2", d2, sz[1])
        e1 = cat1[:, d1:]
        ops.conv3x3(pk["patch_embed"], x, e1, Cin, d1)
        if self.dual_pixel_task:
            e1_in = buf("enc1_in", d1, sz[0])
            e1_in.copy_(e1)
        self._run_stage("encoder_level1", pk, e1, ids[0])
        e2 = cat2[:, d2:]
        ops.conv3x3(pk["down1_2"], e1, e2, d1, d1 // 2, store_mode=1)
        self._run_stage("encoder_level2", pk, e2, ids[1])
        e3 = cat3[:, d3:]
        ops.conv3x3(pk["down2_3"], e2, e3, d2, d2 // 2, store_mode=1)
        self._run_stage("encoder_level3", pk, e3, ids[2])
        ops.conv3x3(pk["down3_4"], e3, lat, d3, d3 // 2, store_mode=1)
        self._run_stage("latent", pk, lat, ids[3])
        ops.conv3x3(pk["up4_3"], lat, cat3[:, :d3], d4, d4 * 2, store_mode=2)
        ops.gemm1x1(pk["reduce_chan_level3"], cat3, dec3, d3, 2 * d3, bias=pk["reduce_chan_level3_b"])
        self._run_stage("decoder_level3", pk, dec3, ids[2])
        ops.conv3x3(pk["up3_2"], dec3, cat2[:, :d2], d3, d3 * 2, store_mode=2)
        ops.gemm1x1(pk["reduce_chan_level2"], cat2, dec2, d2, 2 * d2, bias=pk["reduce_chan_level2_b"])
        self._run_stage("decoder_level2", pk, dec2, ids[1])
        ops.conv3x3(pk["up2_1"], dec2, cat1[:, :d1], d2, d2 * 2, store_mode=2)
        self._run_stage("decoder_level1", pk, cat1, ids[0])
        self._run_stage("refinement", pk, cat1, ids[0])
        out = torch.empty(B, self.out_channels, H, W, dtype=torch.float32, device=dev)
        if self.dual_pixel_task:
            ops.gemm1x1(pk["skip_conv"], e1_in, cat1, d2, d1, res=cat1, bias=pk["skip_conv_b"])
            ops.conv3x3(pk["output"], cat1, out, d2, self.out_channels, bias=pk["output_b"])
        else:
            ops.conv3x3(pk["output"], cat1, out, d2, self.out_channels, bias=pk["output_b"], res=x, res_mode=1)
        return out
