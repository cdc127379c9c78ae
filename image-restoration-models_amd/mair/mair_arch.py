"""Flat MaIR (colour Gaussian denoising) on MI355X - drop-in for src/mair/basicsr/archs/mair_arch.py:493-730
(denoising branch: `upsampler=None`, `resi_connection='1conv'`; same constructor keywords and state_dict
keys).  It reuses the VSSBlock driver of mairunet_arch.py (the reference's RMB is the same block with the MLP
called `conv_blk`); odd blocks of a group use the shifted scan tables (mair_arch.py:455, 379-382).

  (x - mean) * img_range -> conv3x3 conv_first -> LayerNorm (patch_embed.norm) -> groups of RMBs, each closed by
  conv3x3 + residual (RMG, :863-864) -> LayerNorm -> conv_after_body + conv_first output -> conv_last + input ->
  / img_range + mean

The two stand-alone LayerNorms run as an identity-weight irm_gemm1x1_f32 with the LN prologue (1 GFLOP per
128x128 tile, no extra kernel); the mean shifts run on irm_chan_norm_act_f32 with constant "statistics".
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _hip, ops
from . import SYNTH_RULES
from .mairunet_arch import MambaHost, VSSBlock, pack_block, scan_ids

RGB_MEAN = (0.4488, 0.4371, 0.4040)


class _Group(nn.Module):
    """RMG holder: residual_group.blocks.{i}, conv (mair_arch.py:793-864)."""

    def __init__(self, dim, depth, d_state, ssm_ratio, mlp_ratio):
        super().__init__()
        self.residual_group = nn.Module()
        self.residual_group.blocks = nn.ModuleList(
            [VSSBlock(dim, d_state, ssm_ratio, mlp_ratio, mlp_name="conv_blk") for _ in range(depth)])
        self.conv = nn.Conv2d(dim, dim, 3, 1, 1)


class _Norm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)


class MaIR(MambaHost):
    def __init__(self, img_size=64, patch_size=1, in_chans=3, embed_dim=60, depths=(6, 6, 6, 6), drop_rate=0., d_state=16,
                 ssm_ratio=1.5, drop_path_rate=0.1, norm_layer=nn.LayerNorm, patch_norm=True, use_checkpoint=False, upscale=2,
                 img_range=1., upsampler='pixelshuffledirect', resi_connection='1conv', dynamic_ids=False, scan_len=8,
                 mlp_ratio=2, **kwargs):
        super().__init__()
        if upsampler not in (None, '', 'None') or upscale != 1 or resi_connection != '1conv' or patch_size != 1:
            raise NotImplementedError("only the denoising configuration of MaIR (upsampler=None, upscale=1, "
                                      "resi_connection='1conv') is built in the MI355X path")
        self.in_chans, self.embed_dim, self.img_range, self.scan_len = in_chans, embed_dim, float(img_range), scan_len
        self.patch_norm = patch_norm
        self.conv_first = nn.Conv2d(in_chans, embed_dim, 3, 1, 1)
        self.patch_embed = _Norm(embed_dim) if patch_norm else nn.Module()
        self.layers = nn.ModuleList([_Group(embed_dim, d, d_state, ssm_ratio, mlp_ratio) for d in depths])
        self.norm = nn.LayerNorm(embed_dim)
        self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1)
        self.conv_last = nn.Conv2d(embed_dim, in_chans, 3, 1, 1)
        self._init_host()
        self.max_tiles_per_batch = 8
        self.hip_graph = True      # the tiler replays the per-batch forward from a HIP graph (utils.graphed_forward)

    def load_synthetic(self, seed=42):
        from .. import synth
        shapes = {k: tuple(v.shape) for k, v in self.state_dict().items()}
        self.load_state_dict(synth.synth_state_dict(shapes, seed=seed, rules=SYNTH_RULES), strict=True)
        return self

    def _pack(self):
        key = self._param_key()
        if self._packed is not None and self._packed_key == key:
            return self._packed
        f32 = lambda t: None if t is None else t.detach().float().contiguous()       # noqa: E731
        c3 = lambda conv: (_hip.pack_conv3x3(conv.weight), f32(conv.bias))     # noqa: E731
        dev = self.conv_first.weight.device
        E = self.embed_dim
        pk = {name: pack_block(m) for name, m in self.named_modules() if isinstance(m, VSSBlock)}
        pk["conv_first"], pk["conv_after_body"], pk["conv_last"] = c3(self.conv_first), c3(self.conv_after_body), c3(self.conv_last)
        for i, g in enumerate(self.layers):
            pk[f"layers.{i}.conv"] = c3(g.conv)
        pk["eye"] = _hip.pack_gemm_weight(torch.eye(E, device=dev))
        if self.patch_norm:
            pk["pn"] = (f32(self.patch_embed.norm.weight), f32(self.patch_embed.norm.bias))
        pk["fn"] = (f32(self.norm.weight), f32(self.norm.bias))
        mean = torch.tensor(RGB_MEAN if self.in_chans == 3 else [0.0] * self.in_chans, dtype=torch.float32, device=dev)
        r = self.img_range
        # chan_norm_act computes (x - m) * s: shift in = (x - mean) * r ; shift out = (y + mean * r) / r
        pk["shift_in"] = torch.stack([mean, torch.full_like(mean, r)], dim=1).contiguous()
        pk["shift_out"] = torch.stack([-mean * r, torch.full_like(mean, 1.0 / r)], dim=1).contiguous()
        self._packed, self._packed_key = pk, key
        return pk

    def _layer_norm(self, x, y, wb, pk):
        B, C, H, W = x.shape
        stats = self._buf("stats", B * 2 * H * W, x.device)
        ops.ln_stats(x, stats)
        ops.gemm1x1(pk["eye"], x, y, C, C, stats=stats, lnw=wb[0], lnb=wb[1], ln_mode=ops.LN_WITHBIAS)

    @torch.no_grad()
    def forward(self, inp: torch.Tensor) -> torch.Tensor:
        if not inp.is_cuda:
            raise _hip.HipLibraryError("irm_amd MaIR runs on the GPU only (no CPU fallback); move the model and input to 'cuda'")
        x = inp.float().contiguous()
        B, Cin, H, W = x.shape
        dev, E = x.device, self.embed_dim
        pk = self._pack()
        ids = (scan_ids(H, W, self.scan_len, dev), scan_ids(H, W, self.scan_len, dev, self.scan_len // 2))
        new = lambda c: torch.empty(B, c, H, W, dtype=torch.float32, device=dev)      # noqa: E731
        xin = new(Cin)
        ops.chan_norm_act(x, pk["shift_in"].unsqueeze(0).expand(B, -1, -1).contiguous(), xin)
        first = new(E)
        ops.conv3x3(pk["conv_first"][0], xin, first, Cin, E, bias=pk["conv_first"][1])
        t = new(E)
        if self.patch_norm:
            self._layer_norm(first, t, pk["pn"], pk)
        else:
            t.copy_(first)
        g_in = new(E)
        for li, grp in enumerate(self.layers):
            g_in.copy_(t)
            for bi, blk in enumerate(grp.residual_group.blocks):
                self._block(blk, pk[f"layers.{li}.residual_group.blocks.{bi}"], t, ids[bi % 2])
            nxt = new(E)
            ops.conv3x3(pk[f"layers.{li}.conv"][0], t, nxt, E, E, bias=pk[f"layers.{li}.conv"][1], res=g_in, res_mode=1)
            t = nxt
        tn = new(E)
        self._layer_norm(t, tn, pk["fn"], pk)
        res = new(E)
        ops.conv3x3(pk["conv_after_body"][0], tn, res, E, E, bias=pk["conv_after_body"][1], res=first, res_mode=1)
        out = new(Cin)
        ops.conv3x3(pk["conv_last"][0], res, out, E, Cin, bias=pk["conv_last"][1], res=xin, res_mode=1)
        ops.chan_norm_act(out, pk["shift_out"].unsqueeze(0).expand(B, -1, -1).contiguous(), out)
        return out
