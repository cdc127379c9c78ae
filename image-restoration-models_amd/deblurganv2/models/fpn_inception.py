"""FPN-Inception generator of DeblurGANv2 on MI355X: the in-tree part of
src/deblurganv2/models/fpn_inception.py - lateral 1x1 convs, reflect pads and the top-down path of `FPN`
(:142-170), the four heads, smoothing convs and the tanh/clamp output of `FPNInception.forward` (:65-81) - with the
reference's state_dict keys for those layers.

The bottom-up encoder is `timm.create_model('inception_resnet_v2')` (fpn_inception.py:94): third-party code that is
not vendored in the reference and not installed here.  It is therefore NOT rebuilt (and its arithmetic cannot be
pinned); `forward` takes the five encoder feature maps as inputs:

    y = model(x, enc0, enc1, enc2, enc3, enc4)      # enc_i = fpn.enc_i(...) of the reference, float32 on the GPU

Norm layers run in TRAIN mode like the reference's generators (src/deblurganv2/__init__.py:38): InstanceNorm2d
(affine=False) with the statistics of the current tile.  Reflect padding of the small lateral maps is tensor plumbing
(torch.nn.functional.pad on the device); every conv / norm / up-sample runs in libirm_hip.so.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import _hip, ops
from ...convnet_common import PackedCache, require_cuda
from .fpn_mobilenet import FPNHead

ENC_CHANNELS = (32, 64, 192, 1088, 2080)          # conv2d_1a, maxpool_3a, maxpool_5a, mixed_6a, mixed_7a outputs


class _FPNTop(nn.Module):
    """Holder for the in-tree layers of `FPN` (keys fpn.td1.0.weight, fpn.lateral4.weight, ...)."""

    def __init__(self, norm_layer, num_filters=256):
        super().__init__()
        for name in ("td1", "td2", "td3"):
            setattr(self, name, nn.Sequential(nn.Conv2d(num_filters, num_filters, 3, padding=1), norm_layer(num_filters),
                                              nn.ReLU(inplace=True)))
        self.lateral4 = nn.Conv2d(ENC_CHANNELS[4], num_filters, 1, bias=False)
        self.lateral3 = nn.Conv2d(ENC_CHANNELS[3], num_filters, 1, bias=False)
        self.lateral2 = nn.Conv2d(ENC_CHANNELS[2], num_filters, 1, bias=False)
        self.lateral1 = nn.Conv2d(ENC_CHANNELS[1], num_filters, 1, bias=False)
        self.lateral0 = nn.Conv2d(ENC_CHANNELS[0], num_filters // 2, 1, bias=False)


class FPNInceptionDecoder(nn.Module):
    def __init__(self, norm_layer=None, output_ch=3, num_filters=128, num_filters_fpn=256):
        super().__init__()
        if norm_layer is None:
            import functools
            norm_layer = functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=True)   # networks.py:22
        self.nf, self.nfpn = num_filters, num_filters_fpn
        self.fpn = _FPNTop(norm_layer, num_filters_fpn)
        for i in (1, 2, 3, 4):
            setattr(self, f"head{i}", FPNHead(num_filters_fpn, num_filters, num_filters))
        self.smooth = nn.Sequential(nn.Conv2d(4 * num_filters, num_filters, 3, padding=1), norm_layer(num_filters), nn.ReLU())
        self.smooth2 = nn.Sequential(nn.Conv2d(num_filters, num_filters // 2, 3, padding=1), norm_layer(num_filters // 2),
                                     nn.ReLU())
        self.final = nn.Conv2d(num_filters // 2, output_ch, 3, padding=1)
        self._cache = PackedCache(self, self._build)

    def load_synthetic(self, seed=42):
        from ... import synth
        from .. import SYNTH_RULES
        shapes = {k: tuple(v.shape) for k, v in self.state_dict().items()
                  if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
        self.load_state_dict(synth.synth_state_dict(shapes, seed=seed, rules=SYNTH_RULES), strict=False)
        return self

    def _build(self):
        f32 = lambda t: None if t is None else t.detach().float().contiguous()      # noqa: E731
        c3 = lambda conv: (_hip.pack_conv3x3(conv.weight), f32(conv.bias))    # noqa: E731
        pk = {f"lateral{i}": _hip.pack_gemm_weight(getattr(self.fpn, f"lateral{i}").weight) for i in range(5)}
        for n in ("td1", "td2", "td3"):
            pk[n] = c3(getattr(self.fpn, n)[0])
        for i in (1, 2, 3, 4):
            h = getattr(self, f"head{i}")
            pk[f"head{i}"] = (c3(h.block0)[0], c3(h.block1)[0])
        pk["smooth"], pk["smooth2"], pk["final"] = c3(self.smooth[0]), c3(self.smooth2[0]), c3(self.final)
        return pk

    @torch.no_grad()
    def forward(self, x, enc0, enc1, enc2, enc3, enc4):
        require_cuda(x, "FPNInceptionDecoder")
        x = x.float().contiguous()
        B, _, H, W = x.shape
        dev = x.device
        pk = self._cache.get()
        Fp, nf = self.nfpn, self.nf
        new = lambda c, h, w: torch.empty(B, c, h, w, dtype=torch.float32, device=dev)     # noqa: E731
        rpad = lambda t, p: torch.nn.functional.pad(t, p, "reflect").contiguous()          # noqa: E731

        def norm(t):
            st = torch.empty(B, t.shape[1], 2, dtype=torch.float32, device=dev)
            ops.chan_stats(t, st)
            ops.chan_norm_act(t, st, t, act=ops.ACT_RELU)
            return t

        def lateral(i, e):
            e = e.float().contiguous()
            co = Fp // 2 if i == 0 else Fp
            t = new(co, e.shape[2], e.shape[3])
            ops.gemm1x1(pk[f"lateral{i}"], e, t, co, e.shape[1])
            return t

        # lateral connections + reflect pads (fpn_inception.py:153-163)
        l4 = rpad(lateral(4, enc4), (1, 1, 1, 1))
        l3 = rpad(lateral(3, enc3), (1, 1, 1, 1))
        l2 = rpad(lateral(2, enc2), (1, 2, 1, 2))
        l1 = rpad(lateral(1, enc1), (1, 1, 1, 1))
        map0 = rpad(lateral(0, enc0), (0, 1, 0, 1))

        def td(lat, top, name):                       # td(lateral + up2(top))   (:166-168)
            s = torch.empty_like(lat)
            ops.upsample_add(top, s, 2, add=lat)
            o = torch.empty_like(lat)
            ops.conv3x3(pk[name][0], s, o, Fp, Fp, bias=pk[name][1])
            return norm(o)
        map4 = l4
        map3 = td(l3, map4, "td1")
        map2 = td(l2, map3, "td2")
        map1 = td(l1, map2, "td3")
        # heads, nearest up-sampling into the concat buffer [map4 | map3 | map2 | map1] (:68-73)
        h4, w4 = map1.shape[2], map1.shape[3]
        cat = new(4 * nf, h4, w4)
        for slot, (m, name, scale) in enumerate(((map4, "head4", 8), (map3, "head3", 4), (map2, "head2", 2), (map1, "head1", 1))):
            a = new(nf, m.shape[2], m.shape[3])
            ops.conv3x3(pk[name][0], m, a, Fp, nf, relu1=True)
            dst = cat[:, slot * nf:(slot + 1) * nf]
            if scale == 1:
                ops.conv3x3(pk[name][1], a, dst, nf, nf, relu1=True)
            else:
                b2 = new(nf, m.shape[2], m.shape[3])
                ops.conv3x3(pk[name][1], a, b2, nf, nf, relu1=True)
                ops.upsample_add(b2, dst, scale)
        sm = new(nf, h4, w4)
        ops.conv3x3(pk["smooth"][0], cat, sm, 4 * nf, nf, bias=pk["smooth"][1])
        norm(sm)
        s2 = new(nf, 2 * h4, 2 * w4)
        ops.upsample_add(sm, s2, 2, add=map0)                                      # (:75-76)
        sm2 = new(nf // 2, 2 * h4, 2 * w4)
        ops.conv3x3(pk["smooth2"][0], s2, sm2, nf, nf // 2, bias=pk["smooth2"][1])
        norm(sm2)
        if (4 * h4, 4 * w4) != (H, W):
            raise ValueError(f"encoder maps of a {4 * h4}x{4 * w4} image do not belong to this {H}x{W} input")
        up = new(nf // 2, H, W)
        ops.upsample_add(sm2, up, 2)
        out = new(x.shape[1], H, W)
        ops.conv3x3(pk["final"][0], up, out, nf // 2, x.shape[1], bias=pk["final"][1], res=x, res_mode=3)   # clamp(tanh(.) + x)
        return out
