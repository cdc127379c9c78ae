"""FPN-MobileNet generator of DeblurGANv2 on MI355X (drop-in for
src/deblurganv2/models/fpn_mobilenet.py + models/mobilenet_v2.py; same state_dict keys incl. the
`fpn.enc0..enc4` aliases of `fpn.features[...]` and the norm layers' running-stat buffers).

The reference runs this generator in TRAIN mode (src/deblurganv2/__init__.py:38) one tile at a time, so
BatchNorm2d and InstanceNorm2d both normalise with the statistics of the current tile; here that is
chan_stats + chan_norm_act per (sample, channel), which also keeps batched tiles independent.  (Running
statistics are not updated: they never influence the output in this mode.)

Pipeline per InvertedResidual (mobilenet_v2.py:24-56): gemm1x1 -> norm+ReLU6 -> dwconv3x3 (stride 1/2) ->
norm+ReLU6 -> gemm1x1 -> norm (+ skip).  FPN / heads / smoothing (fpn_mobilenet.py:53-70, 121-146): 1x1
laterals, nearest up-sample + add, conv3x3 (+bias) -> InstanceNorm -> ReLU, conv3x3+ReLU heads written
into slices of the concat buffer, final conv3x3 with the tanh(.) + x, clamp epilogue.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import _hip, ops
from ...convnet_common import PackedCache, require_cuda
from .. import SYNTH_RULES

_SETTING = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]


def _conv_bn(inp, oup, stride):
    return nn.Sequential(nn.Conv2d(inp, oup, 3, stride, 1, bias=False), nn.BatchNorm2d(oup), nn.ReLU6(inplace=True))


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        self.inp, self.oup, self.stride, self.t = inp, oup, stride, expand_ratio
        hid = round(inp * expand_ratio)
        self.hidden = hid
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers += [nn.Conv2d(inp, hid, 1, 1, 0, bias=False), nn.BatchNorm2d(hid), nn.ReLU6(inplace=True)]
        layers += [nn.Conv2d(hid, hid, 3, stride, 1, groups=hid, bias=False), nn.BatchNorm2d(hid), nn.ReLU6(inplace=True),
                   nn.Conv2d(hid, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)


def _mobilenet_features():
    feats, inp = [_conv_bn(3, 32, 2)], 32
    for t, c, n, s in _SETTING:
        for i in range(n):
            feats.append(InvertedResidual(inp, c, s if i == 0 else 1, t))
            inp = c
    feats.append(nn.Sequential(nn.Conv2d(inp, 1280, 1, 1, 0, bias=False), nn.BatchNorm2d(1280), nn.ReLU6(inplace=True)))
    return nn.Sequential(*feats)          # features[16:] exist in the checkpoint but are unused by the FPN


class FPNHead(nn.Module):
    def __init__(self, num_in, num_mid, num_out):
        super().__init__()
        self.block0 = nn.Conv2d(num_in, num_mid, 3, padding=1, bias=False)
        self.block1 = nn.Conv2d(num_mid, num_out, 3, padding=1, bias=False)


class FPN(nn.Module):
    def __init__(self, norm_layer, num_filters=128):
        super().__init__()
        self.features = _mobilenet_features()
        self.enc0 = nn.Sequential(*self.features[0:2])
        self.enc1 = nn.Sequential(*self.features[2:4])
        self.enc2 = nn.Sequential(*self.features[4:7])
        self.enc3 = nn.Sequential(*self.features[7:11])
        self.enc4 = nn.Sequential(*self.features[11:16])
        for name in ("td1", "td2", "td3"):
            setattr(self, name, nn.Sequential(nn.Conv2d(num_filters, num_filters, 3, padding=1), norm_layer(num_filters),
                                              nn.ReLU(inplace=True)))
        self.lateral4 = nn.Conv2d(160, num_filters, 1, bias=False)
        self.lateral3 = nn.Conv2d(64, num_filters, 1, bias=False)
        self.lateral2 = nn.Conv2d(32, num_filters, 1, bias=False)
        self.lateral1 = nn.Conv2d(24, num_filters, 1, bias=False)
        self.lateral0 = nn.Conv2d(16, num_filters // 2, 1, bias=False)


class FPNMobileNet(nn.Module):
    def __init__(self, norm_layer=None, output_ch=3, num_filters=64, num_filters_fpn=128, pretrained=False):
        super().__init__()
        if norm_layer is None:
            import functools
            norm_layer = functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=True)   # networks.py:22
        self.nf, self.nfpn = num_filters, num_filters_fpn
        self.fpn = FPN(norm_layer, num_filters_fpn)
        for i in (1, 2, 3, 4):
            setattr(self, f"head{i}", FPNHead(num_filters_fpn, num_filters, num_filters))
        self.smooth = nn.Sequential(nn.Conv2d(4 * num_filters, num_filters, 3, padding=1), norm_layer(num_filters), nn.ReLU())
        self.smooth2 = nn.Sequential(nn.Conv2d(num_filters, num_filters // 2, 3, padding=1), norm_layer(num_filters // 2),
                                     nn.ReLU())
        self.final = nn.Conv2d(num_filters // 2, output_ch, 3, padding=1)
        self._cache = PackedCache(self, self._build)
        self.max_tiles_per_batch = 2
        self.hip_graph = True      # the tiler replays the per-batch forward from a HIP graph (utils.graphed_forward)

    # ------------------------------------------------------------------ weights
    def load_synthetic(self, seed=42):
        from ... import synth
        shapes = {k: tuple(v.shape) for k, v in self.state_dict().items()
                  if not (k.endswith(("running_mean", "running_var", "num_batches_tracked")) or k.startswith("fpn.enc"))}
        self.load_state_dict(synth.synth_state_dict(shapes, seed=seed, rules=SYNTH_RULES), strict=False)
        return self

    def _build(self):
        f32 = lambda t: None if t is None else t.detach().float().contiguous()      # noqa: E731
        g1 = lambda conv: _hip.pack_gemm_weight(conv.weight)                         # noqa: E731
        c3 = lambda conv: (_hip.pack_conv3x3(conv.weight), f32(conv.bias))    # noqa: E731
        pk = {"stem_w": f32(self.fpn.features[0][0].weight), "stem_bn": (f32(self.fpn.features[0][1].weight),
                                                                          f32(self.fpn.features[0][1].bias))}
        blocks = []
        for blk in list(self.fpn.features)[1:16]:
            L = list(blk.conv)
            d = {}
            i = 0
            if blk.t != 1:
                d["pw"], d["pw_bn"] = g1(L[0]), (f32(L[1].weight), f32(L[1].bias))
                i = 3
            d["dw"], d["dw_bn"] = f32(L[i].weight.reshape(-1, 9)), (f32(L[i + 1].weight), f32(L[i + 1].bias))
            d["pwl"], d["pwl_bn"] = g1(L[i + 3]), (f32(L[i + 4].weight), f32(L[i + 4].bias))
            blocks.append(d)
        pk["blocks"] = blocks
        for i in range(5):
            pk[f"lateral{i}"] = g1(getattr(self.fpn, f"lateral{i}"))
        for n in ("td1", "td2", "td3"):
            pk[n] = c3(getattr(self.fpn, n)[0])
        for i in (1, 2, 3, 4):
            h = getattr(self, f"head{i}")
            pk[f"head{i}"] = (c3(h.block0)[0], c3(h.block1)[0])
        pk["smooth"], pk["smooth2"], pk["final"] = c3(self.smooth[0]), c3(self.smooth2[0]), c3(self.final)
        return pk

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x):
        require_cuda(x, "FPNMobileNet")
        x = x.float().contiguous()
        B, _, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError("FPNMobileNet needs H and W to be multiples of 32 (deblurganv2.pad)")
        dev = x.device
        pk = self._cache.get()
        new = lambda c, h, w: torch.empty(B, c, h, w, dtype=torch.float32, device=dev)     # noqa: E731

        def norm(t, wb, act, res=None):
            st = torch.empty(B, t.shape[1], 2, dtype=torch.float32, device=dev)
            ops.chan_stats(t, st)
            ops.chan_norm_act(t, st, t, weight=wb[0] if wb else None, bias=wb[1] if wb else None, res=res, act=act)
            return t

        # ---- MobileNetV2 features[0:16] (mobilenet_v2.py:79-94)
        h, w = H // 2, W // 2
        y = new(32, h, w)
        ops.conv3x3_s2(x, pk["stem_w"], y, 3, 32)
        norm(y, pk["stem_bn"], ops.ACT_RELU6)
        enc = {}
        for idx, (blk, d) in enumerate(zip(list(self.fpn.features)[1:16], pk["blocks"]), start=1):
            inp = y
            t = inp
            if blk.t != 1:
                t = new(blk.hidden, h, w)
                ops.gemm1x1(d["pw"], inp, t, blk.hidden, blk.inp)
                norm(t, d["pw_bn"], ops.ACT_RELU6)
            if blk.stride == 2:
                h, w = h // 2, w // 2
                u = new(blk.hidden, h, w)
                ops.dwconv3x3_s2(t, d["dw"], u)
            else:
                u = new(blk.hidden, h, w)
                ops.dwconv3x3(t, d["dw"], u)
            norm(u, d["dw_bn"], ops.ACT_RELU6)
            o = new(blk.oup, h, w)
            ops.gemm1x1(d["pwl"], u, o, blk.oup, blk.hidden)
            norm(o, d["pwl_bn"], ops.ACT_NONE, res=inp if blk.use_res_connect else None)
            y = o
            enc[idx] = o
        e0, e1, e2, e3, e4 = enc[1], enc[3], enc[6], enc[10], enc[15]
        # ---- FPN (fpn_mobilenet.py:121-146)
        F, nf = self.nfpn, self.nf
        lat = []
        for i, e in enumerate((e0, e1, e2, e3, e4)):
            co = F // 2 if i == 0 else F
            t = new(co, e.shape[2], e.shape[3])
            ops.gemm1x1(pk[f"lateral{i}"], e, t, co, e.shape[1])
            lat.append(t)

        def td(lateral, top, name):
            s = torch.empty_like(lateral)
            ops.upsample_add(top, s, 2, add=lateral)
            o = torch.empty_like(lateral)
            ops.conv3x3(pk[name][0], s, o, F, F, bias=pk[name][1])
            return norm(o, None, ops.ACT_RELU)
        map4 = lat[4]
        map3 = td(lat[3], map4, "td1")
        map2 = td(lat[2], map3, "td2")
        map1 = td(lat[1], map2, "td3")
        # ---- heads + smoothing (fpn_mobilenet.py:53-70)
        h4, w4 = H // 4, W // 4
        cat = new(4 * nf, h4, w4)
        for slot, (m, name, scale) in enumerate(((map4, "head4", 8), (map3, "head3", 4), (map2, "head2", 2), (map1, "head1", 1))):
            a = new(nf, m.shape[2], m.shape[3])
            ops.conv3x3(pk[name][0], m, a, F, nf, relu1=True)
            dst = cat[:, slot * nf:(slot + 1) * nf]
            if scale == 1:
                ops.conv3x3(pk[name][1], a, dst, nf, nf, relu1=True)
            else:
                b2 = new(nf, m.shape[2], m.shape[3])
                ops.conv3x3(pk[name][1], a, b2, nf, nf, relu1=True)
                ops.upsample_add(b2, dst, scale)
        sm = new(nf, h4, w4)
        ops.conv3x3(pk["smooth"][0], cat, sm, 4 * nf, nf, bias=pk["smooth"][1])
        norm(sm, None, ops.ACT_RELU)
        s2 = new(nf, H // 2, W // 2)
        ops.upsample_add(sm, s2, 2, add=lat[0])
        sm2 = new(nf // 2, H // 2, W // 2)
        ops.conv3x3(pk["smooth2"][0], s2, sm2, nf, nf // 2, bias=pk["smooth2"][1])
        norm(sm2, None, ops.ACT_RELU)
        up = new(nf // 2, H, W)
        ops.upsample_add(sm2, up, 2)
        out = new(3, H, W)
        ops.conv3x3(pk["final"][0], up, out, nf // 2, 3, bias=pk["final"][1], res=x, res_mode=3)   # clamp(tanh(.) + x)
        return out
