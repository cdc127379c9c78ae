"""REDNet forward on MI355X (drop-in for src/rednet/rednet.py:15-136).

Same state_dict keys (conv1..15, deconv1..15).  ConvTranspose2d(k=3,s=1,p=1) is
run as a conv3x3 with the weight transposed and flipped at pack time; the
symmetric skips ``relu(relu(deconv) + c)`` and the final ``+ x`` are conv epilogues."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _hip
from ..convnet_common import PackedCache, conv3x3, require_cuda


class REDNet(nn.Module):
    def __init__(self, num_channels=1, num_features=128):
        super().__init__()
        self.num_channels, self.num_features = num_channels, num_features
        for i in range(1, 16):
            setattr(self, f"conv{i}", nn.Conv2d(num_channels if i == 1 else num_features, num_features, 3, padding=1))
        for i in range(1, 16):
            setattr(self, f"deconv{i}", nn.ConvTranspose2d(num_features, num_channels if i == 15 else num_features,
                                                           3, padding=1))
        self._cache = PackedCache(self, self._build)
        self.hip_graph = True      # the tiler replays the per-batch forward from a HIP graph (utils.graphed_forward)

    def _build(self):
        enc = [(_hip.pack_conv3x3(m.weight), m.bias.detach().float().contiguous(), m.in_channels, m.out_channels)
               for m in (getattr(self, f"conv{i}") for i in range(1, 16))]
        dec = [(_hip.pack_conv3x3(_hip.deconv_as_conv_weight(m.weight)), m.bias.detach().float().contiguous(),
                m.in_channels, m.out_channels) for m in (getattr(self, f"deconv{i}") for i in range(1, 16))]
        return enc, dec

    def load_synthetic(self, seed=42):
        from .. import synth
        from . import SYNTH_RULES
        shapes = {k: tuple(v.shape) for k, v in self.state_dict().items()}
        self.load_state_dict(synth.synth_state_dict(shapes, seed=seed, rules=SYNTH_RULES), strict=True)
        return self

    @torch.no_grad()
    def forward(self, x):
        require_cuda(x, "REDNet")
        x = x.float().contiguous()
        B, _, H, W = x.shape
        enc, dec = self._cache.get()
        F = self.num_features

        def new(c=F):
            return torch.empty(B, c, H, W, dtype=torch.float32, device=x.device)

        feats, cur = [], x
        for (wp, b, ci, co) in enc:                       # c1..c15 (rednet.py:66-80)
            nxt = new()
            conv3x3(wp, cur, nxt, ci, co, bias=b, relu1=True)
            feats.append(nxt)
            cur = nxt
        d = cur
        for i in range(1, 15):                            # deconv1..14 (rednet.py:84-130)
            wp, b, ci, co = dec[i - 1]
            nxt = new()
            if i % 2 == 1:                                # relu(relu(deconv) + c_{15-i})
                conv3x3(wp, d, nxt, ci, co, bias=b, relu1=True, res=feats[14 - i], res_mode=1, relu2=True)
            else:
                conv3x3(wp, d, nxt, ci, co, bias=b, relu1=True)
            d = nxt
        wp, b, ci, co = dec[14]
        out = new(self.num_channels)
        conv3x3(wp, d, out, ci, co, bias=b, res=x, res_mode=1)      # d15 + x (rednet.py:133-136)
        return out
