"""DnCNN forward on MI355X (drop-in for src/dncnn/models/network_dncnn.py:40-71).

Same constructor and state_dict keys as the reference (``model.<2i>.weight/bias``:
B.conv with mode 'C'+'R' puts a ReLU module after every conv but the last,
basicblock.py:61-98).  Every layer is one irm_conv3x3_f32 launch with bias+ReLU
in the epilogue; the last layer also folds the residual ``x - n``."""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import _hip
from ...convnet_common import PackedCache, conv3x3, require_cuda


class DnCNN(nn.Module):
    def __init__(self, in_nc=1, out_nc=1, nc=64, nb=17, act_mode='BR'):
        super().__init__()
        if 'B' in act_mode:
            raise NotImplementedError("inference uses BN-merged weights (act_mode='R', src/dncnn/__init__.py:8)")
        assert 'R' in act_mode, 'only ReLU activation is used by the reference loader'
        self.in_nc, self.out_nc, self.nc, self.nb = in_nc, out_nc, nc, nb
        layers = []
        chans = [in_nc] + [nc] * (nb - 1) + [out_nc]
        for i in range(nb):
            layers.append(nn.Conv2d(chans[i], chans[i + 1], 3, padding=1, bias=True))
            if i + 1 < nb:
                layers.append(nn.ReLU(inplace=True))
        self.model = nn.Sequential(*layers)      # parameter holder: never called
        self._cache = PackedCache(self, self._build)
        self.hip_graph = True      # the tiler replays the per-batch forward from a HIP graph (utils.graphed_forward)
        self._ws = {}

    def _convs(self):
        return [m for m in self.model if isinstance(m, nn.Conv2d)]

    def _build(self):
        return [(_hip.pack_conv3x3(m.weight), m.bias.detach().float().contiguous(),
                 m.in_channels, m.out_channels) for m in self._convs()]

    def load_synthetic(self, seed=42):
        from ... import synth
        from .. import SYNTH_RULES
        shapes = {k: tuple(v.shape) for k, v in self.state_dict().items()}
        self.load_state_dict(synth.synth_state_dict(shapes, seed=seed, rules=SYNTH_RULES), strict=True)
        return self

    def release_workspace(self):
        self._ws = {}

    @torch.no_grad()
    def forward(self, x):
        require_cuda(x, "DnCNN")
        x = x.float().contiguous()
        B, _, H, W = x.shape
        layers = self._cache.get()
        key = (B, H, W, str(x.device))
        if self._ws.get("key") != key:
            self._ws = {"key": key,
                        "a": torch.empty(B, self.nc, H, W, dtype=torch.float32, device=x.device),
                        "b": torch.empty(B, self.nc, H, W, dtype=torch.float32, device=x.device)}
        cur, nxt = x, self._ws["a"]
        out = torch.empty(B, self.out_nc, H, W, dtype=torch.float32, device=x.device)
        for i, (wp, bias, ci, co) in enumerate(layers):
            if i + 1 < len(layers):
                conv3x3(wp, cur, nxt, ci, co, bias=bias, relu1=True)
                cur, nxt = nxt, (self._ws["b"] if nxt is self._ws["a"] else self._ws["a"])
            else:
                conv3x3(wp, cur, out, ci, co, bias=bias, res=x, res_mode=2)    # x - n
        return out
