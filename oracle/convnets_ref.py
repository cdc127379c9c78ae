"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatements (functional PyTorch fp32) of the two plain conv stacks of the
reference, driven by state dicts with the reference's parameter names.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def dncnn_forward(x, p):
    """src/dncnn/models/network_dncnn.py:40-71 built by basicblock.conv (:61-98)
    with act_mode='R' (src/dncnn/__init__.py:8): conv3x3+bias+ReLU ... conv3x3+bias,
    output x - n.  Sequential indices are 0,2,4,... (ReLU modules sit between)."""
    idx = sorted({int(k.split(".")[1]) for k in p if k.startswith("model.")})
    n = x
    for j, i in enumerate(idx):
        n = F.conv2d(n, p[f"model.{i}.weight"], p[f"model.{i}.bias"], padding=1)
        if j + 1 < len(idx):
            n = torch.relu(n)
    return x - n


def rednet_forward(x, p):
    """src/rednet/rednet.py:64-136: 15 conv+ReLU, 15 deconv(+ReLU), skip
    relu(d_i + c_{15-i}) after every odd deconv, final + x."""
    feats = []
    c = x
    for i in range(1, 16):
        c = torch.relu(F.conv2d(c, p[f"conv{i}.weight"], p[f"conv{i}.bias"], padding=1))
        feats.append(c)
    d = c
    for i in range(1, 15):
        d = torch.relu(F.conv_transpose2d(d, p[f"deconv{i}.weight"], p[f"deconv{i}.bias"], padding=1))
        if i % 2 == 1:
            d = torch.relu(d + feats[14 - i])      # c14, c12, ..., c2
    d = F.conv_transpose2d(d, p["deconv15.weight"], p["deconv15.bias"], padding=1)
    return d + x
