"""MaIR (U-shaped Mamba) on MI355X - loader with the reference's call surface (src/mair/__init__.py:62-74)."""

#: synthetic-weight rules (synth.py) in the spirit of the reference's initialisers (mairunet_arch.py:176-224)
SYNTH_RULES = (
    (r"A_logs$", "range", (0.0, 1.5)),
    (r"Ds$", "range", (0.5, 1.5)),
    (r"dt_projs_bias$", "range", (-4.0, -2.0)),
    (r"^output\.weight$", "gain", 0.02),
    (r"^conv_last\.weight$", "gain", 0.05),
    (r"gating\.gating\.1\.weight$", "gain", 2.0),
)

import os as _os

import numpy as _np
import torch as _torch
import yaml as _yaml

from .mairunet_arch import MaIRUNet  # noqa: E402
from .mair_arch import MaIR  # noqa: E402

__all__ = ["MaIRUNet", "MaIR", "get_model", "SYNTH_RULES"]


def get_model(opt_path: str):
    """yml -> network_g -> MaIRUNet, weights from path.pretrain_network_g under key 'params' with an
    optional 'module.' prefix (BasicSR load_network, base_model.py:277-304), eval mode, on the GPU iff
    num_gpu != 0 and one is present (base_model.py:18).  No device argument, like the reference."""
    with open(opt_path, mode="r") as f:
        opt = _yaml.safe_load(f)
    net_opt = dict(opt["network_g"])
    kind = net_opt.pop("type")
    if kind not in ("MaIRUNet", "MaIR"):
        raise NotImplementedError(f"network type {kind} is not built in the MI355X path")
    model = MaIRUNet(**net_opt) if kind == "MaIRUNet" else MaIR(**net_opt)
    weights_path = _os.path.expanduser(opt["path"]["pretrain_network_g"])
    ckpt = _torch.load(weights_path, map_location="cpu", weights_only=True)
    sd = ckpt["params"] if "params" in ckpt else ckpt
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=bool(opt["path"].get("strict_load_g", True)))
    if opt.get("num_gpu", 1) != 0 and _torch.cuda.is_available():
        model.to("cuda")
    model.eval()
    print(f"Successfully loaded {_np.sum([p.numel() for p in model.parameters()]):,} parameters from {weights_path}")
    return model
