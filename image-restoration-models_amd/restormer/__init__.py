"""Loader with the reference's call surface (src/restormer/__init__.py:8-20):
``get_model(opt_path, device)`` reads ``network_g`` (minus ``type``) and
``path.pretrain_network_g`` from the yml, loads ``checkpoint['params']`` and
returns the eval-mode model on ``device``."""
import numpy as np
import torch
import yaml

from .restormer import Restormer

__all__ = ["Restormer", "get_model"]


def get_model(opt_path: str, device: torch.device):
    with open(opt_path, mode="r") as f:
        opt = yaml.safe_load(f)
    net_opt = dict(opt["network_g"])
    net_opt.pop("type", None)
    model = Restormer(**net_opt)
    weights_path = opt["path"]["pretrain_network_g"]
    # weights_only: nothing in the checkpoint is executed (FileNotFoundError propagates like the reference)
    checkpoint = torch.load(weights_path, map_location="cpu", weights_only=True)
    model.load_state_dict(checkpoint["params"])
    model.to(device)
    model.eval()
    print(f"Successfully loaded {np.sum([p.numel() for p in model.parameters()]):,} parameters from {weights_path}")
    return model
