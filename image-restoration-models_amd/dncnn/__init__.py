"""Loader with the reference's call surface (src/dncnn/__init__.py:7-15)."""
import numpy as np
import torch

#: synthetic-weight rules: small last layer so x - n stays near the image range
SYNTH_RULES = ((r"^model\.\d+\.weight$", "gain", 1.15),)

from .models.network_dncnn import DnCNN  # noqa: E402

__all__ = ["DnCNN", "get_model", "SYNTH_RULES"]


def get_model(weights_path: str, n_channels: int, nb: int, device: torch.device):
    model = DnCNN(in_nc=n_channels, out_nc=n_channels, nc=64, nb=nb, act_mode='R')
    model.load_state_dict(torch.load(weights_path, map_location="cpu", weights_only=True), strict=True)
    model.eval()
    model.to(device)
    print(f"Successfully loaded {np.sum([p.numel() for p in model.parameters()]):,} parameters from {weights_path}")
    return model
