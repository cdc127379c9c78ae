"""Loader with the reference's call surface (src/rednet/__init__.py:7-26)."""
import numpy as np
import torch

SYNTH_RULES = ((r"^deconv15\.weight$", "gain", 0.02),)

from .rednet import REDNet  # noqa: E402

__all__ = ["REDNet", "get_model", "SYNTH_RULES"]


def get_model(weights_path: str, device: torch.device):
    model = REDNet()
    state_dict = torch.load(weights_path, map_location="cpu", weights_only=True)
    model.load_state_dict(state_dict, strict=False)
    model.to(device)
    model.eval()
    print(f"Successfully loaded {np.sum([p.numel() for p in model.parameters()]):,} parameters from {weights_path}")
    return model
