"""DeblurGANv2 (FPN-MobileNet generator) on MI355X - call surface of src/deblurganv2/__init__.py:11-41."""

#: synthetic-weight rules (synth.py)
SYNTH_RULES = (
    (r"^final\.weight$", "gain", 0.3),
    (r"features\.\d+\.(\d+\.)?(conv\.)?\d+\.weight$", "gain", 1.4),
)

import os as _os

import numpy as _np
import torch as _torch

from .models.fpn_mobilenet import FPNMobileNet  # noqa: E402
from .models.fpn_inception import FPNInceptionDecoder  # noqa: E402

__all__ = ["FPNMobileNet", "FPNInceptionDecoder", "get_model", "normalize", "pad", "postprocess", "SYNTH_RULES"]


def normalize(x: _np.ndarray):
    """albumentations Normalize(mean=.5, std=.5) (aug.py:31-39): (x - 127.5) * (1/127.5) in float32.
    Restated from the library's published arithmetic (albumentations is not installed: unpinned)."""
    mean = _np.float32(0.5) * _np.float32(255.0)
    denom = _np.float32(1.0) / (_np.float32(0.5) * _np.float32(255.0))
    return ((x.astype(_np.float32) - mean) * denom).astype(_np.float32)


def pad(x: _torch.Tensor):
    """src/deblurganv2/__init__.py:16-24: zero pad to (h//32+1)*32 - always at least one more block."""
    h, w = x.shape[-2:]
    return _torch.nn.functional.pad(x, (0, (w // 32 + 1) * 32 - w, 0, (h // 32 + 1) * 32 - h), 'constant', 0)


def postprocess(x: _torch.Tensor):
    return (x + 1) / 2.0


def get_model(weights_path: str, device: _torch.device):
    """src/deblurganv2/__init__.py:31-41: generator name = file stem; checkpoint['model'] carries the
    DataParallel 'module.' prefix; the model is returned in TRAIN mode like the reference."""
    name = _os.path.basename(weights_path).split('.')[0]
    if name != 'fpn_mobilenet':
        raise NotImplementedError(f"generator {name!r}: only fpn_mobilenet is built end to end in the MI355X path.  "
                                  "fpn_inception's encoder is timm's InceptionResNetV2 (third-party, not vendored in the "
                                  "reference, absent here); its in-tree decoder is deblurganv2.FPNInceptionDecoder, which "
                                  "takes the encoder's five feature maps as inputs")
    model = FPNMobileNet()
    sd = _torch.load(weights_path, map_location="cpu", weights_only=True)['model']
    model.load_state_dict({(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()})
    model.to(device)
    model.train(True)
    print(f"Successfully loaded {_np.sum([p.numel() for p in model.parameters()]):,} parameters from {weights_path}")
    return model
