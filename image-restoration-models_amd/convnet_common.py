"""Shared driver for the plain conv3x3 stacks (DnCNN, REDNet) on the HIP conv kernel."""
from __future__ import annotations

import torch

from . import _hip, ops


def conv3x3(wp, x, y, ci, co, bias=None, relu1=False, res=None, res_mode=0, relu2=False):
    """y = epilogue(conv3x3(x)) through irm_conv3x3_f32 (include/irm_hip.h)."""
    ops.conv3x3(wp, x, y, ci, co, bias=bias, relu1=relu1, res=res, res_mode=res_mode, relu2=relu2)


class PackedCache:
    """Rebuilds packed weights when the parameters of `module` change."""

    def __init__(self, module, build):
        self.module, self.build, self.key, self.value = module, build, None, None

    def get(self):
        key = _hip.param_key(self.module)
        if self.value is None or key != self.key:
            self.value, self.key = self.build(), key
        return self.value


def require_cuda(x, what):
    if not x.is_cuda:
        raise _hip.HipLibraryError(f"irm_amd {what} runs on the GPU only (no CPU fallback); "
                                   "move the model and input to 'cuda'")
