#!/usr/bin/env python3
"""Pin the oracle against the reference and write tests/golden/*.

Runs ONLY in the build container (it imports the reference from
/root/reference/src, CPU).  The reference never travels to the GPU box: what
travels are the small expected-output fixtures written here, the deterministic
weight/input generators of the product package (synth.py) and this script.

What it does
  1. imports the reference modules (Restormer, DnCNN, REDNet, src/utils.py with
     empty stubs for the absent third-party modules cv2 / skimage / deblurganv2 /
     mair), loads synthetic weights into them;
  2. checks every oracle restatement in this directory against them and records
     the max-abs differences in tests/golden/MANIFEST.json;
  3. stores the REFERENCE outputs (not the oracle's) as golden vectors.

Usage: python oracle/gen_golden.py [--only restormer|tiler|convnets|ops|deblurgan|fpn_inception|fullsize|fullsize_frame|fullsize_c3|fullsize_c5|demo|mair]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

import irm_amd  # noqa: E402
from irm_amd import synth  # noqa: E402
from oracle import convnets_ref, deblurgan_ref, mair_ref, restormer_ref, tiler_ref  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


# ---------------------------------------------------------------------------
def import_reference():
    """sys.path import of the reference with stubs for absent third-party deps."""
    sys.path.insert(0, REF_SRC)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Absent:  # placeholder classes used only in isinstance() checks
        pass

    stub("cv2")
    stub("skimage")
    stub("skimage.metrics", peak_signal_noise_ratio=None, structural_similarity=None)
    stub("deblurganv2", normalize=None, pad=None, postprocess=None)
    stub("deblurganv2.models")
    stub("deblurganv2.models.fpn_inception", FPNInception=type("FPNInception", (_Absent,), {}))
    stub("deblurganv2.models.fpn_mobilenet", FPNMobileNet=type("FPNMobileNet", (_Absent,), {}))
    stub("mair")
    stub("mair.basicsr")
    stub("mair.basicsr.archs")
    stub("mair.basicsr.archs.mair_arch", MaIR=type("MaIR", (_Absent,), {}))
    stub("mair.realDenoising")
    stub("mair.realDenoising.basicsr")
    stub("mair.realDenoising.basicsr.models")
    stub("mair.realDenoising.basicsr.models.archs")
    stub("mair.realDenoising.basicsr.models.archs.mairunet_arch", MaIRUNet=type("MaIRUNet", (_Absent,), {}))
    import restormer as ref_restormer            # noqa
    import restormer.restormer as ref_rmod       # noqa
    import dncnn.models.network_dncnn as ref_dn  # noqa
    import rednet.rednet as ref_red              # noqa
    import utils as ref_utils                    # noqa
    return types.SimpleNamespace(rmod=ref_rmod, dncnn=ref_dn, rednet=ref_red, utils=ref_utils)


def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def maxabs(a, b) -> float:
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


RESTORMER_CFGS = {
    # name: (ctor kwargs, input channels)
    "deblur_withbias": dict(inp_channels=3, out_channels=3, LayerNorm_type="WithBias"),
    "denoise_biasfree": dict(inp_channels=3, out_channels=3, LayerNorm_type="BiasFree"),
    "gray_biasfree": dict(inp_channels=1, out_channels=1, LayerNorm_type="BiasFree"),
    "dualpixel_withbias": dict(inp_channels=6, out_channels=3, LayerNorm_type="WithBias", dual_pixel_task=True),
}


def synth_input(name, shape, lo=0.0, hi=1.0):
    return synth.uniform(7, name, shape, lo, hi)


# ---------------------------------------------------------------------------
def gen_restormer(ref, manifest):
    from irm_amd.restormer import restormer as prod
    out = {}
    for name, kw in RESTORMER_CFGS.items():
        net = ref.rmod.Restormer(**kw).eval()
        shapes = shapes_of(net)
        manifest.setdefault("restormer_param_shapes", {})[name] = {k: list(v) for k, v in shapes.items()}
        sd = synth.synth_state_dict(shapes, seed=42, rules=prod.SYNTH_RULES)
        net.load_state_dict(sd, strict=True)
        for tag, (h, w) in {"64x64": (64, 64), "40x56": (40, 56)}.items():
            if tag == "40x56" and name != "deblur_withbias":
                continue
            x = synth_input(f"restormer_in_{name}_{tag}", (1, kw["inp_channels"], h, w))
            y_ref = net(x)
            y_orc = restormer_ref.restormer_forward(x, sd, dual_pixel_task=kw.get("dual_pixel_task", False))
            d = maxabs(y_ref, y_orc)
            manifest.setdefault("oracle_vs_reference", {})[f"restormer/{name}/{tag}"] = d
            print(f"restormer {name} {tag}: oracle-vs-reference max-abs {d:.3e}  out range "
                  f"[{float(y_ref.min()):.3f},{float(y_ref.max()):.3f}] |y-x| mean "
                  f"{float((y_ref - x[:, :y_ref.shape[1]]).abs().mean()) if y_ref.shape[1] <= x.shape[1] else -1:.4f}")
            assert d <= 2e-5, d
            out[f"{name}_{tag}"] = y_ref.numpy()
    np.savez_compressed(os.path.join(GOLD, "restormer_forward.npz"), **out)


def gen_ops(ref, manifest):
    """Per-op goldens from the reference's own sub-modules."""
    from irm_amd.restormer import restormer as prod
    out = {}
    R = ref.rmod
    cases = [(48, 1, 16, 24, "WithBias"), (48, 1, 16, 24, "BiasFree"), (96, 2, 16, 16, "WithBias"),
             (96, 1, 8, 40, "BiasFree"), (192, 4, 8, 8, "BiasFree"), (384, 8, 8, 8, "WithBias")]
    for (c, heads, h, w, ln) in cases:
        tag = f"c{c}_h{heads}_{h}x{w}_{ln}"
        blk = R.TransformerBlock(dim=c, num_heads=heads, ffn_expansion_factor=2.66, bias=False,
                                 LayerNorm_type=ln).eval()
        sd = synth.synth_state_dict(shapes_of(blk), seed=11, rules=prod.SYNTH_RULES)
        blk.load_state_dict(sd, strict=True)
        x = synth_input("tb_in_" + tag, (2, c, h, w), -1.0, 1.0)
        n1 = blk.norm1(x)
        a = blk.attn(n1)
        x1 = x + a
        n2 = blk.norm2(x1)
        f = blk.ffn(n2)
        y = blk(x)
        # oracle pieces
        pre = ""
        d = max(
            maxabs(n1, restormer_ref.layer_norm_c(x, sd["norm1.body.weight"], sd.get("norm1.body.bias"))),
            maxabs(a, restormer_ref.mdta(n1, sd, "attn.", heads)),
            maxabs(f, restormer_ref.gdfn(n2, sd, "ffn.")),
            maxabs(y, restormer_ref.transformer_block(x, sd, pre, heads)))
        manifest.setdefault("oracle_vs_reference", {})["ops/tb_" + tag] = d
        print(f"ops {tag}: oracle-vs-reference max-abs {d:.3e}")
        assert d <= 2e-5
        out["tb_" + tag + "_attn"] = a.numpy()
        out["tb_" + tag + "_ffn"] = f.numpy()
        out["tb_" + tag + "_out"] = y.numpy()
    # resampling modules
    for c, h, w in [(48, 16, 24), (96, 8, 16)]:
        dn = R.Downsample(c).eval()
        sd = synth.synth_state_dict(shapes_of(dn), seed=12)
        dn.load_state_dict(sd)
        x = synth_input(f"down_in_{c}", (2, c, h, w), -1, 1)
        out[f"down_{c}_{h}x{w}"] = dn(x).numpy()
        up = R.Upsample(c).eval()
        sd = synth.synth_state_dict(shapes_of(up), seed=13)
        up.load_state_dict(sd)
        out[f"up_{c}_{h}x{w}"] = up(x).numpy()
    np.savez_compressed(os.path.join(GOLD, "restormer_ops.npz"), **out)


def gen_convnets(ref, manifest):
    from irm_amd.dncnn import SYNTH_RULES as DN_RULES
    from irm_amd.rednet import SYNTH_RULES as RED_RULES
    out = {}
    for tag, (nch, nb) in {"gray17": (1, 17), "gray20": (1, 20), "color20": (3, 20)}.items():
        net = ref.dncnn.DnCNN(in_nc=nch, out_nc=nch, nc=64, nb=nb, act_mode="R").eval()
        shapes = shapes_of(net)
        manifest.setdefault("dncnn_param_shapes", {})[tag] = {k: list(v) for k, v in shapes.items()}
        sd = synth.synth_state_dict(shapes, seed=42, rules=DN_RULES)
        net.load_state_dict(sd, strict=True)
        for h, w in [(32, 32), (40, 72)]:
            x = synth_input(f"dncnn_in_{tag}_{h}x{w}", (1, nch, h, w))
            y = net(x)
            d = maxabs(y, convnets_ref.dncnn_forward(x, sd))
            manifest.setdefault("oracle_vs_reference", {})[f"dncnn/{tag}/{h}x{w}"] = d
            print(f"dncnn {tag} {h}x{w}: oracle-vs-reference {d:.3e} |y-x| {float((y-x).abs().mean()):.4f}")
            assert d <= 2e-5
            out[f"dncnn_{tag}_{h}x{w}"] = y.numpy()
    net = ref.rednet.REDNet().eval()
    shapes = shapes_of(net)
    manifest["rednet_param_shapes"] = {k: list(v) for k, v in shapes.items()}
    sd = synth.synth_state_dict(shapes, seed=42, rules=RED_RULES)
    net.load_state_dict(sd, strict=True)
    for h, w in [(32, 32), (24, 40)]:
        x = synth_input(f"rednet_in_{h}x{w}", (1, 1, h, w))
        y = net(x)
        d = maxabs(y, convnets_ref.rednet_forward(x, sd))
        manifest.setdefault("oracle_vs_reference", {})[f"rednet/{h}x{w}"] = d
        print(f"rednet {h}x{w}: oracle-vs-reference {d:.3e} |y-x| {float((y-x).abs().mean()):.4f}")
        assert d <= 2e-5
        out[f"rednet_{h}x{w}"] = y.numpy()
    np.savez_compressed(os.path.join(GOLD, "convnets_forward.npz"), **out)


def gen_tiler(ref, manifest):
    """The reference's run_model_inference (src/utils.py:353-454) itself, on CPU."""
    from irm_amd import configs as prod_cfg
    from irm_amd.dncnn import SYNTH_RULES as DN_RULES
    from irm_amd.restormer import restormer as prod
    U = ref.utils
    out = {}
    # (5) tile index lists for every PATCH_CONFIG entry x sizes
    idx = {}
    flat = []
    for k, v in prod_cfg.PATCH_CONFIG.items():
        for i, e in enumerate(v if isinstance(v, list) else [v]):
            flat.append((f"{k}[{i}]", e["patch_size"], e["patch_overlap"]))
    for name, ps0, ov in flat:
        for (h, w) in [(256, 256), (512, 512), (720, 1280), (481, 321), (300, 420)]:
            ps = min(ps0, max(h, w))
            stride = max(ps - ov, 1)
            ref_h = list(range(0, h - ps, stride)) + [max(h - ps, 0)]
            ref_w = list(range(0, w - ps, stride)) + [max(w - ps, 0)]
            assert ref_h == tiler_ref.tile_origins(h, ps, ov) and ref_w == tiler_ref.tile_origins(w, ps, ov)
            idx[f"{name}|{h}x{w}"] = [ps, ref_h, ref_w]
    manifest["tile_origins"] = idx
    # windows / pad / noise
    for n in (64, 128, 200):
        wref = U.get_gaussian_weights(n, n, 3)
        assert np.array_equal(wref, tiler_ref.gaussian_window(n, n, 3))
    out["window_64"] = U.get_gaussian_weights(64, 64, 1)[:, :, 0]
    t = synth_input("padcheck", (1, 3, 37, 50))
    assert torch.equal(U.pad(t), tiler_ref.reflect_pad8(t))
    tl = synth_input("noisecheck", (40, 48, 3)).numpy()
    assert np.array_equal(U.add_gaussian_noise(tl.copy(), 25), tiler_ref.degrade(tl, 25))
    out["noise_40x48x3_s25"] = U.add_gaussian_noise(tl.copy(), 25)

    # (3a) DnCNN through the tiler on a 150x210 gray image with a shrunk patch config + noise
    net = ref.dncnn.DnCNN(in_nc=1, out_nc=1, nc=64, nb=17, act_mode="R").eval()
    sd = synth.synth_state_dict(shapes_of(net), seed=42, rules=DN_RULES)
    net.load_state_dict(sd)
    img, _ = synth.synth_image_pair(1, 150, 210, 1, seed_base=3000, blur=0)
    for tag, kw in {"dncnn_tiled": dict(patch_size=64, patch_overlap=16),
                    "dncnn_tiled_noise": dict(patch_size=64, patch_overlap=16, need_degradation=True, noise_level=25),
                    "dncnn_whole": dict(patch_size=None)}.items():
        pred_ref, _ = U.run_model_inference(net, img, torch.device("cpu"), **kw)
        pred_orc = tiler_ref.tiled_inference(lambda t: convnets_ref.dncnn_forward(t, sd), img, **kw)
        nd = int((pred_ref.astype(int) != pred_orc.astype(int)).sum())
        manifest.setdefault("oracle_vs_reference", {})[f"tiler/{tag}/u8_mismatches"] = nd
        print(f"tiler {tag}: u8 mismatches oracle-vs-reference {nd} of {pred_ref.size}")
        assert nd == 0
        out[tag] = pred_ref
    # (3b) Restormer through the tiler incl. utils.pad: 100x136 image, patch 64 / overlap 16 -> edge tiles
    rnet = ref.rmod.Restormer(LayerNorm_type="WithBias").eval()
    rsd = synth.synth_state_dict(shapes_of(rnet), seed=42, rules=prod.SYNTH_RULES)
    rnet.load_state_dict(rsd)
    img3, tgt3 = synth.synth_image_pair(2, 100, 136, 3, seed_base=1000, blur=7)
    tiles_ref = []
    orig_call = rnet.forward

    def spy(x):
        y = orig_call(x)
        tiles_ref.append(y.numpy().copy())
        return y
    rnet.forward = spy
    pred_ref, _ = U.run_model_inference(rnet, img3, torch.device("cpu"), pad=U.pad, patch_size=64, patch_overlap=16)
    rnet.forward = orig_call
    pred_orc = tiler_ref.tiled_inference(lambda t: restormer_ref.restormer_forward(t, rsd), img3,
                                         pad=tiler_ref.reflect_pad8, patch_size=64, patch_overlap=16)
    nd = int((pred_ref.astype(int) != pred_orc.astype(int)).sum())
    manifest["oracle_vs_reference"]["tiler/restormer_tiled/u8_mismatches"] = nd
    print(f"tiler restormer_tiled: u8 mismatches {nd} of {pred_ref.size}; psnr vs target "
          f"{tiler_ref.psnr(tgt3, pred_ref):.4f}")
    assert nd <= 2
    out["restormer_tiled"] = pred_ref
    out["restormer_tiled_tile0"] = tiles_ref[0]
    manifest["restormer_tiled_psnr"] = tiler_ref.psnr(tgt3, pred_ref)
    np.savez_compressed(os.path.join(GOLD, "tiler.npz"), **out)


MAIR_NET_G = dict(inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4,
                  ssm_ratio=2.0, flp_ratio=4.0, mlp_ratio=1.5, bias=False, dual_pixel_task=False, img_size=32,
                  scan_len=4, batch_size=1, dynamic_ids=False)      # test_MaIR_RealDN.yml network_g (img_size shrunk)


def import_reference_mairunet():
    """Import the reference's mairunet_arch.py itself.  Absent third-party modules are stubbed: timm.layers
    (DropPath -> identity, to_2tuple, trunc_normal_), the BasicSR registry, and mamba_ssm's CUDA
    selective_scan_fn, for which oracle/mair_ref.selective_scan stands in (that op stays unpinned)."""
    import importlib
    import torch.nn as nn
    base = "/root/reference/src/mair"
    for k in [k for k in sys.modules if k == "mair" or k.startswith("mair.")]:
        del sys.modules[k]          # drop the empty stubs import_reference() installed for src/utils.py

    def pkg(name, path=None, **attrs):
        m = types.ModuleType(name)
        if path:
            m.__path__ = [path]
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    pkg("timm")
    pkg("timm.layers", DropPath=DropPath, to_2tuple=lambda v: (v, v) if not isinstance(v, (tuple, list)) else tuple(v),
        trunc_normal_=lambda *a, **k: None)
    pkg("mamba_ssm")
    pkg("mamba_ssm.ops")
    pkg("mamba_ssm.ops.selective_scan_interface", selective_scan_fn=mair_ref.selective_scan,
        selective_scan_ref=mair_ref.selective_scan)

    class _Reg:
        def register(self, *a, **k):
            return lambda cls: cls
    pkg("mair", base)
    pkg("mair.basicsr", base + "/basicsr")
    pkg("mair.basicsr.utils", base + "/basicsr/utils")
    pkg("mair.basicsr.utils.registry", ARCH_REGISTRY=_Reg())
    pkg("mair.basicsr.archs", base + "/basicsr/archs")
    pkg("mair.realDenoising", base + "/realDenoising")
    pkg("mair.realDenoising.basicsr", base + "/realDenoising/basicsr")
    pkg("mair.realDenoising.basicsr.models", base + "/realDenoising/basicsr/models")
    pkg("mair.realDenoising.basicsr.models.archs", base + "/realDenoising/basicsr/models/archs")
    flat = importlib.import_module("mair.basicsr.archs.mair_arch")
    unet = importlib.import_module("mair.realDenoising.basicsr.models.archs.mairunet_arch")
    unet.flat = flat
    return unet


def gen_mair(ref, manifest):
    from irm_amd.mair import SYNTH_RULES as MAIR_RULES
    arch = import_reference_mairunet()
    out = {}
    # (6) scan-id tables straight from the reference's generator
    ssu = sys.modules["mair.realDenoising.basicsr.models.archs.shift_scanf_util"]
    for (h, w, sl) in [(4, 8, 4), (16, 16, 4), (6, 10, 4), (5, 7, 4), (32, 24, 4)]:
        a, ai = ssu.mair_ids_generate((1, 1, h, w), scan_len=sl)
        b, bi = mair_ref.scan_ids(h, w, sl)
        assert torch.equal(a.reshape(4, -1), b) and torch.equal(ai.reshape(4, -1), bi)
        out[f"ids_{h}x{w}_s{sl}"] = a.reshape(4, -1).numpy().astype(np.int32)
    net = arch.MaIRUNet(**MAIR_NET_G)
    shapes = shapes_of(net)
    manifest["mairunet_param_shapes"] = {k: list(v) for k, v in shapes.items()}
    sd = synth.synth_state_dict(shapes, seed=42, rules=MAIR_RULES)
    net.load_state_dict(sd, strict=True)
    net.train()      # eval-mode forward cannot run on CPU (ids only bound under cuda, mairunet_arch.py:667-671);
    #                  train mode is the same arithmetic (no BN, no dropout, DropPath(0))
    for (h, w) in [(32, 32), (24, 40)]:
        x = synth_input(f"mair_in_{h}x{w}", (1, 3, h, w))
        net.trainig_img_size = -1
        y_ref = net(x)
        y_orc = mair_ref.mairunet_forward(x, sd, scan_len=4)
        d = maxabs(y_ref, y_orc)
        manifest.setdefault("oracle_vs_reference", {})[f"mairunet/{h}x{w}(scan op = oracle stand-in)"] = d
        print(f"mairunet {h}x{w}: oracle-vs-reference max-abs {d:.3e} |y-x| mean {float((y_ref - x).abs().mean()):.4f}"
              f" range [{float(y_ref.min()):.3f},{float(y_ref.max()):.3f}]")
        assert d <= 2e-5
        out[f"mairunet_{h}x{w}"] = y_ref.numpy()
    # one VSSBlock per level shape, reference sub-module outputs
    for (c, n, ratio, h, w) in [(48, 4, 4.0, 16, 24), (96, 8, 1.5, 8, 16), (384, 32, 1.5, 8, 8)]:
        blk = arch.VSSBlock(hidden_dim=c, drop_path=0.0, norm_layer=torch.nn.LayerNorm, attn_drop_rate=0,
                            ssm_ratio=2.0, d_state=n, mlp_ratio=ratio)
        bsd = synth.synth_state_dict(shapes_of(blk), seed=21, rules=MAIR_RULES)
        blk.load_state_dict(bsd, strict=True)
        ids, inv = ssu.mair_ids_generate((1, 1, h, w), scan_len=4)
        x = synth_input(f"vss_in_{c}", (2, h * w, c), -1.0, 1.0)
        y = blk(x, (ids, inv, None, None), (h, w))
        d = maxabs(y, mair_ref.vss_block(x, bsd, "", (h, w), ids.reshape(4, -1), inv.reshape(4, -1)))
        manifest["oracle_vs_reference"][f"mairunet/vssblock_c{c}"] = d
        print(f"vssblock c{c}: oracle-vs-reference {d:.3e}")
        assert d <= 2e-5
        out[f"vss_c{c}_{h}x{w}"] = y.numpy()
    # ---- flat MaIR (colour Gaussian denoising, options/test_MaIR_CDN_s*.yml) incl. the shifted scan tables
    for (h, w, sl) in [(4, 8, 4), (16, 16, 4), (6, 10, 4), (5, 7, 4), (12, 9, 4)]:
        a, ai = ssu.mair_shift_ids_generate((1, 1, h, w), scan_len=sl, shift_len=sl // 2)
        b, bi = mair_ref.scan_ids(h, w, sl, sl // 2)
        assert torch.equal(a.reshape(4, -1), b) and torch.equal(ai.reshape(4, -1), bi), (h, w)
        out[f"shift_ids_{h}x{w}_s{sl}"] = a.reshape(4, -1).numpy().astype(np.int32)
    flat_cfg = dict(upscale=1, in_chans=3, img_range=1., d_state=16, depths=[2, 2], embed_dim=180, ssm_ratio=1.3,
                    mlp_ratio=2.0, upsampler=None, resi_connection='1conv', img_size=16, dynamic_ids=False,
                    batch_size=1, scan_len=4)       # test_MaIR_CDN_s25.yml network_g with fewer groups/blocks
    fnet = arch.flat.MaIR(**flat_cfg)
    fshapes = shapes_of(fnet)
    manifest["mair_flat_param_shapes_depths2x2"] = {k: list(v) for k, v in fshapes.items()}
    fsd = synth.synth_state_dict(fshapes, seed=42, rules=MAIR_RULES)
    fnet.load_state_dict(fsd, strict=True)
    fnet.train()
    for (h, w) in [(16, 16), (24, 20)]:
        x = synth_input(f"mairflat_in_{h}x{w}", (1, 3, h, w))
        y_ref = fnet(x)
        y_orc = mair_ref.mair_forward(x, fsd, scan_len=4)
        d = maxabs(y_ref, y_orc)
        manifest["oracle_vs_reference"][f"mair_flat/{h}x{w}(scan op = oracle stand-in)"] = d
        print(f"mair flat {h}x{w}: oracle-vs-reference {d:.3e} |y-x| mean {float((y_ref - x).abs().mean()):.4f}")
        assert d <= 2e-5
        out[f"mairflat_{h}x{w}"] = y_ref.detach().numpy()
    np.savez_compressed(os.path.join(GOLD, "mair.npz"), **out)


def gen_deblurgan(ref, manifest):
    """FPNMobileNet: the reference's own fpn_mobilenet.py / mobilenet_v2.py (torch only) imported under a
    stub `deblurganv2` package (its __init__ needs albumentations), run in train mode like the reference."""
    import functools
    import importlib
    import torch.nn as nn
    from irm_amd.deblurganv2 import SYNTH_RULES as DG_RULES
    base = "/root/reference/src/deblurganv2"
    for k in [k for k in sys.modules if k == "deblurganv2" or k.startswith("deblurganv2.")]:
        del sys.modules[k]
    for name, path in (("deblurganv2", base), ("deblurganv2.models", base + "/models")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    fm = importlib.import_module("deblurganv2.models.fpn_mobilenet")
    norm = functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=True)     # networks.py:22
    net = fm.FPNMobileNet(norm_layer=norm, pretrained=False)
    net.train(True)                                                                          # __init__.py:38
    # fpn.enc0..enc4 are nn.Sequential views of fpn.features[...] (fpn_mobilenet.py:90-94): the same
    # tensors appear under two names; synthetic values are generated for the canonical names only
    params = {k: tuple(v.shape) for k, v in net.state_dict().items()
              if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")
                      or k.startswith("fpn.enc"))}
    manifest["fpn_mobilenet_param_shapes"] = {k: list(v) for k, v in params.items()}
    manifest["fpn_mobilenet_state_keys"] = sorted(net.state_dict().keys())
    sd = synth.synth_state_dict(params, seed=42, rules=DG_RULES)
    net.load_state_dict(sd, strict=False)
    out = {}
    for (h, w) in [(64, 64), (96, 160)]:
        x = synth_input(f"dg_in_{h}x{w}", (1, 3, h, w), -1.0, 1.0)
        y_ref = net(x)
        y_orc = deblurgan_ref.fpn_mobilenet_forward(x, sd)
        d = maxabs(y_ref, y_orc)
        manifest.setdefault("oracle_vs_reference", {})[f"fpn_mobilenet/{h}x{w}"] = d
        print(f"fpn_mobilenet {h}x{w}: oracle-vs-reference {d:.3e} |y-x| mean {float((y_ref - x).abs().mean()):.4f}")
        assert d <= 2e-5
        out[f"fpn_mobilenet_{h}x{w}"] = y_ref.detach().numpy()
    np.savez_compressed(os.path.join(GOLD, "deblurgan.npz"), **out)


def gen_fpn_inception(ref, manifest):
    """FPN-Inception decoder: the reference's own fpn_inception.py (FPNInception + FPN classes) with `timm` /
    `torchsummary` stubbed (absent third-party packages) and the five encoder stages replaced by modules that return
    fixed synthetic feature maps, so that everything the reference itself defines - laterals, reflect pads, top-down
    path, heads, smoothing, output - runs unmodified.  The encoder's arithmetic (timm InceptionResNetV2) stays
    unpinned."""
    import functools
    import importlib
    import torch.nn as nn
    from irm_amd.deblurganv2 import SYNTH_RULES as DG_RULES
    base = "/root/reference/src/deblurganv2"
    for k in [k for k in sys.modules if k == "deblurganv2" or k.startswith("deblurganv2.")]:
        del sys.modules[k]
    for name, path in (("deblurganv2", base), ("deblurganv2.models", base + "/models")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m

    class _Backbone(nn.Module):                       # what timm.create_model returns, as far as FPN.__init__ touches it
        def __init__(self):
            super().__init__()
            for n in ("conv2d_1a", "conv2d_2a", "conv2d_2b", "maxpool_3a", "conv2d_3b", "conv2d_4a", "maxpool_5a",
                      "mixed_5b", "repeat", "mixed_6a", "repeat_1", "mixed_7a", "classif"):
                setattr(self, n, nn.Identity())
    timm = types.ModuleType("timm")
    timm.create_model = lambda *a, **k: _Backbone()
    ts = types.ModuleType("torchsummary")
    ts.summary = lambda *a, **k: None
    sys.modules["timm"], sys.modules["torchsummary"] = timm, ts
    fi = importlib.import_module("deblurganv2.models.fpn_inception")
    norm = functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=True)     # networks.py:22
    net = fi.FPNInception(norm_layer=norm)
    net.train(True)

    class _Const(nn.Module):
        def __init__(self, t):
            super().__init__()
            self.t = t

        def forward(self, _):
            return self.t
    params = {k: tuple(v.shape) for k, v in net.state_dict().items()
              if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    manifest["fpn_inception_decoder_param_shapes"] = {k: list(v) for k, v in params.items()}
    sd = synth.synth_state_dict(params, seed=42, rules=DG_RULES)
    net.load_state_dict(sd, strict=False)
    out = {}
    chans = (32, 64, 192, 1088, 2080)
    # encoder map sizes of InceptionResNetV2 for an H x W input: H/2-1, H/4-2, H/8-3, H/16-2, H/32-2 (H = 256: 127, 62, 29, 14, 6)
    for (h, w) in [(128, 160), (256, 128)]:
        sizes = [(h // 2 - 1, w // 2 - 1), (h // 4 - 2, w // 4 - 2), (h // 8 - 3, w // 8 - 3), (h // 16 - 2, w // 16 - 2),
                 (h // 32 - 2, w // 32 - 2)]
        x = synth_input(f"fi_in_{h}x{w}", (1, 3, h, w), -1.0, 1.0)
        encs = [synth_input(f"fi_enc{i}_{h}x{w}", (1, chans[i]) + sizes[i], -1.0, 1.0) for i in range(5)]
        for i in range(5):
            setattr(net.fpn, f"enc{i}", _Const(encs[i]))
        with torch.no_grad():
            y_ref = net(x)
            y_orc = deblurgan_ref.fpn_inception_decoder(x, encs, sd)
        d = maxabs(y_ref, y_orc)
        manifest.setdefault("oracle_vs_reference", {})[f"fpn_inception_decoder/{h}x{w}"] = d
        print(f"fpn_inception decoder {h}x{w}: oracle-vs-reference {d:.3e} |y-x| mean {float((y_ref - x).abs().mean()):.4f}")
        assert d <= 2e-5
        out[f"fi_{h}x{w}"] = y_ref.numpy()
        out[f"fi_sizes_{h}x{w}"] = np.array(sizes, dtype=np.int32)
    np.savez_compressed(os.path.join(GOLD, "fpn_inception.npz"), **out)


def gen_fullsize(ref, manifest):
    """BASELINE.json configs[3] (the headline workload) at its full tile size: the reference Restormer (motion-deblur configuration,
    synthetic weights seed 42) on tile 0 (512x512) of the benchmark's first synthetic 1280x720 frame.  The
    fixture keeps every 8th output pixel (49 KB) plus whole-tile moments."""
    from irm_amd.restormer import restormer as prod
    kw = RESTORMER_CFGS["deblur_withbias"]
    net = ref.rmod.Restormer(**kw).eval()
    sd = synth.synth_state_dict(shapes_of(net), seed=42, rules=prod.SYNTH_RULES)
    net.load_state_dict(sd, strict=True)
    inp, _ = synth.synth_image_pair(0, 720, 1280, 3, seed_base=1000, blur=15)
    x = torch.from_numpy(np.ascontiguousarray(inp[:512, :512].transpose(2, 0, 1))).float().div(255.0)[None]
    import time
    t0 = time.time()
    with torch.no_grad():
        y = net(x)[0].numpy()
    print(f"fullsize: reference forward on 1x3x512x512 took {time.time() - t0:.0f} s; out range "
          f"[{y.min():.3f},{y.max():.3f}]")
    np.savez_compressed(os.path.join(GOLD, "restormer_fullsize.npz"), sub8=y[:, ::8, ::8],
                        mean=y.mean(axis=(1, 2)), sqmean=(y.astype(np.float64) ** 2).mean(axis=(1, 2)),
                        row100=y[:, 100, :], in_sha=np.frombuffer(bytes.fromhex(sha(x.numpy())), dtype=np.uint8))
    manifest["restormer_fullsize"] = {"input": "synth_image_pair(0,720,1280,3,seed_base=1000,blur=15)[0][:512,:512]/255",
                                      "config": "deblur_withbias", "weights_seed": 42}


def gen_fullsize_frame(ref, manifest):
    """BASELINE.json configs[3] end to end: the reference's OWN tiled-patch loop (src/utils.py:353-454, through
    get_model_prediction) with the reference Restormer (motion-deblur configuration, synthetic weights seed 42) on the
    benchmark's first synthetic 1280x720 frame, PATCH_CONFIG 512 / 96 (6 tiles).  Fixture: the whole uint8 frame, its
    sha256 and its PSNR against the synthetic target.  The same run also taps the 96-channel output of `refinement`
    (restormer.py:274, the tensor BEFORE the 0.02-gain `output` conv of the synthetic weights) on tile 0: every 16th
    pixel + one row + moments - full-size parity on an un-attenuated tensor."""
    from irm_amd.restormer import restormer as prod
    U = ref.utils
    kw = RESTORMER_CFGS["deblur_withbias"]
    net = ref.rmod.Restormer(**kw).eval()
    sd = synth.synth_state_dict(shapes_of(net), seed=42, rules=prod.SYNTH_RULES)
    net.load_state_dict(sd, strict=True)
    inp, tgt = synth.synth_image_pair(0, 720, 1280, 3, seed_base=1000, blur=15)
    cfg = U.get_patch_config("deblurring", "motion", "Restormer")
    taps = []
    hook = net.refinement.register_forward_hook(lambda m, i, o: taps.append(o[0].numpy().copy()) if not taps else None)
    import time
    t0 = time.time()
    pred, _ = U.get_model_prediction(net, inp, torch.device("cpu"), **cfg)
    hook.remove()
    dt = time.time() - t0
    assert pred.shape == inp.shape and pred.dtype == np.uint8 and len(taps) == 1 and taps[0].shape == (96, 512, 512)
    psnr = tiler_ref.psnr(tgt, pred)
    psnr_in = tiler_ref.psnr(tgt, inp)
    r = taps[0]
    print(f"fullsize frame: reference run_model_inference on 1280x720 (cfg {cfg}) took {dt:.0f} s; PSNR {psnr:.4f} dB "
          f"(input {psnr_in:.4f}); refinement tap range [{r.min():.3f},{r.max():.3f}] rms {np.sqrt((r.astype(np.float64) ** 2).mean()):.3f}")
    np.savez_compressed(os.path.join(GOLD, "restormer_fullsize_frame.npz"), pred_u8=pred,
                        sha256=np.frombuffer(bytes.fromhex(sha(pred)), dtype=np.uint8), psnr=np.float64(psnr),
                        psnr_input=np.float64(psnr_in),
                        refine_sub16=r[:, ::16, ::16], refine_row100=r[:, 100, :], refine_mean=r.mean(axis=(1, 2)),
                        refine_sqmean=(r.astype(np.float64) ** 2).mean(axis=(1, 2)))
    manifest["restormer_fullsize_frame"] = {
        "input": "synth_image_pair(0,720,1280,3,seed_base=1000,blur=15)", "config": "deblur_withbias", "weights_seed": 42,
        "patch_config": cfg, "reference_seconds": round(dt), "psnr_db": psnr, "sha256_u8": sha(pred),
        "refinement_tap": "net.refinement output of the first tile (origin 0,0), forward hook"}


def gen_fullsize_c3(ref, manifest):
    """BASELINE.json configs[2] (Restormer colour blind-denoise, BiasFree LayerNorm) at its full tile size: the
    reference model on tile 0 (256x256, PATCH_CONFIG denoising) of a 512x512 synthetic frame with sigma = 25 noise
    (seed 0, as run_model_inference adds it).  Fixture: every 4th output pixel + one row + moments (50 KB)."""
    from irm_amd.restormer import restormer as prod
    net = ref.rmod.Restormer(**RESTORMER_CFGS["denoise_biasfree"]).eval()
    sd = synth.synth_state_dict(shapes_of(net), seed=42, rules=prod.SYNTH_RULES)
    net.load_state_dict(sd, strict=True)
    inp, _ = synth.synth_image_pair(3, 512, 512, 3, seed_base=1000, blur=0)
    x01 = inp[:256, :256].astype(np.float32) / 255.0
    rng = np.random.RandomState(0)
    x01 = x01 + rng.normal(0, 25 / 255.0, x01.shape).astype(np.float32)       # (src/utils.py:386-389 shape of the noise step)
    x = torch.from_numpy(np.ascontiguousarray(x01.transpose(2, 0, 1)))[None].float()
    import time
    t0 = time.time()
    with torch.no_grad():
        y = net(x)[0].numpy()
    print(f"fullsize c3: reference BiasFree forward on 1x3x256x256 took {time.time() - t0:.0f} s; range [{y.min():.3f},{y.max():.3f}]")
    np.savez_compressed(os.path.join(GOLD, "restormer_fullsize_c3.npz"), x=x[0].numpy().astype(np.float32), sub4=y[:, ::4, ::4],
                        mean=y.mean(axis=(1, 2)), sqmean=(y.astype(np.float64) ** 2).mean(axis=(1, 2)), row77=y[:, 77, :])
    manifest["restormer_fullsize_c3"] = {"config": "denoise_biasfree", "weights_seed": 42,
                                         "input": "stored in the fixture (tile 0 of synth frame 3 + N(0, 25/255), RandomState(0))"}


def gen_fullsize_c5(ref, manifest):
    """BASELINE.json configs[4] (MaIRUNet real-denoise) at its full tile size 256x256: the reference's own
    mairunet_arch.MaIRUNet (scan op = oracle stand-in, as everywhere: mamba_ssm is absent) - scan length
    L = 65 536 at level 1.  Fixture: every 4th output pixel + one row + moments."""
    from irm_amd.mair import SYNTH_RULES as MAIR_RULES
    arch = import_reference_mairunet()
    cfg = dict(MAIR_NET_G, img_size=256)
    net = arch.MaIRUNet(**cfg)
    sd = synth.synth_state_dict(shapes_of(net), seed=42, rules=MAIR_RULES)
    net.load_state_dict(sd, strict=True)
    net.train()
    net.trainig_img_size = -1
    x = synth_input("mair_in_256x256", (1, 3, 256, 256))
    import time
    t0 = time.time()
    with torch.no_grad():
        y = net(x)[0].numpy()
    print(f"fullsize c5: reference MaIRUNet forward on 1x3x256x256 took {time.time() - t0:.0f} s; range [{y.min():.3f},{y.max():.3f}]")
    np.savez_compressed(os.path.join(GOLD, "mair_fullsize_c5.npz"), sub4=y[:, ::4, ::4], mean=y.mean(axis=(1, 2)),
                        sqmean=(y.astype(np.float64) ** 2).mean(axis=(1, 2)), row77=y[:, 77, :])
    manifest["mair_fullsize_c5"] = {"input": "synth.uniform(7, 'mair_in_256x256', (1,3,256,256), 0, 1)", "weights_seed": 42,
                                    "note": "scan op = oracle stand-in (mamba_ssm absent): parity of the scan arithmetic stays unpinned"}


def gen_demo(ref, manifest):
    """BASELINE.json configs[0]: the reference's CPU-runnable demo case (scripts/test_demo.py) - DnCNN gray
    Gaussian sigma 25 on its 256x256 demo image, through the reference's own get_patch_config +
    get_model_prediction on the CPU.  The checkpoints are not available offline, so the network carries the
    synthetic weights (seed 42); the image is the reference's data file demo/denoising_gaussian_gray_blind_noisy.bmp
    (already noisy: no degradation is added), stored in the fixture as a uint8 array."""
    from PIL import Image
    from irm_amd.dncnn import SYNTH_RULES as DN_RULES
    U = ref.utils
    img = np.array(Image.open(os.path.join(os.path.dirname(REF_SRC), "demo", "denoising_gaussian_gray_blind_noisy.bmp")).convert("L"))
    assert img.shape == (256, 256) and img.dtype == np.uint8
    img = img[:, :, None]
    out = {"noisy_u8": img}
    for tag, nb in (("nonblind_nb17", 17), ("blind_nb20", 20)):
        net = ref.dncnn.DnCNN(in_nc=1, out_nc=1, nc=64, nb=nb, act_mode="R").eval()
        sd = synth.synth_state_dict(shapes_of(net), seed=42, rules=DN_RULES)
        net.load_state_dict(sd)
        cfg = U.get_patch_config("denoising", "gaussian", "DnCNN")
        pred, _ = U.get_model_prediction(net, img, torch.device("cpu"), **cfg)
        orc = tiler_ref.tiled_inference(lambda t: convnets_ref.dncnn_forward(t, sd), img,
                                        patch_size=cfg["patch_size"], patch_overlap=cfg["patch_overlap"])
        nd = int((pred.astype(int) != orc.astype(int)).sum())
        manifest.setdefault("oracle_vs_reference", {})[f"demo/{tag}/u8_mismatches"] = nd
        print(f"demo {tag}: patch config {cfg}, u8 mismatches oracle-vs-reference {nd} of {pred.size}")
        assert nd == 0
        out[tag] = pred
    np.savez_compressed(os.path.join(GOLD, "demo_c1.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    mpath = os.path.join(GOLD, "MANIFEST.json")
    manifest = json.load(open(mpath)) if os.path.exists(mpath) else {}
    ref = import_reference()
    steps = {"ops": gen_ops, "restormer": gen_restormer, "convnets": gen_convnets, "tiler": gen_tiler,
             "deblurgan": gen_deblurgan, "fpn_inception": gen_fpn_inception, "fullsize": gen_fullsize, "fullsize_frame": gen_fullsize_frame, "fullsize_c3": gen_fullsize_c3, "demo": gen_demo, "mair": gen_mair,
             "fullsize_c5": gen_fullsize_c5}      # mair last: it re-stubs the `mair` package for the reference's arch file
    for k, fn in steps.items():
        if args.only in (None, k):
            fn(ref, manifest)
    manifest["generated_by"] = "oracle/gen_golden.py (reference imported from /root/reference/src, torch CPU fp32)"
    manifest["torch"] = torch.__version__
    json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)
    print("wrote", mpath)


if __name__ == "__main__":
    main()
