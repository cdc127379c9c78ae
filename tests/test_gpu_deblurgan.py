"""GPU parity tests of the DeblurGANv2 FPN-MobileNet path (train-mode norms) against PyTorch references of
each op, the golden outputs of the imported reference module, and the oracle tiler with the model's hooks."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from irm_amd import _hip, deblurganv2, ops, synth, utils
from oracle import deblurgan_ref, tiler_ref

pytestmark = pytest.mark.gpu
TOL = 2e-4


def rnd(name, shape, lo=-1.0, hi=1.0):
    return synth.uniform(555, name, shape, lo, hi)


def gin(name, shape, lo=0.0, hi=1.0):
    return synth.uniform(7, name, shape, lo, hi)


@pytest.mark.parametrize("B,C,H,W,act,res,affine", [(2, 32, 16, 24, 4, False, True), (1, 24, 23, 41, 0, True, True),
                                                     (2, 64, 8, 8, 1, False, False)])
def test_chan_norm(dev, B, C, H, W, act, res, affine):
    x = rnd(f"cn{C}", (B, C, H, W), -2, 3)
    w, b = rnd(f"cnw{C}", (C,), 0.5, 1.5), rnd(f"cnb{C}", (C,))
    r = rnd(f"cnr{C}", (B, C, H, W))
    ref = F.instance_norm(x.double(), None, None, w.double() if affine else None, b.double() if affine else None, True, 0.1, 1e-5)
    ref = F.relu6(ref) if act == 4 else F.relu(ref) if act == 1 else ref
    if res:
        ref = ref + r.double()
    xg = x.to(dev)
    st = torch.empty(B, C, 2, device=dev)
    ops.chan_stats(xg, st)
    ops.chan_norm_act(xg, st, xg, weight=w.to(dev) if affine else None, bias=b.to(dev) if affine else None,
                      res=r.to(dev) if res else None, act=act)
    assert (xg.cpu().double() - ref).abs().max() < TOL


@pytest.mark.parametrize("B,C,H,W,lo,hi", [(1, 8, 96, 128, -2.0, 3.0), (1, 3, 200, 324, 100.0, 101.0),
                                            (2, 32, 360, 640, -1.0, 1.0), (1, 5, 37, 51, 0.0, 1.0)])
def test_chan_stats_split_planes(dev, B, C, H, W, lo, hi):
    """Planes large enough to be split over several workgroups (and a large common offset): mean / rstd
    against float64, and bit-identical on a second run."""
    x = rnd(f"cs{C}{H}", (B, C, H, W), lo, hi)
    xg = x.to(dev)
    st, st2 = torch.empty(B, C, 2, device=dev), torch.empty(B, C, 2, device=dev)
    ops.chan_stats(xg, st)
    ops.chan_stats(xg, st2)
    assert torch.equal(st, st2)
    xd = x.double().reshape(B, C, -1)
    mean, var = xd.mean(-1), xd.var(-1, unbiased=False)
    got = st.cpu().double()
    assert (got[..., 0] - mean).abs().max() < 1e-5 * max(1.0, abs(hi))
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    assert ((got[..., 1] - rstd).abs() / rstd).max() < 1e-4


@pytest.mark.parametrize("H,W", [(32, 48), (23, 41)])
def test_stride2_convs(dev, H, W):
    x = rnd("s2x", (2, 3, H, W))
    w = rnd("s2w", (32, 3, 3, 3), -0.3, 0.3)
    ref = F.conv2d(x.double(), w.double(), None, 2, 1)
    y = torch.empty(2, 32, ref.shape[2], ref.shape[3], device=dev)
    ops.conv3x3_s2(x.to(dev), w.to(dev), y, 3, 32)
    assert (y.cpu().double() - ref).abs().max() < TOL
    xd = rnd("s2d", (2, 96, H, W))
    wd = rnd("s2dw", (96, 1, 3, 3))
    refd = F.conv2d(xd.double(), wd.double(), None, 2, 1, groups=96)
    yd = torch.empty(2, 96, refd.shape[2], refd.shape[3], device=dev)
    ops.dwconv3x3_s2(xd.to(dev), wd.reshape(96, 9).to(dev), yd)
    assert (yd.cpu().double() - refd).abs().max() < TOL


def test_upsample_add_and_tanh_epilogue(dev):
    src, add = rnd("ua", (2, 16, 5, 7)), rnd("ub", (2, 16, 10, 14))
    out = torch.empty(2, 16, 10, 14, device=dev)
    ops.upsample_add(src.to(dev), out, 2, add=add.to(dev))
    assert torch.equal(out.cpu(), add + F.interpolate(src, scale_factor=2, mode="nearest"))
    big = torch.full((2, 40, 40, 56), 3.0, device=dev)
    ops.upsample_add(src.to(dev), big[:, 8:24], 8)
    assert torch.equal(big[:, 8:24].cpu(), F.interpolate(src, scale_factor=8, mode="nearest"))
    assert torch.all(big[:, :8] == 3.0) and torch.all(big[:, 24:] == 3.0)
    x = rnd("tx", (1, 32, 16, 32))
    w, b = rnd("tw", (3, 32, 3, 3), -0.3, 0.3), rnd("tb", (3,))
    r = rnd("tr", (1, 3, 16, 32))
    ref = torch.clamp(torch.tanh(F.conv2d(x.double(), w.double(), b.double(), padding=1)) + r.double(), -1, 1)
    y = torch.empty(1, 3, 16, 32, device=dev)
    ops.conv3x3(_hip.pack_conv3x3_weight(w).to(dev), x.to(dev), y, 32, 3, bias=b.to(dev), res=r.to(dev), res_mode=3)
    assert (y.cpu().double() - ref).abs().max() < TOL


@pytest.mark.parametrize("h,w", [(64, 64), (96, 160)])
def test_fpn_mobilenet_vs_golden(dev, golden, h, w):
    model = deblurganv2.FPNMobileNet().load_synthetic(42).to(dev)
    model.train(True)
    x = gin(f"dg_in_{h}x{w}", (1, 3, h, w), -1.0, 1.0)
    y = model(x.to(dev)).cpu().numpy()
    err = np.abs(y - golden("deblurgan")[f"fpn_mobilenet_{h}x{w}"]).max()
    print(f"fpn_mobilenet {h}x{w}: max-abs vs reference golden {err:.3e}")
    assert err <= 1e-3


def test_fpn_mobilenet_batched_tiles_are_independent(dev):
    model = deblurganv2.FPNMobileNet().load_synthetic(42).to(dev)
    x = gin("dg_batch", (2, 3, 64, 96), -1.0, 1.0).to(dev)
    yb = model(x)
    assert torch.equal(yb, model(x))
    assert (model(x[1:2]) - yb[1:2]).abs().max() < 5e-5       # train-mode norms use per-sample statistics


def test_deblurgan_tiled_vs_oracle(dev):
    """get_model_prediction (uint8 in/out) with DeblurGANv2's normalize / zero-pad-to-32 / postprocess hooks and
    PATCH_CONFIG entry, against the oracle tiler + oracle model."""
    model = deblurganv2.FPNMobileNet().load_synthetic(42).to(dev)
    img, tgt = synth.synth_image_pair(5, 100, 150, 3, seed_base=1000, blur=9)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref = tiler_ref.tiled_inference(lambda t: deblurgan_ref.fpn_mobilenet_forward(t, sd), img, patch_size=96,
                                    patch_overlap=32, pad=deblurgan_ref.pad32, normalize=deblurgan_ref.normalize,
                                    postprocess=deblurgan_ref.postprocess)
    pred, ms = utils.run_model_inference(model, img, dev, normalize=deblurganv2.normalize, pad=deblurganv2.pad,
                                         postprocess=deblurganv2.postprocess, patch_size=96, patch_overlap=32)
    diff = np.abs(pred.astype(int) - ref.astype(int))
    p_gpu, p_ref = tiler_ref.psnr(tgt, pred), tiler_ref.psnr(tgt, ref)
    print(f"deblurgan tiled: {int((diff > 0).sum())}/{ref.size} u8 values differ (max {diff.max()}), PSNR {p_gpu:.4f} vs {p_ref:.4f}")
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01 and abs(p_gpu - p_ref) < 0.01
    cfg = utils.get_patch_config("deblurring", "motion", "DeblurGANv2 (MobileNet)")
    pred2, _ = utils.get_model_prediction(model, img, dev, **cfg)           # one 100x150 tile padded to 128x160
    assert pred2.shape == img.shape and pred2.dtype == np.uint8


@pytest.mark.parametrize("h,w", [(128, 160), (256, 128)])
def test_fpn_inception_decoder_vs_golden(dev, golden, h, w):
    """The in-tree part of FPN-Inception (laterals, reflect pads, top-down path, heads, smoothing, tanh/clamp output,
    fpn_inception.py:65-81, 153-170) on synthetic encoder maps against the reference class run with constant-output
    encoder stages (gen_golden.py --only fpn_inception).  The timm encoder itself is absent: unpinned."""
    g = golden("fpn_inception")
    model = deblurganv2.FPNInceptionDecoder().load_synthetic(42).to(dev).train(True)
    x = gin(f"fi_in_{h}x{w}", (1, 3, h, w), -1.0, 1.0)
    chans = (32, 64, 192, 1088, 2080)
    encs = [gin(f"fi_enc{i}_{h}x{w}", (1, chans[i]) + tuple(int(v) for v in g[f"fi_sizes_{h}x{w}"][i]), -1.0, 1.0).to(dev)
            for i in range(5)]
    y = model(x.to(dev), *encs).cpu().numpy()
    err = np.abs(y - g[f"fi_{h}x{w}"]).max()
    print(f"fpn_inception decoder {h}x{w}: max-abs vs reference golden {err:.3e}")
    assert err <= 2e-4
    # batched tiles stay independent (train-mode norms use per-sample statistics)
    y2 = model(torch.cat([x, x.flip(-1)]).to(dev), *[torch.cat([e, e.flip(-1)]) for e in encs])[:1].cpu().numpy()
    assert np.abs(y2 - y).max() <= 1e-5
