"""Checks at BASELINE.json's full sizes (configs[3]: Restormer motion-deblur on 1280x720 frames, six
512x512 tiles with overlap 96).

The CPU oracle needs ~1 minute per 512x512 tile on a many-core host, so at this size the tests use
(a) a golden of the imported reference itself on tile 0 of the benchmark's first frame
    (tests/golden/restormer_fullsize.npz, written by oracle/gen_golden.py --only fullsize: every 8th output
    pixel, one full row and whole-tile moments), and
(b) size-independent properties of the path: run-to-run bit identity, independence of a tile's result from
    the batch it travels in, the tiler's exact round trip of an identity model, and its bit-exact agreement
    with the numpy tiler of the oracle on a cheap per-pixel model.
Tolerances as in BASELINE.json north_star: 1e-3 max-abs on float outputs, integer work bit-exact."""
import numpy as np
import pytest
import torch

from irm_amd import restormer, synth, utils
from oracle import tiler_ref

pytestmark = pytest.mark.gpu
H, W, PS, OV = 720, 1280, 512, 96


@pytest.fixture(scope="module")
def model(dev):
    return restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)


@pytest.fixture(scope="module")
def frame():
    return synth.synth_image_pair(0, H, W, 3, seed_base=1000, blur=15)      # bench.py's first frame


def _tile0(frame, dev):
    inp = frame[0]
    return torch.from_numpy(np.ascontiguousarray(inp[:PS, :PS].transpose(2, 0, 1))).float().div(255.0)[None].to(dev)


def test_fullsize_tile_vs_reference_golden(dev, model, frame, golden):
    g = golden("restormer_fullsize")
    y = model(_tile0(frame, dev))[0].cpu().numpy()
    err = max(np.abs(y[:, ::8, ::8] - g["sub8"]).max(), np.abs(y[:, 100, :] - g["row100"]).max())
    m_err = np.abs(y.mean(axis=(1, 2)) - g["mean"]).max()
    s_err = np.abs((y.astype(np.float64) ** 2).mean(axis=(1, 2)) - g["sqmean"]).max()
    print(f"512x512 tile vs reference golden: max-abs {err:.3e} on the sampled pixels, channel mean {m_err:.2e}, "
          f"mean square {s_err:.2e}")
    assert err <= 1e-3 and m_err <= 1e-5 and s_err <= 1e-5


def test_fullsize_whole_frame_vs_reference_run_model_inference(dev, model, frame, golden):
    """BASELINE configs[3] end to end against the REFERENCE's own tiled-patch loop (src/utils.py:353-454 through
    get_model_prediction, reference Restormer, CPU; gen_golden.py --only fullsize_frame, 134 s): the 1280x720 uint8
    frame of the GPU path differs from the reference's by at most 1 in fewer than 0.1 % of its bytes, and the PSNR
    against the synthetic target agrees within north_star's 0.01 dB."""
    g = golden("restormer_fullsize_frame")
    inp, tgt = frame
    out, sse = utils.tiled_forward_device(model, torch.from_numpy(inp).to(dev), PS, OV, pad8=True,
                                          target_dev=torch.from_numpy(tgt).to(dev), max_batch=8)
    out = out.cpu().numpy()
    ref = g["pred_u8"]
    assert out.shape == ref.shape == (H, W, 3) and out.dtype == np.uint8
    diff = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    psnr_gpu = 10 * np.log10(255.0 ** 2 / (float(sse.item()) / out.size))
    assert abs(psnr_gpu - tiler_ref.psnr(tgt, out)) < 1e-9                      # device SSE == host PSNR of the same bytes
    print(f"whole frame vs the reference's run_model_inference: {int((diff > 0).sum())} of {diff.size} bytes differ "
          f"(max {int(diff.max())}); PSNR gpu {psnr_gpu:.5f} dB, reference {float(g['psnr']):.5f} dB")
    assert int(diff.max()) <= 1 and float((diff > 0).mean()) < 1e-3
    assert abs(psnr_gpu - float(g["psnr"])) < 0.01
    # the product's reference-shaped host call gives the same bytes as the device pipeline
    pred, _ = utils.get_model_prediction(model, inp, dev, **utils.get_patch_config("deblurring", "motion", "Restormer"))
    assert np.array_equal(pred, out)


def test_fullsize_unattenuated_trunk_vs_reference_golden(dev, model, frame, golden):
    """The whole-model goldens are taken behind the 0.02 gain of the synthetic `output` conv (restormer.py SYNTH_RULES),
    which attenuates every trunk error ~50x.  Here the 96-channel output of `refinement`
    (reference restormer.py:274, values up to +-19, rms 3.3) of the full 512x512 tile 0 is compared directly with the
    reference's tensor (forward hook in gen_golden.py --only fullsize_frame): north_star's 1e-3 max-abs on an
    un-attenuated tensor, relative to its rms as well."""
    g = golden("restormer_fullsize_frame")
    tap = {}
    model._tap = tap
    try:
        model(_tile0(frame, dev))
    finally:
        model._tap = None
    r = tap["refinement"][0].cpu().numpy()
    assert r.shape == (96, PS, PS)
    err = max(np.abs(r[:, ::16, ::16] - g["refine_sub16"]).max(), np.abs(r[:, 100, :] - g["refine_row100"]).max())
    rms = float(np.sqrt((g["refine_sub16"].astype(np.float64) ** 2).mean()))
    m_err = np.abs(r.mean(axis=(1, 2)) - g["refine_mean"]).max()
    s_err = np.abs((r.astype(np.float64) ** 2).mean(axis=(1, 2)) - g["refine_sqmean"]).max()
    print(f"refinement output (96 x 512 x 512, rms {rms:.2f}, max |.| {np.abs(g['refine_sub16']).max():.1f}) vs reference: "
          f"max-abs {err:.3e} ({err / rms:.2e} of rms), channel mean {m_err:.2e}, mean square {s_err:.2e}")
    assert err <= 1e-3 and m_err <= 1e-4 and s_err <= 1e-3


def _check_sub(y, g, sub, row):
    err = max(np.abs(y[:, ::sub, ::sub] - g[f"sub{sub}"]).max(), np.abs(y[:, row, :] - g[f"row{row}"]).max())
    m_err = np.abs(y.mean(axis=(1, 2)) - g["mean"]).max()
    s_err = np.abs((y.astype(np.float64) ** 2).mean(axis=(1, 2)) - g["sqmean"]).max()
    return err, m_err, s_err


def test_fullsize_c3_biasfree_tile_vs_reference_golden(dev, golden):
    """BASELINE configs[2]: Restormer colour blind-denoise (BiasFree LayerNorm) on one full 256x256 tile of the 512x512
    frame, against the imported reference's output on the same (noisy) tile (gen_golden.py --only fullsize_c3)."""
    g = golden("restormer_fullsize_c3")
    model = restormer.Restormer(LayerNorm_type="BiasFree").load_synthetic(42).eval().to(dev)
    x = torch.from_numpy(g["x"])[None].to(dev)
    y = model(x)[0].cpu().numpy()
    err, m_err, s_err = _check_sub(y, g, 4, 77)
    print(f"c3 256x256 BiasFree tile vs reference golden: max-abs {err:.3e}, channel mean {m_err:.2e}, mean square {s_err:.2e}")
    assert err <= 1e-3 and m_err <= 1e-5 and s_err <= 1e-5


def test_fullsize_c3_nine_tiles_properties(dev):
    """configs[2] whole frame (512x512, 9 tiles of 256 with overlap): run-to-run bit identity and batched == per-tile
    after requantisation."""
    model = restormer.Restormer(LayerNorm_type="BiasFree").load_synthetic(42).eval().to(dev)
    inp, _ = synth.synth_image_pair(3, 512, 512, 3, seed_base=1000, blur=0)
    cfg = utils.get_patch_config("denoising", "gaussian", "Restormer")          # 256 / 48
    img = torch.from_numpy(inp).to(dev)
    a, _ = utils.tiled_forward_device(model, img, cfg["patch_size"], cfg["patch_overlap"], pad8=True, max_batch=9)
    b, _ = utils.tiled_forward_device(model, img, cfg["patch_size"], cfg["patch_overlap"], pad8=True, max_batch=9)
    assert torch.equal(a, b)
    c, _ = utils.tiled_forward_device(model, img, cfg["patch_size"], cfg["patch_overlap"], pad8=True, max_batch=1)
    diff = (a.int() - c.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3


def test_fullsize_c5_mairunet_vs_reference_golden(dev, golden):
    """BASELINE configs[4]: MaIRUNet on a full 256x256 crop (scan length 65 536 at level 1, the product scan_plan
    chunking) against the imported reference module's output (scan op = oracle stand-in: that arithmetic stays
    unpinned)."""
    from irm_amd.mair import mairunet_arch
    g = golden("mair_fullsize_c5")
    model = mairunet_arch.MaIRUNet(inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4,
                                   ssm_ratio=2.0, flp_ratio=4.0, mlp_ratio=1.5, bias=False, dual_pixel_task=False,
                                   img_size=256, scan_len=4, batch_size=1, dynamic_ids=False)
    model = model.load_synthetic(42).eval().to(dev)
    x = synth.uniform(7, "mair_in_256x256", (1, 3, 256, 256), 0.0, 1.0).to(dev)
    y = model(x)[0].cpu().numpy()
    err, m_err, s_err = _check_sub(y, g, 4, 77)
    print(f"c5 MaIRUNet 256x256 vs reference golden: max-abs {err:.3e}, channel mean {m_err:.2e}, mean square {s_err:.2e}")
    assert err <= 1e-3 and m_err <= 1e-4 and s_err <= 1e-4


def test_fullsize_batch_independence_and_determinism(dev, model, frame):
    """Tile 0 alone, tile 0 inside the batch of 6, and a repeated run."""
    img = torch.from_numpy(frame[0]).to(dev)
    keep = []
    out1, _ = utils.tiled_forward_device(model, img, PS, OV, pad8=True, max_batch=8, keep_tiles=keep)
    preds = keep[0]
    ys, xs = tiler_ref.tile_origins(H, PS, OV), tiler_ref.tile_origins(W, PS, OV)
    assert preds.shape == (len(ys) * len(xs), 3, PS, PS) and preds.shape[0] == 6
    out2, _ = utils.tiled_forward_device(model, img, PS, OV, pad8=True, max_batch=8)
    assert torch.equal(out1, out2), "same frame twice must give the same bytes"
    single = model(_tile0(frame, dev))
    d = float((single - preds[:1]).abs().max())
    print(f"tile 0 alone vs inside the batch of 6: max-abs {d:.3e}")
    assert d <= 2e-5
    # one tile per launch (the reference's loop shape) against the batched launch, after requantisation
    out3, _ = utils.tiled_forward_device(model, img, PS, OV, pad8=True, max_batch=1)
    diff = (out1.int() - out3.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3


def test_fullsize_graph_replay_and_experimental_two_streams(dev, model, frame, monkeypatch):
    """HIP-graph replay of the per-batch forward gives the frame of the plain path bit for bit (same launches).
    Tile groups on two HIP streams are EXPERIMENTAL (refused without IRM_EXPERIMENTAL_STREAMS=1; not in bench.py's JSON):
    the Gram chunk plan follows the batch size (3 instead of 6 tiles per launch), so the frame may differ from the
    one-stream frame by 1 in a few hundred bytes, and on some GPUs of the pool overlapping forwards were not
    bit-reproducible run to run (cause not established; round 3: four runs bit-identical with and
    without an agent-scope acquire at every kernel entry - the defect did not show on that GPU at all)."""
    img = torch.from_numpy(frame[0]).to(dev)
    monkeypatch.setenv("IRM_NO_GRAPH", "1")
    base, _ = utils.tiled_forward_device(model, img, PS, OV, pad8=True, max_batch=8)
    base = base.clone()
    monkeypatch.delenv("IRM_NO_GRAPH")
    assert model.hip_graph
    model.__dict__.pop("_irm_graphs", None)
    for _ in range(2):                      # capture, then replay
        g, _ = utils.tiled_forward_device(model, img, PS, OV, pad8=True, max_batch=8)
        assert torch.equal(g, base)
    assert len(model._irm_graphs) == 1
    model.num_streams = 2
    try:
        with pytest.raises(ValueError):
            utils.tiled_forward_device(model, img, PS, OV, pad8=True, max_batch=8)
        monkeypatch.setenv("IRM_EXPERIMENTAL_STREAMS", "1")
        runs = [utils.tiled_forward_device(model, img, PS, OV, pad8=True, max_batch=8)[0].clone() for _ in range(3)]
    finally:
        model.num_streams = 1
    for two in runs:
        diff = (two.int() - base.int()).abs()
        print(f"two streams vs one: {int((diff > 0).sum())} of {diff.numel()} bytes differ, max {int(diff.max())}; "
              f"equal to the first two-stream run: {bool(torch.equal(two, runs[0]))}")
        assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 5e-3


class _PerPixel:
    """Stand-in model with exactly representable arithmetic: y = 1 - x (correctly rounded on both sides)."""
    num_streams = 1

    def __call__(self, t):
        return 1.0 - t


def test_fullsize_tiler_round_trip_and_oracle(dev, frame):
    inp = frame[0]
    img = torch.from_numpy(inp).to(dev)
    ident, _ = utils.tiled_forward_device(lambda t: t, img, PS, OV, pad8=True, max_batch=8)
    assert torch.equal(ident.cpu(), torch.from_numpy(inp)), "identity model: extract -> blend -> requantise is lossless"
    got, sse = utils.tiled_forward_device(_PerPixel(), img, PS, OV, pad8=True, target_dev=img, max_batch=8)
    ref = tiler_ref.tiled_inference(lambda t: 1.0 - t, inp, patch_size=PS, patch_overlap=OV, pad=tiler_ref.reflect_pad8)
    assert np.array_equal(got.cpu().numpy(), ref), "device tiler vs numpy oracle at 1280x720"
    want = int(((ref.astype(np.int64) - inp.astype(np.int64)) ** 2).sum())
    assert int(sse.item()) == want, "integer squared error accumulates exactly"


def test_fullsize_two_frames_in_one_batch(dev, model, golden):
    """utils.tiled_forward_device_batch (bench.py's step: the 12 tiles of two 1280x720 frames in one batched forward):
    every frame equals its single-frame result up to the batch-dependent Gram summation order (+-1 in < 0.1 % of the
    bytes, PSNR within 0.001 dB), the batched call reproduces bit for bit, and frame 0 stays within the whole-frame
    golden's bar."""
    pairs = [synth.synth_image_pair(i, H, W, 3, seed_base=1000, blur=15) for i in range(2)]
    imgs = [torch.from_numpy(p[0]).to(dev) for p in pairs]
    tgts = [torch.from_numpy(p[1]).to(dev) for p in pairs]
    assert model.max_tiles_per_batch >= 12
    a = utils.tiled_forward_device_batch(model, imgs, PS, OV, pad8=True, targets_dev=tgts, max_batch=model.max_tiles_per_batch)
    b = utils.tiled_forward_device_batch(model, imgs, PS, OV, pad8=True, targets_dev=tgts, max_batch=model.max_tiles_per_batch)
    for (o1, s1), (o2, s2) in zip(a, b):
        assert torch.equal(o1, o2) and int(s1.item()) == int(s2.item())
    for k in range(2):
        single, sse = utils.tiled_forward_device(model, imgs[k], PS, OV, pad8=True, target_dev=tgts[k], max_batch=8)
        diff = (a[k][0].int() - single.int()).abs()
        p_b = 10 * np.log10(255.0 ** 2 / (float(a[k][1].item()) / single.numel()))
        p_s = 10 * np.log10(255.0 ** 2 / (float(sse.item()) / single.numel()))
        print(f"frame {k}: batched vs alone {int((diff > 0).sum())} bytes differ (max {int(diff.max())}), PSNR {p_b:.5f} vs {p_s:.5f}")
        assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3 and abs(p_b - p_s) < 1e-3
    g = golden("restormer_fullsize_frame")
    d0 = np.abs(a[0][0].cpu().numpy().astype(np.int32) - g["pred_u8"].astype(np.int32))
    assert int(d0.max()) <= 1 and float((d0 > 0).mean()) < 1e-3
