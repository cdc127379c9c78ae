"""GPU parity tests of the assembled models and the device tiler against
(a) the committed golden vectors (outputs of the imported reference, written by
oracle/gen_golden.py) and (b) the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): float outputs within 1e-3 max-abs of the
reference's PyTorch-CPU fp32 forward; PSNR within 0.01 dB; the tiler's integer
and float32 blend arithmetic bit-exact."""
import numpy as np
import pytest
import torch

from irm_amd import _hip, dncnn, ops, rednet, restormer, synth, utils
from oracle import convnets_ref, restormer_ref, tiler_ref

pytestmark = pytest.mark.gpu
TOL = 1e-3

CFGS = {
    "deblur_withbias": dict(inp_channels=3, out_channels=3, LayerNorm_type="WithBias"),
    "denoise_biasfree": dict(inp_channels=3, out_channels=3, LayerNorm_type="BiasFree"),
    "gray_biasfree": dict(inp_channels=1, out_channels=1, LayerNorm_type="BiasFree"),
    "dualpixel_withbias": dict(inp_channels=6, out_channels=3, LayerNorm_type="WithBias", dual_pixel_task=True),
}


def gin(name, shape, lo=0.0, hi=1.0):
    return synth.uniform(7, name, shape, lo, hi)      # same generator as oracle/gen_golden.py


# --------------------------------------------------------------------------- Restormer
@pytest.mark.parametrize("cfg", list(CFGS))
def test_restormer_vs_golden(dev, golden, cfg):
    kw = CFGS[cfg]
    model = restormer.Restormer(**kw).load_synthetic(42).eval().to(dev)
    x = gin(f"restormer_in_{cfg}_64x64", (1, kw["inp_channels"], 64, 64))
    y = model(x.to(dev)).cpu().numpy()
    ref = golden("restormer_forward")[f"{cfg}_64x64"]
    err = np.abs(y - ref).max()
    print(f"{cfg}: max-abs vs reference golden {err:.3e}")
    assert err <= TOL


def test_restormer_nonsquare_odd_levels_vs_golden(dev, golden):
    """40x56 input: level 4 is 5x7, so every kernel takes its unaligned (scalar) path - golden from the reference."""
    model = restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
    x = gin("restormer_in_deblur_withbias_40x56", (1, 3, 40, 56))
    y = model(utils.pad(x.to(dev))).cpu().numpy()[:, :, :40, :56]
    assert np.abs(y - golden("restormer_forward")["deblur_withbias_40x56"]).max() <= TOL


def test_restormer_batch_matches_single_and_is_deterministic(dev):
    model = restormer.Restormer(LayerNorm_type="BiasFree").load_synthetic(42).eval().to(dev)
    x = gin("batchcheck", (3, 3, 64, 72)).to(dev)
    yb = model(x).clone()
    yb2 = model(x).clone()
    assert torch.equal(yb, yb2), "two runs on the same input must be bit-identical (no atomics in the path)"
    for i in range(3):
        yi = model(x[i:i + 1])
        assert (yi - yb[i:i + 1]).abs().max() <= 2e-5


def test_restormer_vs_oracle_128(dev):
    model = restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
    x = gin("orc128", (1, 3, 128, 128))
    y = model(x.to(dev)).cpu()
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = restormer_ref.restormer_forward(x, sd)
    err = float((y - ref).abs().max())
    print(f"restormer 128x128 max-abs vs oracle {err:.3e}")
    assert err <= TOL


TB_CASES = [(48, 1, 16, 24, "WithBias"), (48, 1, 16, 24, "BiasFree"), (96, 2, 16, 16, "WithBias"),
            (96, 1, 8, 40, "BiasFree"), (192, 4, 8, 8, "BiasFree"), (384, 8, 8, 8, "WithBias")]


@pytest.mark.parametrize("c,heads,h,w,ln", TB_CASES)
def test_transformer_block_vs_golden(dev, golden, c, heads, h, w, ln):
    tag = f"c{c}_h{heads}_{h}x{w}_{ln}"
    host = restormer.Restormer(LayerNorm_type=ln)          # only used for its block driver / workspace
    blk = restormer.restormer.TransformerBlock(c, heads, 2.66, False, ln)
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    blk.load_state_dict(synth.synth_state_dict(shapes, seed=11, rules=restormer.restormer.SYNTH_RULES))
    blk = blk.to(dev)
    x = gin("tb_in_" + tag, (2, c, h, w), -1.0, 1.0).to(dev)
    a, ff = blk.attn, blk.ffn
    f32 = lambda t: None if t is None else t.detach().float().contiguous()  # noqa: E731
    wts = dict(qkv=_hip.pack_gemm_weight(a.qkv.weight), qkv_b=None, qkv_dw=f32(a.qkv_dwconv.weight.reshape(-1, 9)),
               qkv_dw_b=None, wout=f32(a.project_out.weight.reshape(c, c)), wout_b=None,
               temp=f32(a.temperature.reshape(-1)), pin=_hip.pack_gemm_weight(ff.project_in.weight), pin_b=None,
               ffn_dw=f32(ff.dwconv.weight.reshape(-1, 9)), ffn_dw_b=None,
               pout=_hip.pack_gemm_weight(ff.project_out.weight), pout_b=None,
               n1w=f32(blk.norm1.w), n1b=f32(blk.norm1.b), n2w=f32(blk.norm2.w), n2b=f32(blk.norm2.b))
    y = x.clone()
    host._block(blk, wts, y)
    g = golden("restormer_ops")
    err = np.abs(y.cpu().numpy() - g["tb_" + tag + "_out"]).max()
    print(f"block {tag}: max-abs vs reference golden {err:.3e}")
    assert err <= 2e-4


@pytest.mark.parametrize("c,heads,h,w,ln", TB_CASES)
def test_transformer_block_product_path_vs_golden(dev, golden, c, heads, h, w, ln):
    """The same reference goldens through the path users and bench.py run: weights packed by Restormer._pack()
    (fp16 hi/lo splits, range guard, fused branch kernels for C <= 96) and the stage driver _run_stage."""
    import os
    import torch.nn as nn
    if os.environ.get("IRM_GEMM_EXACT"):
        pytest.skip("IRM_GEMM_EXACT=1 selects the f32-MFMA entries: not the path this test is about")
    tag = f"c{c}_h{heads}_{h}x{w}_{ln}"
    host = restormer.Restormer(LayerNorm_type=ln)
    blk = restormer.restormer.TransformerBlock(c, heads, 2.66, False, ln)
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    blk.load_state_dict(synth.synth_state_dict(shapes, seed=11, rules=restormer.restormer.SYNTH_RULES))
    host.encoder_level1 = nn.Sequential(blk)
    host = host.to(dev)
    pk = host._pack()
    wt = pk["encoder_level1.0"]
    assert host._split and "qkv_s" in wt and "pin_s" in wt and "pout_s" in wt
    assert ("gdfn_f" in wt) == (c <= 96)
    x = gin("tb_in_" + tag, (2, c, h, w), -1.0, 1.0).to(dev)
    y = x.clone()
    host._run_stage("encoder_level1", pk, y)
    g = golden("restormer_ops")
    err = np.abs(y.cpu().numpy() - g["tb_" + tag + "_out"]).max()
    print(f"block {tag} (product path): max-abs vs reference golden {err:.3e}")
    assert err <= 2e-4


@pytest.mark.parametrize("c,h,w", [(48, 16, 24), (96, 8, 16)])
def test_resample_vs_golden(dev, golden, c, h, w):
    g = golden("restormer_ops")
    x = gin(f"down_in_{c}", (2, c, h, w), -1, 1).to(dev)
    wd = synth.synth_state_dict({"body.0.weight": (c // 2, c, 3, 3)}, seed=12)["body.0.weight"]
    y = torch.empty(2, 2 * c, h // 2, w // 2, device=dev)
    ops.conv3x3(_hip.pack_conv3x3_weight(wd).to(dev), x, y, c, c // 2, store_mode=1)
    assert np.abs(y.cpu().numpy() - g[f"down_{c}_{h}x{w}"]).max() <= 2e-4
    wu = synth.synth_state_dict({"body.0.weight": (2 * c, c, 3, 3)}, seed=13)["body.0.weight"]
    y = torch.empty(2, c // 2, 2 * h, 2 * w, device=dev)
    ops.conv3x3(_hip.pack_conv3x3_weight(wu).to(dev), x, y, c, 2 * c, store_mode=2)
    assert np.abs(y.cpu().numpy() - g[f"up_{c}_{h}x{w}"]).max() <= 2e-4


# --------------------------------------------------------------------------- DnCNN / REDNet
@pytest.mark.parametrize("tag,nch,nb", [("gray17", 1, 17), ("gray20", 1, 20), ("color20", 3, 20)])
def test_dncnn_vs_golden(dev, golden, tag, nch, nb):
    model = dncnn.DnCNN(nch, nch, 64, nb, "R").load_synthetic(42).eval().to(dev)
    for h, w in [(32, 32), (40, 72)]:
        x = gin(f"dncnn_in_{tag}_{h}x{w}", (1, nch, h, w))
        y = model(x.to(dev)).cpu().numpy()
        err = np.abs(y - golden("convnets_forward")[f"dncnn_{tag}_{h}x{w}"]).max()
        print(f"dncnn {tag} {h}x{w}: max-abs {err:.3e}")
        assert err <= TOL


def test_rednet_vs_golden(dev, golden):
    model = rednet.REDNet().load_synthetic(42).eval().to(dev)
    for h, w in [(32, 32), (24, 40)]:
        x = gin(f"rednet_in_{h}x{w}", (1, 1, h, w))
        y = model(x.to(dev)).cpu().numpy()
        err = np.abs(y - golden("convnets_forward")[f"rednet_{h}x{w}"]).max()
        print(f"rednet {h}x{w}: max-abs {err:.3e}")
        assert err <= TOL


def test_dncnn_odd_size_vs_oracle(dev):
    model = dncnn.DnCNN(1, 1, 64, 17, "R").load_synthetic(42).eval().to(dev)
    x = gin("dn_odd", (2, 1, 37, 53))
    y = model(x.to(dev)).cpu()
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = convnets_ref.dncnn_forward(x, sd)
    assert float((y - ref).abs().max()) <= TOL


# --------------------------------------------------------------------------- tiler
class _Replay(torch.nn.Module):
    """Stands in for a model: records the tiles it is given and returns canned predictions."""

    def __init__(self, preds):
        super().__init__()
        self.preds, self.seen, self.i = preds, [], 0

    def forward(self, t):
        self.seen.append(t.detach().cpu().clone())
        n = t.shape[0]
        out = self.preds[self.i:self.i + n].to(t.device)
        self.i += n
        return out


@pytest.mark.parametrize("h,w,c,ps,ov,pad8,sigma,dtype", [
    (100, 136, 3, 64, 16, True, None, np.uint8),
    (75, 61, 3, 50, 10, True, 25, np.uint8),          # tiles 50x50 -> padded 56x56, noise path
    (150, 210, 1, 64, 16, False, 15, np.uint8),
    (40, 90, 3, 64, 16, True, None, np.uint16),       # image shorter than the patch
    (64, 64, 3, None, 32, False, None, np.uint8),
])
def test_device_tiler_bit_exact_vs_oracle(dev, h, w, c, ps, ov, pad8, sigma, dtype):
    rng = np.random.default_rng(5)
    peak = 255 if dtype == np.uint8 else 65535
    img = rng.integers(0, peak + 1, size=(h, w, c)).astype(dtype)
    # oracle run with canned per-tile predictions
    ps_eff = min(ps, max(h, w)) if ps else max(h, w)
    th, tw = min(ps_eff, h), min(ps_eff, w)
    ph = (th // 8 + 1) * 8 if (pad8 and th % 8) else th
    pw = (tw // 8 + 1) * 8 if (pad8 and tw % 8) else tw
    nt = len(tiler_ref.tile_origins(h, ps_eff, ov)) * len(tiler_ref.tile_origins(w, ps_eff, ov)) if ps else 1
    preds = torch.from_numpy(rng.uniform(-0.2, 1.2, size=(nt, min(3, c), ph, pw)).astype(np.float32))
    seen_ref, k = [], [0]

    def fake(t):
        seen_ref.append(t.clone())
        o = preds[k[0]:k[0] + 1]
        k[0] += 1
        return o
    ref = tiler_ref.tiled_inference(fake, img, patch_size=ps, patch_overlap=ov,
                                    need_degradation=sigma is not None, noise_level=sigma,
                                    pad=tiler_ref.reflect_pad8 if pad8 else None)
    # device run
    rep = _Replay(preds)
    src = img if dtype == np.uint8 else img.view(np.int16)
    out, _ = utils.tiled_forward_device(rep, torch.from_numpy(src.copy()).to(dev), ps, ov, pad8, sigma, max_batch=4)
    got = out.cpu().numpy()
    if dtype == np.uint16:
        got = got.view(np.uint16)
    tiles_dev = torch.cat(rep.seen)
    tiles_ref = torch.cat(seen_ref)
    assert torch.equal(tiles_dev, tiles_ref), "tile extraction (normalise / noise / reflect pad) must be bit-exact"
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {ref.size} output values differ"


def test_tiled_restormer_vs_golden(dev, golden, manifest):
    """End to end: u8 image -> device tiler -> Restormer -> blend, vs the reference's run_model_inference."""
    model = restormer.Restormer(LayerNorm_type="WithBias").load_synthetic(42).eval().to(dev)
    img, tgt = synth.synth_image_pair(2, 100, 136, 3, seed_base=1000, blur=7)
    pred, ms = utils.run_model_inference(model, img, dev, pad=utils.pad, patch_size=64, patch_overlap=16)
    ref = golden("tiler")["restormer_tiled"]
    diff = np.abs(pred.astype(int) - ref.astype(int))
    p_gpu, p_ref = tiler_ref.psnr(tgt, pred), manifest["restormer_tiled_psnr"]
    print(f"tiled restormer: {int((diff > 0).sum())}/{ref.size} u8 values differ (max {diff.max()}), "
          f"PSNR gpu {p_gpu:.4f} ref {p_ref:.4f}")
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01
    assert abs(p_gpu - p_ref) < 0.01


@pytest.mark.parametrize("tag,kw", [("dncnn_tiled", dict(patch_size=64, patch_overlap=16)),
                                    ("dncnn_tiled_noise", dict(patch_size=64, patch_overlap=16, need_degradation=True,
                                                               noise_level=25)),
                                    ("dncnn_whole", dict(patch_size=None))])
def test_tiled_dncnn_vs_golden(dev, golden, tag, kw):
    model = dncnn.DnCNN(1, 1, 64, 17, "R").load_synthetic(42).eval().to(dev)
    img, _ = synth.synth_image_pair(1, 150, 210, 1, seed_base=3000, blur=0)
    pred, _ = utils.run_model_inference(model, img, dev, **kw)
    ref = golden("tiler")[tag]
    diff = np.abs(pred.astype(int) - ref.astype(int))
    print(f"{tag}: {int((diff > 0).sum())}/{ref.size} u8 values differ (max {diff.max()})")
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01


@pytest.mark.parametrize("tag,nb", [("nonblind_nb17", 17), ("blind_nb20", 20)])
def test_demo_case_vs_reference_golden(dev, golden, tag, nb):
    """BASELINE.json configs[0] (scripts/test_demo.py): DnCNN gray on the reference's 256x256 demo image through
    get_patch_config + get_model_prediction; golden = the reference's own functions on the CPU (synthetic weights)."""
    g = golden("demo_c1")
    model = dncnn.DnCNN(1, 1, 64, nb, "R").load_synthetic(42).eval().to(dev)
    cfg = utils.get_patch_config("denoising", "gaussian", "DnCNN")
    pred, ms = utils.get_model_prediction(model, g["noisy_u8"], dev, **cfg)
    diff = np.abs(pred.astype(int) - g[tag].astype(int))
    print(f"demo {tag}: {int((diff > 0).sum())}/{diff.size} u8 values differ (max {diff.max()})")
    assert pred.shape == (256, 256, 1) and diff.max() <= 1 and (diff > 0).mean() < 0.01


def test_get_model_prediction_surface(dev):
    """Same call as scripts/tests.py:391: returns (uint8 HWC, ms)."""
    model = restormer.Restormer(LayerNorm_type="BiasFree").load_synthetic(42).eval().to(dev)
    img, _ = synth.synth_image_pair(3, 72, 88, 3, seed_base=2000, blur=0)
    cfg = utils.get_patch_config("denoising", "gaussian", "Restormer")
    pred, ms = utils.get_model_prediction(model, img, dev, **cfg, need_degradation=True, noise_level=25)
    assert pred.dtype == np.uint8 and pred.shape == img.shape and ms > 0
    with pytest.raises(_hip.HipLibraryError):
        model(torch.zeros(1, 3, 64, 64))            # CPU input: no silent fallback


@pytest.mark.parametrize("family", ["dncnn", "rednet", "restormer", "mair"])
def test_graph_replay_equals_eager(dev, family, monkeypatch):
    """utils.graphed_forward: the HIP-graph replay of the per-batch forward gives the same bytes as eager launches,
    also on the second and third image (static buffers reused) and after the weights changed (re-capture)."""
    if family == "dncnn":
        model, c, cfg = dncnn.DnCNN(1, 1, 64, 17, "R").load_synthetic(1).eval().to(dev), 1, dict(ps=64, ov=16, pad8=False)
    elif family == "rednet":
        model, c, cfg = rednet.REDNet().load_synthetic(1).eval().to(dev), 1, dict(ps=64, ov=16, pad8=False)
    elif family == "restormer":
        model, c, cfg = restormer.Restormer(LayerNorm_type="BiasFree").load_synthetic(1).eval().to(dev), 3, dict(ps=64, ov=16, pad8=True)
    else:
        from irm_amd import mair
        model = mair.MaIRUNet(dim=48, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, ssm_ratio=2.0, flp_ratio=4.0,
                              mlp_ratio=1.5, scan_len=4).load_synthetic(1).eval().to(dev)
        c, cfg = 3, dict(ps=64, ov=16, pad8=True)
    assert model.hip_graph
    imgs = [torch.from_numpy(synth.synth_image_pair(i, 96, 112, c, seed_base=50, blur=3)[0]).to(dev) for i in range(3)]

    def run(i):
        return utils.tiled_forward_device(model, imgs[i], cfg["ps"], cfg["ov"], pad8=cfg["pad8"], max_batch=2)[0].clone()
    monkeypatch.setenv("IRM_NO_GRAPH", "1")
    eager = [run(i) for i in range(3)]
    monkeypatch.delenv("IRM_NO_GRAPH")
    model.__dict__.pop("_irm_graphs", None)
    graphed = [run(i) for i in range(3)] + [run(0)]
    assert len(model._irm_graphs) >= 1
    for a, b in zip(eager + [eager[0]], graphed):
        assert torch.equal(a, b)
    # new weights -> the old graph must not be replayed
    model.load_synthetic(2)
    monkeypatch.setenv("IRM_NO_GRAPH", "1")
    e2 = run(1)
    monkeypatch.delenv("IRM_NO_GRAPH")
    assert torch.equal(run(1), e2) and not torch.equal(e2, eager[1])


def test_graphs_own_their_workspace_across_alternating_shapes(dev, monkeypatch):
    """DnCNN keeps layer buffers per (B, H, W); a 256^2 image (1 tile), then a 512^2 image (9 tiles at
    patch 256 / overlap 32: batches of 8 + 1), then the 256^2 image again replay the B = 1 graph after the buffers it
    was captured with were replaced.  Every frame must equal the eager bytes; so must a frame after a MIDDLE layer's
    Parameter object was swapped (the weight fingerprint has to see it)."""
    from irm_amd import dncnn
    model = dncnn.DnCNN(1, 1, 64, 17, "R").load_synthetic(42).eval().to(dev)
    cfg = utils.get_patch_config("denoising", "gaussian", "DnCNN")
    imgs = [torch.from_numpy(synth.synth_image_pair(i, s, s, 1, seed_base=60, blur=3)[0]).to(dev)
            for i, s in enumerate((256, 512, 256, 512, 256))]

    def run(i):
        return utils.tiled_forward_device(model, imgs[i], cfg["patch_size"], cfg["patch_overlap"], pad8=False,
                                          noise_sigma=25, max_batch=8)[0].clone()
    monkeypatch.setenv("IRM_NO_GRAPH", "1")
    eager = [run(i) for i in range(len(imgs))]
    monkeypatch.delenv("IRM_NO_GRAPH")
    model.__dict__.pop("_irm_graphs", None)
    graphed = [run(i) for i in range(len(imgs))]
    assert len(model._irm_graphs) == 2                   # B = 1 (also the ninth tile of the 512^2 image) and B = 8
    for i, (a, b) in enumerate(zip(eager, graphed)):
        assert torch.equal(a, b), f"image {i}: graph replay differs from the eager bytes"
    conv = [m for m in model.model if isinstance(m, torch.nn.Conv2d)][8]
    conv.weight = torch.nn.Parameter(conv.weight.detach() * 0.5)
    monkeypatch.setenv("IRM_NO_GRAPH", "1")
    e2 = run(0)
    monkeypatch.delenv("IRM_NO_GRAPH")
    g2 = run(0)
    assert torch.equal(g2, e2) and not torch.equal(e2, eager[0])
    del model                                            # the graphs and their pools die with the model object
