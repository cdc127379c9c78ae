"""GPU end-to-end tests of the call surface around the hot path:
the benchmark harness row (scripts/tests.py:389-424) through the real GPU path, and the weight-file layouts of the two
families whose loaders had no load test (src/deblurganv2/__init__.py:35-36, BasicSR load_network
base_model.py:277-304) followed by a GPU forward."""
import csv
import os

import numpy as np
import pytest
import torch
import yaml

from irm_amd import deblurganv2, dncnn, harness, mair, parallel, synth, utils
from irm_amd.deblurganv2.models.fpn_mobilenet import FPNMobileNet
from oracle import convnets_ref, tiler_ref

pytestmark = pytest.mark.gpu


def test_harness_evaluate_rows_vs_oracle_tiler(dev, tmp_path):
    """harness.evaluate on 2 synthetic frames (DnCNN colour, 3 x 2 tiles of 256 with overlap): each PSNR equals the
    oracle tiler + oracle model on the same frame within 0.01 dB, times are positive, and save_results writes the
    reference's columns with those values."""
    model = dncnn.DnCNN(3, 3, 64, 20, "R").load_synthetic(42).eval().to(dev)
    cfg = utils.get_patch_config("denoising", "gaussian", "DnCNN")
    frames = list(harness.synthetic_loader(2, h=300, w=560, c=3, seed_base=77, blur=3))
    row = harness.evaluate(model, iter(frames), dev, cfg, task="denoising", subtask="gaussian", dataset="synthetic",
                           model_name="DnCNN", sigma=25, need_degradation=True, noise_level=25)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}

    def oracle_model(t):
        with torch.no_grad():
            return convnets_ref.dncnn_forward(t, sd)
    want = []
    for inp, tgt, _ in frames:
        pred = tiler_ref.tiled_inference(oracle_model, inp, patch_size=cfg["patch_size"], patch_overlap=cfg["patch_overlap"],
                                         need_degradation=True, noise_level=25)
        want.append(tiler_ref.psnr(tgt, pred))
    assert abs(row["PSNR"] - float(np.mean(want))) <= 0.01 and abs(row["Std_PSNR"] - float(np.std(want))) <= 0.01
    assert row["Avg_Time_ms"] > 0 and row["Std_Time_ms"] >= 0 and 0.0 < row["SSIM"] <= 1.0
    assert row["Model_Params"] == 668227 and row["Task"] == "Denoising" and row["Sigma"] == 25
    path = harness.save_results([row], out_dir=str(tmp_path))
    with open(path) as f:
        rd = list(csv.DictReader(f))
    assert list(rd[0].keys()) == harness.COLUMNS and abs(float(rd[0]["PSNR"]) - row["PSNR"]) < 1e-9


def test_harness_reports_a_failed_frame_and_keeps_the_rest(dev, tmp_path):
    """a frame that raises (here: a 2-D float array the tiler cannot index) is listed in the row's
    'Failed' entry with its name, the statistics cover the frames that ran, the CSV keeps the reference's columns;
    skip_failed=False propagates like the reference's loop (scripts/tests.py:389-398 catches nothing)."""
    model = dncnn.DnCNN(1, 1, 64, 17, "R").load_synthetic(42).eval().to(dev)
    cfg = utils.get_patch_config("denoising", "gaussian", "DnCNN")
    good = list(harness.synthetic_loader(2, h=64, w=96, c=1, seed_base=5, blur=3))
    bad = (np.zeros((64, 96), np.float32), good[0][1], "broken.png")
    kw = dict(task="denoising", subtask="gaussian", dataset="synthetic", model_name="DnCNN", with_ssim=False)
    row = harness.evaluate(model, iter([good[0], bad, good[1]]), dev, cfg, **kw)
    assert [n for n, _ in row["Failed"]] == ["broken.png"] and np.isfinite(row["PSNR"])
    ref = harness.evaluate(model, iter(good), dev, cfg, **kw)
    assert ref["Failed"] == [] and abs(ref["PSNR"] - row["PSNR"]) < 1e-9 and abs(ref["Std_PSNR"] - row["Std_PSNR"]) < 1e-9
    with open(harness.save_results([row], out_dir=str(tmp_path))) as f:
        assert list(csv.DictReader(f).fieldnames) == harness.COLUMNS
    with pytest.raises(Exception):
        harness.evaluate(model, iter([bad]), dev, cfg, skip_failed=False, **kw)


def test_gather_carries_failed_image_ids(dev):
    t, table, failed = parallel.gather_results(0.5, [(0, 30.0), (2, 31.0)], dev, failed_ids=[1])
    assert t == 0.5 and table.shape == (2, 2) and failed == [1]


def test_deblurgan_checkpoint_layout_then_forward(dev, tmp_path):
    """{'model': {'module.<key>': tensor}} written the way the reference's checkpoints are, loaded by get_model,
    then one GPU forward equal to the directly constructed model."""
    direct = FPNMobileNet().load_synthetic(42)
    sd = {"module." + k: v.clone() for k, v in direct.state_dict().items()}
    path = os.path.join(tmp_path, "fpn_mobilenet.h5")
    torch.save({"model": sd}, path)
    loaded = deblurganv2.get_model(path, dev)
    assert loaded.training                                                   # reference returns it in train mode
    x = synth.uniform(7, "dg_in", (1, 3, 64, 96), -1.0, 1.0).to(dev)
    assert torch.equal(loaded(x), direct.to(dev).train(True)(x))
    with pytest.raises(NotImplementedError):
        deblurganv2.get_model(os.path.join(tmp_path, "fpn_inception.h5"), dev)


@pytest.mark.parametrize("prefix", ["", "module."])
def test_mair_checkpoint_layout_then_forward(dev, tmp_path, prefix):
    """BasicSR layout {'params': {...}} with and without the DataParallel prefix + an option file -> get_model."""
    net_g = dict(type="MaIRUNet", inp_channels=3, out_channels=3, dim=48, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1,
                 ssm_ratio=2.0, flp_ratio=4.0, mlp_ratio=1.5, bias=False, dual_pixel_task=False, img_size=32, scan_len=4,
                 batch_size=1, dynamic_ids=False)
    kw = dict(net_g)
    kw.pop("type")
    direct = mair.MaIRUNet(**kw).load_synthetic(5)
    wpath = os.path.join(tmp_path, "net_g.pth")
    torch.save({"params": {prefix + k: v.clone() for k, v in direct.state_dict().items()}}, wpath)
    opt = os.path.join(tmp_path, "opt.yml")
    with open(opt, "w") as f:
        yaml.safe_dump({"name": "t", "num_gpu": 1, "network_g": net_g,
                        "path": {"pretrain_network_g": wpath, "strict_load_g": True}}, f)
    loaded = mair.get_model(opt)
    assert not loaded.training and next(loaded.parameters()).is_cuda
    x = synth.uniform(7, "mair_ld", (1, 3, 32, 32), 0.0, 1.0).to(dev)
    assert torch.equal(loaded(x), direct.eval().to(dev)(x))
