"""CPU-only tests (python -m pytest tests -m "not gpu"): the oracle against the
committed golden vectors (reference outputs), the host-side logic of the
product package, and the C-ABI library (loads, exports every declared symbol -
no compute calls without a GPU)."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from irm_amd import _hip, configs, dncnn, rednet, restormer, synth, utils
from oracle import convnets_ref, mair_ref, restormer_ref, tiler_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gin(name, shape, lo=0.0, hi=1.0):
    return synth.uniform(7, name, shape, lo, hi)


# --------------------------------------------------------------------------- oracle vs golden
def test_oracle_was_pinned_to_reference(manifest):
    """oracle/gen_golden.py recorded oracle-vs-imported-reference differences: all must be ~0."""
    diffs = manifest["oracle_vs_reference"]
    assert len(diffs) >= 20
    assert max(diffs.values()) <= 2e-5


@pytest.mark.parametrize("cfg,kw", [("deblur_withbias", dict(LayerNorm_type="WithBias")),
                                    ("gray_biasfree", dict(inp_channels=1, out_channels=1, LayerNorm_type="BiasFree"))])
def test_oracle_restormer_vs_golden(golden, manifest, cfg, kw):
    shapes = {k: tuple(v) for k, v in manifest["restormer_param_shapes"][cfg].items()}
    sd = synth.synth_state_dict(shapes, seed=42, rules=restormer.restormer.SYNTH_RULES)
    x = gin(f"restormer_in_{cfg}_64x64", (1, kw.get("inp_channels", 3), 64, 64))
    with torch.no_grad():
        y = restormer_ref.restormer_forward(x, sd).numpy()
    assert np.abs(y - golden("restormer_forward")[f"{cfg}_64x64"]).max() <= 1e-5


def test_oracle_convnets_vs_golden(golden, manifest):
    shapes = {k: tuple(v) for k, v in manifest["dncnn_param_shapes"]["gray17"].items()}
    sd = synth.synth_state_dict(shapes, seed=42, rules=dncnn.SYNTH_RULES)
    x = gin("dncnn_in_gray17_32x32", (1, 1, 32, 32))
    with torch.no_grad():
        y = convnets_ref.dncnn_forward(x, sd).numpy()
    assert np.abs(y - golden("convnets_forward")["dncnn_gray17_32x32"]).max() <= 1e-5
    shapes = {k: tuple(v) for k, v in manifest["rednet_param_shapes"].items()}
    sd = synth.synth_state_dict(shapes, seed=42, rules=rednet.SYNTH_RULES)
    x = gin("rednet_in_24x40", (1, 1, 24, 40))
    with torch.no_grad():
        y = convnets_ref.rednet_forward(x, sd).numpy()
    assert np.abs(y - golden("convnets_forward")["rednet_24x40"]).max() <= 1e-5


def test_oracle_tiler_vs_golden(golden, manifest):
    """numpy tiler restatement + DnCNN oracle == the reference's run_model_inference output (u8 exact)."""
    shapes = {k: tuple(v) for k, v in manifest["dncnn_param_shapes"]["gray17"].items()}
    sd = synth.synth_state_dict(shapes, seed=42, rules=dncnn.SYNTH_RULES)
    img, _ = synth.synth_image_pair(1, 150, 210, 1, seed_base=3000, blur=0)
    pred = tiler_ref.tiled_inference(lambda t: convnets_ref.dncnn_forward(t, sd), img, patch_size=64,
                                     patch_overlap=16, need_degradation=True, noise_level=25)
    assert np.array_equal(pred, golden("tiler")["dncnn_tiled_noise"])
    assert np.array_equal(tiler_ref.gaussian_window(64, 64, 1)[:, :, 0], golden("tiler")["window_64"])
    tl = gin("noisecheck", (40, 48, 3)).numpy()
    assert np.array_equal(tiler_ref.degrade(tl, 25), golden("tiler")["noise_40x48x3_s25"])


def test_oracle_demo_case_vs_golden(golden, manifest):
    """BASELINE.json configs[0]: the reference's CPU demo case (DnCNN gray on its 256x256 demo image through
    get_patch_config + get_model_prediction, synthetic weights) - oracle tiler + oracle DnCNN, u8 exact."""
    g = golden("demo_c1")
    shapes = {k: tuple(v) for k, v in manifest["dncnn_param_shapes"]["gray17"].items()}
    sd = synth.synth_state_dict(shapes, seed=42, rules=dncnn.SYNTH_RULES)
    pred = tiler_ref.tiled_inference(lambda t: convnets_ref.dncnn_forward(t, sd), g["noisy_u8"], patch_size=256,
                                     patch_overlap=48)
    assert np.array_equal(pred, g["nonblind_nb17"])


def test_selective_scan_oracle_vs_independent_float64():
    """The scan op is unpinned by any reference artefact (mamba_ssm is absent): cross-check the oracle's
    fp32 restatement against an independent float64 evaluation written from the published recurrence."""
    torch.manual_seed(0)
    b, k, d, n, L = 2, 4, 5, 3, 37
    u, dt = torch.randn(b, k * d, L), torch.randn(b, k * d, L)
    A = -torch.rand(k * d, n) * 2 - 0.1
    B, C = torch.randn(b, k, n, L), torch.randn(b, k, n, L)
    D, bias = torch.randn(k * d), torch.randn(k * d)
    y = mair_ref.selective_scan(u, dt, A, B, C, D, delta_bias=bias, delta_softplus=True)
    ref = np.zeros((b, k * d, L))
    for bi in range(b):
        for r in range(k * d):
            h = np.zeros(n)
            for t in range(L):
                x = float(dt[bi, r, t]) + float(bias[r])
                dl = x if x > 20 else np.log1p(np.exp(x))
                h = np.exp(dl * A[r].double().numpy()) * h + dl * B[bi, r // d, :, t].double().numpy() * float(u[bi, r, t])
                ref[bi, r, t] = h @ C[bi, r // d, :, t].double().numpy() + float(D[r]) * float(u[bi, r, t])
    assert np.abs(y.numpy() - ref).max() < 1e-4


def test_mair_oracle_and_ids_vs_golden(golden, manifest):
    for key in golden("mair").files:
        if key.startswith("ids_"):
            h, w = map(int, key.split("_")[1].split("x"))
            sl = int(key.split("_s")[1])
            assert np.array_equal(mair_ref.scan_ids(h, w, sl)[0].numpy(), golden("mair")[key])
    from irm_amd.mair import mairunet_arch
    assert np.array_equal(mairunet_arch.scan_ids(6, 10, 4, "cpu").numpy(), golden("mair")["ids_6x10_s4"])
    for key in golden("mair").files:
        if key.startswith("shift_ids_"):
            h, w = map(int, key.split("_")[2].split("x"))
            assert np.array_equal(mair_ref.scan_ids(h, w, 4, 2)[0].numpy(), golden("mair")[key])
            assert np.array_equal(mairunet_arch.scan_ids(h, w, 4, "cpu", 2).numpy(), golden("mair")[key])
    shapes = {k: tuple(v) for k, v in manifest["mairunet_param_shapes"].items()}
    from irm_amd import mair
    assert {k: list(v.shape) for k, v in mair.MaIRUNet(
        dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4, ssm_ratio=2.0, flp_ratio=4.0, mlp_ratio=1.5,
        scan_len=4).state_dict().items()} == manifest["mairunet_param_shapes"]
    sd = synth.synth_state_dict(shapes, seed=42, rules=mair.SYNTH_RULES)
    x = gin("mair_in_24x40", (1, 3, 24, 40))
    with torch.no_grad():
        y = mair_ref.mairunet_forward(x, sd, scan_len=4).numpy()
    assert np.abs(y - golden("mair")["mairunet_24x40"]).max() <= 2e-5


# --------------------------------------------------------------------------- host logic
def test_tile_origins_match_reference(manifest):
    for key, (ps, ys, xs) in manifest["tile_origins"].items():
        name, size = key.split("|")
        h, w = map(int, size.split("x"))
        fam, idx = name[:-3], int(name[-2])
        entry = configs.PATCH_CONFIG[fam]
        entry = entry[idx] if isinstance(entry, list) else entry
        p = min(entry["patch_size"], max(h, w))
        assert p == ps
        assert utils.tile_origins(h, p, entry["patch_overlap"]) == ys
        assert utils.tile_origins(w, p, entry["patch_overlap"]) == xs
    # the headline case: 1280x720, Restormer deblur -> 2 x 3 tiles of 512
    assert utils.tile_origins(720, 512, 96) == [0, 208] and utils.tile_origins(1280, 512, 96) == [0, 416, 768]


def test_host_helpers_equal_oracle():
    assert np.array_equal(utils.get_gaussian_weights(96, 96, 3), tiler_ref.gaussian_window(96, 96, 3))
    t = gin("padcheck", (1, 3, 37, 50))
    assert torch.equal(utils.pad(t), tiler_ref.reflect_pad8(t))
    assert utils.pad(torch.zeros(1, 1, 16, 24)).shape == (1, 1, 16, 24)
    img = np.arange(24, dtype=np.uint8).reshape(2, 4, 3)
    assert np.array_equal(utils.normalize(img), tiler_ref.to_unit_range(img))
    tl = gin("n2", (8, 8, 3)).numpy()
    assert np.array_equal(utils.add_gaussian_noise(tl.copy(), 50), tiler_ref.degrade(tl, 50))


def test_patch_config_dispatch():
    assert utils.get_patch_config("denoising", "gaussian", "Restormer") == {"patch_size": 256, "patch_overlap": 48}
    assert utils.get_patch_config("deblurring", "motion", "Restormer") == {"patch_size": 512, "patch_overlap": 96}
    assert utils.get_patch_config("deblurring", "motion", "DeblurGANv2 (Inception)")["patch_size"] == 768
    assert utils.get_patch_config("deblurring", "motion", "DeblurGANv2 (MobileNet)")["patch_size"] == 2048
    assert utils.get_patch_config("denoising", "gaussian", "MaIR")["patch_size"] == 128
    assert utils.get_patch_config("denoising", "real", "MaIR")["patch_size"] == 384
    assert utils.get_patch_config("denoising", "gaussian", "DnCNN")["patch_overlap"] == 48
    assert utils.get_patch_config("denoising", "gaussian", "REDNet")["patch_size"] == 128
    assert utils.get_patch_config("denoising", "gaussian", "nothing") is None


def test_model_factory_errors(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)                       # no weights/ directory here
    with pytest.raises(FileNotFoundError):            # reference: propagates, callers skip (tests.py:48)
        utils.get_model_instance("deblurring", "motion", "Restormer", torch.device("cpu"))
    with pytest.raises(FileNotFoundError):
        utils.get_model_instance("denoising", "gaussian", "DnCNN", torch.device("cpu"), gray=True)
    with pytest.raises(ValueError, match="No model instance"):
        utils.get_model_instance("deblurring", "motion", "DnCNN", torch.device("cpu"))
    with pytest.raises(FileNotFoundError):
        utils.get_model_instance("deblurring", "motion", "DeblurGANv2 (MobileNet)", torch.device("cpu"))
    with pytest.raises(FileNotFoundError):
        utils.get_model_instance("denoising", "real", "MaIR", torch.device("cpu"))
    with pytest.raises(NotImplementedError):      # needs timm's InceptionResNetV2 (not vendored in the reference)
        open(tmp_path / "x", "w").close()
        os.makedirs("weights/DeblurGANv2")
        torch.save({"model": {}}, "weights/DeblurGANv2/fpn_inception.h5")
        utils.get_model_instance("deblurring", "motion", "DeblurGANv2 (Inception)", torch.device("cpu"))


def test_checkpoint_formats_load(tmp_path, monkeypatch, manifest):
    """restormer: yml + {'params': sd}; dncnn: raw state dict strict; rednet: strict=False."""
    monkeypatch.chdir(tmp_path)
    os.makedirs("weights/Restormer/deblurring")
    os.makedirs("weights/DnCNN")
    os.makedirs("weights/REDNet")
    m = restormer.Restormer().load_synthetic(1)
    torch.save({"params": m.state_dict()}, "weights/Restormer/deblurring/motion_deblurring.pth")
    got = utils.get_model_instance("deblurring", "motion", "Restormer", torch.device("cpu"))
    assert isinstance(got, restormer.Restormer) and not got.training
    assert all(torch.equal(a, b) for a, b in zip(got.state_dict().values(), m.state_dict().values()))
    d = dncnn.DnCNN(1, 1, 64, 20, "R").load_synthetic(2)
    torch.save(d.state_dict(), "weights/DnCNN/dncnn_gray_blind.pth")
    got = utils.get_model_instance("denoising", "gaussian", "DnCNN", torch.device("cpu"), gray=True)
    assert got.nb == 20 and torch.equal(got.model[0].weight, d.model[0].weight)
    r = rednet.REDNet().load_synthetic(3)
    sd = r.state_dict()
    sd.pop("conv3.bias")                              # strict=False tolerates a missing key
    torch.save(sd, "weights/REDNet/25.pt")
    got = utils.get_model_instance("denoising", "gaussian", "REDNet", torch.device("cpu"), sigma=25)
    assert torch.equal(got.deconv15.weight, r.deconv15.weight)


def test_product_modules_have_reference_parameter_layout(manifest):
    for cfg, kw in {"deblur_withbias": dict(LayerNorm_type="WithBias"),
                    "denoise_biasfree": dict(LayerNorm_type="BiasFree"),
                    "gray_biasfree": dict(inp_channels=1, out_channels=1, LayerNorm_type="BiasFree"),
                    "dualpixel_withbias": dict(inp_channels=6, dual_pixel_task=True)}.items():
        mine = {k: list(v.shape) for k, v in restormer.Restormer(**kw).state_dict().items()}
        assert mine == manifest["restormer_param_shapes"][cfg]
    assert sum(p.numel() for p in restormer.Restormer().parameters()) == 26126644
    for tag, (nch, nb) in {"gray17": (1, 17), "gray20": (1, 20), "color20": (3, 20)}.items():
        mine = {k: list(v.shape) for k, v in dncnn.DnCNN(nch, nch, 64, nb, "R").state_dict().items()}
        assert mine == manifest["dncnn_param_shapes"][tag]
    assert {k: list(v.shape) for k, v in rednet.REDNet().state_dict().items()} == manifest["rednet_param_shapes"]


def test_weight_packing_layout():
    w = synth.uniform(1, "pw", (37, 29), -1, 1)
    wp = _hip.pack_gemm_weight(w)
    mt, ks = 3, 8
    assert wp.numel() == mt * ks * 64
    for (m, k) in [(0, 0), (5, 7), (36, 28), (17, 13)]:
        lane = (m & 15) + 16 * (k & 3)
        assert wp[((m >> 4) * ks + (k >> 2)) * 64 + lane] == w[m, k]
    assert wp.abs().sum() == pytest.approx(float(w.abs().sum()), rel=1e-6)      # padding is zero
    cw = synth.uniform(1, "cw", (20, 5, 3, 3), -1, 1)
    cp = _hip.pack_conv3x3_weight(cw)
    mt, ks = 2, 2
    assert cp.numel() == 9 * mt * ks * 64
    for (co, ci, ky, kx) in [(0, 0, 0, 0), (19, 4, 2, 1), (7, 3, 1, 2)]:
        tap = ky * 3 + kx
        lane = (co & 15) + 16 * (ci & 3)
        assert cp[((tap * mt + (co >> 4)) * ks + (ci >> 2)) * 64 + lane] == cw[co, ci, ky, kx]
    wt = synth.uniform(1, "dw", (4, 6, 3, 3), -1, 1)
    x = synth.uniform(1, "dx", (1, 4, 7, 9), -1, 1)
    ref = torch.nn.functional.conv_transpose2d(x, wt, padding=1)
    got = torch.nn.functional.conv2d(x, _hip.deconv_as_conv_weight(wt), padding=1)
    assert (ref - got).abs().max() < 1e-5
    assert _hip.choose_ct(9) == 9 and _hip.choose_ct(16) == 8 and _hip.choose_ct(3) == 3 and _hip.choose_ct(12) == 6


def test_metrics():
    a = np.random.default_rng(0).integers(0, 256, (32, 40, 3)).astype(np.uint8)
    b = np.clip(a.astype(int) + np.random.default_rng(1).integers(-3, 4, a.shape), 0, 255).astype(np.uint8)
    p, s = utils.calculate_metrics(b, a)
    assert p == pytest.approx(tiler_ref.psnr(a, b), abs=1e-12) and 0.9 < s <= 1.0
    assert utils.calculate_metrics(a, a)[0] == float("inf")
    assert utils.calculate_metrics(a, a)[1] == pytest.approx(1.0)


def test_harness_aggregate_and_csv(tmp_path):
    from irm_amd import harness
    row = harness.aggregate([30.0, 32.0], [0.9, 0.8], [10.0, 14.0], task="deblurring", subtask="motion", dataset="GoPro",
                            sigma="N/A", model_name="Restormer", params=26126644)
    assert list(row) == harness.COLUMNS
    assert row["PSNR"] == 31.0 and row["Std_PSNR"] == 1.0 and row["Avg_Time_ms"] == 12.0 and row["Std_Time_ms"] == 2.0
    assert row["Task"] == "Deblurring" and row["Type"] == "Motion"
    path = harness.save_results([row], str(tmp_path))
    lines = open(path).read().strip().splitlines()
    assert lines[0] == ",".join(harness.COLUMNS) and lines[1].startswith("Deblurring,Motion,GoPro,N/A,Restormer,26126644,31.0")
    frames = list(harness.synthetic_loader(2, 24, 32))
    assert len(frames) == 2 and frames[0][0].shape == (24, 32, 3) and frames[1][2].endswith(".png")


def test_new_model_families_parameter_layout(manifest):
    from irm_amd import deblurganv2
    m = deblurganv2.FPNMobileNet()
    assert sorted(m.state_dict().keys()) == manifest["fpn_mobilenet_state_keys"]
    assert sum(p.numel() for p in m.parameters()) == 3312707
    with pytest.raises(_hip.HipLibraryError):
        m(torch.zeros(1, 3, 32, 32))
    # hooks restate src/deblurganv2/__init__.py:11-28
    from oracle import deblurgan_ref
    img = np.arange(60, dtype=np.uint8).reshape(4, 5, 3)
    assert np.array_equal(deblurganv2.normalize(img), deblurgan_ref.normalize(img))
    t = torch.zeros(1, 3, 64, 100)
    assert deblurganv2.pad(t).shape == (1, 3, 96, 128) and torch.equal(deblurganv2.pad(t), deblurgan_ref.pad32(t))
    assert torch.equal(deblurganv2.postprocess(t), (t + 1) / 2)


def test_synth_is_deterministic():
    a = synth.uniform(42, "x.weight", (4, 5), -1, 1)
    assert torch.equal(a, synth.uniform(42, "x.weight", (4, 5), -1, 1))
    assert not torch.equal(a, synth.uniform(43, "x.weight", (4, 5), -1, 1))
    i1, t1 = synth.synth_image_pair(0, 48, 64)
    i2, t2 = synth.synth_image_pair(0, 48, 64)
    assert np.array_equal(i1, i2) and np.array_equal(t1, t2) and i1.dtype == np.uint8 and i1.shape == (48, 64, 3)
    assert 15 < tiler_ref.psnr(t1, i1) < 45


# --------------------------------------------------------------------------- C ABI
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "irm_hip.h")).read()
    return sorted(set(re.findall(r"^\s*int\s+(irm_\w+)\s*\(", text, flags=re.M)))


def test_c_abi_library_exports_every_declared_symbol():
    names = _declared_symbols()
    assert len(names) >= 10 and set(names) == set(_hip.SIGNATURES)
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for n in names:
        assert getattr(lib, n) is not None
    assert _hip.load().irm_version() == 1


def test_asan_host_build_rejects_bad_arguments(tmp_path):
    """`make asan` (host side under AddressSanitizer; the GPU side cannot be sanitised on this pool): every C-ABI entry
    point called with null pointers / zero sizes must return IRM_EINVAL before touching HIP, with no ASan report.
    Runs in a child process because the ASan runtime has to be the first library of the process (LD_PRELOAD)."""
    import glob
    import subprocess
    import sys
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("no ASan runtime in this ROCm image")
    csrc = os.path.join(os.path.dirname(_hip.LIB_PATH), "csrc")
    lib = os.path.join(os.path.dirname(_hip.LIB_PATH), "libirm_hip_asan.so")
    subprocess.run(["make", "-C", csrc, "asan", "-j4"], check=True, capture_output=True, timeout=900)
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "from irm_amd import _hip\n"
        "lib = C.CDLL(%r)\n"
        "bad = []\n"
        "for name, sig in _hip.SIGNATURES.items():\n"
        "    if name == 'irm_version': continue\n"
        "    f = getattr(lib, name); f.argtypes = sig; f.restype = C.c_int\n"
        "    rc = f(*[t(0) for t in sig])\n"
        "    if rc != -1: bad.append((name, rc))\n"
        "print('checked', len(_hip.SIGNATURES) - 1, 'bad', bad)\n"
        "sys.exit(1 if bad else 0)\n" % (os.path.dirname(os.path.dirname(_hip.LIB_PATH)), lib))
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr


def test_product_path_fails_loudly_without_gpu_or_library(monkeypatch):
    m = restormer.Restormer()
    with pytest.raises(_hip.HipLibraryError):
        m(torch.zeros(1, 3, 8, 8))
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libirm_hip.so")
    with pytest.raises(_hip.HipLibraryError, match="no CPU fallback"):
        _hip.load()


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` outside a launcher starts N ranks through torch.distributed.run (the driver's
    command shape) instead of exiting; checked without a GPU by intercepting the child process."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("RANK", raising=False)
    args = bench.parse()
    assert bench.self_launch(args) == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_split_range_guard_falls_back_per_layer():
    """The unscaled fp16 hi/lo split of the streaming GEMMs is used only inside its safe range; outside it the layer
    keeps the exact f32 weights."""
    w = torch.randn(64, 48) * 0.1
    assert _hip.split_is_safe(w) and _hip.split_is_safe(w, torch.ones(48), torch.zeros(48))
    assert not _hip.split_is_safe(w * 1e6)                                   # hi overflows
    assert not _hip.split_is_safe(w * 1e-4)                                  # every weight below the fp16 resolution window
    assert not _hip.split_is_safe(w, torch.full((48,), 6000.0), None)        # LN gain * sqrt(K) leaves fp16
    assert not _hip.split_is_safe(w * float("nan"))
    m = restormer.Restormer(dim=16, num_blocks=(1, 1, 1, 1), num_refinement_blocks=1, heads=(1, 1, 1, 1)).load_synthetic(3)
    with torch.no_grad():
        m.latent[0].norm1.body.weight.mul_(1e4)                              # a "trained" outlier in ONE layer
        m.latent[0].ffn.project_out.weight.mul_(1e-5)
    pk = m._pack()
    assert "qkv_s" not in pk["latent.0"] and "pout_s" not in pk["latent.0"] and "pin_s" in pk["latent.0"]
    assert "qkv_s" in pk["encoder_level3.0"] and "pout_s" in pk["encoder_level3.0"]


def test_presplit_host_side_packing_and_plans():
    """Host side of the pre-split GEMM path (gemm_ps.hip): fragment packing round trip (incl. zero K padding), the
    power-of-two scales, and launch plans without empty workgroup groups."""
    w = gin("psw", (570, 192), -0.3, 0.3)
    frag, s_w = _hip.pack_gemm_weight_presplit(w)
    assert np.log2(s_w) == int(np.log2(s_w)) and 2.0 ** 13 <= float(w.abs().max()) * s_w < 2.0 ** 14
    h = frag.view(torch.float16).double().view(-1, 6, 2, 4, 16, 8)                 # [tile][ks][hi|lo][g][m][e]
    back = (h[:, :, 0] + h[:, :, 1]).permute(0, 3, 1, 2, 4).reshape(-1, 192)
    assert back.shape[0] == 576 and float(back[570:].abs().max()) == 0.0         # rows beyond M are zero
    assert float((back[:570] / s_w - w.double()).abs().max()) <= 2.0 ** -21 * float(w.abs().max())
    w2 = gin("psw2", (192, 510), -0.2, 0.2)
    frag2, s2 = _hip.pack_gemm_weight_presplit(w2, k_pad=512)
    h2 = frag2.view(torch.float16).double().view(-1, 16, 2, 4, 16, 8)
    back2 = (h2[:, :, 0] + h2[:, :, 1]).permute(0, 3, 1, 2, 4).reshape(-1, 512)
    assert float(back2[:, 510:].abs().max()) == 0.0
    assert float((back2[:192, :510] / s2 - w2.double()).abs().max()) <= 2.0 ** -21 * float(w2.abs().max())
    # operand scale of the LayerNorm output: the static bound stays inside fp16 for any gain
    for gain, bias in ((1.0, 0.1), (30.0, 2.0), (1e-3, 0.0), (1e4, 50.0)):
        lnw, lnb = torch.full((192,), gain), torch.full((192,), bias)
        s = _hip.ln_split_scale(lnw, lnb, 192, True)
        assert np.log2(s) == int(np.log2(s)) and (gain * 191 ** 0.5 + bias) * s < 2.0 ** 15
        assert _hip.ln_split_scale(lnw, None, 192, False) == _hip.ln_split_scale(lnw, None, 192, True) / 16.0
    for mtiles in (1, 9, 36, 64, 72, 128):
        for npt in (1, 16, 576, 1536, 2304, 6144):
            for K in (192, 384):
                ct, mg, shape = _hip.plan_presplit(mtiles, npt, K)
                chunks = -(-mtiles // ct)
                assert ct in (4, 6, 8) and 1 <= mg <= chunks and (mg - 1) * -(-chunks // mg) < chunks
                assert shape == (43 if K == 192 else 81)
    assert _hip.plan_presplit(64, 6144, 192) == (4, 1, 43) and _hip.plan_presplit(128, 1536, 384) == (8, 4, 81)


def test_apply_fusion_host_side_packing():
    """Host side of irm_attn_gdfn_fused_f16x3_f32: the folded-matrix fragments round trip (hi + lo = the fp16x2 value,
    zero K / M padding), and pack_gdfn_fused(kperm=True) holds the same project_in weights with its input channels in
    the order the apply MFMA leaves x' in the registers (k-slot 32 ks + 8 g + j <-> channel 16 (2 ks + (j >> 2)) + 4 g + (j & 3))."""
    for B, C in ((2, 96), (3, 48), (1, 64)):
        m = gin(f"mf{C}", (B, C, C), -0.4, 0.4)
        frag = _hip.pack_mfold_frag(m)
        KS = (C + 31) // 32
        assert frag.numel() == B * 2 * KS * KS * 512
        hi = m.half().float()
        assert torch.equal(_hip.unpack_mfold_frag(frag, B, C), hi + (m - hi).half().float())
        h = frag.view(torch.float16).view(B, 2 * KS, KS, 2, 4, 16, 8).float()          # [t][ks][hi|lo][g][m][e]
        full = (h[:, :, :, 0] + h[:, :, :, 1]).permute(0, 1, 4, 2, 3, 5).reshape(B, 32 * KS, 32 * KS)
        assert float(full[:, C:].abs().max() if C < 32 * KS else 0.0) == 0.0 and float(full[:, :, C:].abs().max() if C < 32 * KS else 0.0) == 0.0
    C, hid = 96, 255
    args = (gin("kp1", (2 * hid, C), -0.3, 0.3), None, gin("kp2", (2 * hid, 9), -0.4, 0.4), None, gin("kp3", (C, hid), -0.3, 0.3),
            gin("kp4", (C,), 0.5, 1.5), gin("kp5", (C,), -0.2, 0.2))
    rec0, w20, i10, i20 = _hip.pack_gdfn_fused(*args)
    rec1, w21, i11, i21 = _hip.pack_gdfn_fused(*args, kperm=True)
    assert torch.equal(w20, w21) and (i10, i20) == (i11, i21)
    S, KS = (hid + 15) // 16, 3
    r0 = rec0.view(S + 1, KS * 1024 + 512)
    r1 = rec1.view(S + 1, KS * 1024 + 512)
    assert torch.equal(r0[:, KS * 1024:], r1[:, KS * 1024:])                              # taps, biases: unchanged
    # [S][hct][KS][hi|lo][g][m][8 j] halves -> [.., channel slot 32 ks + 8 g + j]
    def slots(r):
        h = r[:S, :KS * 1024].contiguous().view(torch.float16).view(S, 2, KS, 2, 4, 16, 8)
        return h.permute(0, 1, 3, 5, 2, 4, 6).reshape(S, 2, 2, 16, 32 * KS)
    slot = torch.arange(32 * KS)
    ks_, g_, j_ = slot // 32, (slot % 32) // 8, slot % 8
    chan = 16 * (2 * ks_ + (j_ >> 2)) + 4 * g_ + (j_ & 3)
    assert sorted(chan.tolist()) == list(range(32 * KS))
    assert torch.equal(slots(r1), slots(r0)[..., chan])


def test_qkv_gram_plan_matches_the_record_layout_of_the_header():
    """Host side of irm_qkv_gram_cm_f16x3_f32: C = 48 with one head and whole chunks of 4 tiles only; the records per image it
    announces to mdta_fold are H W / 1024 (include/irm_hip.h), each of the Gram pass's own size."""
    from irm_amd import ops
    assert ops.QKV_GRAM_NCH * 256 == 1024
    assert ops.can_qkv_gram(48, 1, 512, 512) and ops.can_qkv_gram(48, 1, 32, 32)
    assert not ops.can_qkv_gram(96, 1, 512, 512) and not ops.can_qkv_gram(48, 2, 512, 512)
    assert not ops.can_qkv_gram(48, 1, 40, 96)          # 15 tiles: no whole chunks
    assert not ops.can_qkv_gram(48, 1, 36, 64) and not ops.can_qkv_gram(48, 1, 32, 48)
    _, _, rec = ops.mdta_plan(2, 48, 1, 512 * 512)
    assert rec == 48 * 48 + 2 * 48
    sig = _hip.SIGNATURES["irm_qkv_gram_cm_f16x3_f32"]
    assert len(sig) == 17                               # the header's 16 arguments + the stream


def test_fullsize_frame_fixture_is_self_consistent():
    """tests/golden/restormer_fullsize_frame.npz (the reference's run_model_inference on bench frame 0): the stored
    sha256 and PSNR are those of the stored uint8 frame against the regenerated synthetic target, and the input it
    was produced from is the frame bench.py feeds to rank 0 (PSNR of the raw input recorded beside it)."""
    import hashlib
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "restormer_fullsize_frame.npz"))
    pred = g["pred_u8"]
    assert pred.shape == (720, 1280, 3) and pred.dtype == np.uint8
    assert hashlib.sha256(np.ascontiguousarray(pred).tobytes()).digest() == g["sha256"].tobytes()
    inp, tgt = synth.synth_image_pair(0, 720, 1280, 3, seed_base=1000, blur=15)
    assert abs(tiler_ref.psnr(tgt, pred) - float(g["psnr"])) < 1e-9
    assert abs(tiler_ref.psnr(tgt, inp) - float(g["psnr_input"])) < 1e-9
    assert g["refine_sub16"].shape == (96, 32, 32) and g["refine_row100"].shape == (96, 512)
