"""GPU parity tests of the MaIR path: the selective-scan kernel against the CPU oracle restatement of the
recurrence (the reference's own implementation is the third-party mamba_ssm CUDA wheel, absent: parity of
that op is UNPINNED - see oracle/mair_ref.py), the LoSh2D glue kernels, and MaIRUNet / VSSBlock against
golden outputs of the imported reference module (everything but the scan op pinned by reference code)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from irm_amd import _hip, mair, ops, synth
from irm_amd.mair import mairunet_arch as arch
from oracle import mair_ref

pytestmark = pytest.mark.gpu

NET_G = dict(inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4, ssm_ratio=2.0,
             flp_ratio=4.0, mlp_ratio=1.5, bias=False, dual_pixel_task=False, img_size=128, scan_len=4, batch_size=8,
             dynamic_ids=False)


def rnd(name, shape, lo=-1.0, hi=1.0):
    return synth.uniform(321, name, shape, lo, hi)


def gin(name, shape, lo=0.0, hi=1.0):
    return synth.uniform(7, name, shape, lo, hi)


@pytest.mark.parametrize("B,R,C", [(2, 37, 50), (1, 96, 1024), (3, 5, 7)])
def test_transpose(dev, B, R, C):
    x = rnd("tr", (B, R + 2, C))
    xg = x.to(dev)
    out = torch.empty(B, C, R, device=dev)
    ops.transpose(xg[:, 1:1 + R], out, R, C)
    assert torch.equal(out.cpu(), x[:, 1:1 + R].transpose(1, 2).contiguous())


@pytest.mark.parametrize("B,D,N,R,H,W,chunk", [(1, 96, 4, 3, 16, 24, 64), (2, 192, 8, 6, 8, 16, 32), (1, 384, 16, 12, 8, 8, 32),
                                               (1, 768, 32, 24, 4, 8, 32), (1, 96, 4, 3, 5, 7, 32), (1, 192, 8, 6, 32, 32, 1024)])
def test_selective_scan_vs_oracle(dev, B, D, N, R, H, W, chunk):
    L, J = H * W, R + 2 * N
    ids, inv = mair_ref.scan_ids(H, W, 4)
    x = rnd(f"su{D}", (B, D, L))                               # planar u
    proj = rnd(f"sp{D}", (B, 4, J, L))                         # per direction [dt_raw | B | C] in PIXEL order
    dtw = rnd(f"sw{D}", (4, D, R), -0.5, 0.5)
    dtb = rnd(f"sb{D}", (4, D), -4, -2)
    A = -torch.exp(rnd(f"sa{D}", (4 * D, N), 0, 1.5))
    Ds = rnd(f"sd{D}", (4 * D,), 0.5, 1.5)
    # oracle: gather, project dt, scan, inverse gather
    xs = torch.stack([x.index_select(-1, ids[k]) for k in range(4)], 1)                     # (B,4,D,L)
    pg = torch.stack([proj[:, k].index_select(-1, ids[k]) for k in range(4)], 1)            # (B,4,J,L)
    dts = torch.einsum("bkrl,kdr->bkdl", pg[:, :, :R], dtw)
    y = mair_ref.selective_scan(xs.reshape(B, -1, L), dts.reshape(B, -1, L), A, pg[:, :, R:R + N], pg[:, :, R + N:],
                                Ds, delta_bias=dtb.reshape(-1), delta_softplus=True).view(B, 4, D, L)
    y_img = torch.stack([y[:, k].index_select(-1, inv[k]) for k in range(4)], 1)            # (B,4,D,L) pixel order
    # device
    xT = x.transpose(1, 2).contiguous().to(dev)                                             # (B,L,D)
    pT = proj.reshape(B, 4 * J, L).transpose(1, 2).contiguous().to(dev)                     # (B,L,4J)
    nchunk, DB = -(-L // chunk), -(-D // 64)
    yT = torch.full((B, 4, L, D), float("nan"), device=dev)
    state = torch.empty(2 * B * 4 * DB * nchunk * N * 64, device=dev)
    sdt = torch.empty(B * 4 * DB * nchunk * 64, device=dev)
    ysum = torch.empty(B * 4 * DB * nchunk * 64, device=dev)
    ops.selective_scan(xT, pT, ids.int().to(dev), dtw.to(dev), dtb.to(dev), A.to(dev), Ds.to(dev), yT, state, sdt, ysum,
                       B, L, D, N, R, chunk)
    got = yT.cpu().permute(0, 1, 3, 2)                                                       # (B,4,D,L)
    err = float((got - y_img).abs().max())
    print(f"scan D{D} N{N} L{L} chunk{chunk}: max-abs vs oracle {err:.3e} (|y| max {float(y_img.abs().max()):.2f})")
    assert err <= 2e-4 * max(1.0, float(y_img.abs().max()))
    # per-chunk sums of y feed the ShuffleAttn mean
    s = ysum.cpu().view(B, 4, DB, nchunk, 64).sum(3).reshape(B, 4, DB * 64)[:, :, :D]
    assert (s / L - y_img.mean(-1)).abs().max() <= 1e-4 * max(1.0, float(y_img.abs().max()))


def test_losh_combine(dev):
    B, D, H, W = 2, 96, 8, 12
    L = H * W
    y = rnd("cy", (B, 4, D, L))
    z = rnd("cz", (B, D + 3, H, W))[:, 1:1 + D]
    gw, gb = rnd("cgw", (4 * D, 4), -2, 2), rnd("cgb", (4 * D,))
    nw, nb = rnd("cnw", (D,), 0.5, 1.5), rnd("cnb", (D,), -0.2, 0.2)
    m = y.double().mean(-1)                                                                  # (B,4,D)
    g = torch.sigmoid(torch.einsum("dqk,bkd->bqd", gw.double().view(D, 4, 4), m) + gb.double().view(D, 4).t())
    v = (y.double() * g.unsqueeze(-1)).sum(1)                                                # (B,D,L)
    ref = F.layer_norm(v.transpose(1, 2), (D,), nw.double(), nb.double(), 1e-5).transpose(1, 2)
    ref = ref * F.silu(z.double().reshape(B, D, L))
    nchunk = 3
    ysum = torch.zeros(B, 4, 2, nchunk, 64)
    tot = y.sum(-1)                                                                          # split the sums over chunks
    for c in range(nchunk):
        part = tot * (0.5 if c == 0 else 0.25)
        ysum[:, :, 0, c, :] = part[:, :, :64]
        ysum[:, :, 1, c, :32] = part[:, :, 64:]
    gate = torch.empty(B, 4, D, device=dev)
    ysum = ysum.to(dev)
    zg = rnd("cz", (B, D + 3, H, W)).to(dev)[:, 1:1 + D]
    out = torch.empty(B, D, H, W, device=dev)
    ops.losh_combine(ysum, gw.to(dev), gb.to(dev), gate, y.permute(0, 1, 3, 2).contiguous().to(dev), nw.to(dev),
                     nb.to(dev), zg, out, B, L, D, nchunk)
    assert (gate.cpu().double() - g).abs().max() < 1e-5
    assert (out.cpu().double().reshape(B, D, L) - ref).abs().max() < 2e-4


@pytest.mark.parametrize("c,n,ratio,h,w", [(48, 4, 4.0, 16, 24), (96, 8, 1.5, 8, 16), (384, 32, 1.5, 8, 8)])
def test_vss_block_vs_golden(dev, golden, c, n, ratio, h, w):
    blk = arch.VSSBlock(c, n, 2.0, ratio)
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    blk.load_state_dict(synth.synth_state_dict(shapes, seed=21, rules=mair.SYNTH_RULES))
    # reuse the real packer / block driver through a MaIRUNet whose first stage is this block
    wrap = mair.MaIRUNet(**{**NET_G, "num_blocks": [1, 0, 0, 0], "num_refinement_blocks": 0})
    wrap.encoder_level1 = torch.nn.ModuleList([blk])
    wrap = wrap.to(dev)
    pk = wrap._pack()["encoder_level1.0"]
    x_tok = gin(f"vss_in_{c}", (2, h * w, c), -1.0, 1.0)
    x = x_tok.transpose(1, 2).reshape(2, c, h, w).contiguous().to(dev)
    wrap._block(blk, pk, x, arch.scan_ids(h, w, 4, dev))
    got = x.cpu().reshape(2, c, h * w).transpose(1, 2).numpy()
    err = np.abs(got - golden("mair")[f"vss_c{c}_{h}x{w}"]).max()
    print(f"vss block c{c}: max-abs vs reference golden {err:.3e}")
    assert err <= 5e-4


@pytest.mark.parametrize("h,w", [(32, 32), (24, 40)])
def test_mairunet_vs_golden(dev, golden, h, w):
    model = mair.MaIRUNet(**NET_G).load_synthetic(42).eval().to(dev)
    x = gin(f"mair_in_{h}x{w}", (1, 3, h, w))
    y = model(x.to(dev)).cpu().numpy()
    err = np.abs(y - golden("mair")[f"mairunet_{h}x{w}"]).max()
    print(f"mairunet {h}x{w}: max-abs vs reference golden {err:.3e}")
    assert err <= 1e-3


def test_mairunet_batch_and_determinism(dev):
    model = mair.MaIRUNet(**NET_G).load_synthetic(42).eval().to(dev)
    x = gin("mair_batch", (2, 3, 64, 64)).to(dev)
    y1, y2 = model(x).clone(), model(x).clone()
    assert torch.equal(y1, y2)
    assert (model(x[1:2]) - y1[1:2]).abs().max() <= 5e-5


FLAT_CFG = dict(upscale=1, in_chans=3, img_range=1., d_state=16, depths=[2, 2], embed_dim=180, ssm_ratio=1.3, mlp_ratio=2.0,
                upsampler=None, resi_connection='1conv', img_size=16, dynamic_ids=False, batch_size=1, scan_len=4)


@pytest.mark.parametrize("h,w", [(16, 16), (24, 20)])
def test_mair_flat_vs_golden(dev, golden, h, w):
    """Flat MaIR (shifted scan tables on odd blocks, C=180 / D=234 not multiples of 16 or 64) vs reference goldens."""
    model = mair.MaIR(**FLAT_CFG).load_synthetic(42).eval().to(dev)
    x = gin(f"mairflat_in_{h}x{w}", (1, 3, h, w))
    y = model(x.to(dev)).cpu().numpy()
    err = np.abs(y - golden("mair")[f"mairflat_{h}x{w}"]).max()
    print(f"mair flat {h}x{w}: max-abs vs reference golden {err:.3e}")
    assert err <= 1e-3


def test_shifted_scan_tables_vs_golden(golden):
    for key in golden("mair").files:
        if key.startswith("shift_ids_"):
            h, w = map(int, key.split("_")[2].split("x"))
            sl = int(key.split("_s")[-1])
            assert np.array_equal(arch.scan_ids(h, w, sl, "cpu", sl // 2).numpy(), golden("mair")[key])
