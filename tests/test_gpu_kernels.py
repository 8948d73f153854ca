"""Per-kernel parity on a real MI355X: every C-ABI entry point (include/smx.h) against the
ATen CPU op / oracle function it replaces, on seeded inputs.  fp32 tolerances are written
next to each check; integer outputs (VQ indices, masks, uint8) are compared exactly."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import reenact_oracle as O
from synergize_motion_appearance_amd.synth import synth_input
from tests.util import maxabs, weights, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "needs an MI355X"
    from synergize_motion_appearance_amd import ops as _ops
    from synergize_motion_appearance_amd import lib
    lib.load()
    return _ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rnd(name, shape, scale=1.0):
    return synth_input(name, shape) * scale


# ---------------------------------------------------------------------------------------
# implicit-GEMM convolution
# ---------------------------------------------------------------------------------------
CONV_CASES = [
    # (B, Cin, Cout, H, k, stride, pad(t,l) or None, out_hw, up2, act, tile, tag)
    (2, 64, 64, 32, 3, 1, None, None, False, 0, 0, "3x3 64->64"),
    (1, 128, 128, 64, 3, 1, None, None, False, 0, 1, "3x3 128->128 tile1"),
    (1, 128, 128, 64, 3, 1, None, None, False, 0, 4, "3x3 128->128 tile4"),
    (2, 256, 128, 16, 3, 1, None, None, False, 1, 0, "3x3 256->128 relu"),
    (1, 64, 64, 32, 3, 1, None, None, False, 0, 2, "tile2"),
    (1, 64, 64, 32, 3, 1, None, None, False, 0, 5, "tile5"),
    (2, 32, 32, 32, 3, 1, None, None, False, 3, 3, "3x3 32->32 swish tile3"),
    (2, 64, 64, 33, 3, 2, (0, 0), (16, 16), False, 0, 0, "downsample-like odd"),
    (2, 32, 32, 64, 3, 2, (0, 0), (32, 32), False, 0, 0, "Downsample pad(0,1,0,1) s2"),
    (2, 64, 64, 16, 3, 1, None, None, True, 0, 0, "Upsample nearest x2 folded"),
    (2, 3, 64, 32, 3, 1, None, None, False, 0, 0, "3x3 3->64 generic path"),
    (2, 2, 32, 32, 3, 1, None, None, False, 0, 0, "3x3 2->32 generic"),
    (2, 2, 128, 24, 7, 1, None, None, False, 1, 0, "7x7 2->128 pad3 relu"),
    (2, 35, 15, 32, 7, 1, (0, 0), None, False, 0, 0, "7x7 valid 35->15"),
    (2, 36, 76, 32, 7, 1, (0, 0), None, False, 0, 0, "7x7 valid 36->76 float4 gather, slices straddle taps"),
    (2, 12, 40, 24, 3, 1, None, None, False, 1, 0, "3x3 12->40 K=108 float4 gather, K%32!=0"),
    (2, 20, 24, 16, 1, 1, None, None, False, 0, 0, "1x1 20->24 K<32 float4"),
    (2, 64, 128, 32, 3, 1, None, None, False, 0, 8, "tile8 128x128 8 waves"),
    (2, 64, 128, 32, 3, 1, None, None, False, 1, 9, "tile9 128x128 16 waves relu"),
    (2, 64, 256, 32, 1, 1, None, None, False, 0, 12, "tile12 256x128 16 waves (the B=60 1x1 GEMMs)"),
    (3, 128, 128, 24, 1, 1, None, None, False, 3, 12, "tile12 ragged M swish"),
    (1, 64, 192, 32, 1, 1, None, None, False, 1, 6, "tile6 256x128 8 waves"),
    (1, 64, 64, 32, 3, 1, None, None, False, 0, 7, "tile7"),
    (1, 64, 64, 32, 3, 1, None, None, False, 0, 10, "tile10"),
    (1, 64, 128, 32, 3, 1, None, None, False, 0, 11, "tile11"),
    (1, 64, 64, 32, 3, 1, None, None, False, 0, 13, "tile13"),
    (1, 64, 32, 32, 3, 1, None, None, False, 0, 15, "tile15"),
    (1, 40, 64, 20, 3, 2, (0, 0), (10, 10), False, 0, 5, "3x3 s2 40->64 float4 gather tile5"),
    (1, 128, 16, 32, 7, 1, None, None, False, 0, 0, "7x7 128->16"),
    (1, 128, 1, 32, 7, 1, None, None, False, 5, 0, "7x7 128->1 sigmoid"),
    (2, 256, 192, 32, 1, 1, None, None, False, 1, 0, "1x1 256->192 relu"),
    (2, 15, 32, 32, 1, 1, None, None, False, 1, 0, "1x1 15->32 generic"),
    (1, 160, 126, 32, 3, 1, None, None, False, 1, 0, "3x3 160->126 odd Cout"),
    (1, 64, 3, 64, 3, 1, None, None, False, 0, 0, "3x3 64->3"),
    (1, 256, 512, 32, 3, 1, None, None, False, 4, 0, "3x3 256->512 gelu"),
    (1, 1024, 512, 4, 3, 1, None, None, True, 1, 0, "hourglass deep up2"),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[-1] for c in CONV_CASES])
def test_conv(ops, case):
    B, Cin, Cout, H, k, stride, pad, out_hw, up2, act, tile, tag = case
    x = rnd("cx" + tag, (B, Cin, H, H))
    w = rnd("cw" + tag, (Cout, Cin, k, k), 1.0 / math.sqrt(Cin * k * k))
    b = rnd("cb" + tag, (Cout,), 0.1)
    xe = F.interpolate(x, scale_factor=2.0, mode="nearest") if up2 else x
    if pad is None:
        ref = F.conv2d(xe, w, b, stride=stride, padding=k // 2)
    else:
        He = xe.shape[2]
        if out_hw is not None:       # asymmetric: pad bottom/right so that exactly out_hw rows come out
            need = (out_hw[0] - 1) * stride + k - He - pad[0]
            xe = F.pad(xe, (pad[1], max(need, 0), pad[0], max(need, 0)))
            ref = F.conv2d(xe, w, b, stride=stride)[:, :, :out_hw[0], :out_hw[1]]
        else:
            ref = F.conv2d(xe, w, b, stride=stride, padding=pad)
    ref = {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, 0.2), 3: O.swish, 4: F.gelu, 5: torch.sigmoid}[act](ref)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    y = ops.conv(nhwc(x), cv, stride=stride, pad=pad, out_hw=out_hw, up2=up2, act=act, tile=tile)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (B, ref.shape[2], ref.shape[3], Cout)
    assert maxabs(nchw(y), ref) < 2e-5, tag


LOADER_CASES = [  # (B, Cin, Cout, H, W, k, stride, pad, tile, res, tag)
    (2, 64, 128, 24, 40, 1, 1, (0, 0), 0, False, "1x1 ragged M"), (3, 256, 192, 17, 19, 1, 1, (0, 0), 12, True, "1x1 tile12 residual, N tail"),
    (2, 32, 48, 33, 21, 3, 2, (1, 1), 5, False, "3x3 s2 pad1 odd grid"), (1, 128, 17, 30, 26, 7, 1, (3, 3), 3, False, "7x7 128->17 small N"),
    (2, 96, 64, 9, 33, 5, 1, (2, 2), 8, True, "5x5 tile8 residual"), (5, 64, 64, 4, 4, 3, 1, (1, 1), 1, False, "4x4 maps: a tile spans several images"),
    (1, 128, 128, 32, 32, 3, 1, (1, 1), 6, False, "tile6"), (1, 64, 64, 16, 48, 3, 1, (0, 2), 10, False, "asymmetric pad tile10"),
]


@pytest.mark.parametrize("case", LOADER_CASES, ids=[c[-1] for c in LOADER_CASES])
def test_gemm_conv_buffer_loader_equals_the_gather(ops, case, tuning):
    """gemm_conv's LMODE 2 loader (round 5: one buffer_load per operand quad, SGPR tile base + slice offset, padding / tails as out-of-range offsets;
    knob `gemm_loader`) against the float4 gather it replaces: the same products in the same order -> bit-identical, and both equal F.conv2d."""
    B, Cin, Cout, H, W, k, stride, pad, tile, res, tag = case
    x = rnd("lx" + tag, (B, Cin, H, W))
    w = rnd("lw" + tag, (Cout, Cin, k, k), 1.0 / math.sqrt(Cin * k * k))
    b = rnd("lb" + tag, (Cout,), 0.1)
    ref = F.leaky_relu(F.conv2d(x, w, b, stride=stride, padding=pad), 0.2)
    r = rnd("lr" + tag, tuple(ref.shape)) if res else None
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    ys = []
    for knob in (1, 0):
        tuning("gemm_loader", knob)
        ys.append(ops.conv(nhwc(x), cv, stride=stride, pad=pad, act=2, tile=tile, res=nhwc(r) if res else None, direct=True).clone())
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ys[1]), tag
    assert maxabs(nchw(ys[0]), ref + r if res else ref) < 2e-5, tag


WINO_CASES = [(2, 64, 64, 32, False, 0, False), (1, 128, 128, 64, False, 3, True), (2, 256, 512, 32, False, 4, False),
              (1, 128, 64, 64, False, 0, True), (2, 64, 64, 16, True, 0, False), (1, 160, 126, 32, False, 1, False),
              (1, 128, 96, 32, False, 1, False), (3, 32, 32, 32, False, 2, True), (1, 64, 64, 256, False, 0, True)]


@pytest.fixture
def tuning(ops):
    """set libsmx launch-selection knobs for one test, restore afterwards."""
    saved = []

    def _set(name, value):
        saved.append((name, ops.set_tuning(name, value)))
    yield _set
    for name, old in reversed(saved):
        ops.set_tuning(name, old)


@pytest.mark.parametrize("nw,wide", [(-1, 1), (1, 1), (2, 0), (2, 1)], ids=["auto", "nw1", "nw2_8wave", "nw2_wide"])
@pytest.mark.parametrize("case", WINO_CASES, ids=[f"B{c[0]}_{c[1]}to{c[2]}_H{c[3]}_up{int(c[4])}_act{c[5]}_res{int(c[6])}" for c in WINO_CASES])
def test_winograd_conv3x3(ops, case, nw, wide, tuning):
    """fused Winograd F(2x2,3x3) == F.conv2d (3x3, s1, p1); also through channel-slice operands.  Every block
    shape the launcher can select (4-wave N=32 blocks, 8-wave N=64 blocks, 4-wave "wide" N=64 blocks -- the last is
    what the B=60 bench runs) is forced explicitly, not left to the size thresholds."""
    B, Cin, Cout, H, up2, act, with_res = case
    tuning("wino_nw", nw)
    tuning("wino_wide", wide)
    x = rnd(f"wx{case}", (B, Cin, H, H))
    w = rnd(f"ww{case}", (Cout, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd(f"wb{case}", (Cout,), 0.1)
    xe = F.interpolate(x, scale_factor=2.0, mode="nearest") if up2 else x
    ref = F.conv2d(xe, w, b, padding=1)
    ref = {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, 0.2), 3: O.swish, 4: F.gelu}[act](ref)
    r = rnd(f"wr{case}", tuple(ref.shape)) if with_res else None
    if with_res:
        ref = ref + r
    assert ops.WINOGRAD
    xin = torch.zeros((B, H, H, Cin + 32), device="cuda")
    xin[..., 32:] = nhwc(x)
    out = torch.full((B, ref.shape[2], ref.shape[3], Cout + 8), 5.0, device="cuda")
    ops.conv(xin[..., 32:], ops.Conv.from_torch(w.cuda(), b.cuda()), out=out[..., 4:4 + Cout], up2=up2, act=act,
             res=None if r is None else nhwc(r))
    assert maxabs(nchw(out[..., 4:4 + Cout]), ref) < 5e-5
    assert float(out[..., :4].min()) == 5.0 and float(out[..., 4 + Cout:].max()) == 5.0
    direct = ops.conv(xin[..., 32:], ops.Conv.from_torch(w.cuda(), b.cuda()), up2=up2, act=act, res=None if r is None else nhwc(r), tile=5)
    assert maxabs(nchw(direct), ref) < 3e-5


def test_winograd_fused_groupnorm_loader(ops):
    """GN(32, eps 1e-6) + swish folded into the Winograd region loader == group_norm -> swish -> conv2d."""
    for (B, C, Co, H, sw) in ((2, 64, 64, 32, True), (1, 128, 64, 64, True), (2, 256, 128, 16, False), (1, 32, 32, 32, True)):
        x = rnd(f"gx{C}{H}", (B, C, H, H)) * 1.5 + 0.2
        g, bt = 1 + 0.1 * rnd(f"gg{C}", (C,)), 0.1 * rnd(f"gb{C}", (C,))
        w = rnd(f"gw{C}{Co}", (Co, C, 3, 3), 1.0 / math.sqrt(9 * C))
        b = rnd(f"gbb{Co}", (Co,), 0.1)
        hn = F.group_norm(x, 32, g, bt, 1e-6)
        ref = F.conv2d(O.swish(hn) if sw else hn, w, b, padding=1)
        xin = nhwc(x)
        ss = ops.groupnorm_stats(xin, g.cuda(), bt.cuda())
        y = ops.conv(xin, ops.Conv.from_torch(w.cuda(), b.cuda()), in_ss=ss, in_swish=sw)
        assert maxabs(nchw(y), ref) < 5e-5
        y2 = ops.conv(xin, ops.Conv.from_torch(w.cuda(), b.cuda()), in_ss=ss, in_swish=sw, tile=5)   # non-fused fallback path
        assert maxabs(nchw(y2), ref) < 5e-5


@pytest.mark.parametrize("B,C,Co,H,W,act,res", [(2, 64, 64, 32, 32, 0, True), (3, 128, 128, 16, 32, 3, False), (1, 32, 96, 64, 64, 0, True),
                                                 (2, 64, 126, 8, 16, 1, False)])
@pytest.mark.parametrize("epi", [0, 1, 11], ids=["n32_blocks", "n64_8wave_blocks", "wide_blocks"])
def test_winograd_epilogue_emits_groupnorm_partials(ops, B, C, Co, H, W, act, res, epi, monkeypatch, tuning):
    """want_stats: the conv's epilogue emits per-block {mean, M2} of what it stores (after bias, activation,
    residual); groupnorm_stats on the tagged output is then a finalize only and must equal the two-pass
    statistics of the same tensor, and GroupNorm through it must match F.group_norm."""
    tuning("wino_nw", 1 if epi == 0 else 2)
    tuning("wino_wide", 1 if epi > 10 else 0)
    x = rnd(f"ws{C}{Co}{H}", (B, C, H, W))
    w = rnd(f"wsw{C}{Co}", (Co, C, 3, 3), 1.0 / math.sqrt(9 * C))
    b = rnd(f"wsb{Co}", (Co,), 0.1)
    r = rnd(f"wsr{Co}{H}", (B, Co, H, W)) if res else None
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    y = ops.conv(nhwc(x), cv, act=act, res=None if r is None else nhwc(r), want_stats=True)
    part = y._gn_part
    assert part is not None and tuple(part.shape) == (B, (H // 8) * (W // 16), Co, 2)
    yc = y.cpu().double()
    blocks = yc.view(B, H // 8, 8, W // 16, 16, Co).permute(0, 1, 3, 5, 2, 4).reshape(B, -1, Co, 128)
    bm = blocks.mean(-1)                                       # Welford form: {mean, M2 = sum (v - mean)^2} per block and channel
    assert maxabs(part[..., 0].cpu(), bm) < 2e-6 and maxabs(part[..., 1].cpu(), ((blocks - bm[..., None]) ** 2).sum(-1)) < 2e-4
    if Co & (Co - 1) == 0:                                    # the two-pass kernel takes power-of-two C
        g, bt = rnd(f"wsg{Co}", (Co,)) * 0.2 + 1.0, rnd(f"wsbt{Co}", (Co,), 0.1)
        ss_fused = ops.groupnorm_stats(y, g.cuda(), bt.cuda())
        monkeypatch.setattr(ops, "EPILOGUE_STATS", False)
        ss_two_pass = ops.groupnorm_stats(y, g.cuda(), bt.cuda())
        monkeypatch.setattr(ops, "EPILOGUE_STATS", True)
        assert maxabs(ss_fused.cpu(), ss_two_pass.cpu()) < 2e-5
        ref = F.group_norm(nchw(y), 32, g, bt, 1e-6)
        assert maxabs(nchw(ops.groupnorm_apply(y, ss_fused, swish=False)), ref) < 5e-5
    # a reused output buffer loses the tag when a later conv without statistics overwrites it
    ops.conv(nhwc(x), cv, out=y, act=act)
    assert y._gn_part is None


@pytest.mark.parametrize("B,Cin,Co,H,W,act,gn", [(2, 64, 3, 32, 32, 0, 0), (1, 256, 3, 16, 16, 0, 0), (2, 128, 1, 16, 24, 5, 0),
                                                 (2, 64, 3, 32, 48, 0, 1), (1, 64, 4, 8, 20, 1, 2), (1, 256, 2, 8, 12, 0, 0)])
def test_conv3x3_small_n_vector_alu_kernel(ops, B, Cin, Co, H, W, act, gn, monkeypatch):
    """C_out <= 4 layers (decoder image head, RefineFlow outputs) run on the VALU kernel: vs F.conv2d, with
    the fused GroupNorm (gn=1) / GroupNorm+swish (gn=2) loader, channel-slice input / output views, and
    against the matrix-core path on the same operands."""
    x = rnd(f"sn{Cin}{H}{W}", (B, Cin, H, W))
    w = rnd(f"snw{Cin}{Co}", (Co, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd(f"snb{Co}", (Co,), 0.1)
    xin = x
    ss = None
    if gn:
        g, bt = rnd(f"sng{Cin}", (Cin,)) * 0.2 + 1.0, rnd(f"snbt{Cin}", (Cin,), 0.1)
        xin = F.group_norm(x, 32, g, bt, 1e-6)
        if gn == 2:
            xin = O.swish(xin)
        ss = ops.groupnorm_stats(nhwc(x), g.cuda(), bt.cuda())
    ref = {0: lambda t: t, 1: F.relu, 5: torch.sigmoid}[act](F.conv2d(xin, w, b, padding=1))
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    wide = torch.zeros((B, H, W, Cin + 8), device="cuda")
    wide[..., 4:4 + Cin] = nhwc(x)
    outw = torch.full((B, H, W, Co + 5), 7.0, device="cuda")
    with ops.profile() as rec:
        y = ops.conv(wide[..., 4:4 + Cin], cv, out=outw[..., 2:2 + Co], act=act, in_ss=ss, in_swish=gn == 2)
    assert [r[0] for r in rec.rows] == ["conv_small_n"]
    assert maxabs(nchw(y.contiguous()), ref) < 2e-5
    assert float(outw[..., :2].min()) == 7.0 and float(outw[..., 2 + Co:].min()) == 7.0      # neighbours untouched
    monkeypatch.setattr(ops, "SMALLN", False)
    y2 = ops.conv(nhwc(x), cv, act=act, in_ss=ss, in_swish=gn == 2)
    assert maxabs(y2.cpu(), y.cpu()) < 2e-5


def test_conv_residual_and_slices(ops):
    """output into a channel slice of a concat buffer, input from a slice, fused residual."""
    x = rnd("sx", (2, 96, 32, 32))
    w = rnd("sw", (64, 64, 3, 3), 1 / 24.0)
    b = rnd("sb", (64,), 0.1)
    r = rnd("sr", (2, 64, 32, 32))
    ref = F.conv2d(x[:, 32:], w, b, padding=1) + r
    buf = torch.full((2, 32, 32, 160), 7.0, device="cuda")
    xin = nhwc(x)
    ops.conv(xin[..., 32:], ops.Conv.from_torch(w.cuda(), b.cuda()), out=buf[..., 64:128], res=nhwc(r))
    torch.cuda.synchronize()
    assert maxabs(nchw(buf[..., 64:128]), ref) < 2e-5
    assert float(buf[..., :64].min()) == 7.0 and float(buf[..., 128:].max()) == 7.0


def test_patch_embed_and_unpatchify(ops):
    """Rearrange+Linear == pxp stride-p conv; Linear+Rearrange == 1x1 conv + depth-to-space store."""
    for s, C in ((64, 128), (128, 128), (256, 64)):
        p = s // 32
        x = rnd(f"pe{s}", (1, C, s, s))
        w = rnd(f"pw{s}", (256, C * p * p), 1 / math.sqrt(C * p * p))
        b = rnd(f"pb{s}", (256,), 0.1)
        ref = F.linear(O.patchify(x, p), w, b)                                   # [1,1024,256]
        cv = ops.Conv(w.cuda().contiguous(), b.cuda(), p, p, C, 256)
        y = ops.conv(nhwc(x), cv, stride=p, pad=(0, 0))
        assert maxabs(y.reshape(1, 1024, 256).cpu(), ref) < 3e-5, s
        w2 = rnd(f"uw{s}", (C * p * p, 256), 1 / 16.0)
        b2 = rnd(f"ub{s}", (C * p * p,), 0.1)
        t = rnd(f"ut{s}", (1, 1024, 256))
        ref2 = O.unpatchify(F.linear(t, w2, b2), p, C)
        y2 = ops.conv(t.view(1, 32, 32, 256).cuda(), ops.Conv.from_torch(w2.cuda(), b2.cuda()), d2s=(p, C))
        assert tuple(y2.shape) == (1, s, s, C)
        assert maxabs(nchw(y2), ref2) < 3e-5, s


@pytest.mark.parametrize("B,K,N,act,with_res,sliced", [(8, 256, 256, 0, True, False), (4, 256, 512, 4, False, True), (8, 128, 128, 0, False, False),
                                                      (4, 128, 256, 1, True, True), (5, 256, 128, 0, True, False)])
def test_row_panel_gemm_f32(ops, monkeypatch, B, K, N, act, with_res, sliced):
    """csrc/gemm_rp_f32.hip (persistent row-panel kernel for the K = 128 / 256 1x1 layers, fp32 MFMA) against the fp64 product; bias /
    activation / residual; channel-slice views (ld > C); == the implicit GEMM to summation order; and it is the kernel that ran."""
    H = W = 64
    monkeypatch.setattr(ops, "GEMM_RP_BF3", 0)      # this kernel, not its split-bf16 successor (tests/test_gpu_gemm_bf3.py)
    rows = ops.GEMM16_RP_MIN_ROWS
    ops.GEMM16_RP_MIN_ROWS = 1024
    try:
        xw = rnd(f"rfx{K}{N}", (B, H, W, K + (24 if sliced else 0))).cuda()
        x = xw[..., 8:8 + K] if sliced else xw
        w = rnd(f"rfw{K}{N}", (N, K), 1.0 / math.sqrt(K))
        b = rnd(f"rfb{K}{N}", (N,), 0.2)
        rw = rnd(f"rfr{K}{N}", (B, H, W, N + (16 if sliced else 0))).cuda()
        res = (rw[..., 16:] if sliced else rw) if with_res else None
        cv = ops.Conv(w.cuda().contiguous(), b.cuda(), 1, 1, K, N)
        with ops.profile() as rec:
            y = ops.conv(x, cv, act=act, res=res)
        assert [r[1].get("rp") for r in rec.rows] == [1], rec.rows
        ref = x.cpu().reshape(-1, K).double() @ w.double().T + b.double()
        if act == 1:
            ref = torch.relu(ref)
        elif act == 4:
            ref = F.gelu(ref)
        if res is not None:
            ref = ref + res.cpu().reshape(-1, N).double()
        assert maxabs(y.cpu().reshape(-1, N), ref.float()) < 2e-5 * max(1.0, float(ref.abs().max()))
        ops.GEMM_RP = 0
        try:
            y0 = ops.conv(x, cv, act=act, res=res)
        finally:
            ops.GEMM_RP = 1
        assert maxabs(y0, y) < 2e-5 * max(1.0, float(ref.abs().max()))
    finally:
        ops.GEMM16_RP_MIN_ROWS = rows


def test_row_panel_gemm_f32_unpatchify_store(ops, monkeypatch):
    """the fp32 row-panel kernel with the un-patchify (depth-to-space) store == the implicit GEMM's d2s store (summation order only)."""
    monkeypatch.setattr(ops, "GEMM_RP_BF3", 0)
    rows = ops.GEMM16_RP_MIN_ROWS
    ops.GEMM16_RP_MIN_ROWS = 1024
    try:
        for (p_, C_, B) in ((8, 64, 3), (4, 128, 2), (2, 64, 5)):
            N, K = p_ * p_ * C_, 256
            x = rnd(f"fd2x{p_}", (B, 32, 32, K)).cuda()
            cv = ops.Conv(rnd(f"fd2w{p_}", (N, K), 1.0 / math.sqrt(K)).cuda().contiguous(), rnd(f"fd2b{p_}", (N,), 0.2).cuda(), 1, 1, K, N)
            with ops.profile() as rec:
                y = ops.conv(x, cv, d2s=(p_, C_))
            assert [r[1].get("rp") for r in rec.rows] == [1] and tuple(y.shape) == (B, 32 * p_, 32 * p_, C_)
            ops.GEMM_RP = 0
            try:
                y0 = ops.conv(x, cv, d2s=(p_, C_))
            finally:
                ops.GEMM_RP = 1
            assert maxabs(y0, y) < 2e-5 * max(1.0, float(y0.abs().max()))
    finally:
        ops.GEMM16_RP_MIN_ROWS = rows


@pytest.mark.parametrize("B,H,W,Cin,N,pad", [(17, 64, 64, 128, 17, 3), (17, 64, 64, 36, 76, 0), (48, 24, 40, 20, 33, 3)])
def test_conv7x7_heads_region_kernel_f32(ops, monkeypatch, B, H, W, Cin, N, pad):
    """conv7_f32_kernel (the motion estimator's 7x7 heads in the fp32 configuration: region-direct, exact fp32 products on the fp32 MFMA)
    against the fp64 convolution and the implicit GEMM it replaces; and it is the kernel that ran."""
    monkeypatch.setattr(ops, "CONV7_F16", 0)          # this kernel, not its f16x3 successor (test below)
    x = rnd(f"c7fx{Cin}{N}", (B, Cin, H, W))
    w = rnd(f"c7fw{Cin}{N}", (N, Cin, 7, 7), 1.0 / math.sqrt(49 * Cin))
    b = rnd(f"c7fb{Cin}{N}", (N,), 0.1)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=pad).float()
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    with ops.profile() as rec:
        y = ops.conv(xin, cv, pad=(pad, pad))
    assert (rec.rows[0][1].get("mfma_flops", 0) > 0) == (Cin % 16 == 0)  # only the region kernel reports its padded executed flops; C_in % 16 != 0 stays on the implicit GEMM
    if Cin % 16:
        from synergize_motion_appearance_amd import lib as L                # the kernel itself handles ragged channel counts: call it directly
        yk = torch.empty_like(y)
        L.check(L.load().smx_conv7_f32(xin.data_ptr(), Cin, cv.w7_f32.data_ptr(), cv.b.data_ptr(), yk.data_ptr(), N, B, H, W, Cin, N, pad, 0,
                                       torch.cuda.current_stream().cuda_stream), "smx_conv7_f32")
        assert maxabs(yk, y) < 2e-5 * max(1.0, float(ref.abs().max()))
    scale = max(1.0, float(ref.abs().max()))
    assert maxabs(y.permute(0, 3, 1, 2).cpu(), ref) < 2e-5 * scale
    ops.CONV7_F32 = 0
    try:
        y0 = ops.conv(xin, cv, pad=(pad, pad))
    finally:
        ops.CONV7_F32 = 1
    assert maxabs(y0, y) < 2e-5 * scale


@pytest.mark.parametrize("name,B,H,W,Cin,N,pad,mk", [("mask head", 17, 64, 64, 128, 17, 3, lambda x: x), ("keypoint head", 17, 64, 64, 36, 76, 0, lambda x: x),
                                                      ("ragged", 48, 24, 40, 20, 33, 3, lambda x: x), ("tiny", 17, 64, 64, 64, 17, 3, lambda x: x * 1e-4),
                                                      ("huge", 17, 64, 64, 64, 17, 3, lambda x: x * 3e3),
                                                      ("channels of very different scale", 17, 64, 64, 128, 17, 3, lambda x: torch.cat([x[:, :16] * 1e-3, x[:, 16:] * 100.0], 1))])
def test_conv7x7_heads_f16x3(ops, monkeypatch, name, B, H, W, Cin, N, pad, mk):
    """the 7x7 heads in the f16x3 arithmetic (conv7_bf16x3_kernel<NT, true>: two half levels, three products; weights scaled at pack time, the input by the block from
    a first pass over its region): it is the kernel that runs for big launches of the fp32 configuration, and against the fp64 convolution its error is within 1.25x of
    conv7_f32_kernel's / the implicit GEMM's at every input scale."""
    x = mk(rnd(f"c7fx{Cin}{N}", (B, Cin, H, W)))
    w = rnd(f"c7fw{Cin}{N}", (N, Cin, 7, 7), 1.0 / math.sqrt(49 * Cin))
    b = rnd(f"c7fb{Cin}{N}", (N,), 0.1) * float(x.abs().mean())
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=pad)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    with ops.profile() as rec:
        y16 = ops.conv(xin, cv, pad=(pad, pad), act=0)
    assert rec.rows[0][1].get("bf3") == 4 and rec.rows[0][1].get("k") == 7
    monkeypatch.setattr(ops, "CONV7_F16", 0)
    with ops.profile() as rec:
        y32 = ops.conv(xin, cv, pad=(pad, pad), act=0)
    assert rec.rows[0][1].get("bf3") is None
    sc_ = float(ref.pow(2).mean().sqrt())
    e16 = float((y16.permute(0, 3, 1, 2).cpu().double() - ref).abs().max()) / sc_; e32 = float((y32.permute(0, 3, 1, 2).cpu().double() - ref).abs().max()) / sc_
    r16 = float((y16.permute(0, 3, 1, 2).cpu().double() - ref).pow(2).mean().sqrt()) / sc_; r32 = float((y32.permute(0, 3, 1, 2).cpu().double() - ref).pow(2).mean().sqrt()) / sc_
    print(f"\n{name}: relative max error  fp32 {e32:.3e}  f16x3 {e16:.3e}   rms {r32:.3e} {r16:.3e}")
    assert bool(torch.isfinite(y16).all()) and e16 <= 1.25 * e32 + 1e-7 and r16 <= 1.1 * r32 + 1e-8
    # sigmoid epilogue (the occlusion / mask use) and determinism
    monkeypatch.setattr(ops, "CONV7_F16", 1)
    ys = ops.conv(xin, cv, pad=(pad, pad), act=5)
    assert maxabs(ys.permute(0, 3, 1, 2).cpu(), torch.sigmoid(ref).float()) < 2e-5 * max(1.0, sc_) and torch.equal(ys, ops.conv(xin, cv, pad=(pad, pad), act=5))


def test_conv7x7_two_channel_flow_encoder_f32(ops):
    """conv7_c2_f32_kernel (BasicMotionEncoder.convf1 in the fp32 configuration: one fp32 MFMA per tap, k pair = channel pair) against the
    fp64 convolution and against the implicit GEMM it replaces."""
    B, H, W, N = 9, 64, 64, 128
    x = rnd("c72fx", (B, 2, H, W)) * 3.0
    w = rnd("c72fw", (N, 2, 7, 7), 1.0 / math.sqrt(98))
    b = rnd("c72fb", (N,), 0.1)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    with ops.profile() as rec:
        y = ops.conv(xin, cv, act=1)
    assert y.dtype == torch.float32 and rec.rows[0][1]["K"] == 98
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=3)).float()
    assert maxabs(y.permute(0, 3, 1, 2).cpu(), ref) < 2e-5 * max(1.0, float(ref.abs().max()))
    ops.CONV7_C2 = 0
    try:
        y0 = ops.conv(xin, cv, act=1)
    finally:
        ops.CONV7_C2 = 1
    assert maxabs(y0, y) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_gemm_nt_batched_heads(ops):
    """attention-shaped batched NT GEMMs incl. d_head = 4 (generic path) and per-row bias."""
    for dh, E in ((32, 256), (4, 32)):
        B, H, N, S = 2, 8, 1024, 256
        q = rnd(f"gq{dh}", (B, N, E)).cuda()
        k = rnd(f"gk{dh}", (S, E)).cuda()
        sc = torch.empty((B, H, N, S), device="cuda")
        ops.gemm_nt(q, k, sc, M=N, N=S, K=dh, lda=E, ldb=E, ldc=S, nb0=B, nb1=H, a_bs=(N * E, dh), bt_bs=(0, dh),
                    c_bs=(H * N * S, N * S), alpha=dh ** -0.5)
        ref = torch.einsum("bnhd,shd->bhns", q.cpu().view(B, N, H, dh), k.cpu().view(S, H, dh)) * dh ** -0.5
        assert maxabs(sc.cpu(), ref) < 2e-5, dh
    a = rnd("ga", (64, 64)).cuda()
    bt = rnd("gb", (3, 100, 64)).cuda()
    bias = rnd("gbias", (64,)).cuda()
    c = torch.empty((3, 64, 100), device="cuda")
    ops.gemm_nt(a, bt, c, M=64, N=100, K=64, lda=64, ldb=64, ldc=100, nb0=3, bt_bs=(100 * 64, 0), c_bs=(64 * 100, 0),
                bias=bias, bias_per_row=True)
    ref = torch.einsum("mk,bnk->bmn", a.cpu(), bt.cpu()) + bias.cpu().view(1, 64, 1)
    assert maxabs(c.cpu(), ref) < 2e-5


def test_attnblock_fused_f32(ops):
    """engine_netg._Attn in the fp32 configuration: the core as ONE kernel (smx_attnblock_f32: exact fp32 products on the fp32 MFMA, online softmax,
    no [B, N, N] score tensor) against (i) the AttnBlock of the reference restated in fp64 (archs/vqgan_arch.py:229-253: GroupNorm, q / k / v 1x1
    convolutions, softmax(q k^T / sqrt(C)) v, proj_out, residual) and (ii) the three-launch form it replaces (QK^T GEMM, softmax_rows, PV GEMM);
    large, badly centred logits included (the running maximum has to move)."""
    from synergize_motion_appearance_amd import engine_netg as E
    C_, B = 256, 3
    P = {"a.norm.weight": 1 + 0.1 * rnd("af_g", (C_,)), "a.norm.bias": 0.1 * rnd("af_b", (C_,))}
    for n in ("q", "k", "v", "proj_out"):
        P[f"a.{n}.weight"] = rnd("af_w" + n, (C_, C_, 1, 1), (3.0 if n in ("q", "k") else 1.5) / math.sqrt(C_))
        P[f"a.{n}.bias"] = rnd("af_bias" + n, (C_,), 0.1)
    x = rnd("af_x", (B, 32, 32, C_))
    # fp64 restatement
    xd = x.permute(0, 3, 1, 2).double()
    hn = F.group_norm(xd, 32, P["a.norm.weight"].double(), P["a.norm.bias"].double(), eps=1e-6)
    q, k, v = (F.conv2d(hn, P[f"a.{n}.weight"].double(), P[f"a.{n}.bias"].double()).reshape(B, C_, -1) for n in ("q", "k", "v"))
    w_ = torch.softmax(torch.bmm(q.permute(0, 2, 1), k) * (int(C_) ** (-0.5)), dim=2)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(B, C_, 32, 32)
    ref = (xd + F.conv2d(h_, P["a.proj_out.weight"].double(), P["a.proj_out.bias"].double())).permute(0, 2, 3, 1).float()
    blk = E._Attn({k_: v_.cuda() for k_, v_ in P.items()}, "a")
    with ops.profile() as rec:
        fused = blk(x.cuda())
    names = [r[0] for r in rec.rows]
    assert "attnblock" in names and "softmax" not in names, names
    E.ATTNBLOCK_FUSED32 = 0
    try:
        with ops.profile() as rec3:
            three = blk(x.cuda())
    finally:
        E.ATTNBLOCK_FUSED32 = 1
    assert "attnblock" not in [r[0] for r in rec3.rows] and "softmax" in [r[0] for r in rec3.rows]
    scale = float(ref.abs().max())
    assert maxabs(fused.cpu(), ref) < 2e-5 * scale, (maxabs(fused.cpu(), ref), scale)
    assert maxabs(fused.cpu(), three.cpu()) < 2e-5 * scale
    # the kernel alone on hostile scores: one query's logits span +-60 (exp2 range) and the maximum arrives in the LAST tile
    N = 1024
    qk = torch.zeros((2, N, 2 * C_))
    qk[..., :C_] = rnd("af_q2", (2, N, C_), 1.0)
    qk[..., C_:] = rnd("af_k2", (2, N, C_), 1.0)
    qk[:, -1, C_:] *= 6.0                                                    # the last key dominates many rows
    vt = rnd("af_v2", (2, C_, N), 1.0)
    sc = 0.25
    qd, kd = qk[..., :C_].double(), qk[..., C_:].double()
    o_ref = torch.bmm(torch.softmax(torch.bmm(qd, kd.transpose(1, 2)) * sc, dim=2), vt.double().transpose(1, 2)).float()
    from synergize_motion_appearance_amd import lib as L_
    qkc, vtc = qk.cuda().contiguous(), vt.cuda().contiguous()
    o = torch.empty((2, N, C_), device="cuda")
    L_.check(L_.load().smx_attnblock_f32(qkc.data_ptr(), 2 * C_, N * 2 * C_, qkc.data_ptr() + 4 * C_, 2 * C_, N * 2 * C_, vtc.data_ptr(), N, C_ * N,
                                         o.data_ptr(), C_, N * C_, 2, N, N, C_, sc, ops._stream()), "smx_attnblock_f32")
    assert maxabs(o.cpu(), o_ref) < 3e-5 * float(o_ref.abs().max()), maxabs(o.cpu(), o_ref)
    assert L_.load().smx_attnblock_f32(qkc.data_ptr(), 2 * C_, N * 2 * C_, qkc.data_ptr() + 4 * C_, 2 * C_, N * 2 * C_, vtc.data_ptr(), N, C_ * N,
                                       o.data_ptr(), C_, N * C_, 2, N - 32, N, C_, sc, ops._stream()) != 0     # L % 128 != 0 is refused


@pytest.mark.parametrize("dh,E,S,shared,masked", [(32, 256, 1024, False, True), (32, 256, 256, True, False), (32, 256, 768, True, False),
                                                   (4, 32, 1024, False, False), (4, 32, 512, True, False), (4, 32, 1024, False, True)])
def test_fused_attention(ops, dh, E, S, shared, masked):
    """smx_attention_f32 vs the explicit nn.MultiheadAttention core of the oracle (softmax(q k^T) v per head)."""
    B, H, N = 2, 8, 1024
    q = rnd(f"aq{dh}{S}", (B, N, E))
    kv = rnd(f"akv{dh}{S}", ((1 if shared else B), 1024, 2 * E))
    mask = None
    if masked:
        mask = torch.zeros((B, S), dtype=torch.bool)
        mask[0, 5::9] = True
        mask[1, :40] = True
    k, v = kv[..., :E][:, :S], kv[..., E:][:, :S]
    qh = (q * dh ** -0.5).view(B, N, H, dh).transpose(1, 2)
    kh = k.expand(B, -1, -1).reshape(B, S, H, dh).transpose(1, 2)
    vh = v.expand(B, -1, -1).reshape(B, S, H, dh).transpose(1, 2)
    sc = qh @ kh.transpose(-1, -2)
    if mask is not None:
        sc = sc.masked_fill(mask.view(B, 1, 1, S), float("-inf"))
    ref = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B, N, E)
    kvd = kv.cuda()
    kd, vd = (kvd[0, :, :E], kvd[0, :, E:]) if shared else (kvd[..., :E], kvd[..., E:])
    o = ops.attention(q.cuda(), kd, vd, H, dh, S, k_shared=shared, mask=None if mask is None else mask.to(torch.uint8).cuda())
    assert maxabs(o.cpu(), ref) < 5e-6


def test_fused_attention_fully_masked_row_is_nan_like_reference(ops):
    B, H, N, E, dh = 1, 8, 1024, 256, 32
    q, kv = rnd("nq", (B, N, E)).cuda(), rnd("nkv", (B, N, 2 * E)).cuda()
    o = ops.attention(q, kv[..., :E], kv[..., E:], H, dh, N, mask=torch.ones((B, N), dtype=torch.uint8, device="cuda"))
    assert bool(torch.isnan(o).all())


def test_conv_split_k(ops):
    """deep hourglass layer shapes take the split-K path (few tiles, K = 9*1024)."""
    for (B, Cin, Cout, H, up2) in ((2, 1024, 1024, 4, False), (3, 2048, 512, 4, True), (10, 512, 1024, 4, False)):
        x = rnd(f"skx{Cin}{up2}", (B, Cin, H, H))
        w = rnd(f"skw{Cin}{up2}", (Cout, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
        b = rnd(f"skb{Cin}", (Cout,), 0.1)
        xe = F.interpolate(x, scale_factor=2.0, mode="nearest") if up2 else x
        ref = F.relu(F.conv2d(xe, w, b, padding=1))
        y = ops.conv(nhwc(x), ops.Conv.from_torch(w.cuda(), b.cuda()), up2=up2, act=1)
        assert maxabs(nchw(y), ref) < 3e-5


def test_gemm_rejects_bad_descriptor(ops):
    from synergize_motion_appearance_amd.lib import SmxError
    a = torch.zeros((4, 4), device="cuda")
    with pytest.raises(SmxError):
        ops.gemm_nt(a, a, a, M=4, N=4, K=8, lda=4, ldb=4, ldc=4)      # ldb < K
    with pytest.raises(SmxError):
        ops.conv(torch.zeros(1, 4, 4, 8), ops.Conv.from_torch(torch.zeros(8, 8, 3, 3).cuda(), None))   # CPU tensor


# ---------------------------------------------------------------------------------------
# normalisation / softmax
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,H", [(32, 32), (64, 64), (128, 64), (256, 32), (64, 256)])
def test_groupnorm_swish(ops, C, H):
    x = rnd(f"gn{C}{H}", (2, C, H, H)) * 1.7 + 0.3
    g = 1 + 0.1 * rnd(f"gg{C}", (C,))
    b = 0.1 * rnd(f"gb{C}", (C,))
    ref = F.group_norm(x, 32, g, b, 1e-6)
    y = ops.groupnorm(nhwc(x), g.cuda(), b.cuda(), swish=False)
    assert maxabs(nchw(y), ref) < 1e-5
    y = ops.groupnorm(nhwc(x), g.cuda(), b.cuda(), swish=True)
    assert maxabs(nchw(y), O.swish(ref)) < 1e-5


@pytest.mark.parametrize("C,H", [(64, 32), (128, 16), (32, 64)])
def test_groupnorm_large_mean_small_std(ops, C, H):
    """|mean| >> std (activations of real checkpoints after stacked ResBlocks): sum / sum-of-squares statistics cancel
    in fp32 (E[x^2] - mean^2 = 2500.01 - 2500 with 2.4e-4 resolution); the Welford-form partials must not.  Both the
    standalone statistics pass and the partials emitted by the Winograd epilogue, against float64 group_norm of the
    same fp32 tensor.  Tolerance: the fp32 input itself resolves (x - mean)/std to 50 * 6e-8 / 0.1 = 3e-5."""
    x = rnd(f"gnoff{C}{H}", (2, C, H, H)) * 0.1 + 50.0
    g, b = 1 + 0.1 * rnd(f"gog{C}", (C,)), 0.1 * rnd(f"gob{C}", (C,))
    ref = F.group_norm(x.double(), 32, g.double(), b.double(), 1e-6)
    y = ops.groupnorm(nhwc(x), g.cuda(), b.cuda(), swish=False)
    assert maxabs(nchw(y), ref) < 3e-4
    ss = ops.groupnorm_stats(nhwc(x), g.cuda(), b.cuda())
    assert maxabs(nchw(ops.groupnorm_apply(nhwc(x), ss, swish=False)), ref) < 3e-4
    # epilogue partials: a conv whose output sits at 50 +- 0.1 (bias 50, small weights)
    w = rnd(f"gow{C}", (C, C, 3, 3), 0.1 / math.sqrt(9 * C))
    cv = ops.Conv.from_torch(w.cuda(), torch.full((C,), 50.0, device="cuda"))
    z = ops.conv(nhwc(rnd(f"goz{C}{H}", (2, C, H, H))), cv, want_stats=True)
    assert z._gn_part is not None
    refz = F.group_norm(nchw(z).double(), 32, g.double(), b.double(), 1e-6)
    assert float(refz.abs().max()) > 1.0                       # a real spread survives the normalisation
    assert maxabs(nchw(ops.groupnorm_apply(z, ops.groupnorm_stats(z, g.cuda(), b.cuda()), swish=False)), refz) < 3e-4


@pytest.mark.parametrize("E", [32, 256])
def test_layernorm_pos(ops, E):
    x = rnd(f"ln{E}", (2, 1024, E)) * 2 + 0.5
    g, b, pos = 1 + 0.1 * rnd("lg", (E,)), 0.1 * rnd("lb", (E,)), 0.2 * rnd("lp", (1024, E))
    ref = F.layer_norm(x, (E,), g, b, 1e-5)
    y, yp = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), pos=pos.cuda())
    assert maxabs(y.cpu(), ref) < 2e-6 and maxabs(yp.cpu(), ref + pos) < 2e-6


@pytest.mark.parametrize("S", [256, 512, 768, 1024])
def test_softmax_rows_masked(ops, S):
    s = rnd(f"sm{S}", (2, 4, 64, S)) * 3
    mask = torch.zeros((2, S), dtype=torch.bool)
    mask[1, ::7] = True
    ref = torch.softmax((s * 0.5).masked_fill(mask.view(2, 1, 1, S), float("-inf")), -1)
    d = s.cuda().contiguous()
    ops.softmax_rows(d, S, 0.5, mask.to(torch.uint8).cuda(), 4 * 64)
    assert maxabs(d.cpu(), ref) < 2e-7


# ---------------------------------------------------------------------------------------
# warp / resize / pooling / anti-alias
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,s", [(256, 32), (128, 64), (128, 128), (64, 256)])
def test_warp_four_scales(ops, C, s):
    """A7 vs deform_input + occlude_input (both ATen and the explicit restatement). Smooth features:
    two correct fp32 implementations differ by flow rounding x (s-1)/2 x feature gradient."""
    B = 2
    feat = F.interpolate(rnd(f"wf{s}", (1, C, 8, 8)), size=(s, s), mode="bicubic", align_corners=True)
    flow = O.make_coordinate_grid(64, 64, torch.float32)[None] + 0.25 * F.interpolate(
        rnd(f"wfl{s}", (B, 2, 6, 6)), size=(64, 64), mode="bicubic", align_corners=True).permute(0, 2, 3, 1)
    occ = torch.sigmoid(rnd(f"wo{s}", (B, 1, 64, 64)))
    assert float(((flow > 1) | (flow < -1)).float().mean()) > 0.005       # zero padding exercised
    ref = O.occlude_input(O.deform_input(feat.repeat(B, 1, 1, 1), flow), occ)
    y = ops.warp(nhwc(feat), flow.cuda(), occ.view(B, 64, 64).cuda())
    tol = 2e-5 if s <= 64 else 3e-4      # ~1e-4 between two correct fp32 warps at s=256
    assert maxabs(nchw(y), ref) < tol
    y2 = ops.warp(nhwc(feat.repeat(B, 1, 1, 1)), flow.cuda())
    assert maxabs(nchw(y2), O.deform_input(feat.repeat(B, 1, 1, 1), flow)) < tol


@pytest.mark.parametrize("C,s,B,fs", [(64, 256, 8, 64), (128, 128, 16, 64), (256, 32, 20, 64), (128, 64, 9, 64), (64, 64, 16, 64)])
def test_warp_row_chunk_kernel_equals_per_lane_kernel_and_oracle(ops, C, s, B, fs, monkeypatch):
    """Large launches take `warp_rows_kernel` (coordinates once per pixel, shuffled to the pixel's lanes);
    it must agree with the per-lane kernel -- bit for bit where no flow resize is involved, to flow-ulp
    level otherwise (the two resize code shapes contract into different FMAs) -- with broadcast and
    per-frame features, flow at 64x64 and at the feature size, and match the oracle."""
    feat = F.interpolate(rnd(f"wr{s}{C}", (1, C, 8, 8)), size=(s, s), mode="bicubic", align_corners=True)
    fs = s if (C, s) == (64, 64) else fs                     # one case with flow size == feature size (no resize)
    flow = O.make_coordinate_grid(fs, fs, torch.float32)[None] + 0.25 * F.interpolate(
        rnd(f"wrf{s}{C}", (B, 2, 6, 6)), size=(fs, fs), mode="bicubic", align_corners=True).permute(0, 2, 3, 1)
    flow[0, :2, :2] = float("nan")                           # NaN coordinates: all taps skipped, as in ATen
    flow[1, 3, 3] = 1e9
    occ = torch.sigmoid(rnd(f"wro{s}{C}", (B, 1, fs, fs)))
    fd, od, xs = flow.cuda().contiguous(), occ.view(B, fs, fs).cuda().contiguous(), nhwc(feat)
    xb = nhwc(feat.repeat(B, 1, 1, 1) * torch.linspace(0.5, 1.5, B).view(B, 1, 1, 1))
    new = [ops.warp(xs, fd, od), ops.warp(xs, fd), ops.warp(xb, fd, od)]
    prev = ops.set_tuning("warp_rows", 0)                      # the per-lane-coordinates kernel
    try:
        old = [ops.warp(xs, fd, od), ops.warp(xs, fd), ops.warp(xb, fd, od)]
    finally:
        ops.set_tuning("warp_rows", prev)
    for a, b in zip(new, old):
        a, b = torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)
        # two correct fp32 flow resizes differ by an ulp of the flow, amplified by (s-1)/2 x feature gradient
        assert torch.equal(a, b) if fs == s else maxabs(a.cpu(), b.cpu()) < (3e-4 if s >= 128 else 5e-5)
    fl = flow.clone()
    ref = O.occlude_input(O.deform_input(feat.repeat(B, 1, 1, 1), fl), occ)
    ok = torch.isfinite(ref)
    assert maxabs(torch.where(ok, nchw(new[0]), torch.zeros(())), torch.where(ok, ref, torch.zeros(()))) < (3e-4 if s >= 128 else 5e-5)


@pytest.mark.parametrize("C,s", [(64, 512), (128, 256), (128, 128), (256, 64)])
def test_warp_config4_512_kernel_level(ops, C, s):
    """BASELINE.json configs[3] (512x512: 4x flow + warp grid).  The reference itself raises at 512
    (SURVEY.md section 8d), so parity is kernel-level: A7 with a 128x128 flow against the ATen semantics."""
    B = 2
    feat = F.interpolate(rnd(f"w5f{s}", (1, C, 8, 8)), size=(s, s), mode="bicubic", align_corners=True)
    flow = O.make_coordinate_grid(128, 128, torch.float32)[None] + 0.2 * F.interpolate(
        rnd(f"w5fl{s}", (B, 2, 6, 6)), size=(128, 128), mode="bicubic", align_corners=True).permute(0, 2, 3, 1)
    occ = torch.sigmoid(rnd(f"w5o{s}", (B, 1, 128, 128)))
    ref = O.occlude_input(O.deform_input(feat.repeat(B, 1, 1, 1), flow), occ)
    y = ops.warp(nhwc(feat), flow.cuda(), occ.view(B, 128, 128).cuda())
    assert maxabs(nchw(y), ref) < (5e-4 if s >= 256 else 5e-5)


def test_sparse_motion_config4_128_grid(ops):
    """configs[3]: A4-A6b stage on a 128x128 grid (512x512 input, scale 0.25)."""
    from synergize_motion_appearance_amd.synth import synth_keypoints
    B, K, G = 1, 15, 128
    kps, kpd = synth_keypoints(B, seed=3)
    src = F.interpolate(rnd("sm5_src", (1, 3, 16, 16)), size=(G, G), mode="bicubic", align_corners=True)
    ident = O.make_coordinate_grid(G, G, torch.float32).view(1, 1, G, G, 2)
    cg = ident - kpd["value"].view(B, K, 1, 1, 2)
    jac = torch.matmul(kps["jacobian"], torch.inverse(kpd["jacobian"])).unsqueeze(-3).unsqueeze(-3)
    d2s = torch.matmul(jac, cg.unsqueeze(-1)).squeeze(-1) + kps["value"].view(B, K, 1, 1, 2)
    sparse = torch.cat([ident.repeat(B, 1, 1, 1, 1), d2s], 1)
    deformed = F.grid_sample(src.repeat(K + 1, 1, 1, 1), sparse.view(K + 1, G, G, 2), align_corners=False).view(B, K + 1, 3, G, G)
    heat = O.kp2gaussian(kpd["value"], G, G) - O.kp2gaussian(kps["value"], G, G)
    hg = torch.empty((B, G, G, 64), device="cuda")
    sp, dh = ops.sparse_motion(nhwc(src), kpd["value"].cuda(), kpd["jacobian"].reshape(B, K, 4).cuda(),
                               kps["value"].cuda(), kps["jacobian"].reshape(B, K, 4).cuda(), hg, B, K)
    got = hg.cpu().view(B, G, G, 16, 4).permute(0, 3, 4, 1, 2)
    assert maxabs(sp.cpu(), sparse) < 2e-6 and maxabs(got[:, 1:, 0], heat) < 1e-6 and maxabs(got[:, :, 1:4], deformed) < 1e-4
    ml = rnd("ml5", (B, 16, G, G))
    d, _, _ = ops.mask_deformation(nhwc(ml), sp)
    ref = (sparse.permute(0, 1, 4, 2, 3) * torch.softmax(ml, 1).unsqueeze(2)).sum(1).permute(0, 2, 3, 1)
    assert maxabs(d.cpu(), ref) < 2e-6


@pytest.mark.parametrize("B,C,N,s", [(2, 64, 192, 256), (1, 32, 48, 128), (2, 16, 20, 72)])
def test_per_pixel_op_at_the_bilinear_taps_only(ops, B, C, N, s):
    """relu(conv1x1(x)) followed by the align_corners=True down-sampling == the same op evaluated only at the 4 taps
    of each output pixel and blended (to_context at 256x256: a quarter of the pixels): equal up to the FMA contraction
    of the blend (<= 1 ulp), and vs F.interpolate."""
    x = rnd(f"tp{C}{s}", (B, C, s, s))
    w, b = rnd(f"tpw{C}{N}", (N, C, 1, 1), 1.0 / math.sqrt(C)), rnd(f"tpb{N}", (N,), 0.1)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    full = ops.resize(ops.conv(nhwc(x), cv, act=1), 64, 64)
    taps = ops.resize_taps_gather(nhwc(x), 64, 64)
    assert tuple(taps.shape) == (B, 64, 256, C)
    sampled = ops.resize_taps_combine(ops.conv(taps, cv, act=1), s, s)
    assert maxabs(full.cpu(), sampled.cpu()) < 1e-6          # same source indices (unfused scale*o); blend FMAs may differ by 1 ulp
    ref = F.interpolate(F.relu(F.conv2d(x, w, b)), size=(64, 64), mode="bilinear", align_corners=True)
    assert maxabs(nchw(sampled), ref) < 2e-5
    wide = torch.zeros((B, 64, 64, N + 8), device="cuda")
    ops.resize_taps_combine(ops.conv(taps, cv, act=1), s, s, out=wide[..., 4:4 + N])
    assert torch.equal(wide[..., 4:4 + N], sampled) and float(wide[..., :4].abs().max()) == 0.0


def test_resize_avgpool_antialias(ops):
    x = rnd("rs", (2, 15, 64, 64))
    assert maxabs(nchw(ops.resize(nhwc(x), 32, 32)), O.resize_ac(x, (32, 32))) < 1e-6
    x = rnd("rs2", (2, 32, 32, 32))
    assert maxabs(nchw(ops.resize(nhwc(x), 64, 64)), O.resize_ac(x, (64, 64))) < 1e-6
    x = rnd("rs3", (1, 64, 256, 256))
    assert maxabs(nchw(ops.resize(nhwc(x), 32, 32)), O.resize_ac(x, (32, 32))) < 1e-6
    x = rnd("ap", (2, 64, 32, 32))
    assert maxabs(nchw(ops.avgpool2(nhwc(x))), F.avg_pool2d(x, 2)) < 1e-6
    Pm = weights("network_motion_estimator")
    img = rnd("aa", (2, 3, 256, 256)).clamp(-1, 1)
    ref = O.antialias_down(Pm, "kp_detector.down", img)
    y = ops.antialias_down(img.cuda(), Pm["kp_detector.down.weight"].reshape(3, 13, 13).cuda())
    assert maxabs(nchw(y), ref) < 1e-6


# ---------------------------------------------------------------------------------------
# motion stages
# ---------------------------------------------------------------------------------------
def test_kp_head(ops):
    B, K = 2, 15
    logits = rnd("kl", (B, K, 58, 58))
    jm = rnd("kj", (B, 4 * K, 58, 58))
    heat = torch.softmax(logits.view(B, K, -1) / 0.1, 2).view(B, K, 58, 58)
    grid = O.make_coordinate_grid(58, 58, torch.float32)
    value = (heat.unsqueeze(-1) * grid.view(1, 1, 58, 58, 2)).sum((2, 3))
    jac = (heat.unsqueeze(2) * jm.view(B, K, 4, 58, 58)).view(B, K, 4, -1).sum(-1).view(B, K, 2, 2)
    v, j = ops.kp_head(nhwc(logits), nhwc(jm), K, 0.1)
    assert maxabs(v.cpu(), value) < 2e-6 and maxabs(j.cpu(), jac) < 5e-6
    both = nhwc(torch.cat([jm, logits, torch.zeros(B, 1, 58, 58)], 1))              # stacked [jac 60 | kp 15 | pad] heads
    v, j = ops.kp_head(both[..., 60:75], both[..., :60], K, 0.1)
    assert maxabs(v.cpu(), value) < 2e-6 and maxabs(j.cpu(), jac) < 5e-6


def test_normalize_kp_kernel_vs_reference(ops):
    """A0 on the device against the reference fixture (4 flag combinations of demo.normalize_kp)."""
    from synergize_motion_appearance_amd.driver import normalize_kp
    g, gn = golden("kp.npz"), golden("normalize_kp.npz")
    kp = lambda v, j: {"value": torch.from_numpy(v).cuda(), "jacobian": torch.from_numpy(j).cuda()}
    kp_s, kp_0, kp_3 = kp(g["src_value"], g["src_jacobian"]), kp(g["drv_value"][0:1], g["drv_jacobian"][0:1]), kp(g["drv_value"][3:4], g["drv_jacobian"][3:4])
    for rel in (0, 1):
        for ad in (0, 1):
            r = normalize_kp(kp_s, kp_3, kp_0, bool(ad), bool(rel), bool(rel))
            assert maxabs(r["value"].cpu(), gn[f"value_r{rel}a{ad}"]) < 1e-6
            assert maxabs(r["jacobian"].cpu(), gn[f"jacobian_r{rel}a{ad}"]) < 1e-5
    kp_all = kp(g["drv_value"], g["drv_jacobian"])                       # batched == per-frame
    rb = normalize_kp(kp_s, kp_all, kp_0, True, True, True)
    assert maxabs(rb["value"][3:4].cpu(), gn["value_r1a1"]) < 1e-6 and maxabs(rb["jacobian"][3:4].cpu(), gn["jacobian_r1a1"]) < 1e-5


def test_sparse_motion_and_mask_deformation(ops):
    from synergize_motion_appearance_amd.synth import synth_keypoints
    B, K = 2, 15
    kps, kpd = synth_keypoints(B, seed=11)
    src = rnd("sm_src", (1, 3, 64, 64))
    # oracle pieces (dense_motion_arch.py:65-116)
    heat = O.kp2gaussian(kpd["value"], 64, 64) - O.kp2gaussian(kps["value"], 64, 64)
    ident = O.make_coordinate_grid(64, 64, torch.float32).view(1, 1, 64, 64, 2)
    cg = ident - kpd["value"].view(B, K, 1, 1, 2)
    jac = torch.matmul(kps["jacobian"], torch.inverse(kpd["jacobian"])).unsqueeze(-3).unsqueeze(-3)
    d2s = torch.matmul(jac, cg.unsqueeze(-1)).squeeze(-1) + kps["value"].view(B, K, 1, 1, 2)
    sparse = torch.cat([ident.repeat(B, 1, 1, 1, 1), d2s], 1)
    rep = src.repeat(B * (K + 1), 1, 1, 1)
    deformed = F.grid_sample(rep, sparse.view(B * (K + 1), 64, 64, 2), align_corners=False).view(B, K + 1, 3, 64, 64)
    hg = torch.empty((B, 64, 64, 128), device="cuda")
    sp, dh = ops.sparse_motion(nhwc(src), kpd["value"].cuda(), kpd["jacobian"].reshape(B, K, 4).cuda(),
                               kps["value"].cuda(), kps["jacobian"].reshape(B, K, 4).cuda(), hg[..., 64:], B, K)
    assert maxabs(sp.cpu(), sparse) < 2e-6
    assert maxabs(dh.cpu().permute(0, 3, 1, 2), O.kp2gaussian(kpd["value"], 64, 64)) < 1e-6
    got = hg[..., 64:].cpu().view(B, 64, 64, 16, 4).permute(0, 3, 4, 1, 2)          # [B,16,4,H,W]
    assert maxabs(got[:, 1:, 0], heat) < 1e-6 and float(got[:, 0, 0].abs().max()) == 0.0
    # white-noise source: 2e-6 flow noise (closed-form vs LU 2x2 inverse) x 32 px x O(1)/px gradient
    assert maxabs(got[:, :, 1:4], deformed) < 1e-4
    ml = rnd("ml", (B, 16, 64, 64)) * 2
    mask = torch.softmax(ml, 1)
    deformation = (sparse.permute(0, 1, 4, 2, 3) * mask.unsqueeze(2)).sum(1).permute(0, 2, 3, 1)
    d, m, _ = ops.mask_deformation(nhwc(ml), sp, want_mask=True)
    assert maxabs(d.cpu(), deformation) < 2e-6 and maxabs(nchw(m), mask) < 1e-6
    ml17 = torch.cat([ml, rnd("mlocc", (B, 1, 64, 64))], 1)                      # stacked mask + occlusion logits
    d, _, occ = ops.mask_deformation(nhwc(ml17), sp, K1=16, fused_occ=True)
    assert maxabs(d.cpu(), deformation) < 2e-6 and maxabs(occ.cpu(), torch.sigmoid(ml17[:, 16])) < 1e-6


def test_flow_stage_kernels(ops):
    B = 2
    flow = O.make_coordinate_grid(64, 64, torch.float32)[None] + 0.2 * rnd("ff", (B, 64, 64, 2))
    xx = torch.linspace(-1., 1., 64)
    gx, gy = torch.meshgrid(xx, xx, indexing="xy")
    grid = torch.stack([gx, gy], -1)[None]
    assert maxabs(ops.flow_to_residual(flow.cuda()).cpu(), (flow - grid) * 31.5) < 1e-5
    r = rnd("fr", (B, 64, 64, 3))
    occ = torch.sigmoid(rnd("fo", (B, 64, 64)))
    m, rn, o = ops.flow_occ_update(flow.cuda(), r.cuda(), occ.cuda())
    assert maxabs(rn.cpu(), r[..., :2] / 31.5) < 1e-7 and maxabs(m.cpu(), flow + r[..., :2] / 31.5) < 1e-6
    assert maxabs(o.cpu(), torch.sigmoid(occ + r[..., 2])) < 1e-6
    big = O.make_coordinate_grid(64, 64, torch.float32)[None] * 1.2 + 0.1 * rnd("fb", (B, 64, 64, 2))
    m32 = O.resize_ac(big.permute(0, 3, 1, 2), (32, 32)).reshape(B, 2, 1024)
    ign = ((m32 > 1) | (m32 < -1)).any(1)
    got = ops.motion_ignore(big.cuda()).cpu().bool()
    # flips only allowed where |coord| is within fp32 noise of the threshold 1.0
    diff = got != ign
    near = ((m32.abs() - 1).abs() < 1e-5).any(1)
    assert not (diff & ~near).any() and 0.05 < ign.float().mean() < 0.9
    d, sc, sh = rnd("s1", (2, 8, 8, 64)), rnd("s2", (2, 8, 8, 64)), rnd("s3", (2, 8, 8, 64))
    assert maxabs(ops.sft_combine(d.cuda(), sc.cuda(), sh.cuda(), 0.7).cpu(), d + 0.7 * (d * sc + sh)) < 1e-6
    wide = torch.cat([sh, d], -1).cuda()                      # dec as the upper channel half of an [enc|dec] buffer
    assert maxabs(ops.sft_combine(wide[..., 64:], sc.cuda(), sh.cuda(), 0.7).cpu(), d + 0.7 * (d * sc + sh)) < 1e-6
    assert maxabs(ops.add(d.cuda(), sc.cuda()).cpu(), d + sc) == 0.0


def test_layout_and_uint8(ops):
    x = rnd("lay", (2, 5, 33, 17))
    assert maxabs(ops.nchw_to_nhwc(x.cuda()).cpu(), x.permute(0, 2, 3, 1)) == 0.0
    assert maxabs(ops.nhwc_to_nchw(x.permute(0, 2, 3, 1).contiguous().cuda()).cpu(), x) == 0.0
    t = synth_input("tensor2img", (3, 64, 64)) * 0.8
    got = ops.to_uint8(t.permute(1, 2, 0).contiguous().cuda()).cpu().numpy()
    assert np.array_equal(got, golden("tensor2img.npz")["img"])                  # reference tensor2img output
    edge = torch.tensor([-2.0, -1.0, -0.996078431, 0.0, 0.00392156862, 1.0, 3.0]).cuda()
    assert np.array_equal(ops.to_uint8(edge).cpu().numpy(), O.tensor2img(edge.cpu().view(1, 1, -1).repeat(3, 1, 1))[0, :, 0])


# ---------------------------------------------------------------------------------------
# A12 VQ
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,key,D,scale", [("m256", "quantize_motion", 32, 0.25), ("m1024", "quantize_motion", 32, 1.0),
                                             ("a512", "quantize_app", 256, 0.5), ("a1024", "quantize_app", 256, None)])
def test_vq_bit_exact_indices(ops, tag, key, D, scale):
    """indices must equal the REFERENCE's (golden fixture) -- N(0,1) codebooks keep top-2 gaps wide."""
    Pg = weights("network_g")
    g = golden("vq.npz")
    z = synth_input(f"vq_{tag}", (2, D, 32, 32))
    cb = Pg[f"{key}.embedding.weight"]
    Ks = cb.shape[0] if scale is None else int(scale * cb.shape[0])
    zt = z.permute(0, 2, 3, 1).reshape(-1, D).contiguous()
    idx, zq, dmin, sq = ops.vq_nearest(zt.cuda(), cb.cuda(), Ks)
    assert np.array_equal(idx.cpu().numpy().reshape(-1, 1), g[f"{tag}_indices"])
    r = O.vector_quantizer(z, cb, scale)
    assert maxabs(zq.cpu(), r["z_q"].permute(0, 2, 3, 1).reshape(-1, D)) == 0.0   # z + (e - z), bit exact
    assert maxabs(dmin.cpu(), r["d"].min(1).values) < 1e-3 * float(r["d"].min(1).values.abs().max())
    loss = 1.25 * float(sq) / zt.numel()
    assert abs(loss - float(g[f"{tag}_loss"])) < 1e-4 * abs(float(g[f"{tag}_loss"]))


def test_vq_ties_and_ragged(ops):
    """first-minimum rule on exact ties; N not a multiple of the 128-token block; Ks not a multiple of 32."""
    cb = rnd("tie_cb", (100, 32))
    cb[57] = cb[13]
    cb[90] = cb[13]
    z = cb[[13, 57, 90, 5, 99]].clone() + 1e-3
    z = torch.cat([z, rnd("tie_z", (196, 32))])
    d = (z ** 2).sum(1, keepdim=True) + (cb ** 2).sum(1) - 2 * z @ cb.t()
    idx, zq, _, _ = ops.vq_nearest(z.cuda(), cb.cuda(), 100)
    got = idx.cpu()
    assert got[:3].tolist() == [13, 13, 13]
    ref = d.argmin(1)
    bad = got != ref
    if bad.any():   # tie-aware: a differing index must be distance-equivalent within fp32 noise
        assert float((d[bad, got[bad]] - d[bad, ref[bad]]).abs().max()) < 1e-4


@pytest.mark.parametrize("N,D,Ks", [(1024, 256, 1024), (4096, 256, 1000), (1000, 32, 1024), (4133, 64, 300), (16384, 128, 256), (130, 32, 513)])
def test_vq_split_sweep_equals_one_sweep(ops, tuning, N, D, Ks):
    """few tokens: the codebook sweep is split over blockIdx.y and a combine kernel folds the shares (knob vq_split).  Same distances, same
    tie rule: indices, minimum distances and z_q must equal the one-sweep kernel's bit for bit (ragged N, Ks not a multiple of 32, a
    codebook with exact duplicates in different shares); the loss differs only by its summation order."""
    cb = rnd(f"vqs_cb{D}", (1024, D))
    if Ks > 70:
        cb[Ks - 3] = cb[5]
        cb[Ks // 2] = cb[5]
    z = rnd(f"vqs_z{N}_{D}", (N, D))
    z[:7] = cb[5] + 1e-3
    tuning("vq_split", 0)
    a = ops.vq_nearest(z.cuda(), cb.cuda(), Ks)
    tuning("vq_split", 1)
    b = ops.vq_nearest(z.cuda(), cb.cuda(), Ks)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert abs(float(a[3]) - float(b[3])) <= 2e-6 * abs(float(a[3]))
    if Ks > 70:
        assert b[0][:7].tolist() == [5] * 7


@pytest.mark.parametrize("B,C,H,W,nw", [(2, 64, 32, 32, 1), (2, 64, 32, 32, 2), (3, 128, 16, 32, 2), (1, 96, 8, 16, 1)])
def test_conv_sft_epilogue_equals_conv_then_sft_combine(ops, B, C, H, W, nw, tuning):
    """Fuse_sft_block's `dec + w * (dec * scale + shift)` as the epilogue of the shift branch's 3x3 conv
    (smx_winograd_conv3x3_sft_f32), on both Winograd block shapes and with the operands being channel slices of
    wider buffers (ss = [scale.0 | shift.0], cat = [enc | dec]) as in the engine -- against F.conv2d + the formula."""
    tuning("wino_nw", nw)
    tuning("wino_wide", 1)
    ss = rnd(f"cs{C}{H}", (B, H, W, 2 * C))
    cat = rnd(f"cc{C}{H}", (B, H, W, 2 * C))
    scale = rnd(f"cq{C}{H}", (B, H, W, C))
    w = rnd(f"cw{C}", (C, C, 3, 3), 1.0 / math.sqrt(9 * C))
    b = rnd(f"cb{C}", (C,), 0.1)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    ssd, catd = ss.cuda(), cat.cuda()
    y = ops.conv_sft(ssd[..., C:], cv, catd[..., C:], scale.cuda(), 0.7)
    shift = F.conv2d(ss[..., C:].permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1)
    dec = cat[..., C:]
    assert maxabs(y.cpu(), dec + 0.7 * (dec * scale + shift)) < 2e-5
    # and it is the same value the unfused pair produces (identical conv, one rounding of the modulation)
    ref = ops.sft_combine(catd[..., C:], scale.cuda(), ops.conv(ssd[..., C:], cv), 0.7)
    assert maxabs(y, ref) < 2e-6


# ---------------------------------------------------------------------------------------
# Size-independent properties AT THE BENCHMARK'S SIZES (B = 60 frames of 256x256: the oracle / F.conv2d on CPU are too slow
# there, and these are the launches that pick the big-block kernel variants by themselves)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("Cin,Cout,s", [(64, 64, 256), (128, 128, 128), (256, 512, 32)])
def test_full_size_conv_linearity_and_locality(ops, Cin, Cout, s):
    """B = 60 launches of the fused Winograd kernel (wide blocks selected by the launcher itself):
    conv(a x + y) == a conv(x) + conv(y) - bias-free part (linearity), a zero input gives exactly the bias, and an
    impulse in one frame leaves every other frame untouched (frames never mix)."""
    B = 60
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((B, s, s, Cin), device="cuda", generator=g)
    y = torch.randn((B, s, s, Cin), device="cuda", generator=g)
    cv = ops.Conv.from_torch(torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) / (3 * Cin ** 0.5), torch.randn(Cout, device="cuda", generator=g) * 0.1)
    with ops.profile() as rec:
        cx = ops.conv(x, cv)
    assert [r[0] for r in rec.rows] == ["gemm_conv"] and rec.rows[0][1].get("wino") == 1
    cy, cz = ops.conv(y, cv), ops.conv(torch.zeros_like(x), cv)
    assert float((cz - cv.b.view(1, 1, 1, -1)).abs().max()) == 0.0                      # exact: every product is 0
    lin = ops.conv(2.0 * x + y, cv)
    err = float((lin - (2.0 * (cx - cz) + (cy - cz) + cz)).abs().max())
    assert err < 5e-5 * max(1.0, float(cx.abs().max())), err
    imp = torch.zeros_like(x)
    imp[17, s // 2, s // 2, 3] = 1.0
    ci = ops.conv(imp, cv) - cz
    assert float(ci[:17].abs().max()) == 0.0 and float(ci[18:].abs().max()) == 0.0      # other frames: bias only
    lit = ci[17].abs().amax(-1) > 0
    assert int(lit.sum()) == 9 and bool(lit[s // 2 - 1:s // 2 + 2, s // 2 - 1:s // 2 + 2].all())   # exactly the 3x3 support


def test_full_size_attention_and_vq_invariants(ops):
    """B = 60: attention with all keys equal returns V's mean regardless of the queries (d_head 4 MFMA kernel and d_head 32);
    a one-hot-dominant key returns its value row; VQ of codebook rows returns their own indices (idempotence) at N = 61,440."""
    B, H, N = 60, 8, 1024
    g = torch.Generator(device="cuda").manual_seed(11)
    for dh in (4, 32):
        E = H * dh
        q = torch.randn((B, N, E), device="cuda", generator=g)
        k = torch.zeros((B, 256, E), device="cuda")
        v = torch.randn((B, 256, E), device="cuda", generator=g)
        o = ops.attention(q, k, v, H, dh, 256)
        assert float((o - v.mean(1, keepdim=True)).abs().max()) < 2e-6
        k2 = torch.zeros((B, 256, E), device="cuda")
        k2[:, 77] = 100.0                                                         # score of query 0 (all ones) with key 77: 100 * sqrt(d_head) >= 200
        q[:, 0] = 1.0
        o2 = ops.attention(q, k2, v, H, dh, 256)
        assert float((o2[:, 0] - v[:, 77]).abs().max()) < 1e-3
    cb = torch.randn((1024, 256), device="cuda", generator=g)
    idx = torch.randint(0, 1024, (B * 1024,), device="cuda", generator=g)
    got, zq, dmin, _ = ops.vq_nearest(cb[idx].contiguous(), cb, 1024)
    assert torch.equal(got.view(-1), idx) and torch.equal(zq, cb[idx]) and float(dmin.abs().max()) < 1e-3


def test_vq_loss_is_bit_reproducible_and_full_size(ops):
    """round-3: sum (z_q - z)^2 leaves the kernel as per-block partials summed in a fixed order (no atomics): two runs on
    the same inputs give the SAME bits; at the bench's size (N = 245,760 tokens) the result still equals the fp64 sum of the
    gathered rows to fp32 accuracy, and every index is the reference expression's argmin (tie-aware)."""
    for D, K, N in ((256, 1024, 60 * 4 * 1024), (32, 768, 4096 + 37)):
        z = rnd(f"vqrep_z{D}", (N, D)).cuda()
        cb = rnd(f"vqrep_cb{D}", (K, D)).cuda()
        a = ops.vq_nearest(z, cb, K)
        b = ops.vq_nearest(z, cb, K)
        assert torch.equal(a[3], b[3]) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        e = cb[a[0]]
        ref = float(((e.double() - z.double()) ** 2).sum())
        assert abs(float(a[3]) - ref) < 2e-6 * ref
        assert torch.equal(a[1], z + (e - z))
        sub = slice(0, 4096)
        d = (z[sub] ** 2).sum(1, keepdim=True) + (cb ** 2).sum(1) - 2 * z[sub] @ cb.t()
        best = d.min(1).values
        got = d.gather(1, a[0][sub].view(-1, 1)).view(-1)
        assert float((got - best).max()) < 1e-3


def test_fingerprint_sees_single_bit_changes(ops):
    """the source-cache key hashes raw bit patterns -- equal contents in another buffer give the same key, a 1-ulp
    change of one element (far below the resolution of an fp32 sum over 196k values) gives a different one."""
    x = rnd("fp_src", (1, 3, 256, 256)).cuda()
    k0 = ops.fingerprint(x)
    assert ops.fingerprint(x.clone()) == k0
    y = x.clone()
    y.view(-1)[123457] = torch.nextafter(y.view(-1)[123457], torch.tensor(10.0, device="cuda"))
    assert ops.fingerprint(y) != k0
    y2 = x.clone()
    y2.view(-1)[[5, 6]] = y2.view(-1)[[6, 5]]            # a permutation of two unequal elements changes it too
    assert ops.fingerprint(y2) != k0


W43_CASES = [(2, 64, 64, 32, 64, 0, True), (1, 128, 128, 64, 64, 3, False), (2, 256, 128, 16, 32, 1, True), (1, 32, 96, 32, 32, 0, False),
             (3, 32, 32, 16, 32, 2, True), (1, 64, 64, 256, 256, 0, True)]


@pytest.mark.parametrize("B,Cin,Cout,H,W,act,with_res", W43_CASES, ids=[f"B{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}_act{c[5]}_res{int(c[6])}" for c in W43_CASES])
def test_winograd43_conv3x3(ops, B, Cin, Cout, H, W, act, with_res, monkeypatch):
    """round 3: fused Winograd F(4x4,3x3) (csrc/winograd43.hip) == F.conv2d (3x3, s1, p1) at 1e-4 (transform coefficients up to 8: the
    error budget the DESIGN quotes), through channel-slice operands, with bias / activation / residual, border blocks, one block per
    image and many; forced on small inputs (the launcher only picks it for >= 512 blocks)."""
    monkeypatch.setattr(ops, "WINO43_MIN_BLOCKS", 1)
    monkeypatch.setattr(ops, "WINO43", 1)
    tag = f"{B}{Cin}{Cout}{H}{W}"
    x = rnd(f"w43x{tag}", (B, Cin, H, W))
    w = rnd(f"w43w{tag}", (Cout, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd(f"w43b{tag}", (Cout,), 0.1)
    ref = F.conv2d(x, w, b, padding=1)
    ref = {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, 0.2), 3: O.swish}[act](ref)
    r = rnd(f"w43r{tag}", tuple(ref.shape)) if with_res else None
    if with_res:
        ref = ref + r
    xin = torch.zeros((B, H, W, Cin + 32), device="cuda")
    xin[..., 32:] = nhwc(x)
    out = torch.full((B, H, W, Cout + 8), 5.0, device="cuda")
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    with ops.profile() as rec:
        ops.conv(xin[..., 32:], cv, out=out[..., 4:4 + Cout], act=act, res=None if r is None else nhwc(r))
    assert rec.rows[0][1].get("w43") == 1                              # the F(4x4,3x3) kernel ran, not a fallback
    assert maxabs(nchw(out[..., 4:4 + Cout]), ref) < 1e-4
    assert float(out[..., :4].min()) == 5.0 and float(out[..., 4 + Cout:].max()) == 5.0
    monkeypatch.setattr(ops, "WINO43", 0)
    old = ops.conv(xin[..., 32:], cv, act=act, res=None if r is None else nhwc(r))
    assert maxabs(nchw(old), ref) < 5e-5


def test_winograd43_fused_groupnorm_loader_and_stats(ops, monkeypatch):
    """the F(4x4,3x3) kernel with the producing GroupNorm(+swish) folded into its staging pass and the Welford partials of its own
    output for the next GroupNorm (the ResBlock form of the big launches) == group_norm -> swish -> conv2d -> group_norm statistics."""
    monkeypatch.setattr(ops, "WINO43_MIN_BLOCKS", 1)
    monkeypatch.setattr(ops, "WINO43", 1)
    for (B, C, Co, H, W, sw) in ((2, 64, 64, 32, 64, True), (1, 128, 64, 64, 64, True), (2, 256, 128, 16, 32, False)):
        x = rnd(f"g43x{C}{H}", (B, C, H, W)) * 1.5 + 0.2
        g, bt = 1 + 0.1 * rnd(f"g43g{C}", (C,)), 0.1 * rnd(f"g43b{C}", (C,))
        w = rnd(f"g43w{C}{Co}", (Co, C, 3, 3), 1.0 / math.sqrt(9 * C))
        b = rnd(f"g43bb{Co}", (Co,), 0.1) + 2.0                        # a large common mean: the partials must not cancel
        hn = F.group_norm(x, 32, g, bt, 1e-6)
        ref = F.conv2d(O.swish(hn) if sw else hn, w, b, padding=1)
        xin = nhwc(x)
        ss = ops.groupnorm_stats(xin, g.cuda(), bt.cuda())
        with ops.profile() as rec:
            y = ops.conv(xin, ops.Conv.from_torch(w.cuda(), b.cuda()), in_ss=ss, in_swish=sw, want_stats=True)
        assert [r for r in rec.rows if r[0] == "gemm_conv"][0][1].get("w43") == 1
        err = maxabs(nchw(y), ref)
        assert err < 1e-4, (C, Co, H, W, err)
        assert y._gn_part is not None and tuple(y._gn_part.shape) == (B, (H // 16) * (W // 32), Co, 2)
        g2, b2 = 1 + 0.1 * rnd(f"g43g2{Co}", (Co,)), 0.1 * rnd(f"g43b2{Co}", (Co,))
        ss_fused = ops.groupnorm_stats(y, g2.cuda(), b2.cuda())          # finalize over the epilogue's partials
        y_plain = y.clone()                                               # no partials attached: the statistics pass reads the tensor
        ss_plain = ops.groupnorm_stats(y_plain, g2.cuda(), b2.cuda())
        d = maxabs(ss_fused.cpu(), ss_plain.cpu())
        assert d < 2e-5 * float(ss_plain.abs().max()), (C, Co, H, W, d)
