"""BASELINE configs[2] (bf16 storage / bf16 MFMA, fp32 accumulate) on a real MI355X.

Kernel level: every bf16 entry point against the SAME op evaluated in fp32 on the bf16-rounded operands, so the only
differences are accumulation order and the final round-to-bf16 of the stored result (tolerances below are written in
units of that rounding: one bf16 ulp = 2^-8 relative).  Pipeline level: the bf16 path against the reference's fp32
output, judged by the error the REFERENCE ITSELF makes under CPU autocast(bfloat16) on the same inputs (fixture
tests/golden/autocast_bf16.npz, produced by tests/golden/make_golden_r2.py) -- SURVEY 8(d) config 3: 1e-3 is not claimable
in bf16; "same as reference-under-autocast" is."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import reenact_oracle as O
from synergize_motion_appearance_amd.synth import synth_input
from tests.util import maxabs, weights, golden, clip

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "needs an MI355X"
    from synergize_motion_appearance_amd import ops as _ops
    from synergize_motion_appearance_amd import lib
    lib.load()
    return _ops


def rnd(name, shape, scale=1.0):
    return synth_input(name, shape) * scale


def r16(t):
    """round to bf16 and back: the value a bf16-stored operand really has."""
    return t.to(BF).float()


def nhwc16(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda().to(BF)


def nchw32(t):
    return t.float().permute(0, 3, 1, 2).contiguous().cpu()


def close16(got, ref, ulps=1.0, floor=1e-3):
    """|got - ref| <= ulps * |ref| * 2^-8 (+ a small absolute floor for values near zero).  |ref| * 2^-8 is the worst-case
    round-to-nearest error of storing ref in bf16 (half a unit in the last of 8 significand bits), so ulps=1.0 with a tiny
    floor means "the stored value is the correctly rounded fp32 result"."""
    got, ref = torch.as_tensor(got).double(), torch.as_tensor(ref).double()
    tol = ulps * ref.abs() * 2.0 ** -8 + floor
    bad = (got - ref).abs() > tol
    return not bool(bad.any()), float(((got - ref).abs() / tol).max())


# ---------------------------------------------------------------------------------------
# MFMA fragment layout first: A = I against an ASYMMETRIC B catches a swapped row/col or a wrong k-half mapping
# ---------------------------------------------------------------------------------------
def test_bf16_mfma_fragment_layout_identity_times_asymmetric(ops):
    M = N = K = 64
    a = torch.eye(M, K)
    bt = r16(torch.arange(N)[:, None] * 3.0 + torch.arange(K)[None, :] * 0.25 - 7.0)    # Bt[n][k], asymmetric (bf16-representable values)
    c = torch.empty((M, N), device="cuda", dtype=torch.float32)
    ops.gemm_nt(a.cuda().to(BF), bt.cuda().to(BF), c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
    assert maxabs(c.cpu(), a @ bt.t()) == 0.0


GEMM16_CASES = [
    # (B, Cin, Cout, H, k, stride, pad, out_hw, up2, act, tile, a_f32, c_f32, tag)
    (2, 64, 64, 32, 3, 1, None, None, False, 0, 0, False, False, "3x3 64->64"),
    (1, 128, 128, 32, 3, 1, None, None, False, 3, 1, False, False, "3x3 128->128 swish tile1"),
    (1, 64, 64, 32, 3, 1, None, None, False, 0, 2, False, True, "tile2 fp32 out"),
    (1, 64, 64, 32, 3, 1, None, None, False, 1, 3, False, False, "tile3 relu"),
    (2, 64, 32, 16, 3, 1, None, None, False, 0, 4, False, False, "tile4 N=32"),
    (1, 64, 128, 32, 3, 1, None, None, False, 0, 5, False, False, "tile5"),
    (1, 64, 64, 32, 3, 1, None, None, False, 0, 6, False, False, "tile6"),
    (2, 128, 128, 32, 3, 1, None, None, False, 1, 7, False, False, "tile7 128x128 / 4 waves relu"),
    (2, 256, 256, 32, 1, 1, None, None, False, 4, 7, False, True, "tile7 1x1 short K gelu fp32 out"),
    (2, 32, 32, 64, 3, 2, (0, 0), (32, 32), False, 0, 0, False, False, "stride 2 pad(0,1,0,1), Cin=32 (chunks inside a 64-slice)"),
    (2, 64, 64, 16, 3, 1, None, None, True, 0, 0, False, False, "nearest x2 folded"),
    (2, 256, 192, 32, 1, 1, None, None, False, 1, 0, False, False, "1x1 256->192 relu"),
    (2, 160, 126, 32, 3, 1, None, None, False, 1, 0, False, False, "3x3 160->126 odd Cout, Cin%64!=0"),
    (2, 3, 64, 32, 3, 1, None, None, False, 0, 0, True, False, "3x3 3->64 generic path, fp32 input"),
    (2, 2, 128, 24, 7, 1, None, None, False, 1, 0, True, False, "7x7 2->128 fp32 input"),
    (2, 15, 32, 32, 1, 1, None, None, False, 1, 0, True, False, "1x1 15->32 fp32 input"),
    (2, 64, 64, 32, 3, 1, None, None, False, 0, 0, True, True, "fp32 in / fp32 out on the bf16 MFMA (hourglass form)"),
    (1, 1024, 512, 4, 3, 1, None, None, True, 1, 0, True, True, "hourglass deep up2, split-K"),
    (1, 128, 128, 32, 2, 2, (0, 0), None, False, 0, 0, False, False, "patchify 2x2 stride 2"),
    (1, 256, 512, 32, 3, 1, None, None, False, 4, 0, False, False, "3x3 256->512 gelu"),
]


@pytest.mark.parametrize("case", GEMM16_CASES, ids=[c[-1] for c in GEMM16_CASES])
def test_gemm_conv_bf16(ops, case):
    B, Cin, Cout, H, k, stride, pad, out_hw, up2, act, tile, a_f32, c_f32, tag = case
    x = rnd("bx" + tag, (B, Cin, H, H))
    w = rnd("bw" + tag, (Cout, Cin, k, k), 1.0 / math.sqrt(Cin * k * k))
    b = rnd("bb" + tag, (Cout,), 0.1)
    xq, wq = r16(x), r16(w)                                   # what the MFMA sees (fp32 inputs are rounded while staging)
    xe = F.interpolate(xq, scale_factor=2.0, mode="nearest") if up2 else xq
    if pad is None:
        ref = F.conv2d(xe.double(), wq.double(), b.double(), stride=stride, padding=k // 2)
    elif out_hw is not None:
        need = (out_hw[0] - 1) * stride + k - xe.shape[2] - pad[0]
        ref = F.conv2d(F.pad(xe, (pad[1], max(need, 0), pad[0], max(need, 0))).double(), wq.double(), b.double(), stride=stride)[:, :, :out_hw[0], :out_hw[1]]
    else:
        ref = F.conv2d(xe.double(), wq.double(), b.double(), stride=stride, padding=pad)
    ref = {0: lambda t: t, 1: F.relu, 3: O.swish, 4: F.gelu}[act](ref).float()
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    xin = xin if a_f32 else xin.to(BF)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    with ops.profile() as rec:
        y = ops.conv(xin, cv, stride=stride, pad=pad, out_hw=out_hw, up2=up2, act=act, tile=tile, mfma16=True,
                     out_dtype=torch.float32 if c_f32 else None)
    names = [r[0] for r in rec.rows]
    assert names in (["gemm_bf16"], ["conv3x3_bf16"], ["conv3x3_mfma16"]) and (tile == 0 or names == ["gemm_bf16"])   # a forced tile always means the implicit GEMM
    if a_f32 and c_f32 and k == 3 and stride == 1 and Cin % 64 == 0 and H % 8 == 0 and tile == 0:
        assert names == ["conv3x3_mfma16"], names              # fp32 storage on both sides: the region-direct kernel's F32 form
    assert y.dtype == (torch.float32 if c_f32 else BF)
    if c_f32:
        assert maxabs(nchw32(y), ref) < 2e-4 * max(1.0, float(ref.abs().max())), tag
    else:
        ok, worst = close16(nchw32(y), ref, ulps=1.0)
        assert ok, (tag, worst)


def test_gemm_conv_bf16_fused_groupnorm_residual_slices_and_d2s(ops):
    """in_ss (GroupNorm + swish while staging), bf16 / fp32 residuals, channel-slice operands, un-patchify store."""
    B, C, Co, H = 2, 64, 64, 32
    x = r16(rnd("g16x", (B, C, H, H)) * 1.5 + 0.2)
    g, bt = 1 + 0.1 * rnd("g16g", (C,)), 0.1 * rnd("g16b", (C,))
    w = rnd("g16w", (Co, C, 3, 3), 1.0 / math.sqrt(9 * C))
    b = rnd("g16bb", (Co,), 0.1)
    res = r16(rnd("g16r", (B, Co, H, H)))
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    wide = torch.zeros((B, H, H, C + 16), device="cuda", dtype=BF)
    wide[..., 8:8 + C] = nhwc16(x)
    xin = wide[..., 8:8 + C]
    ss = ops.groupnorm_stats(xin, g.cuda(), bt.cuda())
    ss_ref = ops.groupnorm_stats(x.permute(0, 2, 3, 1).contiguous().cuda(), g.cuda(), bt.cuda())
    assert maxabs(ss.cpu(), ss_ref.cpu()) < 1e-5                                  # same fp32 statistics from bf16 storage
    hn = O.swish(F.group_norm(x, 32, g, bt, 1e-6))
    ref = F.conv2d(r16(hn).double(), r16(w).double(), b.double(), padding=1).float() + res
    outw = torch.full((B, H, H, Co + 24), 5.0, device="cuda", dtype=BF)
    for rt in (nhwc16(res), res.permute(0, 2, 3, 1).contiguous().cuda()):         # bf16 and fp32 residual
        ops.conv(xin, cv, out=outw[..., 8:8 + Co], in_ss=ss, in_swish=True, res=rt)
        # the staged value is swish(GN(x)) rounded to bf16 from an fp32 evaluation: allow 2 ulp end to end
        ok, worst = close16(nchw32(outw[..., 8:8 + Co]), ref, ulps=2.0, floor=4e-3)
        assert ok, worst
        assert float(outw[..., :8].float().min()) == 5.0 and float(outw[..., 8 + Co:].float().max()) == 5.0
    # un-patchify (depth-to-space) store: Linear(256 -> p*p*C) + rearrange == 1x1 conv with d2s
    p_, Cc = 4, 64
    t = r16(rnd("d2sx", (1, 256, 32, 32)))
    wl = rnd("d2sw", (p_ * p_ * Cc, 256, 1, 1), 1.0 / 16)
    bl = rnd("d2sb", (p_ * p_ * Cc,), 0.1)
    lin = F.conv2d(t.double(), r16(wl).double(), bl.double()).float()             # [1, p*p*C, 32, 32], n = (p1*p+p2)*C + c
    ref2 = lin.view(1, p_, p_, Cc, 32, 32).permute(0, 3, 4, 1, 5, 2).reshape(1, Cc, 32 * p_, 32 * p_)
    y = ops.conv(nhwc16(t), ops.Conv.from_torch(wl.cuda(), bl.cuda()), d2s=(p_, Cc))
    ok, worst = close16(nchw32(y), ref2, ulps=1.0)
    assert ok, worst


R3_CASES = [(2, 64, 64, 32, 32, False, 0, None), (1, 128, 128, 64, 32, False, 3, "bf16"), (2, 256, 128, 16, 16, False, 1, "f32"),
            (2, 64, 64, 16, 16, True, 0, "bf16"), (1, 128, 96, 32, 48, False, 0, None), (1, 512, 256, 32, 32, False, 4, None),
            (3, 64, 3 * 64, 16, 32, False, 0, None)]


# the 16x32-tile form (csrc/conv3x3_bf16_t32.hip): W % 32 == 0, C_in % 16 == 0 (48 and 160 are not multiples of 64), ragged C_out (96, 126),
# 8 slices (C_in 128) and an odd slice count (C_in 48: the two-stage loop ends on stage 0), x2 upsampling folded into the loader
T32_CASES = [(2, 64, 64, 32, 32, False, 0, None), (1, 128, 128, 64, 32, False, 3, "bf16"), (2, 256, 128, 16, 32, False, 1, "f32"),
             (2, 64, 64, 32, 64, True, 0, "bf16"), (1, 128, 96, 32, 64, False, 0, None), (1, 512, 256, 32, 32, False, 4, None),
             (3, 64, 3 * 64, 16, 32, False, 0, None), (2, 48, 64, 32, 32, False, 2, "bf16"), (1, 160, 126, 16, 64, False, 0, None)]


def _case_id(c):
    return f"B{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}_up{int(c[5])}_act{c[6]}_res{c[7]}"


@pytest.mark.parametrize("case", T32_CASES, ids=[_case_id(c) for c in T32_CASES])
def test_conv3x3_region_direct_bf16_t32(ops, case, monkeypatch):
    """the big-launch form of the region-direct kernel (16x32-pixel tiles, 16-channel slices through a double-buffered LDS stage, weights by
    LDS-DMA from the fragment-ordered pack, A fragments shared by the three ky taps): same checks as the 16x16 / 8x16 forms below."""
    _region_direct_case(ops, case, 32, monkeypatch)


@pytest.mark.parametrize("tile_h", [16, 116, 8], ids=["16x16_slab", "16x16_per_tap", "8x16"])
@pytest.mark.parametrize("case", R3_CASES, ids=[_case_id(c) for c in R3_CASES])
def test_conv3x3_region_direct_bf16(ops, case, tile_h, monkeypatch):
    """the region-direct 3x3 kernel (input region staged once per 64-channel slice, nine taps read from LDS) against conv2d on
    the bf16-rounded operands, against the implicit-GEMM bf16 kernel on the same operands, through channel-slice views, with
    the fused GroupNorm+swish loader, and with the Welford partials it emits for the next GroupNorm."""
    _region_direct_case(ops, case, tile_h, monkeypatch)


def _region_direct_case(ops, case, tile_h, monkeypatch):
    B, Cin, Cout, H, W, up2, act, resk = case
    t32 = tile_h == 32
    monkeypatch.setattr(ops, "CONV16_T32", int(t32))
    monkeypatch.setattr(ops, "CONV16_T32_MIN_BLOCKS", 1)
    tw = 32 if t32 else 16
    if t32:
        tile_h = 16
    ops.set_tuning("conv16_slab", 0 if tile_h > 100 else 1)     # 16x16 tiles: all nine taps' weights of a 32-channel slice in LDS | one tile per tap
    tile_h %= 100
    monkeypatch.setattr(ops, "CONV16_TILE_H", tile_h)
    x = r16(rnd(f"r3x{case}", (B, Cin, H // (2 if up2 else 1), W // (2 if up2 else 1))))
    w = rnd(f"r3w{case}", (Cout, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd(f"r3b{case}", (Cout,), 0.1)
    xe = F.interpolate(x, scale_factor=2.0, mode="nearest") if up2 else x
    ref = F.conv2d(xe.double(), r16(w).double(), b.double(), padding=1)
    ref = {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, 0.2), 3: O.swish, 4: F.gelu}[act](ref).float()
    res = r16(rnd(f"r3r{case}", tuple(ref.shape))) if resk else None
    if res is not None:
        ref = ref + res
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    xin = torch.zeros((B, x.shape[2], x.shape[3], Cin + 16), device="cuda", dtype=BF)
    xin[..., 8:8 + Cin] = nhwc16(x)
    out = torch.full((B, H, W, Cout + 16), 5.0, device="cuda", dtype=BF)
    rt = None if res is None else (nhwc16(res) if resk == "bf16" else res.permute(0, 2, 3, 1).contiguous().cuda())
    with ops.profile() as rec:
        y = ops.conv(xin[..., 8:8 + Cin], cv, out=out[..., 8:8 + Cout], up2=up2, act=act, res=rt, want_stats=True)
    assert [r[0] for r in rec.rows] == ["conv3x3_bf16"] and bool(rec.rows[0][1].get("t32")) == t32
    ok, worst = close16(nchw32(y), ref, ulps=1.0)
    assert ok, worst
    assert float(out[..., :8].float().min()) == 5.0 and float(out[..., 8 + Cout:].float().max()) == 5.0
    gen = ops.conv(xin[..., 8:8 + Cin], cv, up2=up2, act=act, res=rt, tile=3)          # implicit-GEMM bf16 kernel, same operands
    ok, worst = close16(gen.float().cpu(), y.float().cpu(), ulps=2.0)       # two correctly rounded sums in different k order: <= 2 half-ulps apart
    assert ok, worst
    # Welford partials of the stored tile: {mean, M2} per 16x16 pixels and channel
    part = y._gn_part
    assert part is not None and tuple(part.shape) == (B, (H // tile_h) * (W // tw), Cout, 2)
    yc = y.float().cpu().double()
    blocks = yc.view(B, H // tile_h, tile_h, W // tw, tw, Cout).permute(0, 1, 3, 5, 2, 4).reshape(B, -1, Cout, tile_h * tw)
    bm = blocks.mean(-1)
    assert maxabs(part[..., 0].cpu(), bm) < 5e-6 * max(1.0, float(bm.abs().max()))
    assert maxabs(part[..., 1].cpu(), ((blocks - bm[..., None]) ** 2).sum(-1)) < 1e-3
    if Cout & (Cout - 1) == 0:
        g, bt = (1 + 0.1 * rnd(f"r3g{Cout}", (Cout,))).cuda(), (0.1 * rnd(f"r3bt{Cout}", (Cout,))).cuda()
        dense = y.contiguous()
        assert getattr(dense, "_gn_part", None) is None
        assert maxabs(ops.groupnorm_stats(y, g, bt).cpu(), ops.groupnorm_stats(dense, g, bt).cpu()) < 2e-5      # finalize == two-pass
        # fused loader on the region kernel: GN + swish of THIS tensor while staging the next conv
        ss = ops.groupnorm_stats(y, g, bt)
        w2 = rnd(f"r3w2{case}", (64, Cout, 3, 3), 1.0 / math.sqrt(9 * Cout))
        hn = r16(O.swish(F.group_norm(nchw32(y), 32, g.cpu(), bt.cpu(), 1e-6)))
        ref2 = F.conv2d(hn.double(), r16(w2).double(), None, padding=1).float()
        with ops.profile() as rec2:
            y2 = ops.conv(y, ops.Conv.from_torch(w2.cuda(), None), in_ss=ss, in_swish=True)
        assert bool(rec2.rows[0][1].get("t32")) == t32
        ok, worst = close16(nchw32(y2), ref2, ulps=2.0, floor=4e-3)
        assert ok, worst


def test_gemm_nt_bf16_batched(ops):
    """AttnBlock-style batched products on the bf16 MFMA: Q K^T with an offset Bt view, alpha; P V^T; per-row bias."""
    B, N, C = 2, 256, 64
    qk = r16(rnd("nt16qk", (B, N, 2 * C)))
    s = torch.empty((B, N, N), device="cuda", dtype=BF)
    ops.gemm_nt(qk.cuda().to(BF), qk.cuda().to(BF), s, M=N, N=N, K=C, lda=2 * C, ldb=2 * C, ldc=N, nb0=B, a_bs=(N * 2 * C, 0),
                bt_bs=(N * 2 * C, 0), c_bs=(N * N, 0), bt_off=C, alpha=0.125)
    ref = 0.125 * torch.einsum("bik,bjk->bij", qk[..., :C].double(), qk[..., C:].double()).float()
    ok, worst = close16(s.float().cpu(), ref)
    assert ok, worst
    wv, bv, hn = rnd("nt16w", (C, C), 0.2), rnd("nt16b", (C,), 0.1), r16(rnd("nt16h", (B, N, C)))
    vt = torch.empty((B, C, N), device="cuda", dtype=BF)
    ops.gemm_nt(wv.cuda(), hn.cuda().to(BF), vt, M=C, N=N, K=C, lda=C, ldb=C, ldc=N, nb0=B, bt_bs=(N * C, 0), c_bs=(C * N, 0),
                bias=bv.cuda(), bias_per_row=True)
    ref = torch.einsum("ck,bnk->bcn", r16(wv).double(), hn.double()).float() + bv[None, :, None]
    ok, worst = close16(vt.float().cpu(), ref)
    assert ok, worst


# ---------------------------------------------------------------------------------------
# storage-templated kernels: bf16 variant == fp32 variant on the bf16-rounded operands (+ one output rounding)
# ---------------------------------------------------------------------------------------
def test_warp_bf16_all_scales(ops):
    B = 3
    flow = (O.make_coordinate_grid(64, 64, torch.float32)[None].repeat(B, 1, 1, 1) + 0.3 * rnd("w16f", (B, 64, 64, 2))).cuda()
    occ = torch.sigmoid(rnd("w16o", (B, 64, 64))).cuda()
    for (C, s) in ((256, 32), (128, 64), (128, 128), (64, 256)):
        for Bf in (1, B):
            feat = r16(rnd(f"w16x{C}{s}{Bf}", (Bf, s, s, C))).cuda()
            ref = ops.warp(feat, flow, occ)                                     # fp32 kernel on the same (bf16-valued) features
            got = ops.warp(feat.to(BF), flow, occ)
            assert got.dtype == BF
            ok, worst = close16(got.float().cpu(), ref.cpu(), ulps=1.0, floor=1e-6)   # the stored result is the rounded fp32 result
            assert ok, (C, s, Bf, worst)
    z = ops.warp(r16(rnd("w16z", (1, 64, 64, 128))).cuda().to(BF), flow + 5.0)  # everything out of frame -> exact zeros
    assert float(z.float().abs().max()) == 0.0


def test_elementwise_bf16_variants(ops):
    x = r16(rnd("e16x", (2, 64, 64, 128)) * 1.7 + 0.3).cuda()
    g, b = (1 + 0.1 * rnd("e16g", (128,))).cuda(), (0.1 * rnd("e16b", (128,))).cuda()
    x16 = x.to(BF)
    half = lambda got, ref, what: close16(got.float().cpu(), ref.cpu(), ulps=1.0, floor=1e-6)[0] or pytest.fail(what)   # noqa: E731
    for sw in (False, True):
        half(ops.groupnorm(x16, g, b, swish=sw), ops.groupnorm(x, g, b, swish=sw), f"groupnorm swish={sw}")
    ss = ops.groupnorm_stats(x16, g, b)
    half(ops.groupnorm_apply(x16, ss, swish=True), ops.groupnorm_apply(x, ss, swish=True), "groupnorm_apply")
    half(ops.resize(x16, 32, 32), ops.resize(x, 32, 32), "resize down")
    half(ops.resize(x16[..., 32:96], 32, 32), ops.resize(x[..., 32:96], 32, 32), "resize (channel-slice view)")
    half(ops.resize(x16[:, :32, :32].contiguous(), 64, 64), ops.resize(x[:, :32, :32].contiguous(), 64, 64), "resize up")
    t16, t32 = ops.resize_taps_gather(x16, 16, 16), ops.resize_taps_gather(x, 16, 16)
    assert torch.equal(t16.float(), t32)
    half(ops.resize_taps_combine(t16, 64, 64), ops.resize_taps_combine(t32, 64, 64), "resize_taps_combine")
    y = r16(rnd("e16y", (2, 64, 64, 128))).cuda()
    half(ops.add(x16, y.to(BF)), ops.add(x, y), "add")
    half(ops.sft_combine(x16, y.to(BF), x16, 0.7), ops.sft_combine(x, y, x, 0.7), "sft_combine")
    tok = r16(rnd("e16t", (2, 1024, 256)) * 2 + 0.5).cuda()
    gl, bl, pos = (1 + 0.1 * rnd("e16lg", (256,))).cuda(), (0.1 * rnd("e16lb", (256,))).cuda(), (0.2 * rnd("e16lp", (1024, 256))).cuda()
    a16, ap16 = ops.layernorm(tok.to(BF), gl, bl, pos=pos)
    a32, ap32 = ops.layernorm(tok, gl, bl, pos=pos)
    half(a16, a32, "layernorm")
    half(ap16, ap32, "layernorm + pos")
    s = r16(rnd("e16s", (2, 64, 1024)) * 3).cuda()
    half(ops.softmax_rows(s.to(BF), 1024, 0.5), ops.softmax_rows(s.clone(), 1024, 0.5), "softmax_rows")
    # conversions: fp32 -> bf16 slice -> fp32 round trip, and the NCHW <-> NHWC layout kernels
    buf = torch.zeros((2, 64, 64, 160), device="cuda", dtype=BF)
    ops.copy_slice(x, buf[..., 16:144])
    assert torch.equal(buf[..., 16:144].float(), x) and float(buf[..., :16].float().abs().max()) == 0.0
    back = torch.empty_like(x)
    ops.copy_slice(buf[..., 16:144], back)
    assert torch.equal(back, x)
    img = rnd("e16i", (2, 3, 32, 48)).cuda()
    n16 = ops.nchw_to_nhwc(img, dtype=BF)
    assert n16.dtype == BF and torch.equal(n16.float(), r16(img.cpu()).permute(0, 2, 3, 1).cuda())
    assert torch.equal(ops.nhwc_to_nchw(n16), r16(img.cpu()).cuda())


def test_attention_bf16_storage(ops):
    """bf16 storage on the fp32-MFMA attention kernels (d_head 4 always; d_head 32 with the bf16-MFMA kernel switched off):
    the stored result is the correctly rounded fp32 result."""
    old, old4 = ops.set_tuning("attn16", 0), ops.set_tuning("attn4_mfma", 2)
    try:
        _attention_bf16_storage(ops)
    finally:
        ops.set_tuning("attn16", old)
        ops.set_tuning("attn4_mfma", old4)


def test_attention_d4_bf16_mfma_kernel(ops):
    """d_head 4 on v_mfma_f32_4x4x4_16B_bf16 (attn_mfma4_bf16_kernel: stored bf16 operands straight into the matrix instruction, P rounded
    to bf16 before the PV product): against the fp32 4x4x1 kernel on the same bf16-rounded q / k / v, same error model as the d_head 32
    bf16 kernel (|err| <= 2^-9 max|v| + the output rounding); masked keys, a key count the 32-key step does not divide (falls back to the
    fp32 form: identical result), fully masked rows NaN."""
    B, H, N, dh = 2, 8, 1024, 4
    E = H * dh
    assert ops.set_tuning("attn4_mfma", 1) == 1
    for S, shared in ((1024, False), (512, True), (256, True), (768, True), (320, True)):
        q = r16(rnd(f"d4q{S}", (B, N, E)) * 2.0).cuda()
        kv = r16(rnd(f"d4k{S}", ((1 if shared else B), S, 2 * E)) * 2.0).cuda()
        mask = None
        if not shared:
            mask = torch.zeros((B, S), dtype=torch.uint8)
            mask[1, ::5] = 1
            mask[0, 100:200] = 1
            mask = mask.cuda()
        k32, v32 = (kv[..., :E], kv[..., E:]) if not shared else (kv[0, :, :E], kv[0, :, E:])
        ref = ops.attention(q, k32, v32, H, dh, S, k_shared=shared, mask=mask)
        kv16 = kv.to(BF)
        k16, v16 = (kv16[..., :E], kv16[..., E:]) if not shared else (kv16[0, :, :E], kv16[0, :, E:])
        got = ops.attention(q.to(BF), k16, v16, H, dh, S, k_shared=shared, mask=mask)
        assert got.dtype == BF
        vmax = float(kv[..., E:].abs().max())
        err = float((got.float() - ref).abs().max())
        assert err <= (2.0 ** -9 + 2.0 ** -8) * vmax, (S, shared, err, vmax)
        assert float((got.float() - ref).abs().mean()) < 1e-3 * vmax
        old = ops.set_tuning("attn4_mfma", 2)
        try:
            f32form = ops.attention(q.to(BF), k16, v16, H, dh, S, k_shared=shared, mask=mask)
        finally:
            ops.set_tuning("attn4_mfma", old)
        assert torch.equal(f32form, got) == (S % 128 != 0), S      # the bf16-MFMA form really ran where it applies (and only there)
    mask = torch.zeros((B, 1024), dtype=torch.uint8)
    mask[1] = 1
    q = r16(rnd("d4qn", (B, N, E))).cuda().to(BF)
    kv = r16(rnd("d4kn", (B, 1024, 2 * E))).cuda().to(BF)
    o = ops.attention(q, kv[..., :E], kv[..., E:], H, dh, 1024, mask=mask.cuda())
    assert bool(torch.isnan(o[1].float()).all()) and bool(torch.isfinite(o[0].float()).all())


def test_attention_bf16_mfma_kernel(ops):
    """d_head 32 on v_mfma_f32_32x32x16_bf16 (P = exp2(S - m) rounded to bf16 before the PV product, like every bf16 flash
    attention): against the fp32-MFMA kernel on the same bf16-rounded q / k / v.  Error model: sum_j p_j d_j v_j with
    |d_j| <= 2^-9 and sum p_j = 1  =>  |err| <= 2^-9 max|v| (+ the output rounding); fully masked rows NaN in both."""
    B, H, N, dh = 2, 8, 1024, 32
    E = H * dh
    for S, shared in ((1024, False), (512, True), (256, True), (768, True)):
        q = r16(rnd(f"m16q{S}", (B, N, E))).cuda()
        kv = r16(rnd(f"m16k{S}", ((1 if shared else B), S, 2 * E))).cuda()
        mask = None
        if not shared:
            mask = torch.zeros((B, S), dtype=torch.uint8)
            mask[1, ::5] = 1
            mask[0, 100:200] = 1
            mask = mask.cuda()
        k32, v32 = (kv[..., :E], kv[..., E:]) if not shared else (kv[0, :, :E], kv[0, :, E:])
        ref = ops.attention(q, k32, v32, H, dh, S, k_shared=shared, mask=mask)
        kv16 = kv.to(BF)
        k16, v16 = (kv16[..., :E], kv16[..., E:]) if not shared else (kv16[0, :, :E], kv16[0, :, E:])
        with ops.profile() as rec:
            got = ops.attention(q.to(BF), k16, v16, H, dh, S, k_shared=shared, mask=mask)
        assert got.dtype == BF
        vmax = float(kv[..., E:].abs().max())
        err = float((got.float() - ref).abs().max())
        assert err <= (2.0 ** -9 + 2.0 ** -8) * vmax, (S, shared, err, vmax)
        assert float((got.float() - ref).abs().mean()) < 1e-3 * vmax
    # fully masked row -> NaN like the reference (and like the fp32 kernel)
    mask = torch.zeros((B, 1024), dtype=torch.uint8)
    mask[1] = 1
    q = r16(rnd("m16qn", (B, N, E))).cuda().to(BF)
    kv = r16(rnd("m16kn", (B, 1024, 2 * E))).cuda().to(BF)
    o = ops.attention(q, kv[..., :E], kv[..., E:], H, dh, 1024, mask=mask.cuda())
    assert bool(torch.isnan(o[1].float()).all()) and bool(torch.isfinite(o[0].float()).all())


def _attention_bf16_storage(ops):
    B, H, N = 2, 8, 1024
    for dh, S, shared in ((32, 1024, False), (32, 512, True), (4, 1024, False), (4, 256, True)):
        E = H * dh
        q = r16(rnd(f"a16q{dh}{S}", (B, N, E))).cuda()
        kv = r16(rnd(f"a16k{dh}{S}", ((1 if shared else B), S, 2 * E))).cuda()
        mask = None
        if not shared:
            mask = torch.zeros((B, S), dtype=torch.uint8)
            mask[1, ::5] = 1
            mask = mask.cuda()
        k32, v32 = kv[..., :E], kv[..., E:]
        ref = ops.attention(q, k32 if not shared else k32[0], v32 if not shared else v32[0], H, dh, S, k_shared=shared, mask=mask)
        kv16 = kv.to(BF)
        k16, v16 = kv16[..., :E], kv16[..., E:]
        got = ops.attention(q.to(BF), k16 if not shared else k16[0], v16 if not shared else v16[0], H, dh, S, k_shared=shared, mask=mask)
        assert got.dtype == BF
        ok, worst = close16(got.float().cpu(), ref.cpu(), ulps=1.0, floor=1e-6)
        assert ok, (dh, S, shared, worst)


@pytest.mark.parametrize("B,K,N,act,with_res,sliced", [(8, 256, 256, 0, True, False), (4, 256, 512, 4, False, True), (8, 128, 128, 0, False, False),
                                                      (4, 128, 256, 1, True, True), (5, 256, 128, 0, True, False)])
def test_row_panel_gemm_bf16(ops, B, K, N, act, with_res, sliced):
    """csrc/gemm_rp_bf16.hip (persistent row-panel kernel for the K = 128 / 256 1x1 layers) == the fp32 product of the bf16-rounded operands,
    bias / activation / residual in fp32, ONE rounding at the store; on channel-slice views (ld > C) too; and it is the kernel that ran."""
    H = W = 64
    rows = ops.GEMM16_RP_MIN_ROWS
    ops.GEMM16_RP_MIN_ROWS = 1024
    try:
        xw = r16(rnd(f"rpx{K}{N}", (B, H, W, K + (24 if sliced else 0)))).cuda().to(BF)
        x = xw[..., 8:8 + K] if sliced else xw
        w = rnd(f"rpw{K}{N}", (N, K), 1.0 / math.sqrt(K))
        b = rnd(f"rpb{K}{N}", (N,), 0.2)
        rw = r16(rnd(f"rpr{K}{N}", (B, H, W, N + (16 if sliced else 0)))).cuda().to(BF)
        res = (rw[..., 16:] if sliced else rw) if with_res else None
        cv = ops.Conv(w.cuda().contiguous(), b.cuda(), 1, 1, K, N)
        with ops.profile() as rec:
            y = ops.conv(x, cv, act=act, res=res)
        assert [r[1].get("rp") for r in rec.rows] == [1], rec.rows
        ref = x.float().cpu().reshape(-1, K).double() @ r16(w).double().T + b.double()
        ref = ref.float()
        if act == 1:
            ref = torch.relu(ref)
        elif act == 4:
            ref = F.gelu(ref)
        if res is not None:
            ref = ref + res.float().cpu().reshape(-1, N)
        ok, worst = close16(y.float().cpu().reshape(-1, N), ref, ulps=1.02, floor=2e-3)
        assert ok, worst
        ops.GEMM16_RP = 0
        try:
            y0 = ops.conv(x, cv, act=act, res=res)
        finally:
            ops.GEMM16_RP = 1
        assert float((y0.float() - y.float()).abs().max()) <= 2.0 ** -7 * float(y0.float().abs().max())      # the implicit GEMM: same values up to the last bf16 bit
    finally:
        ops.GEMM16_RP_MIN_ROWS = rows


def test_row_panel_gemm_bf16_unpatchify_store(ops):
    """the row-panel kernel with the un-patchify (depth-to-space) store of the patch Linears == the implicit GEMM's d2s store, bit for bit
    up to summation order (same operands, same single rounding)."""
    rows = ops.GEMM16_RP_MIN_ROWS
    ops.GEMM16_RP_MIN_ROWS = 1024
    try:
        for (p_, C_, B) in ((8, 64, 3), (4, 128, 2), (2, 64, 5)):
            N, K = p_ * p_ * C_, 256
            x = r16(rnd(f"d2x{p_}", (B, 32, 32, K))).cuda().to(BF)
            cv = ops.Conv(rnd(f"d2w{p_}", (N, K), 1.0 / math.sqrt(K)).cuda().contiguous(), rnd(f"d2b{p_}", (N,), 0.2).cuda(), 1, 1, K, N)
            with ops.profile() as rec:
                y = ops.conv(x, cv, d2s=(p_, C_))
            assert [r[1].get("rp") for r in rec.rows] == [1] and tuple(y.shape) == (B, 32 * p_, 32 * p_, C_)
            ops.GEMM16_RP = 0
            try:
                y0 = ops.conv(x, cv, d2s=(p_, C_))
            finally:
                ops.GEMM16_RP = 1
            assert float((y0.float() - y.float()).abs().max()) <= 2.0 ** -7 * float(y0.float().abs().max())
            assert float((y0.float() - y.float()).abs().mean()) < 1e-3 * float(y0.float().abs().mean())
    finally:
        ops.GEMM16_RP_MIN_ROWS = rows


@pytest.mark.parametrize("B,H,W,Cin,N,pad", [(2, 64, 64, 128, 17, 3), (2, 64, 64, 36, 76, 0), (1, 24, 40, 20, 33, 3), (3, 16, 16, 4, 96, 0)])
def test_conv7_bf16x3_heads(ops, B, H, W, Cin, N, pad):
    """csrc/conv7_bf16x3.hip: the 7x7 heads in hi + lo bf16 arithmetic (three bf16 MFMAs per product) against the fp64 convolution --
    fp32-grade: 2e-5 of the largest output (plain bf16 operands would sit at ~4e-3), and against the fp32 implicit GEMM it replaces."""
    x = rnd(f"c7x{Cin}{N}", (B, Cin, H, W))
    w = rnd(f"c7w{Cin}{N}", (N, Cin, 7, 7), 1.0 / math.sqrt(49 * Cin))
    b = rnd(f"c7b{Cin}{N}", (N,), 0.1)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=pad).float()
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    with ops.profile() as rec:
        y = ops.conv7_x3(xin, cv, pad=pad)
    assert [r[0] for r in rec.rows] == ["conv7_x3"] and y.dtype == torch.float32
    scale = max(1.0, float(ref.abs().max()))
    assert maxabs(nchw32(y), ref) < 2e-5 * scale, maxabs(nchw32(y), ref)
    y32 = ops.conv(xin, cv, pad=(pad, pad))
    assert maxabs(y, y32) < 3e-5 * scale
    plain = F.conv2d(r16(x).double(), r16(w).double(), b.double(), padding=pad).float()       # what bf16 operands would give
    assert maxabs(plain, ref) > 20 * maxabs(nchw32(y), ref)


def test_attnblock_fused_bf16(ops):
    """engine_netg._Attn on bf16 storage: the core as ONE kernel (smx_attnblock_bf16) against (i) the three-launch form it replaces (QK^T GEMM with
    fp32 scores, softmax_rows, PV GEMM that rounds the probabilities while staging -- the same arithmetic up to summation order) and (ii) the
    fp32 module evaluated on the bf16-rounded input: within twice the three-launch form's own distance from it."""
    from synergize_motion_appearance_amd import engine_netg as E
    C_, B = 256, 3
    P = {"a.norm.weight": 1 + 0.1 * rnd("ab_g", (C_,)), "a.norm.bias": 0.1 * rnd("ab_b", (C_,))}
    for n in ("q", "k", "v", "proj_out"):
        P[f"a.{n}.weight"] = rnd("ab_w" + n, (C_, C_, 1, 1), 1.5 / math.sqrt(C_))
        P[f"a.{n}.bias"] = rnd("ab_bias" + n, (C_,), 0.1)
    P = {k: v.cuda() for k, v in P.items()}
    blk = E._Attn(P, "a")
    x = r16(rnd("ab_x", (B, 32, 32, C_))).cuda()
    ref32 = blk(x)
    with ops.profile() as rec:
        fused = blk(x.to(BF))
    assert "attnblock" in [r[0] for r in rec.rows] and "softmax" not in [r[0] for r in rec.rows]
    E.ATTNBLOCK_FUSED16 = 0
    try:
        with ops.profile() as rec3:
            three = blk(x.to(BF))
    finally:
        E.ATTNBLOCK_FUSED16 = 1
    assert "attnblock" not in [r[0] for r in rec3.rows]
    d3 = float((three.float() - ref32).abs().max())
    df = float((fused.float() - ref32).abs().max())
    d = float((fused.float() - three.float()).abs().max())
    scale = float(ref32.abs().max())
    assert df <= 2.0 * d3 + 2.0 ** -8 * scale and d <= 2.0 * d3 + 2.0 ** -7 * scale, (df, d3, d, scale)
    assert float((fused.float() - ref32).abs().mean()) <= 1.5 * float((three.float() - ref32).abs().mean()) + 1e-4


def test_conv7x7_two_channel_flow_encoder(ops):
    """csrc/conv7_c2_bf16.hip (BasicMotionEncoder.convf1: 7x7 pad 3, 2 fp32 channels -> 128 bf16, ReLU): the stored value is the correctly
    rounded fp32 result of the bf16-rounded operands; == the implicit GEMM it replaces up to the last bf16 bit."""
    B, H, W, N = 9, 64, 64, 128
    x = rnd("c72x", (B, 2, H, W)) * 3.0
    w = rnd("c72w", (N, 2, 7, 7), 1.0 / math.sqrt(98))
    b = rnd("c72b", (N,), 0.1)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    with ops.profile() as rec:
        y = ops.conv(xin, cv, act=1, mfma16=True)
    assert y.dtype == BF and rec.rows[0][1]["K"] == 98 and rec.rows[0][1].get("k") == 7
    ref = F.relu(F.conv2d(r16(x).double(), r16(w).double(), b.double(), padding=3)).float()
    ok, worst = close16(nchw32(y), ref, ulps=1.0)
    assert ok, worst
    ops.CONV7_C2 = 0
    try:
        y0 = ops.conv(xin, cv, act=1, mfma16=True)
    finally:
        ops.CONV7_C2 = 1
    assert float((y0.float() - y.float()).abs().max()) <= 2.0 ** -7 * float(y0.float().abs().max())


@pytest.mark.parametrize("B,Cin,Co,H,W,act,gn", [(2, 64, 3, 64, 64, 0, True), (1, 128, 2, 32, 96, 1, False), (3, 256, 4, 16, 32, 0, True)])
def test_conv3x3_small_n_mfma(ops, B, Cin, Co, H, W, act, gn):
    """csrc/conv3x3_smalln_mfma16.hip (C_out <= 4 on the bf16 MFMA, one 32-wide N tile, fused GroupNorm + swish loader) against the fp64
    convolution of the bf16-rounded operands (weights are bf16 in this kernel, like every other layer of the configuration), and against
    the VALU kernel it replaces (fp32 weights: differs by the weight rounding only)."""
    x = r16(rnd(f"snm_x{Cin}", (B, Cin, H, W)))
    w = rnd(f"snm_w{Cin}", (Co, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd(f"snm_b{Cin}", (Co,), 0.1)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    ss = None
    xin = x
    if gn:
        ss = torch.stack([1.0 + 0.2 * rnd(f"snm_s{Cin}", (B, Cin)), 0.1 * rnd(f"snm_t{Cin}", (B, Cin))], -1).contiguous()
        xin = O.swish(x * ss[..., 0][:, :, None, None] + ss[..., 1][:, :, None, None])
    old = ops.SMALLN_MFMA_MIN_BLOCKS
    ops.SMALLN_MFMA_MIN_BLOCKS = 1
    try:
        with ops.profile() as rec:
            y = ops.conv(nhwc16(x), cv, act=act, out_dtype=torch.float32, in_ss=None if ss is None else ss.cuda(), in_swish=gn)
        assert [r[0] for r in rec.rows] == ["conv_small_n"] and rec.rows[0][1].get("mfma_flops", 0) > 0 and y.dtype == torch.float32
        ops.SMALLN_MFMA = 0
        try:
            yv = ops.conv(nhwc16(x), cv, act=act, out_dtype=torch.float32, in_ss=None if ss is None else ss.cuda(), in_swish=gn)
        finally:
            ops.SMALLN_MFMA = 1
    finally:
        ops.SMALLN_MFMA_MIN_BLOCKS = old
    ref = F.conv2d(r16(xin).double(), r16(w).double(), b.double(), padding=1)
    ref = (F.relu(ref) if act == 1 else ref).float()
    scale = max(1.0, float(ref.abs().max()))
    assert maxabs(nchw32(y), ref) < (3e-3 if gn else 2e-5) * scale          # gn: the normalised input is rounded to bf16 once more inside the loader
    assert maxabs(y, yv) < 1.2e-2 * scale                                    # VALU kernel: fp32 weights, unrounded normalised input


def test_conv3x3_small_n_bf16_input(ops):
    B, Cin, Co, H = 2, 64, 3, 32
    x = r16(rnd("sn16x", (B, Cin, H, H)))
    w = rnd("sn16w", (Co, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd("sn16b", (Co,), 0.1)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    with ops.profile() as rec:
        y = ops.conv(nhwc16(x), cv, out_dtype=torch.float32)
    assert [r[0] for r in rec.rows] == ["conv_small_n"] and y.dtype == torch.float32
    assert maxabs(nchw32(y), F.conv2d(x, w, b, padding=1)) < 2e-5             # fp32 weights, fp32 math, fp32 output


# ---------------------------------------------------------------------------------------
# pipeline: configs[2] against the reference's fp32 output, judged by the reference-under-autocast error
# ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def nets16():
    import os
    import yaml
    from basicsr.archs import build_network
    from tests.util import HERE
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(HERE), "options/test.yml")))
    net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
    net_g.load_state_dict(weights("network_g"), strict=True)
    me.load_state_dict(weights("network_motion_estimator"), strict=True)
    net_g, me = net_g.eval().cuda(), me.eval().cuda()
    net_g.set_compute_dtype("bf16")
    me.set_compute_dtype("bf16")
    return net_g, me


def test_bf16_netg_vs_fp32_reference_within_the_reference_autocast_error(nets16):
    """net_g on the bf16 path, fed the reference's fp32 dense motion: max / mean |out - fp32 reference| must not exceed
    what the reference itself loses under CPU autocast(bfloat16) on the same inputs (x1.25 margin for the different
    bf16 operator coverage of the two implementations)."""
    net_g, _ = nets16
    g = golden("autocast_bf16.npz")
    src, _ = clip()
    dm = {"deformation": torch.from_numpy(g["deformation_fp32"]).cuda(), "occlusion_map": torch.from_numpy(g["occlusion_fp32"]).cuda(),
          "driving_kp_heatmap": torch.from_numpy(g["heat_fp32"]).cuda()}
    o = net_g(src[None].cuda(), dm, w=1, inference=True)
    out = o["out"].cpu()
    ref = torch.from_numpy(g["out_fp32"])
    emax, emean = float((out - ref).abs().max()), float((out - ref).abs().mean())
    rmax, rmean = [float(v) for v in g["err_out_netg_only"]]
    print(f"bf16 net_g: max {emax:.4f} mean {emean:.5f}   reference under autocast: max {rmax:.4f} mean {rmean:.5f}")
    assert emax <= 1.25 * rmax and emean <= 1.25 * rmean, (emax, emean, rmax, rmean)
    assert torch.isfinite(out).all()


def test_bf16_end_to_end_vs_fp32_reference_within_the_reference_autocast_error(nets16):
    """keypoints -> dense motion -> net_g, all on the configs[2] path (hourglass convolutions on the bf16 MFMA, heads and
    flow math fp32), against the fp32 reference; bar = the reference's own end-to-end autocast error."""
    net_g, me = nets16
    g = golden("autocast_bf16.npz")
    src, drv = clip()
    idx = [int(i) for i in g["frames"]]
    s = src[None].cuda()
    kp_s, kp_d = me.estimate_kp(s), me.estimate_kp(drv[idx].cuda())
    kmax, _ = [float(v) for v in g["err_kp_value"]]
    assert maxabs(kp_d["value"].cpu(), g["kp_value_fp32"]) <= 1.25 * kmax + 1e-4
    dm = me.estimate_motion_w_kp(kp_source=kp_s, kp_driving=kp_d, source_image=s)
    dmax, _ = [float(v) for v in g["err_deformation"]]
    assert maxabs(dm["deformation"].cpu(), g["deformation_fp32"]) <= 1.25 * dmax + 1e-4
    out = net_g(s, dm, w=1, inference=True)["out"].cpu()
    ref = torch.from_numpy(g["out_fp32"])
    emax, emean = float((out - ref).abs().max()), float((out - ref).abs().mean())
    rmax, rmean = [float(v) for v in g["err_out_e2e"]]
    print(f"bf16 e2e: max {emax:.4f} mean {emean:.5f}   reference under autocast: max {rmax:.4f} mean {rmean:.5f}")
    assert emax <= 1.25 * rmax and emean <= 1.25 * rmean, (emax, emean, rmax, rmean)


def test_bf16_batched_driver_and_packed_state_roundtrip(nets16):
    """animate_batched on the bf16 engines (uint8 frames), batch-size independence (<= 2 LSB in bf16) and the packed
    source state carrying bf16 encoder taps as raw bits (14.2 MB instead of 28.3 MB)."""
    from synergize_motion_appearance_amd import driver
    net_g, me = nets16
    src, drv = clip()
    src, drv = src.cuda(), drv.cuda()
    st = driver.encode_source_state(net_g, me, src, drv[0:1], True)
    assert st.cache.feats[256].dtype == BF
    flat = driver.pack_source_state(st.cache, st.src64, st.kp_source, st.kp_initial, st.scale)
    assert flat.dtype == torch.float32 and flat.numel() == driver.cache_numel(BF) == 7077888 // 2 + 12288 + 180 + 1
    st2 = driver.unpack_source_state(flat, BF)
    a = driver.render_frames(st, drv, net_g, me, True, True, batch=8)
    b = driver.render_frames(st2, drv, net_g, me, True, True, batch=8)
    assert torch.equal(a, b) and a.dtype == torch.uint8 and a.shape == (8, 256, 256, 3)
    # a different batch size selects other tile shapes / split-K factors: another summation order, and in bf16 storage a
    # result that lands on the other side of a rounding boundary is 2^-8 off and propagates through ~100 layers.  Two bf16
    # evaluations are therefore as far from each other as each is from fp32; bar = the reference-under-autocast error itself
    # (mean < 1.5x its mean error, worst pixel within its max error)
    c = driver.render_frames(st, drv, net_g, me, True, True, batch=3)
    d = (a.int() - c.int()).abs().float()
    rmax = float(golden("autocast_bf16.npz")["err_out_netg_only"][0])
    rmean = float(golden("autocast_bf16.npz")["err_out_netg_only"][1])
    assert float(d.mean()) < 1.5 * rmean * 127.5 and float(d.max()) <= rmax * 127.5, (float(d.mean()), float(d.max()))


@pytest.mark.parametrize("B,C,H,W,tile_h", [(2, 64, 32, 32, 16), (2, 128, 16, 32, 8), (1, 64, 16, 16, 8), (2, 64, 32, 32, 32), (1, 128, 16, 64, 32)])
def test_conv_sft_epilogue_bf16(ops, B, C, H, W, tile_h, monkeypatch):
    """Fuse_sft_block's `dec + w * (dec * scale + shift)` as the epilogue of the shift branch's 3x3 conv on the bf16 region kernel
    (smx_conv3x3_sft_bf16), operands being channel slices of wider bf16 buffers as in the engine: against conv2d + the formula in
    fp32 on the bf16-rounded operands (one rounding of the result), and it must not run a separate sft_combine pass."""
    monkeypatch.setattr(ops, "CONV16_T32", int(tile_h == 32))          # 32: the 16x32-tile kernel (smx_conv3x3_sft_bf16_t32)
    monkeypatch.setattr(ops, "CONV16_T32_MIN_BLOCKS", 1)
    monkeypatch.setattr(ops, "CONV16_TILE_H", 16 if tile_h == 32 else tile_h)
    ss = r16(rnd(f"bs{C}{H}", (B, H, W, 2 * C)))
    cat = r16(rnd(f"bc{C}{H}", (B, H, W, 2 * C)))
    scale = r16(rnd(f"bq{C}{H}", (B, H, W, C)))
    w = rnd(f"bw{C}", (C, C, 3, 3), 1.0 / math.sqrt(9 * C))
    b = rnd(f"bb{C}", (C,), 0.1)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    ssd, catd = ss.cuda().to(BF), cat.cuda().to(BF)
    with ops.profile() as rec:
        y = ops.conv_sft(ssd[..., C:], cv, catd[..., C:], scale.cuda().to(BF), 0.7)
    assert [r[0] for r in rec.rows] == ["conv3x3_bf16"] and y.dtype == BF and bool(rec.rows[0][1].get("t32")) == (tile_h == 32)
    shift = F.conv2d(ss[..., C:].permute(0, 3, 1, 2).double(), r16(w).double(), b.double(), padding=1).permute(0, 2, 3, 1).float()
    dec = cat[..., C:]
    ok, worst = close16(y.float().cpu(), dec + 0.7 * (dec * scale + shift), ulps=1.0)
    assert ok, worst
