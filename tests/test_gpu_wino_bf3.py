"""fp32 3x3 convolutions on the BF16 matrix pipe with three-way split operands (csrc/winograd_bf3.hip, "bf16x6") on a real MI355X:
the same bars as the fp32-MFMA Winograd kernel's tests (tests/test_gpu_kernels.py: 5e-5 against F.conv2d), every epilogue, the fused
GroupNorm(+swish) loader, border blocks, and a direct fp64 comparison showing the split arithmetic is not less accurate than the
fp32 MFMA kernel it replaces.  Reference call sites: /root/reference/basicsr/archs/vqgan_arch.py:168-191 (ResBlock 3x3 convolutions),
appmotioncodebook_arch.py:49-51 (Fuse_sft_block)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import reenact_oracle as O
from synergize_motion_appearance_amd.synth import synth_input
from tests.util import maxabs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "needs an MI355X"
    from synergize_motion_appearance_amd import ops as _ops
    from synergize_motion_appearance_amd import lib
    lib.load()
    return _ops


@pytest.fixture
def bf3(ops, monkeypatch):
    """route eligible launches to the split kernel (any size), restore afterwards."""
    def _set(nprod):
        monkeypatch.setattr(ops, "WINO_BF3", nprod)
        monkeypatch.setattr(ops, "WINO_F16", 0)              # the bf16 forms; the f16x3 form has its own tests at the end of this file
        monkeypatch.setattr(ops, "WINO_BF3_MIN_BLOCKS", 1)
    return _set


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rnd(name, shape, scale=1.0):
    return synth_input(name, shape) * scale


def ran_bf3(rec):
    return [r[1].get("bf3") for r in rec.rows if r[0] == "gemm_conv"]


# (B, Cin, Cout, H, W, up2, act, res)
CASES = [(2, 64, 64, 32, 32, False, 0, False), (1, 128, 128, 64, 64, False, 3, True), (2, 256, 128, 16, 32, False, 4, False),
         (1, 128, 64, 64, 48, False, 0, True), (2, 64, 64, 16, 16, True, 0, False), (1, 32, 64, 16, 16, False, 1, False),
         (3, 64, 192, 32, 16, False, 2, True), (1, 512, 256, 32, 32, False, 0, True), (1, 96, 128, 48, 80, True, 0, True)]


@pytest.mark.parametrize("case", CASES, ids=[f"B{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}_up{int(c[5])}_act{c[6]}_res{int(c[7])}" for c in CASES])
def test_bf3_conv3x3_vs_conv2d(ops, bf3, case):
    """six-product split kernel == F.conv2d (3x3, s1, p1) at the fp32 Winograd kernel's own bar, through channel-slice operands
    (an input that starts 32 floats into its rows, an output inside a wider buffer whose neighbours must stay untouched)."""
    B, Cin, Cout, H, W, up2, act, with_res = case
    bf3(6)
    x = rnd(f"b3x{case}", (B, Cin, H, W))
    w = rnd(f"b3w{case}", (Cout, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd(f"b3b{case}", (Cout,), 0.1)
    xe = F.interpolate(x, scale_factor=2.0, mode="nearest") if up2 else x
    ref = F.conv2d(xe, w, b, padding=1)
    ref = {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, 0.2), 3: O.swish, 4: F.gelu}[act](ref)
    r = rnd(f"b3r{case}", tuple(ref.shape)) if with_res else None
    if with_res:
        ref = ref + r
    xin = torch.zeros((B, H, W, Cin + 32), device="cuda")
    xin[..., 32:] = nhwc(x)
    out = torch.full((B, ref.shape[2], ref.shape[3], Cout + 8), 5.0, device="cuda")
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    with ops.profile() as rec:
        ops.conv(xin[..., 32:], cv, out=out[..., 4:4 + Cout], up2=up2, act=act, res=None if r is None else nhwc(r))
    assert ran_bf3(rec) == [6]
    assert maxabs(nchw(out[..., 4:4 + Cout]), ref) < 5e-5
    assert float(out[..., :4].min()) == 5.0 and float(out[..., 4 + Cout:].max()) == 5.0


@pytest.mark.parametrize("Cin,Cout,H", [(128, 128, 64), (64, 64, 64), (256, 128, 32), (512, 256, 32)])
def test_bf3_not_less_accurate_than_the_fp32_mfma_kernel(ops, bf3, Cin, Cout, H):
    """against an fp64 convolution of the SAME fp32 operands: the six-product kernel's error is within 1.25x of the fp32-MFMA Winograd
    kernel's (it is usually smaller: products are exact and there are 8x fewer accumulation roundings); the three-product form is the
    2^-17-class arithmetic it is documented as (reported, bounded loosely)."""
    B = 2
    x = rnd(f"a3x{Cin}{H}", (B, Cin, H, H)) * 1.7
    w = rnd(f"a3w{Cin}{Cout}", (Cout, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd(f"a3b{Cout}", (Cout,), 0.1)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    y32 = ops.conv(nhwc(x), cv)
    bf3(6)
    with ops.profile() as rec:
        y6 = ops.conv(nhwc(x), cv)
    assert ran_bf3(rec) == [6]
    bf3(3)
    y3 = ops.conv(nhwc(x), cv)
    e32 = float((nchw(y32).double() - ref).abs().max()); e6 = float((nchw(y6).double() - ref).abs().max()); e3 = float((nchw(y3).double() - ref).abs().max())
    r32 = float((nchw(y32).double() - ref).pow(2).mean().sqrt()); r6 = float((nchw(y6).double() - ref).pow(2).mean().sqrt())
    print(f"\n{Cin}->{Cout} @{H}: max|err| vs fp64  fp32-MFMA {e32:.3e}  bf16x6 {e6:.3e}  bf16x3 {e3:.3e}   rms  fp32-MFMA {r32:.3e}  bf16x6 {r6:.3e}")
    assert e6 <= 1.25 * e32 + 1e-7 and r6 <= 1.1 * r32 + 1e-8
    assert e3 < 1e-4 and e3 > e6


def test_bf3_pack_is_an_exact_three_way_split(ops):
    """hi + mid + lo == u bit for bit (fp32 adds of the three planes are exact here), every plane a bf16 value, fragment order as documented."""
    w = rnd("p3w", (64, 32, 3, 3), 0.3)
    cv = ops.Conv.from_torch(w.cuda(), None)
    u = cv.winograd_u()[:16 * 2 * 4 * 256].view(16, 2, 4, 2, 32, 4).cpu()          # [f][nt][c8][half][row][4]
    u3 = cv.winograd_bf3_u().view(torch.bfloat16).view(16, 2, 2, 3, 64, 8).cpu().float()    # [f][nt][step][split][lane][8]
    tot = (u3[:, :, :, 0].double() + u3[:, :, :, 1].double() + u3[:, :, :, 2].double()).float()      # [f][nt][step][lane][8]
    # lane l <-> row l & 31, channels 16 step + 8 (l >> 5) + e  ==  c8 = 2 step + (l >> 5), half = e >> 2, elem = e & 3
    exp = u.view(16, 2, 2, 2, 2, 32, 4).permute(0, 1, 2, 3, 5, 4, 6).reshape(16, 2, 2, 64, 8)
    assert torch.equal(tot, exp)
    assert float(u3[:, :, :, 1].abs().max()) <= float(u3[:, :, :, 0].abs().max()) * 2 ** -8


def test_bf3_fused_groupnorm_loader(ops, bf3):
    """GN(32, eps 1e-6) (+ swish) folded into the region staging == group_norm -> swish -> conv2d, border blocks included (one block per image)."""
    bf3(6)
    for (B, C, Co, H, sw) in ((2, 64, 64, 32, True), (1, 128, 64, 64, True), (2, 256, 128, 16, False), (1, 32, 64, 16, True)):
        x = rnd(f"g3x{C}{H}", (B, C, H, H)) * 1.5 + 0.2
        g, bt = 1 + 0.1 * rnd(f"g3g{C}", (C,)), 0.1 * rnd(f"g3b{C}", (C,))
        w = rnd(f"g3w{C}{Co}", (Co, C, 3, 3), 1.0 / math.sqrt(9 * C))
        b = rnd(f"g3bb{Co}", (Co,), 0.1)
        hn = F.group_norm(x, 32, g, bt, 1e-6)
        ref = F.conv2d(O.swish(hn) if sw else hn, w, b, padding=1)
        xin = nhwc(x)
        ss = ops.groupnorm_stats(xin, g.cuda(), bt.cuda())
        with ops.profile() as rec:
            y = ops.conv(xin, ops.Conv.from_torch(w.cuda(), b.cuda()), in_ss=ss, in_swish=sw)
        assert ran_bf3(rec) == [6]
        assert maxabs(nchw(y), ref) < 5e-5


@pytest.mark.parametrize("B,C,Co,H,W,act,res", [(2, 64, 64, 32, 32, 0, True), (3, 128, 128, 16, 32, 3, False), (1, 32, 192, 64, 64, 0, True)])
def test_bf3_epilogue_emits_groupnorm_partials(ops, bf3, B, C, Co, H, W, act, res):
    """want_stats: {mean, M2} per 8 x 16-pixel chunk and channel, in the fp32 kernel's chunk format (a block emits two chunks)."""
    bf3(6)
    x = rnd(f"s3{C}{Co}{H}", (B, C, H, W))
    w = rnd(f"s3w{C}{Co}", (Co, C, 3, 3), 1.0 / math.sqrt(9 * C))
    b = rnd(f"s3b{Co}", (Co,), 0.1)
    r = rnd(f"s3r{Co}{H}", (B, Co, H, W)) if res else None
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    with ops.profile() as rec:
        y = ops.conv(nhwc(x), cv, act=act, res=None if r is None else nhwc(r), want_stats=True)
    assert ran_bf3(rec) == [6]
    part = y._gn_part
    assert part is not None and tuple(part.shape) == (B, (H // 8) * (W // 16), Co, 2)
    yc = y.cpu().double()
    blocks = yc.view(B, H // 8, 8, W // 16, 16, Co).permute(0, 1, 3, 5, 2, 4).reshape(B, -1, Co, 128)
    bm = blocks.mean(-1)
    assert maxabs(part[..., 0].cpu(), bm) < 2e-6 and maxabs(part[..., 1].cpu(), ((blocks - bm[..., None]) ** 2).sum(-1)) < 2e-4
    if Co & (Co - 1) == 0:
        g, bt = rnd(f"s3g{Co}", (Co,)) * 0.2 + 1.0, rnd(f"s3bt{Co}", (Co,), 0.1)
        ss = ops.groupnorm_stats(y, g.cuda(), bt.cuda())
        ref = F.group_norm(nchw(y), 32, g, bt, 1e-6)
        assert maxabs(nchw(ops.groupnorm_apply(y, ss, swish=False)), ref) < 5e-5


def test_bf3_sft_epilogue(ops, bf3):
    """y = dec + w (dec * scale + conv3x3(x)) as the convolution's epilogue == the fp32 kernel's fused form and conv2d + the formula."""
    bf3(6)
    B, C, H, W = 2, 128, 32, 32
    x = rnd("f3x", (B, C, H, W)); dec = rnd("f3d", (B, C, H, W)); sc = rnd("f3s", (B, C, H, W))
    w = rnd("f3w", (C, C, 3, 3), 1.0 / math.sqrt(9 * C)); b = rnd("f3b", (C,), 0.1)
    ref = dec + 0.7 * (dec * sc + F.conv2d(x, w, b, padding=1))
    with ops.profile() as rec:
        y = ops.conv_sft(nhwc(x), ops.Conv.from_torch(w.cuda(), b.cuda()), nhwc(dec), nhwc(sc), 0.7)
    assert ran_bf3(rec) == [6]
    assert maxabs(nchw(y), ref) < 5e-5


def test_bf3_is_deterministic_and_leaves_small_launches_alone(ops, bf3, monkeypatch):
    bf3(6)
    x = nhwc(rnd("d3x", (4, 128, 64, 64)))
    cv = ops.Conv.from_torch(rnd("d3w", (128, 128, 3, 3), 0.03).cuda(), None)
    y0 = ops.conv(x, cv).clone()
    for _ in range(3):
        assert torch.equal(ops.conv(x, cv), y0)
    monkeypatch.setattr(ops, "WINO_BF3_MIN_BLOCKS", 512)               # the shipped threshold: one block per CU needs >= 2 rounds of blocks
    with ops.profile() as rec:
        ops.conv(x[:1], cv)
    assert ran_bf3(rec) == [None]


# ---- the f16x3 form (nprod = 4: two IEEE-half levels, three products; csrc/winograd_bf3.hip) ----------------------------------------------------------------
@pytest.fixture
def f16(ops, monkeypatch):
    def _set(level):
        monkeypatch.setattr(ops, "WINO_BF3", 6)
        monkeypatch.setattr(ops, "WINO_F16", level)
        monkeypatch.setattr(ops, "WINO_BF3_MIN_BLOCKS", 1)
    return _set


@pytest.mark.parametrize("case", CASES, ids=[f"B{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}_up{int(c[5])}_act{c[6]}_res{int(c[7])}" for c in CASES])
def test_f16x3_conv3x3_vs_conv2d(ops, f16, case):
    """the half-precision three-product form on RAW inputs (per-block power-of-two input scale) == F.conv2d at the fp32 Winograd kernel's bar, every epilogue."""
    B, Cin, Cout, H, W, up2, act, with_res = case
    f16(2)
    x = rnd(f"b3x{case}", (B, Cin, H, W))
    w = rnd(f"b3w{case}", (Cout, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd(f"b3b{case}", (Cout,), 0.1)
    xe = F.interpolate(x, scale_factor=2.0, mode="nearest") if up2 else x
    ref = F.conv2d(xe, w, b, padding=1)
    ref = {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, 0.2), 3: O.swish, 4: F.gelu}[act](ref)
    r = rnd(f"b3r{case}", tuple(ref.shape)) if with_res else None
    if with_res:
        ref = ref + r
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    with ops.profile() as rec:
        y = ops.conv(nhwc(x), cv, up2=up2, act=act, res=None if r is None else nhwc(r))
    assert ran_bf3(rec) == [4]
    assert maxabs(nchw(y), ref) < 5e-5


def test_f16x3_fused_groupnorm_loader_and_partials(ops, f16):
    """through the fused GroupNorm (+ swish) loader (level 1: only such launches take the form) and with the GroupNorm partials of the consumer."""
    f16(1)
    for (B, C, Co, H, sw) in ((2, 64, 64, 32, True), (1, 128, 64, 64, True), (2, 256, 128, 16, False), (1, 32, 64, 16, True)):
        x = rnd(f"g3x{C}{H}", (B, C, H, H)) * 1.5 + 0.2
        g, bt = 1 + 0.1 * rnd(f"g3g{C}", (C,)), 0.1 * rnd(f"g3b{C}", (C,))
        w = rnd(f"g3w{C}{Co}", (Co, C, 3, 3), 1.0 / math.sqrt(9 * C))
        b = rnd(f"g3bb{Co}", (Co,), 0.1)
        hn = F.group_norm(x, 32, g, bt, 1e-6)
        ref = F.conv2d(O.swish(hn) if sw else hn, w, b, padding=1)
        xin = nhwc(x)
        ss = ops.groupnorm_stats(xin, g.cuda(), bt.cuda())
        cv = ops.Conv.from_torch(w.cuda(), b.cuda())
        with ops.profile() as rec:
            y = ops.conv(xin, cv, in_ss=ss, in_swish=sw, want_stats=True)
            y0 = ops.conv(xin, cv)                                              # a raw launch stays on the six-product form at level 1
        assert ran_bf3(rec) == [4, 6]
        assert maxabs(nchw(y), ref) < 5e-5
        yc = y.cpu().double()
        blocks = yc.view(B, H // 8, 8, H // 16, 16, Co).permute(0, 1, 3, 5, 2, 4).reshape(B, -1, Co, 128)
        bm = blocks.mean(-1)
        assert maxabs(y._gn_part[..., 0].cpu(), bm) < 2e-6 and maxabs(y._gn_part[..., 1].cpu(), ((blocks - bm[..., None]) ** 2).sum(-1)) < 2e-4


@pytest.mark.parametrize("name,Cin,Cout,H,mk", [
    ("plain", 128, 128, 64, lambda x: x * 1.7), ("plain64", 64, 64, 64, lambda x: x * 1.7), ("deep", 512, 256, 32, lambda x: x * 1.7),
    ("tiny", 128, 64, 32, lambda x: x * 1e-4), ("huge", 64, 64, 32, lambda x: x * 3e3),
    ("grows 1e5 after the first slice", 128, 128, 32, lambda x: torch.cat([x[:, :32] * 1e-3, x[:, 32:] * 100.0], 1)),
    ("grows in the last slices", 256, 128, 32, lambda x: torch.cat([x[:, :200], x[:, 200:] * 3e4], 1)),
    ("first slice zero", 64, 64, 32, lambda x: torch.cat([x[:, :32] * 0, x[:, 32:] * 1e-3], 1))])
def test_f16x3_not_less_accurate_than_the_fp32_mfma_kernel(ops, f16, monkeypatch, name, Cin, Cout, H, mk):
    """against an fp64 convolution of the SAME fp32 operands, raw inputs of every scale: the three-product half form's error is within 1.25x of the fp32-MFMA
    Winograd kernel's (measured: below it, and at or below the six-product bf16 form's).  The input scale is the block's own: tiny and huge tensors, a tensor
    whose later channel slices outgrow the first by 1e5 (the exact power-of-two rescale of region and accumulators runs) and a first slice of zeros included."""
    B = 2
    x = mk(rnd(f"h3x{Cin}{H}", (B, Cin, H, H)))
    w = rnd(f"h3w{Cin}{Cout}", (Cout, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd(f"h3b{Cout}", (Cout,), 0.1) * float(x.abs().mean())
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    monkeypatch.setattr(ops, "WINO_BF3", 0)
    y32 = ops.conv(nhwc(x), cv)
    f16(2)
    with ops.profile() as rec:
        y16 = ops.conv(nhwc(x), cv)
    assert ran_bf3(rec) == [4]
    e32 = float((nchw(y32).double() - ref).abs().max()); e16 = float((nchw(y16).double() - ref).abs().max())
    r32 = float((nchw(y32).double() - ref).pow(2).mean().sqrt()); r16 = float((nchw(y16).double() - ref).pow(2).mean().sqrt())
    scale = float(ref.pow(2).mean().sqrt())
    print(f"\n{name}: max|err| vs fp64  fp32-MFMA {e32:.3e}  f16x3 {e16:.3e}   rms {r32:.3e} {r16:.3e}   (output rms {scale:.2e})")
    assert e16 <= 1.25 * e32 + 1e-7 * scale and r16 <= 1.1 * r32 + 1e-8 * scale


def test_f16x3_pack_scales_u_by_a_power_of_two(ops):
    """header {max |U| bits, 1 / scale}; scale = the power of two that puts max |U| into [2^11, 2^12); level 0 + level 1 == U x scale to 2^-21 of each value."""
    w = rnd("p16w", (64, 32, 3, 3), 0.3)
    cv = ops.Conv.from_torch(w.cuda(), None)
    u = cv.winograd_u()[:16 * 2 * 4 * 256].view(16, 2, 4, 2, 32, 4).cpu()          # [f][nt][c8][half][row][4]
    raw = cv.winograd_f16_u().cpu()
    hdr = raw[:16].view(torch.float32)
    umax = float(u.abs().max())
    assert float(raw[:4].view(torch.float32)[0]) == umax
    inv = float(hdr[1]); su = 1.0 / inv
    assert math.log2(su) == round(math.log2(su)) and 2 ** 11 <= umax * su < 2 ** 12
    lv = raw[16:].view(torch.float16).view(16, 2, 2, 3, 64, 8).double()                # [f][nt][step][plane][lane][8]
    tot = lv[:, :, :, 0] + lv[:, :, :, 1]
    exp = (u.double() * su).view(16, 2, 2, 2, 2, 32, 4).permute(0, 1, 2, 3, 5, 4, 6).reshape(16, 2, 2, 64, 8)
    assert float((tot - exp).abs().max()) <= 2 ** -21 * umax * su and float(lv[:, :, :, 2].abs().max()) == 0.0


def test_f16x3_sft_epilogue_and_determinism(ops, f16):
    f16(2)
    B, C, H, W = 2, 128, 32, 32
    x = rnd("f3x", (B, C, H, W)); dec = rnd("f3d", (B, C, H, W)); sc = rnd("f3s", (B, C, H, W))
    w = rnd("f3w", (C, C, 3, 3), 1.0 / math.sqrt(9 * C)); b = rnd("f3b", (C,), 0.1)
    ref = dec + 0.7 * (dec * sc + F.conv2d(x, w, b, padding=1))
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    with ops.profile() as rec:
        y = ops.conv_sft(nhwc(x), cv, nhwc(dec), nhwc(sc), 0.7)
    assert ran_bf3(rec) == [4]
    assert maxabs(nchw(y), ref) < 5e-5
    assert torch.equal(y, ops.conv_sft(nhwc(x), cv, nhwc(dec), nhwc(sc), 0.7))


@pytest.mark.parametrize("B,Cin,Cout,H,W,act,with_res,stats", [(2, 128, 96, 32, 32, 1, True, False), (2, 160, 126, 64, 64, 1, False, False), (1, 64, 72, 16, 32, 0, True, True),
                                                               (3, 32, 200, 16, 16, 2, False, True)])
def test_f16x3_takes_any_output_channel_count(ops, f16, B, Cin, Cout, H, W, act, with_res, stats):
    """C_out % 64 != 0 (the motion estimator's 96- and 126-channel hourglass layers): U padded at pack time, the ragged quad masked in the epilogue; output a
    channel slice of a wider buffer (ld % 4 == 0) whose neighbours must stay untouched; == F.conv2d at the kernel's bar, GroupNorm partials included."""
    f16(2)
    x = rnd(f"rg{Cin}{Cout}", (B, Cin, H, W))
    w = rnd(f"rgw{Cin}{Cout}", (Cout, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd(f"rgb{Cout}", (Cout,), 0.1)
    ref = F.conv2d(x, w, b, padding=1)
    ref = {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, 0.2)}[act](ref)
    ld = 4 * ((Cout + 3) // 4) + 8
    r = rnd(f"rgr{Cout}{H}", (B, Cout, H, W)) if with_res else None
    if with_res:
        ref = ref + r
    rbuf = None
    if with_res:
        rbuf = torch.zeros((B, H, W, ld), device="cuda"); rbuf[..., 4:4 + Cout] = nhwc(r)
    out = torch.full((B, H, W, ld), 5.0, device="cuda")
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    with ops.profile() as rec:
        y = ops.conv(nhwc(x), cv, out=out[..., 4:4 + Cout], act=act, res=None if rbuf is None else rbuf[..., 4:4 + Cout], want_stats=stats)
    assert ran_bf3(rec) == [4]
    assert maxabs(nchw(out[..., 4:4 + Cout]), ref) < 5e-5
    assert float(out[..., :4].min()) == 5.0 and float(out[..., 4 + Cout:].max()) == 5.0
    if stats:
        part = y._gn_part
        assert tuple(part.shape) == (B, (H // 8) * (W // 16), Cout, 2)
        yc = out[..., 4:4 + Cout].cpu().double()
        blocks = yc.reshape(B, H // 8, 8, W // 16, 16, Cout).permute(0, 1, 3, 5, 2, 4).reshape(B, -1, Cout, 128)
        bm = blocks.mean(-1)
        assert maxabs(part[..., 0].cpu(), bm) < 2e-6 and maxabs(part[..., 1].cpu(), ((blocks - bm[..., None]) ** 2).sum(-1)) < 2e-4

