"""The K = 128 / 256 1x1 layers of the fp32 configuration on the BF16 matrix pipe with three-way split operands (csrc/gemm_rp_bf3.hip)
on a real MI355X: the bars of the fp32-MFMA row-panel kernel's tests (tests/test_gpu_kernels.py: 2e-5 relative against the fp64
product), bias / activation / residual, channel-slice operands, the un-patchify store, and a direct fp64 comparison with the kernel it
replaces.  Reference call sites: the token Linears of /root/reference/basicsr/archs/appmotioncodebook_arch.py:69-70, 101-115 and the 1x1
convolutions around them."""
import math

import pytest
import torch
import torch.nn.functional as F

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
def small(ops, monkeypatch):
    monkeypatch.setattr(ops, "GEMM_RP_BF3_MIN_ROWS", 32)
    monkeypatch.setattr(ops, "GEMM16_RP_MIN_ROWS", 1024)
    monkeypatch.setattr(ops, "GEMM_RP_F16", 0)          # the six-product bf16 form; the f16x3 form has its own tests below


def rnd(name, shape, scale=1.0):
    return synth_input(name, shape) * scale


def ran(rec):
    return [(r[1].get("rp"), r[1].get("bf3")) for r in rec.rows if r[0] == "gemm_conv"]


@pytest.mark.parametrize("B,K,N,act,with_res,sliced", [(8, 256, 256, 0, True, False), (4, 256, 512, 4, False, True), (8, 128, 256, 0, False, False),
                                                      (4, 128, 512, 1, True, True), (5, 256, 128, 0, True, False), (1, 256, 384, 3, False, False)])
def test_split_row_panel_gemm(ops, small, monkeypatch, B, K, N, act, with_res, sliced):
    """six bf16 products per fp32 multiply == the fp64 product at the fp32 kernel's bar; == the fp32-MFMA kernel far inside that bar;
    operands that are channel slices of wider buffers (ld > C), output neighbours untouched; and it is the kernel that ran."""
    H, W = 64, 48
    xw = rnd(f"r3x{K}{N}", (B, H, W, K + (24 if sliced else 0))).cuda()
    x = xw[..., 8:8 + K] if sliced else xw
    w = rnd(f"r3w{K}{N}", (N, K), 1.0 / math.sqrt(K))
    b = rnd(f"r3b{K}{N}", (N,), 0.2)
    rw = rnd(f"r3r{K}{N}", (B, H, W, N + (16 if sliced else 0))).cuda()
    res = (rw[..., 16:] if sliced else rw) if with_res else None
    cv = ops.Conv(w.cuda().contiguous(), b.cuda(), 1, 1, K, N)
    out = torch.full((B, H, W, N + 8), 5.0, device="cuda")
    with ops.profile() as rec:
        ops.conv(x, cv, out=out[..., 4:4 + N], act=act, res=res)
    assert ran(rec) == [(1, 6)], rec.rows
    y = out[..., 4:4 + N]
    ref = x.cpu().reshape(-1, K).double() @ w.double().T + b.double()
    ref = {0: lambda t: t, 1: torch.relu, 3: lambda t: t * torch.sigmoid(t), 4: F.gelu}[act](ref)
    if res is not None:
        ref = ref + res.cpu().reshape(-1, N).double()
    bar = 2e-5 * max(1.0, float(ref.abs().max()))
    assert maxabs(y.cpu().reshape(-1, N), ref.float()) < bar
    assert float(out[..., :4].min()) == 5.0 and float(out[..., 4 + N:].max()) == 5.0
    monkeypatch.setattr(ops, "GEMM_RP_BF3", 0)
    with ops.profile() as rec:
        y32 = ops.conv(x, cv, act=act, res=res)
    assert ran(rec) == [(1, None)]
    assert maxabs(y32, y) < bar


def test_split_row_panel_gemm_unpatchify_store(ops, small, monkeypatch):
    """the un-patchify (depth-to-space) store of the split kernel == the fp32 row-panel kernel's and the implicit GEMM's (summation order only)."""
    for (p_, C_, B, K) in ((8, 64, 3, 256), (4, 128, 2, 256), (2, 64, 5, 256), (4, 32, 2, 128), (8, 32, 1, 128)):
        N = p_ * p_ * C_
        x = rnd(f"d3x{p_}{K}", (B, 32, 32, K)).cuda()
        cv = ops.Conv(rnd(f"d3w{p_}{K}", (N, K), 1.0 / math.sqrt(K)).cuda().contiguous(), rnd(f"d3b{p_}", (N,), 0.2).cuda(), 1, 1, K, N)
        monkeypatch.setattr(ops, "GEMM_RP_BF3", 1)
        with ops.profile() as rec:
            y = ops.conv(x, cv, d2s=(p_, C_), act=2)
        assert ran(rec) == [(1, 6)] and tuple(y.shape) == (B, 32 * p_, 32 * p_, C_)
        monkeypatch.setattr(ops, "GEMM_RP_BF3", 0)
        monkeypatch.setattr(ops, "GEMM_RP", 0)
        y0 = ops.conv(x, cv, d2s=(p_, C_), act=2)
        monkeypatch.setattr(ops, "GEMM_RP", 1)
        assert maxabs(y0, y) < 2e-5 * max(1.0, float(y0.abs().max()))


@pytest.mark.parametrize("K,N", [(256, 256), (128, 256), (256, 1024)])
def test_split_gemm_not_less_accurate_than_the_fp32_mfma_kernel(ops, small, monkeypatch, K, N):
    """against the fp64 product of the SAME fp32 operands: the split kernel's error is within 1.25x of the fp32-MFMA row-panel kernel's
    (products exact to 2^-24 relative, eight times fewer accumulator roundings)."""
    x = (rnd(f"a3x{K}", (4, 64, 64, K)) * 1.7).cuda()
    w = rnd(f"a3w{K}{N}", (N, K), 1.0 / math.sqrt(K))
    b = rnd(f"a3b{N}", (N,), 0.1)
    cv = ops.Conv(w.cuda().contiguous(), b.cuda(), 1, 1, K, N)
    ref = x.cpu().reshape(-1, K).double() @ w.double().T + b.double()
    with ops.profile() as rec:
        y6 = ops.conv(x, cv)
    assert ran(rec) == [(1, 6)]
    monkeypatch.setattr(ops, "GEMM_RP_BF3", 0)
    y32 = ops.conv(x, cv)
    e6 = float((y6.cpu().reshape(-1, N).double() - ref).abs().max()); e32 = float((y32.cpu().reshape(-1, N).double() - ref).abs().max())
    r6 = float((y6.cpu().reshape(-1, N).double() - ref).pow(2).mean().sqrt()); r32 = float((y32.cpu().reshape(-1, N).double() - ref).pow(2).mean().sqrt())
    print(f"\nK {K} N {N}: max|err| vs fp64  fp32-MFMA {e32:.3e}  bf16x6 {e6:.3e}   rms  fp32-MFMA {r32:.3e}  bf16x6 {r6:.3e}")
    assert e6 <= 1.25 * e32 + 1e-7 and r6 <= 1.1 * r32 + 1e-8


def test_split_gemm_pack_is_an_exact_three_way_split(ops):
    """hi + mid + lo == w bit for bit, fragment order [N/32][K/16][level][lane][8] with lane l <-> row l & 31, k = 8 (l >> 5) + 0..7."""
    N, K = 128, 256
    w = rnd("p3gw", (N, K), 0.3)
    cv = ops.Conv(w.cuda().contiguous(), None, 1, 1, K, N)
    p3 = cv.w_rp3.view(torch.bfloat16).view(N // 32, K // 16, 3, 64, 8).cpu().double()
    tot = (p3[:, :, 0] + p3[:, :, 1] + p3[:, :, 2]).float()                      # [nt][step][lane][8]
    exp = w.view(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(N // 32, K // 16, 64, 8)
    assert torch.equal(tot, exp)
    assert float(p3[:, :, 1].abs().max()) <= float(p3[:, :, 0].abs().max()) * 2 ** -8


def test_split_gemm_is_deterministic_and_refuses_bad_arguments(ops, small):
    x = rnd("dt3x", (2, 64, 64, 256)).cuda()
    cv = ops.Conv(rnd("dt3w", (256, 256), 1 / 16).cuda().contiguous(), rnd("dt3b", (256,), 0.1).cuda(), 1, 1, 256, 256)
    a = ops.conv(x, cv); b = ops.conv(x, cv)
    assert torch.equal(a, b)
    from synergize_motion_appearance_amd import lib
    Lb = lib.load()
    assert Lb.smx_gemm_rp_bf3_ok(8192, 256, 256) == 1 and Lb.smx_gemm_rp_bf3_ok(8192, 128, 128) == 0 and Lb.smx_gemm_rp_bf3_ok(8200, 256, 256) == 0
    assert Lb.smx_gemm_rp_bf3_ok(8192, 256, 192) == 0
    y = torch.empty((8192, 256), device="cuda")
    assert Lb.smx_gemm_rp_bf3(x.data_ptr() + 4, 256, cv.w_rp3.data_ptr(), None, None, 0, y.data_ptr(), 256, 8192, 256, 256, 0, None) != 0   # misaligned rows
    assert Lb.smx_gemm_rp_bf3(x.data_ptr(), 256, cv.w_rp3.data_ptr(), None, None, 0, y.data_ptr(), 128, 8192, 256, 256, 0, None) != 0      # ldc < N


# ---- the f16x3 form (two IEEE-half levels, three products; per-row input scale) ------------------------------------------------------------------------------
@pytest.fixture
def half(ops, monkeypatch):
    monkeypatch.setattr(ops, "GEMM_RP_BF3_MIN_ROWS", 32)
    monkeypatch.setattr(ops, "GEMM16_RP_MIN_ROWS", 1024)
    monkeypatch.setattr(ops, "GEMM_RP_F16", 1)


@pytest.mark.parametrize("B,K,N,act,with_res,sliced", [(8, 256, 256, 0, True, False), (4, 256, 512, 4, False, True), (8, 128, 256, 0, False, False),
                                                      (4, 128, 512, 1, True, True), (5, 256, 128, 0, True, False)])
def test_f16x3_row_panel_gemm(ops, half, B, K, N, act, with_res, sliced):
    H, W = 64, 48
    xw = rnd(f"r3x{K}{N}", (B, H, W, K + (24 if sliced else 0))).cuda()
    x = xw[..., 8:8 + K] if sliced else xw
    w = rnd(f"r3w{K}{N}", (N, K), 1.0 / math.sqrt(K))
    b = rnd(f"r3b{K}{N}", (N,), 0.2)
    rw = rnd(f"r3r{K}{N}", (B, H, W, N + (16 if sliced else 0))).cuda()
    res = (rw[..., 16:] if sliced else rw) if with_res else None
    cv = ops.Conv(w.cuda().contiguous(), b.cuda(), 1, 1, K, N)
    out = torch.full((B, H, W, N + 8), 5.0, device="cuda")
    with ops.profile() as rec:
        ops.conv(x, cv, out=out[..., 4:4 + N], act=act, res=res)
    assert ran(rec) == [(1, 4)], rec.rows
    ref = x.cpu().reshape(-1, K).double() @ w.double().T + b.double()
    ref = {0: lambda t: t, 1: torch.relu, 4: F.gelu}[act](ref)
    if res is not None:
        ref = ref + res.cpu().reshape(-1, N).double()
    assert maxabs(out[..., 4:4 + N].cpu().reshape(-1, N), ref.float()) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert float(out[..., :4].min()) == 5.0 and float(out[..., 4 + N:].max()) == 5.0


def test_f16x3_unpatchify_store(ops, half, monkeypatch):
    for (p_, C_, B, K) in ((8, 64, 3, 256), (4, 128, 2, 256), (4, 32, 2, 128)):
        N = p_ * p_ * C_
        x = rnd(f"d3x{p_}{K}", (B, 32, 32, K)).cuda()
        cv = ops.Conv(rnd(f"d3w{p_}{K}", (N, K), 1.0 / math.sqrt(K)).cuda().contiguous(), rnd(f"d3b{p_}", (N,), 0.2).cuda(), 1, 1, K, N)
        with ops.profile() as rec:
            y = ops.conv(x, cv, d2s=(p_, C_), act=2)
        assert ran(rec) == [(1, 4)] and tuple(y.shape) == (B, 32 * p_, 32 * p_, C_)
        monkeypatch.setattr(ops, "GEMM_RP", 0)
        y0 = ops.conv(x, cv, d2s=(p_, C_), act=2)
        monkeypatch.setattr(ops, "GEMM_RP", 1)
        assert maxabs(y0, y) < 2e-5 * max(1.0, float(y0.abs().max()))


@pytest.mark.parametrize("name,K,N,mk", [("plain", 256, 256, lambda x: x * 1.7), ("k128", 128, 256, lambda x: x * 1.7), ("wide", 256, 1024, lambda x: x * 1.7),
                                         ("tiny rows and huge rows", 256, 256, lambda x: x * torch.logspace(-6, 5, x.shape[0] * x.shape[1] * x.shape[2]).view(*x.shape[:3], 1)),
                                         ("zero rows", 256, 128, lambda x: x * (torch.arange(x.shape[2]) % 3 != 0).float().view(1, 1, -1, 1))])
def test_f16x3_gemm_not_less_accurate_than_the_fp32_mfma_kernel(ops, half, monkeypatch, name, K, N, mk):
    """against the fp64 product of the SAME fp32 operands; rows of every magnitude in one launch (each row carries its own power-of-two scale): the error of every
    row, relative to that row's own output scale, is within 1.25x of the fp32-MFMA row-panel kernel's."""
    x = mk(rnd(f"a3x{K}", (4, 64, 64, K))).cuda()
    w = rnd(f"a3w{K}{N}", (N, K), 1.0 / math.sqrt(K))
    cv = ops.Conv(w.cuda().contiguous(), None, 1, 1, K, N)
    ref = x.cpu().reshape(-1, K).double() @ w.double().T
    with ops.profile() as rec:
        y16 = ops.conv(x, cv)
    assert ran(rec) == [(1, 4)]
    monkeypatch.setattr(ops, "GEMM_RP_BF3", 0)
    y32 = ops.conv(x, cv)
    rs = ref.pow(2).mean(1).sqrt().clamp_min(1e-30)                      # every row judged against its own scale
    e16 = ((y16.cpu().reshape(-1, N).double() - ref).abs().max(1).values / rs); e32 = ((y32.cpu().reshape(-1, N).double() - ref).abs().max(1).values / rs)
    print(f"\n{name}: worst row-relative max error  fp32-MFMA {float(e32.max()):.3e}  f16x3 {float(e16.max()):.3e}   mean {float(e32.mean()):.3e} {float(e16.mean()):.3e}")
    assert float(e16.max()) <= 1.25 * float(e32.max()) + 1e-7 and float(e16.mean()) <= 1.1 * float(e32.mean()) + 1e-8
    assert bool(torch.isfinite(y16).all())

