"""nn.MultiheadAttention's core (d_head 32) with fp32-grade products on the BF16 matrix pipe (attn_bf3_kernel, csrc/attention.hip) on a real MI355X:
the fp32 kernel's own bar against the explicit softmax(q k^T) v (tests/test_gpu_kernels.py::test_fused_attention: 5e-6), key-padding masks, a
context shared by the batch, fully masked rows, and a direct fp64 comparison with the fp32-MFMA kernel it replaces on big launches.
Reference call sites: /root/reference/basicsr/archs/appmotioncodebook_arch.py:69-70, 101-115."""
import pytest
import torch

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
def knob(ops):
    old = []

    def _set(v):
        old.append(ops.set_tuning("attn_bf3", v))
    yield _set
    if old:
        ops.set_tuning("attn_bf3", old[0])


def rnd(name, shape, scale=1.0):
    return synth_input(name, shape) * scale


def reference(q, k, v, H, dh, mask, dtype=torch.float64):
    B, N, E = q.shape
    S = k.shape[1]
    qh = (q.to(dtype) * dh ** -0.5).view(B, N, H, dh).transpose(1, 2)
    kh = k.to(dtype).expand(B, -1, -1).reshape(B, S, H, dh).transpose(1, 2)
    vh = v.to(dtype).expand(B, -1, -1).reshape(B, S, H, dh).transpose(1, 2)
    sc = qh @ kh.transpose(-1, -2)
    if mask is not None:
        sc = sc.masked_fill(mask.view(B, 1, 1, S), float("-inf"))
    return (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B, N, E)


def ran(rec):
    return [r[1].get("bf3") for r in rec.rows if r[0].startswith("attention")]


@pytest.mark.parametrize("S,shared,masked,np_", [(1024, False, True, 3), (256, True, False, 3), (768, True, False, 3), (1024, False, False, 3),
                                                 (1024, False, True, 2), (512, True, False, 2), (1024, False, True, 3 + 32), (512, True, False, 2 + 32),
                                                 (1024, False, False, 3 + 32), (1024, False, True, 4), (256, True, False, 4), (768, True, False, 4), (1024, False, False, 4),
                                                 (1024, False, True, 4 + 32), (512, True, False, 4 + 32)])
def test_split_attention_vs_explicit_softmax(ops, knob, S, shared, masked, np_):
    """the split kernel == softmax(q k^T / sqrt(d)) v per head at the fp32 kernel's bar; and it is the kernel that ran."""
    B, H, N, E, dh = 2, 8, 1024, 256, 32
    knob(16 + np_)                                  # (+ 32: the 128-query block shape; default 256 queries per block)
    np_ &= 15
    q = rnd(f"bq{S}", (B, N, E))
    kv = rnd(f"bkv{S}", ((1 if shared else B), 1024, 2 * E))
    mask = None
    if masked:
        mask = torch.zeros((B, S), dtype=torch.bool)
        mask[0, 5::9] = True
        mask[1, :40] = True
    k, v = kv[..., :E][:, :S], kv[..., E:][:, :S]
    ref = reference(q, k, v, H, dh, mask).float()
    kvd = kv.cuda()
    kd, vd = (kvd[0, :, :E], kvd[0, :, E:]) if shared else (kvd[..., :E], kvd[..., E:])
    with ops.profile() as rec:
        o = ops.attention(q.cuda(), kd, vd, H, dh, S, k_shared=shared, mask=None if mask is None else mask.to(torch.uint8).cuda())
    assert ran(rec) == [np_]
    assert maxabs(o.cpu(), ref) < (1e-5 if np_ == 2 else 5e-6)


def test_split_attention_not_less_accurate_than_the_fp32_mfma_kernel(ops, knob):
    """against the fp64 attention of the SAME fp32 operands, logits up to ~ +-12: the six-product kernel's error is within 1.25x of the fp32-MFMA
    kernel's; the two-level-P form is reported and bounded."""
    B, H, N, E, dh = 4, 8, 1024, 256, 32
    q = rnd("cq", (B, N, E)) * 2.5
    kv = rnd("ckv", (B, N, 2 * E)) * 1.5
    ref = reference(q, kv[..., :E], kv[..., E:], H, dh, None)
    qd, kvd = q.cuda(), kv.cuda()
    out = {}
    for v in (0, 16 + 3, 16 + 2, 16 + 4):
        knob(v)
        with ops.profile() as rec:
            out[v] = ops.attention(qd, kvd[..., :E], kvd[..., E:], H, dh, N).cpu().double()
        assert ran(rec) == [(v & 15) or None]
    e32, e6, e5, e4 = (float((out[v] - ref).abs().max()) for v in (0, 19, 18, 20))
    r32, r6, r5, r4 = (float((out[v] - ref).pow(2).mean().sqrt()) for v in (0, 19, 18, 20))
    print(f"\nmax|err| vs fp64  fp32-MFMA {e32:.3e}  bf16x6 {e6:.3e}  P on two levels {e5:.3e}  f16x3 {e4:.3e}   rms {r32:.3e} {r6:.3e} {r5:.3e} {r4:.3e}")
    assert e6 <= 1.25 * e32 + 1e-7 and r6 <= 1.1 * r32 + 1e-8
    assert e4 <= 1.25 * e32 + 1e-7 and r4 <= 1.1 * r32 + 1e-8
    assert e5 < 1e-5


@pytest.mark.parametrize("name,mk", [("tiny values", lambda v: v * 1e-4), ("huge values", lambda v: v * 3e3),
                                     ("values grow 1e5 after the first tile", lambda v: torch.cat([v[:, :64] * 1e-3, v[:, 64:] * 100.0], 1)),
                                     ("first tiles zero", lambda v: torch.cat([v[:, :128] * 0, v[:, 128:] * 1e-3], 1))])
def test_f16x3_attention_keeps_its_accuracy_at_every_value_scale(ops, knob, name, mk):
    """the V operand is scaled by the block (first tile -> power of two; a later tile that outgrows the headroom sets a new scale, the accumulators follow exactly):
    against fp64, relative to the output's own scale, within 1.25x of the fp32-MFMA kernel."""
    B, H, N, E, dh = 2, 8, 1024, 256, 32
    q = rnd("vq", (B, N, E)) * 2.0
    k = rnd("vk", (B, N, E))
    v = mk(rnd("vv", (B, N, E)))
    ref = reference(q, k, v, H, dh, None)
    out = {}
    for kn in (0, 16 + 4):
        knob(kn)
        with ops.profile() as rec:
            out[kn] = ops.attention(q.cuda(), k.cuda(), v.cuda(), H, dh, N).cpu().double()
        assert ran(rec) == [(kn & 15) or None]
    sc = float(ref.pow(2).mean().sqrt())
    e32, e4 = float((out[0] - ref).abs().max()) / sc, float((out[20] - ref).abs().max()) / sc
    r32, r4 = float((out[0] - ref).pow(2).mean().sqrt()) / sc, float((out[20] - ref).pow(2).mean().sqrt()) / sc
    print(f"\n{name}: relative max error  fp32-MFMA {e32:.3e}  f16x3 {e4:.3e}   rms {r32:.3e} {r4:.3e}")
    assert bool(torch.isfinite(out[20]).all()) and e4 <= 1.25 * e32 + 1e-7 and r4 <= 1.1 * r32 + 1e-8


def test_split_attention_fully_masked_rows_and_partly_masked_tiles(ops, knob):
    """every key masked -> NaN like the reference (0 / 0); a mask that blanks whole 64-key tiles (the first two and one in the middle) == the reference."""
    B, H, N, E, dh = 2, 8, 1024, 256, 32
    q, kv = rnd("nq3", (B, N, E)), rnd("nkv3", (B, N, 2 * E))
    mask = torch.zeros((B, N), dtype=torch.bool)
    mask[0, :128] = True; mask[0, 512:576] = True; mask[1, 960:] = True; mask[1, 3::2] = True
    ref = reference(q, kv[..., :E], kv[..., E:], H, dh, mask).float()
    for kn in (16 + 3, 16 + 4):
        knob(kn)
        o = ops.attention(q.cuda(), kv.cuda()[..., :E], kv.cuda()[..., E:], H, dh, N, mask=torch.ones((B, N), dtype=torch.uint8, device="cuda"))
        assert bool(torch.isnan(o).all())
        o = ops.attention(q.cuda(), kv.cuda()[..., :E], kv.cuda()[..., E:], H, dh, N, mask=mask.to(torch.uint8).cuda())
        assert maxabs(o.cpu(), ref) < 5e-6


def test_split_attention_dispatch_rule(ops, knob):
    """default knob: big launches only (>= 512 blocks of 128 queries), d_head 32, S % 64 == 0; deterministic."""
    from synergize_motion_appearance_amd import lib
    Lb = lib.load()
    assert ops.set_tuning("attn_bf3", 4) == 4 and Lb.smx_attention_f32_uses_bf3(60, 8, 1024, 1024, 32) == 4          # the default: the f16x3 kernel
    knob(3)
    assert Lb.smx_attention_f32_uses_bf3(60, 8, 1024, 1024, 32) == 3 and Lb.smx_attention_f32_uses_bf3(4, 8, 1024, 1024, 32) == 0
    assert Lb.smx_attention_f32_uses_bf3(60, 8, 1024, 1024, 4) == 0 and Lb.smx_attention_f32_uses_bf3(60, 8, 1024, 1056, 32) == 0
    knob(0)
    assert Lb.smx_attention_f32_uses_bf3(60, 8, 1024, 1024, 32) == 0
    knob(3)
    q, kv = rnd("dq3", (64, 1024, 256)).cuda(), rnd("dkv3", (64, 1024, 512)).cuda()
    with ops.profile() as rec:
        a = ops.attention(q, kv[..., :256], kv[..., 256:], 8, 32, 1024)
    assert ran(rec) == [3]
    assert torch.equal(a, ops.attention(q, kv[..., :256], kv[..., 256:], 8, 32, 1024))


# ---- the fused AttnBlock core (one head of d = 256, V transposed) in the f16x3 arithmetic: attnblock_f16_kernel ----------------------------------------------
def _attnblock(ops, qk, vt, C_, N, sc):
    from synergize_motion_appearance_amd import lib as L_
    B = qk.shape[0]
    o = torch.empty((B, N, C_), device="cuda")
    L_.check(L_.load().smx_attnblock_f32(qk.data_ptr(), 2 * C_, N * 2 * C_, qk.data_ptr() + 4 * C_, 2 * C_, N * 2 * C_, vt.data_ptr(), N, C_ * N,
                                         o.data_ptr(), C_, N * C_, B, N, N, C_, sc, ops._stream()), "smx_attnblock_f32")
    return o


@pytest.mark.parametrize("name,mk", [("plain", lambda v: v), ("one key dominates", lambda v: v), ("tiny values", lambda v: v * 1e-4), ("huge values", lambda v: v * 3e3),
                                     ("values grow 1e5 after the first tile", lambda v: torch.cat([v[..., :32] * 1e-3, v[..., 32:] * 100.0], -1)),
                                     ("first tiles zero", lambda v: torch.cat([v[..., :64] * 0, v[..., 64:] * 1e-3], -1))])
def test_f16x3_attnblock_core(ops, knob, name, mk):
    """softmax(q k^T / 4) v for ONE head of d = 256 (archs/vqgan_arch.py:229-253) against fp64: the f16x3 kernel's error, relative to the output's scale, is
    within 1.25x of the fp32-MFMA kernel's (attnblock32_kernel), for values of every scale (V^T is scaled by the block; re-staged when a tile outgrows it)."""
    C_, N, B = 256, 1024, 2
    qk = torch.zeros((B, N, 2 * C_))
    qk[..., :C_] = rnd("ab_q", (B, N, C_)); qk[..., C_:] = rnd("ab_k", (B, N, C_))
    if name == "one key dominates":
        qk[:, -1, C_:] *= 6.0
    vt = mk(rnd("ab_v", (B, C_, N)))
    sc = 0.25
    ref = torch.bmm(torch.softmax(torch.bmm(qk[..., :C_].double(), qk[..., C_:].double().transpose(1, 2)) * sc, dim=2), vt.double().transpose(1, 2))
    qkc, vtc = qk.cuda().contiguous(), vt.cuda().contiguous()
    out = {}
    for kn in (0, 16 + 4):
        knob(kn)
        out[kn] = _attnblock(ops, qkc, vtc, C_, N, sc).cpu().double()
    s_ = float(ref.pow(2).mean().sqrt())
    e32, e4 = float((out[0] - ref).abs().max()) / s_, float((out[20] - ref).abs().max()) / s_
    r32, r4 = float((out[0] - ref).pow(2).mean().sqrt()) / s_, float((out[20] - ref).pow(2).mean().sqrt()) / s_
    print(f"\n{name}: relative max error  fp32-MFMA {e32:.3e}  f16x3 {e4:.3e}   rms {r32:.3e} {r4:.3e}")
    assert not torch.equal(out[0], out[20])                                                   # two different kernels ran
    assert bool(torch.isfinite(out[20]).all()) and e4 <= 1.25 * e32 + 1e-7 and r4 <= 1.1 * r32 + 1e-8
    assert float((out[20] - ref).abs().max()) < 3e-5 * max(float(ref.abs().max()), 1e-30)

