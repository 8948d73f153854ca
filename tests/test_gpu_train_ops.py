"""SURVEY row N2 (BASELINE configs[4]): every backward kernel of the training step against torch.autograd on the CPU for the
same op (the reference's own backward IS torch.autograd over these ATen ops).  Each test runs a differentiable wrapper of
`train_ops` on the MI355X through the C ABI, seeds the output gradient with a fixed random tensor, runs the tape, and compares the
gradient of every input and parameter.  Tolerances: 2e-4 relative to the largest reference gradient entry unless stated."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from synergize_motion_appearance_amd.synth import synth_input

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    assert torch.cuda.is_available(), "needs an MI355X"
    from synergize_motion_appearance_amd import train_ops
    return train_ops


def rnd(name, shape, scale=1.0):
    return synth_input(name, shape) * scale


def mk_tape(params):
    from synergize_motion_appearance_amd.tape import Tape
    P = {k: v.cuda().contiguous() for k, v in params.items()}
    G = {k: torch.zeros_like(v) for k, v in P.items()}
    return Tape(P, G)


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


CONV_BWD = [
    # B, Cin, Cout, H, k, stride, pad, up2, act, res, tag
    (2, 64, 64, 16, 3, 1, 1, False, 0, False, "3x3"),
    (2, 64, 128, 16, 3, 1, 1, False, 1, False, "3x3 relu"),
    (2, 64, 128, 16, 3, 1, 1, False, 0, True, "3x3 + residual"),
    (1, 128, 64, 16, 1, 1, 0, False, 0, False, "1x1"),
    (2, 2, 32, 16, 3, 1, 1, False, 0, False, "3x3 2->32 (scalar gather)"),
    (2, 2, 128, 16, 7, 1, 3, False, 1, False, "7x7 pad 3 relu"),
    (2, 35, 15, 24, 7, 1, 0, False, 0, False, "7x7 valid (kp head)"),
    (2, 160, 126, 16, 3, 1, 1, False, 2, False, "3x3 160->126 lrelu (odd Cout)"),
    (2, 32, 32, 32, 3, 2, 0, False, 0, False, "Downsample pad(0,1,0,1) stride 2"),
    (2, 64, 64, 8, 3, 1, 1, True, 0, False, "Upsample nearest x2 + conv"),
    (2, 128, 3, 16, 3, 1, 1, False, 0, False, "3x3 128->3 (image head)"),
    (2, 256, 2, 16, 3, 1, 1, False, 0, False, "3x3 256->2 (RefineFlow head)"),
    # Wo % 32 == 0 and 64-multiple channels: the weight gradient runs on the region kernel (train_wgrad_region.hip)
    (2, 64, 64, 32, 3, 1, 1, False, 0, False, "region 64->64 @32"),
    (1, 128, 64, 64, 3, 1, 1, False, 0, True, "region 128->64 @64 + residual (two strips per row)"),
    (3, 64, 128, 16, 3, 1, 1, True, 0, False, "region Upsample x2 + conv 64->128 @16->32"),
]


@pytest.mark.parametrize("B,Cin,Cout,H,k,stride,pad,up2,act,res,tag", CONV_BWD, ids=[c[-1] for c in CONV_BWD])
def test_conv_backward(T, B, Cin, Cout, H, k, stride, pad, up2, act, res, tag):
    x = rnd(f"cbx{tag}", (B, Cin, H, H))
    w = rnd(f"cbw{tag}", (Cout, Cin, k, k), 1.0 / (Cin * k * k) ** 0.5)
    b = rnd(f"cbb{tag}", (Cout,), 0.1)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if up2 else xr
    if stride == 2:
        y = F.conv2d(F.pad(xin, (0, 1, 0, 1)), wr, br, stride=2)
    else:
        y = F.conv2d(xin, wr, br, padding=pad)
    y = F.relu(y) if act == 1 else F.leaky_relu(y, 0.2) if act == 2 else y
    rr = None
    if res:
        rr = rnd(f"cbr{tag}", tuple(y.shape)).requires_grad_()
        y = y + rr
    g = rnd(f"cbg{tag}", tuple(y.shape))
    y.backward(g)

    tp = mk_tape({"w": w, "b": b})
    xd = nhwc(x).cuda()
    rd = None if rr is None else nhwc(rr.detach()).cuda()
    kw = {}
    if stride == 2:
        kw = dict(stride=2, pad=(0, 0), out_hw=(H // 2, H // 2))
    elif pad != k // 2:
        kw = dict(pad=(pad, pad))
    yd = T.conv(tp, xd, "w", "b", up2=up2, act=act, res=rd, **kw)
    assert rel(nchw(yd), y.detach()) < 1e-4
    tp.acc(yd, nhwc(g).cuda())
    tp.backward()
    assert rel(nchw(tp.grad(xd)), xr.grad) < 2e-4, tag
    assert rel(tp.G["w"], wr.grad) < 2e-4, tag
    assert rel(tp.G["b"], br.grad) < 2e-4, tag
    if rr is not None:
        assert rel(nchw(tp.grad(rd)), rr.grad) < 1e-6


def test_conv_weight_gradient_accumulates_and_row_slices(T):
    """a parameter used at two call sites gets the SUM of both weight gradients; a row range of a stacked parameter
    (nn.MultiheadAttention.in_proj_weight[E:2E]) writes only its rows."""
    E = 32
    x1, x2 = rnd("acc_x1", (2, E, 8, 8)), rnd("acc_x2", (1, E, 16, 16))
    w, b = rnd("acc_w", (3 * E, E), 0.2), rnd("acc_b", (3 * E,), 0.1)
    wr, br = w.clone().requires_grad_(), b.clone().requires_grad_()
    sl = slice(E, 2 * E)
    y1 = F.conv2d(x1, wr[sl].view(E, E, 1, 1), br[sl])
    y2 = F.conv2d(x2, wr[sl].view(E, E, 1, 1), br[sl])
    g1, g2 = rnd("acc_g1", tuple(y1.shape)), rnd("acc_g2", tuple(y2.shape))
    (y1 * g1).sum().backward(retain_graph=True)
    (y2 * g2).sum().backward()
    tp = mk_tape({"w": w, "b": b})
    a, c = nhwc(x1).cuda(), nhwc(x2).cuda()
    tp.stop(a), tp.stop(c)
    o1 = T.conv(tp, a, ("w", sl), ("b", sl))
    o2 = T.conv(tp, c, ("w", sl), ("b", sl))
    tp.acc(o1, nhwc(g1).cuda())
    tp.acc(o2, nhwc(g2).cuda())
    tp.backward()
    assert rel(tp.G["w"], wr.grad) < 2e-4 and rel(tp.G["b"], br.grad) < 2e-4
    assert float(tp.G["w"][:E].abs().max()) == 0.0 and float(tp.G["w"][2 * E:].abs().max()) == 0.0


@pytest.mark.parametrize("s,C", [(64, 128), (128, 128), (256, 64)])
def test_patch_embed_and_unpatchify_backward(T, s, C):
    """app_feat_emb_{s} (Rearrange + Linear) and to_app_feat_{s} (Linear + Rearrange), appmotioncodebook_arch.py:218-240."""
    from einops import rearrange
    p = s // 32
    B = 1 if s == 256 else 2
    x = rnd(f"pe_x{s}", (B, C, s, s))
    w1, b1 = rnd(f"pe_w{s}", (256, p * p * C), 1.0 / (p * p * C) ** 0.5), rnd(f"pe_b{s}", (256,), 0.1)
    w2, b2 = rnd(f"up_w{s}", (p * p * C, 256), 1.0 / 16), rnd(f"up_b{s}", (p * p * C,), 0.1)
    xr = x.clone().requires_grad_()
    P = [t.clone().requires_grad_() for t in (w1, b1, w2, b2)]
    tok = F.linear(rearrange(xr, "b c (h p1) (w p2) -> b (h w) (p1 p2 c)", p1=p, p2=p), P[0], P[1])          # [B,1024,256]
    back = rearrange(F.linear(tok, P[2], P[3]), "b (h w) (p1 p2 c) -> b c (h p1) (w p2)", h=32, w=32, p1=p, p2=p, c=C)
    g = rnd(f"pe_g{s}", tuple(back.shape))
    back.backward(g)
    tp = mk_tape({"w1": w1, "b1": b1, "w2": w2, "b2": b2})
    xd = nhwc(x).cuda()
    t = T.conv(tp, xd, "w1", "b1", kind="patch", patch=(p, C))
    assert rel(t.view(B, 1024, 256), tok.detach()) < 1e-4
    o = T.conv(tp, t, "w2", "b2", kind="unpatch", patch=(p, C))
    assert rel(nchw(o), back.detach()) < 1e-4
    tp.acc(o, nhwc(g).cuda())
    tp.backward()
    assert rel(nchw(tp.grad(xd)), xr.grad) < 2e-4
    for name, ref in zip(("w1", "b1", "w2", "b2"), P):
        assert rel(tp.G[name], ref.grad) < 2e-4, name


@pytest.mark.parametrize("C,H,swish", [(64, 16, True), (256, 8, False), (32, 32, True), (128, 24, True)])
def test_groupnorm_backward(T, C, H, swish):
    x = rnd(f"gnb{C}", (2, C, H, H)) * 2 + 0.5
    gm, bt = rnd(f"gng{C}", (C,)) + 1.0, rnd(f"gnbt{C}", (C,), 0.3)
    xr, gr, br = x.clone().requires_grad_(), gm.clone().requires_grad_(), bt.clone().requires_grad_()
    y = F.group_norm(xr, 32, gr, br, eps=1e-6)
    y = y * torch.sigmoid(y) if swish else y
    g = rnd(f"gngr{C}", tuple(y.shape))
    y.backward(g)
    tp = mk_tape({"g": gm, "b": bt})
    xd = nhwc(x).cuda()
    yd = T.groupnorm(tp, xd, "g", "b", swish=swish)
    assert rel(nchw(yd), y.detach()) < 1e-4
    tp.acc(yd, nhwc(g).cuda())
    tp.backward()
    assert rel(nchw(tp.grad(xd)), xr.grad) < 3e-4
    assert rel(tp.G["g"], gr.grad) < 3e-4 and rel(tp.G["b"], br.grad) < 3e-4


@pytest.mark.parametrize("E", [32, 256])
def test_layernorm_pos_backward(T, E):
    B = 2
    x = rnd(f"lnx{E}", (B, 1024, E)) * 1.5 + 0.2
    gm, bt, pos = rnd(f"lng{E}", (E,)) + 1.0, rnd(f"lnb{E}", (E,), 0.2), rnd(f"lnp{E}", (1024, E), 0.02)
    xr, gr, br, pr = (t.clone().requires_grad_() for t in (x, gm, bt, pos))
    y = F.layer_norm(xr, (E,), gr, br, eps=1e-5)
    yp = y + pr
    g1, g2 = rnd(f"lng1{E}", tuple(y.shape)), rnd(f"lng2{E}", tuple(y.shape))
    (y * g1).sum().backward(retain_graph=True)
    (yp * g2).sum().backward()
    tp = mk_tape({"g": gm, "b": bt, "pos": pos})
    xd = x.cuda()
    yd, ypd = T.layernorm(tp, xd, "g", "b", pos=T.leaf(tp, "pos"))
    assert rel(ypd, yp.detach()) < 1e-4
    tp.acc(yd, g1.cuda())
    tp.acc(ypd, g2.cuda())
    tp.backward()
    assert rel(tp.grad(xd), xr.grad) < 3e-4
    assert rel(tp.G["g"], gr.grad) < 3e-4 and rel(tp.G["b"], br.grad) < 3e-4 and rel(tp.G["pos"], pr.grad) < 3e-4


def _mha_ref(q, k, v, H, dh, mask=None):
    B, Lq, E = q.shape
    S = k.shape[1]
    qh = q.view(B, Lq, H, dh).transpose(1, 2) * dh ** -0.5
    kh, vh = k.view(B, S, H, dh).transpose(1, 2), v.view(B, S, H, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :].bool(), float("-inf"))
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Lq, E)


@pytest.mark.parametrize("dh,masked,mfma", [(32, False, 1), (32, True, 1), (32, True, 0), (4, False, 1)])
def test_self_attention_backward(T, dh, masked, mfma):
    """d_head 32: the fp32-MFMA backward kernels (attn_bwd_q_mfma / attn_bwd_kv_mfma) and, with the knob off, the per-thread VALU
    kernels they replaced (still the path for shapes off the 128-row grid); d_head 4: VALU."""
    from synergize_motion_appearance_amd import lib as Lm
    Lm.load().smx_set_tuning(b"attn_bwd_mfma", mfma)
    try:
        _self_attention_backward(T, dh, masked)
    finally:
        Lm.load().smx_set_tuning(b"attn_bwd_mfma", 1)


def _self_attention_backward(T, dh, masked):
    B, H, Lq = 2, 8, 1024
    E = H * dh
    q, k, v = (rnd(f"sa{n}{dh}", (B, Lq, E)) for n in "qkv")
    mask = None
    if masked:
        mask = (synth_input("samask", (B, Lq)) > 0.8).to(torch.uint8)
    qr, kr, vr = (t.clone().double().requires_grad_() for t in (q, k, v))
    o = _mha_ref(qr, kr, vr, H, dh, mask)
    g = rnd(f"sag{dh}", (B, Lq, E))
    o.backward(g.double())
    tp = mk_tape({})
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    od = T.attention(tp, qd, kd, vd, H, dh, Lq, mask=None if mask is None else mask.cuda())
    assert rel(od, o.detach()) < 1e-4
    tp.acc(od, g.cuda())
    tp.backward()
    for a, r in ((qd, qr), (kd, kr), (vd, vr)):
        assert rel(tp.grad(a), r.grad) < 2e-4


@pytest.mark.parametrize("dh,S", [(32, 512), (4, 768)])
def test_cross_attention_backward_shared_context(T, dh, S):
    """context = projected codebook rows shared by the batch: dK / dV are batch sums and land in the first S rows of d ctx."""
    B, H, Lq, K = 3, 8, 1024, 1024
    E = H * dh
    q, ctx = rnd(f"caq{dh}", (B, Lq, E)), rnd(f"cac{dh}", (K, 2 * E))
    qr, cr = q.clone().double().requires_grad_(), ctx.clone().double().requires_grad_()
    o = _mha_ref(qr, cr[:S, :E].expand(B, S, E), cr[:S, E:].expand(B, S, E), H, dh)
    g = rnd(f"cag{dh}", (B, Lq, E))
    o.backward(g.double())
    tp = mk_tape({})
    qd, cd = q.cuda(), ctx.cuda()
    od = T.attention(tp, qd, None, None, H, dh, S, ctx=cd)
    assert rel(od, o.detach()) < 1e-4
    tp.acc(od, g.cuda())
    tp.backward()
    assert rel(tp.grad(qd), qr.grad) < 2e-4
    assert rel(tp.grad(cd), cr.grad) < 2e-4
    assert float(tp.grad(cd)[S:].abs().max()) == 0.0


def test_attnblock_core_backward(T):
    """AttnBlock archs/vqgan_arch.py:236-249 at its real size (1024 tokens x 256 channels)."""
    B, C, H = 2, 256, 32
    q, k, v = (rnd(f"ab{n}", (B, C, H, H)) for n in "qkv")
    qr, kr, vr = (t.clone().double().requires_grad_() for t in (q, k, v))
    N = H * H
    w_ = torch.softmax(torch.bmm(qr.reshape(B, C, N).permute(0, 2, 1), kr.reshape(B, C, N)) * C ** -0.5, dim=2)
    h = torch.bmm(vr.reshape(B, C, N), w_.permute(0, 2, 1)).reshape(B, C, H, H)
    g = rnd("abg", (B, C, H, H))
    h.backward(g.double())
    tp = mk_tape({})
    qd, kd, vd = nhwc(q).cuda(), nhwc(k).cuda(), nhwc(v).cuda()
    hd = T.attn_core(tp, qd, kd, vd, C ** -0.5)
    assert rel(nchw(hd), h.detach()) < 1e-4
    tp.acc(hd, nhwc(g).cuda())
    tp.backward()
    for a, r in ((qd, qr), (kd, kr), (vd, vr)):
        assert rel(nchw(tp.grad(a)), r.grad) < 2e-4


@pytest.mark.parametrize("C,s,with_occ", [(256, 32, True), (128, 64, True), (64, 128, False), (64, 256, True)])
def test_warp_backward(T, C, s, with_occ):
    """deform_input + occlude_input (appmotioncodebook_arch.py:349-362): grid_sampler_2d_backward w.r.t. the features AND the
    64x64 flow (through the align_corners=True resize), and the occlusion map's gradient.  Smooth features (SURVEY appendix B:
    white-noise features amplify fp32 coordinate rounding), flow partly out of frame."""
    B = 2
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, s), torch.linspace(-1, 1, s), indexing="ij")
    feat = torch.stack([torch.sin((c % 7 + 1) * xx + 0.3 * c) * torch.cos((c % 5 + 1) * yy) for c in range(C)]).unsqueeze(0).repeat(B, 1, 1, 1)
    feat = feat + 0.05 * rnd(f"wbf{s}", (B, C, s, s))
    gy, gx = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="ij")
    flow = torch.stack([gx, gy], -1).unsqueeze(0).repeat(B, 1, 1, 1) * 1.05 + 0.08 * rnd(f"wbfl{s}", (B, 64, 64, 2))
    occ = torch.sigmoid(rnd(f"wbo{s}", (B, 1, 64, 64)))
    fr, flr, ocr = feat.clone().requires_grad_(), flow.clone().requires_grad_(), occ.clone().requires_grad_()
    d = flr
    if s != 64:
        d = F.interpolate(flr.permute(0, 3, 1, 2), size=(s, s), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    out = F.grid_sample(fr, d, align_corners=True)
    if with_occ:
        o = ocr if s == 64 else F.interpolate(ocr, size=(s, s), mode="bilinear", align_corners=True)
        out = out * o
    g = rnd(f"wbg{s}", tuple(out.shape))
    out.backward(g)
    tp = mk_tape({})
    fd, fld, od = nhwc(feat).cuda(), flow.cuda().contiguous(), occ.view(B, 64, 64).cuda().contiguous()
    outd = T.warp(tp, fd, fld, od if with_occ else None)
    assert rel(nchw(outd), out.detach()) < 2e-4
    tp.acc(outd, nhwc(g).cuda())
    tp.backward()
    assert rel(nchw(tp.grad(fd)), fr.grad) < 5e-4
    assert rel(tp.grad(fld), flr.grad) < 2e-3          # (s-1)/2 pixels per unit of flow times the feature gradient: fp32 noise scales with s
    if with_occ:
        assert rel(tp.grad(od).view(B, 1, 64, 64), ocr.grad) < 5e-4


@pytest.mark.parametrize("C,hin,hout", [(32, 32, 64), (192, 128, 64), (15, 64, 32), (128, 256, 32)])
def test_resize_backward(T, C, hin, hout):
    x = rnd(f"rz{C}{hin}", (2, C, hin, hin))
    xr = x.clone().requires_grad_()
    y = F.interpolate(xr, size=(hout, hout), mode="bilinear", align_corners=True)
    g = rnd(f"rzg{C}{hin}", tuple(y.shape))
    y.backward(g)
    tp = mk_tape({})
    xd = nhwc(x).cuda()
    yd = T.resize(tp, xd, hout, hout)
    tp.acc(yd, nhwc(g).cuda())
    tp.backward()
    assert rel(nchw(tp.grad(xd)), xr.grad) < 2e-4


@pytest.mark.parametrize("D,Ks", [(32, 512), (256, 768)])
def test_quantize_backward_straight_through_and_codebook_loss(T, D, Ks):
    """archs/vqgan_arch.py:60-76: loss = beta mse(sg(z_q), z) + mse(z_q, sg(z)); z_q = z + sg(z_q - z)."""
    B, beta = 2, 0.25
    z = rnd(f"vqb_z{D}", (B, D, 32, 32))
    cb = rnd(f"vqb_cb{D}", (1024, D))
    zr, cr = z.clone().requires_grad_(), cb.clone().requires_grad_()
    zf = zr.permute(0, 2, 3, 1).reshape(-1, D)
    d = (zf ** 2).sum(1, keepdim=True) + (cr[:Ks] ** 2).sum(1) - 2 * zf @ cr[:Ks].t()
    idx = d.argmin(1)
    zq = cr[:Ks][idx].view(B, 32, 32, D)
    zp = zr.permute(0, 2, 3, 1)
    loss = beta * ((zq.detach() - zp) ** 2).mean() + ((zq - zp.detach()) ** 2).mean()
    zq_st = zp + (zq - zp).detach()
    g = rnd(f"vqb_g{D}", (B, 32, 32, D))
    ((zq_st * g).sum() + 3.0 * loss).backward()
    tp = mk_tape({"cb": cb})
    zd = nhwc(z).cuda()
    zqd, lossd, st = T.quantize(tp, zd, "cb", Ks, beta)
    assert torch.equal(st["min_encoding_indices"].view(-1).cpu(), idx)
    assert abs(float(lossd) - float(loss)) < 1e-5 * abs(float(loss))
    tp.acc(zqd, g.cuda())
    tp.acc(lossd, torch.full((1,), 3.0, device="cuda"))
    tp.backward()
    assert rel(nchw(tp.grad(zd)), zr.grad) < 2e-4
    assert rel(tp.G["cb"], cr.grad) < 2e-4


def test_flow_chain_sft_l1_and_plumbing_backward(T):
    """flow -> residual (pixels) -> update with [dflow | docc] -> sigmoid chain (appmotioncodebook_arch.py:577-601), SFT modulation
    (:49-51), L1 losses, cat / slice / scale / weighted_sum."""
    B = 2
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="xy")
    grid = torch.stack([yy, xx], -1)[None]
    flow = grid + 0.05 * rnd("fc_flow", (B, 64, 64, 2))
    r = rnd("fc_r", (B, 64, 64, 3))
    occ0 = torch.sigmoid(rnd("fc_o", (B, 64, 64)))
    tgt = rnd("fc_t", (B, 64, 64, 2))
    fr, rr, orr = flow.clone().requires_grad_(), r.clone().requires_grad_(), occ0.clone().requires_grad_()
    res = (fr - grid) * 31.5
    m_com = fr + rr[..., :2] / 31.5
    occ = torch.sigmoid(orr + rr[..., 2])
    l1 = 32.0 * (res / 31.5 - tgt).abs().mean()
    total = 0.7 * l1 + 2.0 * (m_com * tgt).sum() + (occ * occ0).sum()
    total.backward()
    tp = mk_tape({})
    fd, rd, od = flow.cuda(), r.cuda(), occ0.cuda()
    resd = T.flow_to_residual(tp, fd)
    assert rel(resd, res.detach()) < 1e-4
    mc, _, oc = T.flow_occ_update(tp, fd, rd, od)
    assert rel(mc, m_com.detach()) < 1e-6 and rel(oc, occ.detach()) < 1e-6
    l1d = T.l1_loss(tp, T.scale(tp, resd, 1 / 31.5), tgt.cuda(), 32.0)
    assert abs(float(l1d) - float(l1)) < 1e-5 * float(l1)
    tot = T.weighted_sum(tp, [(l1d, 0.7)])
    tp.acc(tot, torch.ones(1, device="cuda"))
    tp.acc(mc, 2.0 * tgt.cuda())
    tp.acc(oc, occ0.cuda())
    tp.backward()
    assert rel(tp.grad(fd), fr.grad) < 2e-4 and rel(tp.grad(rd), rr.grad) < 2e-4 and rel(tp.grad(od), orr.grad) < 2e-4
    # SFT + cat / slice
    C = 64
    enc, dec, sc, sh = (rnd(f"sft_{n}", (B, 8, 8, C)) for n in ("e", "d", "s", "h"))
    er, dr, sr, hr = (t.clone().requires_grad_() for t in (enc, dec, sc, sh))
    cat = torch.cat([er, dr], -1)
    out = cat[..., C:] + 0.8 * (cat[..., C:] * sr + hr) + cat[..., :C]
    g = rnd("sft_g", tuple(out.shape))
    out.backward(g)
    tp = mk_tape({})
    ed, dd, sd, hd = enc.cuda(), dec.cuda(), sc.cuda(), sh.cuda()
    catd = T.cat(tp, [ed, dd])
    o1 = T.sft_combine(tp, T.slice_ch(tp, catd, C, 2 * C), sd, hd, 0.8)
    e2 = T.slice_ch(tp, catd, 0, C)
    tp.acc(o1, g.cuda(), owned=False)
    tp.acc(e2, g.cuda(), owned=False)
    tp.backward()
    for a, ref in ((ed, er), (dd, dr), (sd, sr), (hd, hr)):
        assert rel(tp.grad(a), ref.grad) < 1e-5


def test_gelu_and_adam_and_ema(T):
    from synergize_motion_appearance_amd import lib as L
    from synergize_motion_appearance_amd.tape import _stream
    x = rnd("gelu_x", (2, 8, 8, 64)) * 2
    xr = x.clone().requires_grad_()
    y = F.gelu(xr)
    g = rnd("gelu_g", tuple(y.shape))
    y.backward(g)
    tp = mk_tape({})
    xd = x.cuda()
    yd = T.act(tp, xd, T.ACT_GELU)
    assert rel(yd, y.detach()) < 1e-5
    tp.acc(yd, g.cuda())
    tp.backward()
    assert rel(tp.grad(xd), xr.grad) < 1e-5
    # Adam (torch.optim.Adam defaults of options/train.yml: lr 8e-5, betas (0.9, 0.99)) over three steps, then EMA
    p0 = rnd("adam_p", (4097,))
    pr = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pr], lr=8e-5, betas=(0.9, 0.99), weight_decay=0)
    pd, m, v = p0.cuda(), torch.zeros(4097, device="cuda"), torch.zeros(4097, device="cuda")
    lib = L.load()
    for t in range(1, 4):
        gr = rnd(f"adam_g{t}", (4097,))
        pr.grad = gr.clone()
        opt.step()
        L.check(lib.smx_adam_step_f32(pd.data_ptr(), gr.cuda().data_ptr(), m.data_ptr(), v.data_ptr(), 4097, 8e-5, 0.9, 0.99, 1e-8, 0.0, t, 1.0,
                                      _stream()), "adam")
    assert float((pd.cpu() - pr.detach()).abs().max()) < 1e-7
    ema = rnd("ema_e", (4097,)).cuda()
    ref = ema.cpu() * 0.995 + pd.cpu() * 0.005
    L.check(lib.smx_ema_f32(ema.data_ptr(), pd.data_ptr(), 4097, 0.995, _stream()), "ema")
    assert float((ema.cpu() - ref).abs().max()) < 1e-6


BF16_CONV = [c for c in CONV_BWD if c[-1] in ("3x3", "3x3 relu", "1x1", "7x7 pad 3 relu", "7x7 valid (kp head)", "3x3 160->126 lrelu (odd Cout)",
                                              "Downsample pad(0,1,0,1) stride 2", "Upsample nearest x2 + conv", "3x3 128->3 (image head)",
                                              "region 64->64 @32", "region Upsample x2 + conv 64->128 @16->32")]


@pytest.mark.parametrize("B,Cin,Cout,H,k,stride,pad,up2,act,res,tag", BF16_CONV, ids=[c[-1] for c in BF16_CONV])
def test_conv_backward_bf16_compute_mode(T, B, Cin, Cout, H, k, stride, pad, up2, act, res, tag):
    """Tape(mfma16=True): forward, data gradient and weight gradient (smx_wgrad_mfma16_f32) on the bf16 MFMA == the fp32 op evaluated on
    bf16-ROUNDED operands (x, w for the forward; g, w for the data gradient; g, x for the weight gradient) with fp32 accumulation --
    i.e. exactly the rounding torch.autocast(bfloat16) applies to F.conv2d and its backward, to fp32 summation-order tolerance (2e-4).
    The stride-2 data gradient stays on the fp32 zero-insert gather (no bf16 form): compared against the unrounded operands there."""
    from synergize_motion_appearance_amd.tape import Tape
    r16 = lambda t: t.to(torch.bfloat16).float()           # noqa: E731
    x = rnd(f"cbx{tag}", (B, Cin, H, H))
    w = rnd(f"cbw{tag}", (Cout, Cin, k, k), 1.0 / (Cin * k * k) ** 0.5)
    b = rnd(f"cbb{tag}", (Cout,), 0.1)

    def ref_conv(xi, wi):
        xin = F.interpolate(xi, scale_factor=2, mode="nearest") if up2 else xi
        y = F.conv2d(F.pad(xin, (0, 1, 0, 1)), wi, b, stride=2) if stride == 2 else F.conv2d(xin, wi, b, padding=pad)
        return F.relu(y) if act == 1 else F.leaky_relu(y, 0.2) if act == 2 else y
    y_ref = ref_conv(r16(x), r16(w))
    g = rnd(f"cbg{tag}", tuple(y_ref.shape))
    # the activation derivative is taken at the (bf16-product) output; the gradient that enters the contractions is g * act'(y)
    gm = g * ((y_ref > 0).float() if act == 1 else torch.where(y_ref > 0, torch.ones(()), torch.full((), 0.2)) if act == 2 else 1.0)

    def lin(xi, wi):                                       # the linear part only (no bias / activation): its autograd gives both adjoints
        xin = F.interpolate(xi, scale_factor=2, mode="nearest") if up2 else xi
        return F.conv2d(F.pad(xin, (0, 1, 0, 1)), wi, None, stride=2) if stride == 2 else F.conv2d(xin, wi, None, padding=pad)
    xa = r16(x).requires_grad_()                           # dW = adjoint wrt w at (round(x), round(gm))
    wa = w.clone().requires_grad_()
    lin(xa, wa).backward(r16(gm))
    dW = wa.grad.clone()
    xb, wb = x.clone().requires_grad_(), r16(w).requires_grad_()   # dX = adjoint wrt x at (round(w), round(gm)); stride 2: unrounded
    if stride == 2:
        wb = w.clone().requires_grad_()
        lin(xb, wb).backward(gm)
    else:
        lin(xb, wb).backward(r16(gm))
    dX = xb.grad

    P = {"w": w.cuda().contiguous(), "b": b.cuda().contiguous()}
    tp = Tape(P, {n: torch.zeros_like(v) for n, v in P.items()}, mfma16=True)
    xd = nhwc(x).cuda()
    kw = dict(stride=2, pad=(0, 0), out_hw=(H // 2, H // 2)) if stride == 2 else dict(pad=(pad, pad)) if pad != k // 2 else {}
    from synergize_motion_appearance_amd import ops
    with ops.profile() as rec:
        yd = T.conv(tp, xd, "w", "b", up2=up2, act=act, **kw)
        assert yd.dtype == torch.float32 and rel(nchw(yd), y_ref) < 2e-4, tag
        tp.acc(yd, nhwc(g).cuda())
        tp.backward()
    names = [r[0] for r in rec.rows]
    if k == 3 and stride == 1 and Cin % 64 == 0:
        # forward on the fp32-storage form of the region-direct kernel (smx_conv3x3_mfma16_f32); the data gradient too when C_out % 64 == 0
        assert names.count("conv3x3_mfma16") == (2 if Cout % 64 == 0 else 1), (tag, names)
    assert rel(tp.G["w"], dW) < 2e-4, tag
    assert rel(nchw(tp.grad(xd)), dX) < 2e-4, tag
    assert rel(tp.G["b"], gm.sum((0, 2, 3))) < 2e-4, tag
    # and it is NOT the fp32 result: the rounding is really there (guards against a silent fp32 fallback)
    wf = w.clone().requires_grad_()
    lin(x, wf).backward(gm)
    assert rel(tp.G["w"], wf.grad) > 1e-4, tag


@pytest.mark.parametrize("B,Cin,Cout,H,W,up2,bf16", [(4, 128, 128, 64, 64, 0, 0), (2, 64, 192, 40, 96, 0, 0), (1, 256, 64, 7, 32, 0, 1), (2, 64, 64, 24, 32, 1, 1),
                                                     (5, 64, 64, 3, 32, 0, 0)])
def test_region_weight_gradient_equals_the_generic_kernel(B, Cin, Cout, H, W, up2, bf16):
    """csrc/train_wgrad_region.hip against the generic TN GEMM it replaces (same entry point, `wgrad_region` knob 1 / 0) on operands that are
    CHANNEL SLICES of wider buffers (ld > C, 16-byte-aligned offsets), non-square grids, runs that cross strip / image boundaries, accumulate +
    alpha, and the bias gradient from the same pass.  Both sum the same products in different orders: 1e-5 relative (bf16 mode: the same
    rounded operands, fp32 accumulation)."""
    import ctypes as C
    from synergize_motion_appearance_amd import lib as L
    lib = L.load()
    Hin, Win = (H // 2, W // 2) if up2 else (H, W)
    xw = (rnd("rg_x", (B, Hin, Win, Cin + 8)) * 1.0).cuda()
    dyw = rnd("rg_dy", (B, H, W, Cout + 12)).cuda()
    x, dy = xw[..., 4:4 + Cin], dyw[..., 8:8 + Cout]
    M = B * H * W
    st = torch.cuda.current_stream().cuda_stream
    fn = lib.smx_wgrad_mfma16_f32 if bf16 else lib.smx_wgrad_f32
    outs = []
    try:
        for knob in (1, 0):
            L.check(lib.smx_set_tuning(b"wgrad_region", knob), "knob")
            ms = C.c_int(0)
            n = int(lib.smx_wgrad_conv_ws_floats(1, M, Cout, Cin, Hin, Win, H, W, 3, 3, 1, 1, 1, up2, C.byref(ms)))
            ws = torch.empty(n, device="cuda")
            out = torch.full((Cout, Cin, 3, 3), 0.25, device="cuda")
            bias = torch.full((Cout,), -0.5, device="cuda")
            L.check(fn(dy.data_ptr(), Cout + 12, 0, x.data_ptr(), Cin + 8, 0, 1, M, Cout, Hin, Win, Cin, H, W, 3, 3, 1, 1, 1, up2, ws.data_ptr(), ms.value,
                       out.data_ptr(), 0, 0, 0, 1, 0.5, bias.data_ptr(), st), "wgrad")
            torch.cuda.synchronize()
            outs.append((out.cpu(), bias.cpu(), ms.value))
    finally:
        L.check(lib.smx_set_tuning(b"wgrad_region", 1), "knob")
    assert rel(outs[0][0], outs[1][0]) < 1e-5 and rel(outs[0][1], outs[1][1]) < 1e-5
    # and against autograd (fp32 mode)
    if not bf16:
        xr = x.cpu().permute(0, 3, 1, 2)
        xr = F.interpolate(xr, scale_factor=2, mode="nearest") if up2 else xr
        w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
        F.conv2d(xr, w, padding=1).backward(dy.cpu().permute(0, 3, 1, 2))
        assert rel(outs[0][0] - 0.25, 0.5 * w.grad) < 2e-4
        assert rel(outs[0][1] + 0.5, 0.5 * dy.cpu().sum((0, 1, 2))) < 2e-4


def test_batched_split_reduce_equals_the_per_layer_launches(T):
    """train_ops.ReducePlan: the step after the recording one defers every weight gradient's split reduce and finishes a backward piece's
    items with ONE smx_wgrad_reduce_batch launch per wave -- gradients bit-identical to the per-layer launches, including a parameter used at
    two call sites (second wave), the bias gradients, a backward run in two pieces, and a changed call sequence raising instead of
    silently reducing the wrong workspace."""
    from synergize_motion_appearance_amd.tape import Tape
    from synergize_motion_appearance_amd import lib as L
    P = {"w1": rnd("rp_w1", (64, 64, 3, 3), 0.05), "b1": rnd("rp_b1", (64,), 0.1), "w2": rnd("rp_w2", (128, 64, 3, 3), 0.05),
         "b2": rnd("rp_b2", (128,), 0.1), "w3": rnd("rp_w3", (32, 128), 0.1), "b3": rnd("rp_b3", (32,), 0.1)}
    P = {k: v.cuda().contiguous() for k, v in P.items()}
    x = nhwc(rnd("rp_x", (2, 64, 32, 32))).cuda()
    gy = nhwc(rnd("rp_g", (2, 32, 32, 32))).cuda()

    def run(plan, pieces=1, extra=False):
        G = run.G if plan is not None else {k: torch.zeros_like(v) for k, v in P.items()}
        for v in G.values():
            v.zero_()
        tp = Tape(P, G, reduce_plan=plan)
        tp.stop(x)
        h = T.conv(tp, x, "w1", "b1", act=1)
        h = T.conv(tp, h, "w1", "b1")                      # the same parameter again: its reduce goes to the second wave
        cut = tp.mark()
        h = T.conv(tp, h, "w2", "b2", act=2)
        if extra:
            h = T.conv(tp, h, "w3", "b3")
        y = T.conv(tp, h, "w3", "b3")
        tp.acc(y, gy.clone())
        if pieces == 2:
            tp.backward(stop_at=cut)
        tp.backward()
        if plan is not None:
            plan.finish()                                  # what the trainer does at the end of a step that ran through (only such a recording is replayed)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in G.items()}
    ref = run(None)
    for pieces in (1, 2):
        plan = T.ReducePlan()
        run.G = {k: torch.zeros_like(v) for k, v in P.items()}
        rec = run(plan, pieces)                               # records (reduces at once)
        assert not plan.replay and len(plan.segs) == pieces
        rep = run(plan, pieces)                               # replays: deferred, one batch launch per wave and piece
        assert plan.replay
        rep2 = run(plan, pieces)
        for k in P:
            assert torch.equal(rec[k], ref[k]) and torch.equal(rep[k], ref[k]) and torch.equal(rep2[k], ref[k]), (pieces, k)
    assert len(plan.segs[1]["waves"]) == 2                # piece 2 (recorded last = the first convs) holds w1 twice
    with pytest.raises(L.SmxError):
        Tp = T.ReducePlan()
        run.G = {k: torch.zeros_like(v) for k, v in P.items()}
        run(Tp)
        run(Tp, extra=True)


def test_batched_weight_packing_equals_the_per_layer_launches():
    """smx_pack_batch (one launch, a device table of items) against smx_pack_weight_f32 / _bf16 / smx_pack_winograd_u_f32 layer by layer:
    bit for bit, for both modes, odd sizes and totals that end inside a 1024-element block -- through train_ops.PackPlan, the way the
    training step drives it (first tape records, second tape refreshes everything in one launch after the parameters moved)."""
    assert torch.cuda.is_available(), "needs an MI355X"
    from synergize_motion_appearance_amd import train_ops as T
    from synergize_motion_appearance_amd.tape import Tape
    g = torch.Generator().manual_seed(5)
    shapes = {"a": (64, 32, 3, 3), "b": (17, 35, 7, 7), "c": (256, 128, 1, 1), "d": (128, 128, 3, 3), "e": (3, 8, 3, 3), "f": (96, 64, 4, 4)}
    P = {k: torch.randn(*v, generator=g).cuda() for k, v in shapes.items()}
    G = {k: torch.zeros_like(v) for k, v in P.items() if k != "e"}              # "e" has no gradient slot: frozen, packed once
    plan = T.PackPlan()

    def ask(tp):
        out = {}
        for k, (co, ci, kh, kw) in shapes.items():
            for mode in (0, 1):
                out[k, "f32", mode] = T._packed(tp, k, mode, co, ci, kh, kw)
                out[k, "bf16", mode] = T._packed16(tp, k, mode, co, ci, kh, kw)
                if kh == 3 and kw == 3 and (ci if mode == 0 else co) % 8 == 0:
                    out[k, "u", mode] = T._packed_u(tp, k, mode, co, ci)
        return out

    first = ask(Tape(P, G, plan=plan))                                          # owner set by begin(); every request packs by itself and is recorded
    assert plan._n == 0 and len(plan.items) == len(first)
    for k in P:
        if k != "e":
            P[k].mul_(1.5).add_(0.25)                                           # "the optimiser stepped"
    tp2 = Tape(P, G, plan=plan)                                                 # one launch refreshes all recorded packings
    assert plan._n == len([1 for it in plan.items.values() if not it[9]]) and plan._n < len(plan.items)
    assert all(ck in tp2.packed for ck in plan.items) and set(plan._ready) == set(plan.items)
    batched = {k: v.clone() for k, v in ask(tp2).items()}
    ref = ask(Tape(P, G))                                                       # no plan: the per-layer launches at the same parameter values
    torch.cuda.synchronize()
    assert set(ref) == set(batched)
    for k in ref:
        a, b = ref[k], batched[k]
        assert a.dtype == b.dtype and a.shape == b.shape
        assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b), k
    # a tape over a different parameter dict resets the plan instead of packing from stale pointers
    Tape(dict(P), G, plan=plan)
    assert not plan.items


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,stride,pad,up2,bf16", [(2, 64, 128, 24, 40, 1, 1, 0, 0, 0), (3, 32, 48, 17, 19, 3, 2, 1, 0, 0), (1, 128, 20, 30, 26, 7, 1, 3, 0, 0),
                                                                  (2, 64, 64, 20, 24, 3, 1, 1, 1, 0), (2, 96, 64, 9, 33, 5, 1, 2, 0, 1), (4, 256, 256, 8, 8, 1, 1, 0, 0, 1)])
def test_weight_gradient_buffer_loader_equals_the_gather(B, Cin, Cout, H, W, k, stride, pad, up2, bf16):
    """wgrad_kernel's buffer-load loader (round 5: SGPR resource + slice offset, rows past the split / padded taps as out-of-range offsets; knob
    `gemm_loader`) against the float4 gather on channel SLICES of wider buffers, odd grids, strides, nearest-x2 inputs and tail slices: the same
    products in the same order -- bit-identical, bias gradient included."""
    import ctypes as C
    from synergize_motion_appearance_amd import lib as L
    lib = L.load()
    Hin, Win = (H, W)
    He, We = (2 * H, 2 * W) if up2 else (H, W)
    Ho, Wo = (He + 2 * pad - k) // stride + 1, (We + 2 * pad - k) // stride + 1
    xw = rnd("bl_x", (B, Hin, Win, Cin + 8)).cuda()
    dyw = rnd("bl_dy", (B, Ho, Wo, Cout + 12)).cuda()
    x, dy = xw[..., 4:4 + Cin], dyw[..., 8:8 + Cout]
    M = B * Ho * Wo
    st = torch.cuda.current_stream().cuda_stream
    fn = lib.smx_wgrad_mfma16_f32 if bf16 else lib.smx_wgrad_f32
    outs = []
    try:
        L.check(lib.smx_set_tuning(b"wgrad_region", 0), "knob")
        for knob in (1, 0):
            L.check(lib.smx_set_tuning(b"gemm_loader", knob), "knob")
            ms = C.c_int(0)
            n = int(lib.smx_wgrad_conv_ws_floats(1, M, Cout, Cin, Hin, Win, Ho, Wo, k, k, stride, pad, pad, up2, C.byref(ms)))
            ws = torch.empty(n, device="cuda")
            out = torch.full((Cout, Cin, k, k), 0.25, device="cuda")
            bias = torch.full((Cout,), -0.5, device="cuda")
            L.check(fn(dy.data_ptr(), Cout + 12, 0, x.data_ptr(), Cin + 8, 0, 1, M, Cout, Hin, Win, Cin, Ho, Wo, k, k, stride, pad, pad, up2, ws.data_ptr(), ms.value,
                       out.data_ptr(), 0, 0, 0, 1, 0.5, bias.data_ptr(), st), "wgrad")
            torch.cuda.synchronize()
            outs.append((out.cpu(), bias.cpu()))
    finally:
        L.check(lib.smx_set_tuning(b"gemm_loader", 1), "knob")
        L.check(lib.smx_set_tuning(b"wgrad_region", 1), "knob")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    if not bf16:
        xr = x.cpu().permute(0, 3, 1, 2)
        xr = F.interpolate(xr, scale_factor=2, mode="nearest") if up2 else xr
        w = torch.zeros(Cout, Cin, k, k, requires_grad=True)
        F.conv2d(xr, w, stride=stride, padding=pad).backward(dy.cpu().permute(0, 3, 1, 2))
        assert rel(outs[0][0] - 0.25, 0.5 * w.grad) < 2e-4
