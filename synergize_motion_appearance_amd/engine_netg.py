"""HIP execution plan for AppMotionCompFormer.forward(..., inference=True)
(reference `archs/appmotioncodebook_arch.py:546-764`; rows A7-A13 of SURVEY.md section 8a),
on NHWC activations -- tokens [B,1024,E] ARE the NHWC layout at 32x32, so the reference's
reshape/permute traffic between conv maps and transformer tokens does not exist here.

MI355X-first choices:
  * the source encoder (59 GFLOP, frame-invariant; the reference re-runs it every frame,
    demo.py:130) is computed once per source and cached; warps broadcast it over B frames;
  * every conv / Linear / attention contraction is one implicit-GEMM kernel on the fp32 MFMA;
  * patchify+Linear is a pxp stride-p conv, Linear+un-patchify a 1x1 conv with a
    depth-to-space store; concatenations are written in place by their producers;
  * codebook K / V^T projections are input-independent -> precomputed at pack time for the
    full codebook; a scale's prefix is just a smaller N / K on the same buffers;
  * sibling convs sharing an input (RefineFlow conv1/convo1, SFT scale.0/shift.0, MHA q/k)
    are stacked into one wider GEMM;
  * the redundant third warp per scale (`deform_feat_list`, :615/:714) is not recomputed.
"""
import math

import torch

from . import lib as L
from . import ops
from .manifest import encoder_plan, generator_plan, CHANNELS
from .ops import Conv, ACT_RELU, ACT_LRELU02, ACT_GELU

ATTNBLOCK_FUSED16 = ops._knob("SMX_ATTNBLOCK_FUSED16", 1)
ATTNBLOCK_FUSED32 = ops._knob("SMX_ATTNBLOCK_FUSED32", 1)     # fp32 AttnBlock core as one kernel (0 = QK^T GEMM, softmax_rows, PV GEMM)     # bf16 AttnBlock core as one kernel (0 = QK^T GEMM, softmax_rows, PV GEMM)

_SCALE_K = {32: 1, 64: 2, 128: 3, 256: 4}


class _Res:
    def __init__(self, P, pre):
        self.n1 = (P[pre + ".norm1.weight"], P[pre + ".norm1.bias"])
        self.n2 = (P[pre + ".norm2.weight"], P[pre + ".norm2.bias"])
        self.c1 = Conv.from_torch(P[pre + ".conv1.weight"], P[pre + ".conv1.bias"])
        self.c2 = Conv.from_torch(P[pre + ".conv2.weight"], P[pre + ".conv2.bias"])
        self.sc = Conv.from_torch(P[pre + ".conv_out.weight"], P[pre + ".conv_out.bias"]) if (pre + ".conv_out.weight") in P else None

    def __call__(self, x, out=None):
        """ResBlock.forward archs/vqgan_arch.py:180-191."""
        # GN statistics are their own (read-only) pass; the normalise + swish is applied by the
        # consuming conv's region loader, so no normalised copy of the activation is ever written
        # ... and the statistics themselves come out of the producing conv's epilogue where it is the fused
        # Winograd kernel (want_stats): norm2 never re-reads h, the next block's norm1 never re-reads y
        h = ops.conv(x, self.c1, in_ss=ops.groupnorm_stats(x, *self.n1), in_swish=True, want_stats=True)
        skip = x if self.sc is None else ops.conv(x, self.sc)
        return ops.conv(h, self.c2, out=out, res=skip, in_ss=ops.groupnorm_stats(h, *self.n2), in_swish=True, want_stats=True)


class _Attn:
    """AttnBlock archs/vqgan_arch.py:194-253 (single head, C^-0.5 applied to the scores)."""

    def __init__(self, P, pre):
        self.n = (P[pre + ".norm.weight"], P[pre + ".norm.bias"])
        q = Conv.from_torch(P[pre + ".q.weight"], P[pre + ".q.bias"])
        k = Conv.from_torch(P[pre + ".k.weight"], P[pre + ".k.bias"])
        self.qk = Conv.cat([q, k])
        v = Conv.from_torch(P[pre + ".v.weight"], P[pre + ".v.bias"])
        self.wv, self.bv = v.w, v.b
        self.proj = Conv.from_torch(P[pre + ".proj_out.weight"], P[pre + ".proj_out.bias"])
        self.C = q.cout

    def __call__(self, x):
        B, H, W, Cc = x.shape
        N = H * W
        hn = ops.groupnorm(x, *self.n, swish=False)
        qk = ops.conv(hn, self.qk)                                            # [B,H,W,2C]
        vt = torch.empty((B, Cc, N), device=x.device, dtype=x.dtype)          # V^T = Wv . hn^T + bv
        ops.gemm_nt(self.wv, hn, vt, M=Cc, N=N, K=Cc, lda=Cc, ldb=Cc, ldc=N, nb0=B, bt_bs=(N * Cc, 0),
                    c_bs=(Cc * N, 0), bias=self.bv, bias_per_row=True)
        if x.dtype == torch.bfloat16 and ATTNBLOCK_FUSED16 and Cc == 256 and N % 128 == 0:
            # configs[2]: the core as ONE kernel (csrc/attention.hip attnblock16_kernel): no [B,N,N] score tensor, fp32 statistics
            h = torch.empty((B, H, W, Cc), device=x.device, dtype=x.dtype)
            L.check(ops._timed("attnblock", {"flops": 4.0 * B * N * N * Cc, "bf16": 1} if ops._PROFILE is not None else None, L.load().smx_attnblock_bf16,
                               qk.data_ptr(), 2 * Cc, N * 2 * Cc, qk.data_ptr() + 2 * Cc, 2 * Cc, N * 2 * Cc, vt.data_ptr(), N, Cc * N,
                               h.data_ptr(), Cc, N * Cc, B, N, N, Cc, float(int(Cc) ** (-0.5)), ops._stream()), "smx_attnblock_bf16")
            return ops.conv(h, self.proj, res=x)
        if x.dtype == torch.float32 and ATTNBLOCK_FUSED32 and Cc == 256 and N % 128 == 0:
            # configs[1]: the same fusion in the reference's own arithmetic (attnblock32_kernel: exact fp32 products on the fp32 MFMA)
            h = torch.empty((B, H, W, Cc), device=x.device, dtype=x.dtype)
            L.check(ops._timed("attnblock", {"flops": 4.0 * B * N * N * Cc} if ops._PROFILE is not None else None, L.load().smx_attnblock_f32,
                               qk.data_ptr(), 2 * Cc, N * 2 * Cc, qk.data_ptr() + 4 * Cc, 2 * Cc, N * 2 * Cc, vt.data_ptr(), N, Cc * N,
                               h.data_ptr(), Cc, N * Cc, B, N, N, Cc, float(int(Cc) ** (-0.5)), ops._stream()), "smx_attnblock_f32")
            return ops.conv(h, self.proj, res=x)
        # logits and probabilities stay fp32 in both storage modes (bf16 path: c_f32 store, fp32 softmax, a_f32 operand converted
        # while staging): raw q.k sums over C = 256 lose too much in 8 mantissa bits before the softmax
        s = torch.empty((B, N, N), device=x.device, dtype=torch.float32)
        ops.gemm_nt(qk, qk, s, M=N, N=N, K=Cc, lda=2 * Cc, ldb=2 * Cc, ldc=N, nb0=B, a_bs=(N * 2 * Cc, 0),
                    bt_bs=(N * 2 * Cc, 0), c_bs=(N * N, 0), bt_off=Cc)
        ops.softmax_rows(s, N, float(int(Cc) ** (-0.5)))
        h = torch.empty((B, H, W, Cc), device=x.device, dtype=x.dtype)
        ops.gemm_nt(s, vt, h, M=N, N=Cc, K=N, lda=N, ldb=N, ldc=Cc, nb0=B, a_bs=(N * N, 0), bt_bs=(Cc * N, 0),
                    c_bs=(N * Cc, 0))
        return ops.conv(h, self.proj, res=x)


class _Transformer:
    """TransformerLayer archs/appmotioncodebook_arch.py:65-126 with nn.MultiheadAttention
    restated: codebook K / V^T precomputed for all rows."""

    def __init__(self, P, pre, E, nhead, codebook, adt=torch.float32):
        self.E, self.H, self.dh = E, nhead, E // nhead
        W, b = P[pre + ".self_attn.in_proj_weight"], P[pre + ".self_attn.in_proj_bias"]
        self.s_qk = Conv(W[:2 * E].contiguous(), b[:2 * E].contiguous(), 1, 1, E, 2 * E)
        self.s_v = Conv(W[2 * E:].contiguous(), b[2 * E:].contiguous(), 1, 1, E, E)
        self.s_out = Conv.from_torch(P[pre + ".self_attn.out_proj.weight"], P[pre + ".self_attn.out_proj.bias"])
        W, b = P[pre + ".cross_attn.in_proj_weight"], P[pre + ".cross_attn.in_proj_bias"]
        self.c_q = Conv(W[:E].contiguous(), b[:E].contiguous(), 1, 1, E, E)
        self.c_out = Conv.from_torch(P[pre + ".cross_attn.out_proj.weight"], P[pre + ".cross_attn.out_proj.bias"])
        Kc = codebook.shape[0]
        self.Kc = Kc
        # [Kc | Vc] = cb [Wk;Wv]^T + [bk;bv]  -> [Kc, 2E]   (input independent, all codebook rows)
        self.ckv = torch.empty((Kc, 2 * E), device=codebook.device, dtype=torch.float32)
        ops.gemm_nt(codebook, W[E:].contiguous(), self.ckv, M=Kc, N=2 * E, K=E, lda=E, ldb=E, ldc=2 * E, bias=b[E:].contiguous())
        if adt != torch.float32:               # pack time, once: the projected codebook in the activation storage type
            self.ckv = self.ckv.to(adt)
        self.n1 = (P[pre + ".norm1.weight"], P[pre + ".norm1.bias"])
        self.n2 = (P[pre + ".norm2.weight"], P[pre + ".norm2.bias"])
        self.n3 = (P[pre + ".norm3.weight"], P[pre + ".norm3.bias"])
        self.f1 = Conv.from_torch(P[pre + ".conv1.weight"], P[pre + ".conv1.bias"])
        self.f2 = Conv.from_torch(P[pre + ".conv2.weight"], P[pre + ".conv2.bias"])

    def __call__(self, tgt, S, pos, mask=None):
        """tgt [B,g,g,E] (== tokens [B,g*g,E]; g = 32, 64 at img_size 512); S = codebook prefix rows."""
        B, g = tgt.shape[0], tgt.shape[1]
        E, H, dh, N = self.E, self.H, self.dh, g * g
        # self attention: q = k = LN(x)+pos, v = LN(x)
        t2, qk_in = ops.layernorm(tgt, *self.n1, pos=pos)
        qk = ops.conv(qk_in, self.s_qk)                                       # [B,32,32,2E] = [q | k]
        v = ops.conv(t2, self.s_v)                                            # [B,32,32,E]
        o = ops.attention(qk[..., :E], qk[..., E:], v, H, dh, N, mask=mask)
        tgt = ops.conv(o.view(B, g, g, E), self.s_out, res=tgt)
        # cross attention against the codebook prefix
        t2, q_in = ops.layernorm(tgt, *self.n2, pos=pos)
        q = ops.conv(q_in, self.c_q)
        o = ops.attention(q, self.ckv[:, :E], self.ckv[:, E:], H, dh, S, k_shared=True)
        tgt = ops.conv(o.view(B, g, g, E), self.c_out, res=tgt)
        # conv FFN
        t2, _ = ops.layernorm(tgt, *self.n3)
        h = ops.conv(t2, self.f1, act=ACT_GELU)
        return ops.conv(h, self.f2, res=tgt)


class SourceCache:
    """frame-invariant encoder taps of one source (A8 encoder): {32,64,128,256} -> NHWC [1|B,s,s,C]."""
    __slots__ = ("feats", "batch")

    def __init__(self, feats, batch):
        self.feats, self.batch = feats, batch


class NetGEngine:
    def __init__(self, P, cfg, dtype=torch.float32):
        """dtype: activation storage / MFMA input type -- torch.float32 (BASELINE configs[1]) or torch.bfloat16 (configs[2]:
        bf16 NHWC activations and weights on v_mfma_f32_32x32x16_bf16 with fp32 accumulate; flows, occlusion maps, keypoint
        heatmaps, normalisation statistics, softmax and the output image stay fp32)."""
        self.cfg = cfg
        self.adt = dtype
        self.is16 = dtype == torch.bfloat16
        nf, ch_mult, rb = cfg["nf"], tuple(cfg["ch_mult"]), cfg["res_blocks"]
        attn = tuple(cfg["attn_resolutions"])
        eplan, _ = encoder_plan(nf, ch_mult, rb, cfg["img_size"], attn)
        self.enc_kinds = [k for k, _, _ in eplan] + ["conv"]
        self.gen_kinds = [k for k, _, _ in generator_plan(nf, ch_mult, rb, cfg["img_size"], attn, 256)]
        self.enc = [self._block(P, f"encoder.blocks.{i}", k) for i, k in enumerate(self.enc_kinds)]
        self.gen = [self._block(P, f"generator.blocks.{i}", k) for i, k in enumerate(self.gen_kinds)]
        self.nhead = cfg["n_head"]
        Em, Ea = cfg["dim_embd_motion"], cfg["dim_embd_app"]
        self.Em, self.Ea = Em, Ea
        self.cb_motion = P["quantize_motion.embedding.weight"].contiguous()
        self.cb_app = P["quantize_app.embedding.weight"].contiguous()
        self.pos_motion = P["position_emb_motion"].contiguous()
        self.pos_app = P["position_emb_app"].contiguous()
        self.motion_blocks = [_Transformer(P, f"motion_block.{l}", Em, self.nhead, self.cb_motion, dtype) for l in range(cfg["n_layers_motion"])]
        self.app_blocks = [_Transformer(P, f"app_block.{l}", Ea, self.nhead, self.cb_app, dtype) for l in range(cfg["n_layers_app"])]
        cv = lambda n: Conv.from_torch(P[n + ".weight"], P[n + ".bias"])
        self.motion_emb0 = cv("motion_emb.0")
        self.motion_emb1 = cv("motion_emb.1.conv")
        self.motion_emb2 = _Res(P, "motion_emb.2")
        self.mq1, self.mq2 = cv("motion_query_enc_1"), cv("motion_query_enc_2")
        self.kp_enc = cv("driving_kp_enc")
        # N4: every grid scales with k = img_size / 256 (token grid g = 32 k, flow grid 2 g, features at s k); the per-scale modules
        # keep the names of the 256 layout (s = 32 / 64 / 128 / 256: the state_dict is the 256 one but for the g*g position rows)
        self.k = cfg["img_size"] // 256
        self.g = 32 * self.k
        self.sizes = [32] + [int(s) for s in cfg["connect_list"]]
        self.wsrc = {s: cv(f"warped_source_enc_{s}") for s in self.sizes}
        self.to_ctx = {s: cv(f"to_context.{int(math.log2(s)) - 5}") for s in self.sizes}
        self.bme = {n: cv("BasicMotionEncoder." + n) for n in ("convc1", "convc2", "convf1", "convf2", "conv")}
        # bf16: BasicMotionEncoder.conv reads the 160-channel [cor | flo] concat -- 2.5 of the 64-channel slices the region-direct bf16 kernel
        # works in, so it fell to the implicit GEMM (0.14 of its byte bound, 1.0 ms per step).  Padded to 192 input channels (zero weights, a
        # zeroed tail in the concat buffer) it takes the region kernel
        self.bme_conv_pad = None
        if self.is16:
            c = self.bme["conv"]
            w = torch.zeros((c.cout, 9, 192), device=c.w.device, dtype=torch.float32)
            w[:, :, :160] = c.w.view(c.cout, 9, 160)
            self.bme_conv_pad = Conv(w.reshape(c.cout, 9 * 192).contiguous(), c.b, 3, 3, 192, c.cout)
        self.ref_c1 = cv("refine.convc1")
        self.ref_h = Conv.cat([cv("refine.conv1"), cv("refine.convo1")])      # shared input -> N=256
        # conv2 (2 <- h[:128]) and convo2 (1 <- h[128:]) as ONE block-diagonal conv 256 -> 3 writing r directly
        f2, o2 = cv("refine.conv2"), cv("refine.convo2")
        wz = torch.zeros((3, 9, 256), device=f2.w.device, dtype=torch.float32)
        wz[0:2, :, :128] = f2.w.view(2, 9, 128)
        wz[2:3, :, 128:] = o2.w.view(1, 9, 128)
        self.ref_out = Conv(wz.reshape(3, 9 * 256).contiguous(), torch.cat([f2.b, o2.b]).contiguous(), 3, 3, 256, 3)
        self.app_in, self.app_out = {}, {}
        for s in [int(x) for x in cfg["connect_app_list"]]:
            if s == 32:
                self.app_in[s], self.app_out[s] = cv("app_feat_emb_32"), cv("to_app_feat_32")
            else:
                p, c = s // 32, CHANNELS[str(s)]
                w = P[f"app_feat_emb_{s}.1.weight"]                           # [256, (p1 p2 c)] == conv pxp K order
                self.app_in[s] = Conv(w.contiguous(), P[f"app_feat_emb_{s}.1.bias"].contiguous(), p, p, c, w.shape[0])
                self.app_out[s] = Conv.from_torch(P[f"to_app_feat_{s}.0.weight"], P[f"to_app_feat_{s}.0.bias"])
        self.sft = {}
        for s in [int(x) for x in cfg["connect_list"]]:
            pre = f"fuse_convs_dict.{s}"
            self.sft[s] = {"res": _Res(P, pre + ".encode_enc"),
                           "ss0": Conv.cat([cv(pre + ".scale.0"), cv(pre + ".shift.0")]),
                           "scale2": cv(pre + ".scale.2"), "shift2": cv(pre + ".shift.2"),
                           "ms": cv(f"fuse_ms_dict.{s}")}
        # to_motion (appmotioncodebook_arch.py:290-292): Upsample(E) -> ResBlock(E) -> GroupNorm -> conv3x3 E->2; only the
        # training branch runs it (m_recon = to_motion(quant_motion), :426)
        self.to_motion = (cv("to_motion.0.conv"), _Res(P, "to_motion.1"), (P["to_motion.2.weight"], P["to_motion.2.bias"]), cv("to_motion.3"))
        self.beta = float(cfg.get("beta", 0.25))
        self.fuse_after = {9: 64, 12: 128, 15: 256}
        self.taps_after = {2: 256, 5: 128, 8: 64}

    @staticmethod
    def _block(P, pre, kind):
        if kind == "res":
            return _Res(P, pre)
        if kind == "attn":
            return _Attn(P, pre)
        if kind in ("down", "up"):
            return Conv.from_torch(P[pre + ".conv.weight"], P[pre + ".conv.bias"])
        if kind == "gn":
            return (P[pre + ".weight"], P[pre + ".bias"])
        return Conv.from_torch(P[pre + ".weight"], P[pre + ".bias"])

    @staticmethod
    def _run(kind, blk, x, out=None):
        if kind == "res":
            return blk(x, out=out)
        if out is not None:
            raise ValueError(f"in-place output is only planned for res blocks, not {kind}")
        if kind == "conv":
            return ops.conv(x, blk, want_stats=True)
        if kind == "attn":
            return blk(x)
        if kind == "down":      # pad (0,1,0,1) + conv3x3 s2 p0  (vqgan_arch.py:144-153)
            return ops.conv(x, blk, stride=2, pad=(0, 0), out_hw=(x.shape[1] // 2, x.shape[2] // 2))
        if kind == "up":        # nearest x2 folded into the gather (vqgan_arch.py:156-165)
            return ops.conv(x, blk, up2=True, want_stats=True)
        if kind == "gn":
            return ops.groupnorm(x, blk[0], blk[1], swish=False)
        raise ValueError(kind)

    def _run_seq(self, kinds, blocks, x, hook=None, out_for=None):
        """run a block list; a trailing [gn, conv] pair (blocks 17, 18) is fused: GN folded into the conv loader.
        out_for(i, x_in) may hand block i a channel-slice view to write its output into (in-place concat)."""
        i = 0
        while i < len(kinds):
            if kinds[i] == "gn" and i + 1 < len(kinds) and kinds[i + 1] == "conv":
                x = ops.conv(x, blocks[i + 1], in_ss=ops.groupnorm_stats(x, blocks[i][0], blocks[i][1]), in_swish=False,
                             out_dtype=torch.float32 if blocks[i + 1].cout <= 4 else None)      # the image head stays fp32
                i += 2
            else:
                x = self._run(kinds[i], blocks[i], x, None if out_for is None else out_for(i, x))
                i += 1
            if hook is not None:
                x = hook(i - 1, x)
        return x

    # ---- A8 encoder: frame-invariant -------------------------------------------------------
    def _image_in(self, x_nchw):
        """NCHW fp32 image -> NHWC fp32; blocks[0] (3 -> nf) reads it in fp32 either way (converted while staging on the bf16 path)."""
        x = ops.nchw_to_nhwc(x_nchw)
        return ops.conv(x, self.enc[0], want_stats=True, mfma16=self.is16)

    def encode_source(self, x_nchw):
        x = self._image_in(x_nchw)
        feats = {}

        def tap(i, t):
            if i in self.taps_after:
                feats[self.taps_after[i]] = t
            return t
        x = self._run_seq(self.enc_kinds[1:], self.enc[1:], x, lambda i, t: tap(i + 1, t))
        feats[32] = x
        return SourceCache(feats, x_nchw.shape[0])

    def encode_driving(self, x_nchw):
        """taps after encoder blocks 2, 5, 8, 11 keyed by size (appmotioncodebook_arch.py:364-371), NHWC."""
        x = self._image_in(x_nchw)
        out = {}
        for i, (kind, blk) in enumerate(zip(self.enc_kinds, self.enc)):
            if i > 0:
                x = self._run(kind, blk, x)
            if i in (2, 5, 8, 11):
                out[str(x.shape[1] // self.k)] = x                            # keyed by the scale's size in the 256 layout
            if i == 11:
                break
        return out

    def generator_only(self, x_nhwc):
        """Generator.forward without fusion (vqgan_arch.py:344-348): lq_recon = generator(lq_feat)."""
        return self._run_seq(self.gen_kinds, self.gen, x_nhwc)

    # ---- A12 (live in the training branch) -------------------------------------
    def quantize(self, z_nhwc, codebook, Ks):
        """VectorQuantizer.forward (archs/vqgan_arch.py:33-93) on NHWC tokens: -> (z_q NHWC, loss, stats).  One fused kernel:
        distances over the first Ks rows, first-minimum argmin, gather, z_q = z + (e - z), sum (z_q - z)^2;
        loss = beta * mse(sg(z_q), z) + mse(z_q, sg(z)) = (1 + beta) * mean((z_q - z)^2) in the forward pass."""
        B, H, W, D = z_nhwc.shape
        z = z_nhwc.float() if z_nhwc.dtype != torch.float32 else z_nhwc
        if not z.is_contiguous():
            z = ops.copy_slice(z, torch.empty((B, H, W, D), device=z.device, dtype=torch.float32))
        idx, zq, dmin, sq = ops.vq_nearest(z.view(-1, D), codebook, Ks)
        loss = sq * ((1.0 + self.beta) / float(z.numel()))
        return zq.view(B, H, W, D), loss.view(()), {"min_encoding_indices": idx.view(-1, 1), "min_distance": dmin}

    def _to_motion(self, zq):
        up, res, gn, head = self.to_motion
        h = ops.conv(zq, up, up2=True, want_stats=True)                       # [B,64,64,E]
        h = res(h)
        return ops.conv(h, head, in_ss=ops.groupnorm_stats(h, gn[0], gn[1]), in_swish=False, out_dtype=torch.float32)   # [B,64,64,2]

    def app_codebook_loss(self, gt_nchw):
        """appmotioncodebook_arch.py:429-469 on the driving frame `gt`: encoder taps -> patch embedding -> quantize_app at the
        scale's codebook prefix -> un-patchify of the quantised and of the raw embedding.
        -> ([app_recon, app_feat_original, quant_app, app_feat, feat_com] per scale (NHWC), [loss per scale], [stats per scale])"""
        if self.is16:
            raise NotImplementedError("the training branch is fp32 (configs[1] arithmetic)")
        taps = self.encode_driving(gt_nchw)
        recon, losses, stats = [], [], []
        for s in sorted(self.app_in):
            feat = taps[str(s)]
            C = feat.shape[-1]
            app_feat = ops.conv(feat, self.app_in[s]) if s == 32 else ops.conv(feat, self.app_in[s], stride=s // 32, pad=(0, 0))
            zq, loss, st = self.quantize(app_feat, self.cb_app, self.cb_app.shape[0] // 4 * _SCALE_K[s])
            un = (lambda t: ops.conv(t, self.app_out[32])) if s == 32 else (lambda t: ops.conv(t, self.app_out[s], d2s=(s // 32, C)))
            recon.append([un(zq), un(app_feat), zq, app_feat, feat])
            losses.append(loss)
            stats.append(st)
        return recon, losses, stats

    # ---- A9 -------------------------------------------------------------------------------
    def _motion_comp(self, flow_res, mq, warp0, s, train=None):
        B = flow_res.shape[0]
        Em = self.Em
        m1 = ops.conv(flow_res, self.motion_emb0, mfma16=self.is16)           # [B,64,64,32]  (fp32 flow in)
        g, Fg = self.g, flow_res.shape[1]
        m2 = ops.conv(m1, self.motion_emb1, stride=2, pad=(0, 0), out_hw=(g, g))
        qin = torch.empty((B, g, g, 2 * Em), device=flow_res.device, dtype=self.adt)
        self.motion_emb2(m2, out=qin[..., :Em])
        if train is not None:
            # training branch (:379-386, :426): quantise m_feat against the scale's motion-codebook prefix, reconstruct the flow
            zq, loss, st = self.quantize(qin[..., :Em], self.cb_motion, self.cb_motion.shape[0] // 4 * _SCALE_K[s])
            train["motion_recon"].append(self._to_motion(zq))
            train["loss_motion"].append(loss)
            train["stats_motion"].append(st)
        ops.copy_slice(mq, qin[..., Em:])
        q = ops.conv(qin, self.mq2)                                           # tokens [B,1024,32]
        S = self.cb_motion.shape[0] // 4 * _SCALE_K[s]
        for blk in self.motion_blocks:
            q = blk(q, S, self.pos_motion)
        motion_f = ops.resize(q, Fg, Fg)
        cpad = 192 if self.bme_conv_pad is not None else 160
        cf = torch.empty((B, Fg, Fg, cpad), device=q.device, dtype=self.adt)
        if cpad > 160:
            cf[..., 160:].zero_()
        cor = ops.conv(motion_f, self.bme["convc1"], act=ACT_RELU)
        ops.conv(cor, self.bme["convc2"], out=cf[..., :96], act=ACT_RELU)
        flo = ops.conv(flow_res, self.bme["convf1"], act=ACT_RELU, mfma16=self.is16)   # 7x7 pad 3 (fp32 flow in)
        ops.conv(flo, self.bme["convf2"], out=cf[..., 96:160], act=ACT_RELU)
        inp = torch.empty((B, Fg, Fg, 256), device=q.device, dtype=self.adt)
        ops.conv(cf, self.bme_conv_pad or self.bme["conv"], out=inp[..., :126], act=ACT_RELU)
        ops.copy_slice(flow_res, inp[..., 126:128])
        if s > 128:
            # relu(to_context(.)) is per pixel and only its 4 bilinear taps per 64x64 output pixel survive the
            # resize: evaluate it on the gathered taps (1/4 of the 256x256 pixels), then blend (same values up to 1 ulp)
            taps = ops.resize_taps_gather(warp0, Fg, Fg)                      # [B,64,256,C]
            wf = ops.resize_taps_combine(ops.conv(taps, self.to_ctx[s], act=ACT_RELU), s * self.k, s * self.k)
        else:
            wf = ops.conv(warp0, self.to_ctx[s], act=ACT_RELU)                # [B,s,s,192]
            if s != 64:
                wf = ops.resize(wf, Fg, Fg)
        ops.conv(wf, self.ref_c1, out=inp[..., 128:], act=ACT_RELU)
        h = ops.conv(inp, self.ref_h, act=ACT_RELU)                           # [B,64,64,256] = [conv1 | convo1]
        return ops.conv(h, self.ref_out, out_dtype=torch.float32)             # [B,64,64,3] = [dflow(2) | docc(1)], always fp32

    # ---- A10 ------------------------------------------------------------------------------
    def _app_comp(self, feat, m_com, s, out=None):
        C = feat.shape[-1]
        ign = ops.motion_ignore(m_com, self.g, self.g)
        if s == 32:
            q = ops.conv(feat, self.app_in[32])
        else:
            p = s // 32
            q = ops.conv(feat, self.app_in[s], stride=p, pad=(0, 0))
        S = self.cb_app.shape[0] // 4 * _SCALE_K[s]
        for l, blk in enumerate(self.app_blocks):
            q = blk(q, S, self.pos_app, mask=ign if l == 0 else None)
        if s == 32:
            return ops.conv(q, self.app_out[32], out=out)
        return ops.conv(q, self.app_out[s], out=out, d2s=(s // 32, C))

    def _one_scale(self, st, feat, s, first, out=None):
        flow = st["flows"][-1]
        warp0 = ops.warp(feat, flow)
        wsrc = warp0 if s == 32 else ops.resize(warp0, self.g, self.g)
        B = flow.shape[0]
        mqin = torch.empty((B, self.g, self.g, 2 * self.Em), device=flow.device, dtype=self.adt)
        ops.conv(wsrc, self.wsrc[s], out=mqin[..., :self.Em], act=ACT_RELU)
        ops.copy_slice(st["kp_feat"], mqin[..., self.Em:])
        mq = ops.conv(mqin, self.mq1)
        r = self._motion_comp(ops.flow_to_residual(flow), mq, warp0, s, st.get("train"))
        m_com, res_norm, occ = ops.flow_occ_update(flow, r, st["occ"][-1])
        st["flows"].append(m_com)
        st["res"].append(res_norm)
        st["occ"].append(occ)
        warped = ops.warp(feat, m_com, occ)
        st["before"].append(warped)
        comp = self._app_comp(warped, m_com, s, out=out)
        st["comp"].append(comp)
        return comp

    # ---- A13 ------------------------------------------------------------------------------
    def _fuse(self, s, cat, w):
        """cat = [enc | dec] (NHWC, 2C channels): both halves were written in place by their producers
        (the appearance-compensation un-patchify conv and the decoder ResBlock), no concat copy."""
        f = self.sft[s]
        C = cat.shape[-1] // 2
        enc, dec = cat[..., :C], cat[..., C:]
        e = f["res"](cat)
        ss = ops.conv(e, f["ss0"], act=ACT_LRELU02)                           # [.., 2C] = [scale.0 | shift.0]
        scale = ops.conv(ss[..., :C], f["scale2"])
        x = ops.conv_sft(ss[..., C:], f["shift2"], dec, scale, w)             # dec + w * (dec * scale + shift): shift2's epilogue
        return ops.conv(enc, f["ms"], res=x, want_stats=True)                 # x + fuse_ms(enc)

    def forward(self, cache, deformation, occ64, heat_nhwc, w=1.0, train=False):
        """cache: SourceCache; deformation [B,64,64,2]; occ64 [B,64,64]; heat [B,64,64,15] NHWC.
        -> state dict with NHWC 'out' [B,256,256,3] and the intermediate lists.
        train=True: the forward of the training branch (inference=False): st['train'] carries the per-scale motion-codebook
        quantisation (losses, indices, to_motion reconstructions in pixel units) and st['out_lr'] the un-fused decoder pass."""
        st = {"flows": [deformation.contiguous()], "occ": [occ64.contiguous()], "res": [], "before": [], "comp": []}
        if train:
            if self.is16:
                raise NotImplementedError("the training branch is fp32 (configs[1] arithmetic)")
            st["train"] = {"motion_recon": [], "loss_motion": [], "stats_motion": []}
        st["kp_feat"] = ops.conv(ops.resize(heat_nhwc, self.g, self.g), self.kp_enc, act=ACT_RELU, mfma16=self.is16)   # fp32 heatmaps in
        x = self._one_scale(st, cache.feats[32], 32, True)
        st["lq"] = x
        cats = {}

        def out_for(i, x_in):
            if i in self.fuse_after and w > 0 and self.gen_kinds[i] == "res":
                B, H, W_, _ = x_in.shape
                C = self.gen[i].c2.cout
                cats[i] = torch.empty((B, H, W_, 2 * C), device=x_in.device, dtype=self.adt)
                return cats[i][..., C:]
            return None

        def fuse(i, t):
            if i in self.fuse_after and w > 0:
                s = self.fuse_after[i]
                cat = cats.pop(i, None)
                if cat is None:                                               # producer was not a ResBlock: copy
                    C = t.shape[-1]
                    cat = torch.empty(t.shape[:-1] + (2 * C,), device=t.device, dtype=self.adt)
                    ops.copy_slice(t, cat[..., C:])
                C = cat.shape[-1] // 2
                self._one_scale(st, cache.feats[s], s, False, out=cat[..., :C])
                t = self._fuse(s, cat, w)
            return t
        st["out"] = self._run_seq(self.gen_kinds, self.gen, x, fuse, out_for)
        if train:
            st["out_lr"] = self.generator_only(st["lq"])                      # x_lr_32: every generator block, no fusion (:649-659)
        return st
