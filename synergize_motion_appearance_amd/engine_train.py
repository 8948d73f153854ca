"""Training-branch execution of AppMotionCompFormer on the HIP path WITH a gradient tape (BASELINE configs[4]):
`forward(x, dense_motion, w=1, inference=False, gt=driving)` of reference `archs/appmotioncodebook_arch.py:546-764` plus
`app_codebook_loss` (:429-469) and the eight `VectorQuantizer.forward` calls, built from `train_ops` so that `Tape.backward()`
produces the gradient of every parameter and of the dense-motion inputs (deformation, occlusion map, driving keypoint heatmap).

Differences from the inference engine (`engine_netg.NetGEngine`), all forced by "weights change every step / activations are
needed again in the backward":
  * parameters are read in their checkpoint layout every step and packed by a kernel (`smx_pack_weight_f32`); no Winograd-domain
    weights, no stacked sibling convolutions, no precomputed codebook K/V (their projections carry gradients);
  * GroupNorm outputs are materialised (the backward of the consuming convolution needs them as its weight-gradient operand);
  * every source has its own encoder pass (B sources for B driving frames: `feed_data`'s random pairs), nothing is cached;
  * `to_context` at 256x256 runs on all pixels (the tap-sampled form is an inference shortcut).
The arithmetic is the same fp32 MFMA / VALU kernels; results agree with the inference engine to rounding.
"""
import math

import torch

from . import ops
from . import train_ops as T
from .manifest import encoder_plan, generator_plan, CHANNELS
from .train_ops import ACT_RELU, ACT_LRELU02, ACT_GELU

_SCALE_K = {32: 1, 64: 2, 128: 3, 256: 4}


class NetGTrainEngine:
    def __init__(self, cfg):
        self.cfg = cfg
        if cfg["img_size"] != 256:
            raise NotImplementedError("the training step is planned for img_size 256 (BASELINE configs[4]); the 512 variant (N4) is inference")
        nf, ch_mult, rb = cfg["nf"], tuple(cfg["ch_mult"]), cfg["res_blocks"]
        attn = tuple(cfg["attn_resolutions"])
        eplan, _ = encoder_plan(nf, ch_mult, rb, cfg["img_size"], attn)
        self.enc_kinds = [k for k, _, _ in eplan] + ["conv"]
        self.gen_kinds = [k for k, _, _ in generator_plan(nf, ch_mult, rb, cfg["img_size"], attn, 256)]
        self.nhead = cfg["n_head"]
        self.Em, self.Ea = cfg["dim_embd_motion"], cfg["dim_embd_app"]
        self.nl_m, self.nl_a = cfg["n_layers_motion"], cfg["n_layers_app"]
        self.sizes = [32] + [int(s) for s in cfg["connect_list"]]
        self.app_sizes = [int(s) for s in cfg["connect_app_list"]]
        self.beta = float(cfg.get("beta", 0.25))
        self.fuse_after = {9: 64, 12: 128, 15: 256}
        self.taps_after = {2: 256, 5: 128, 8: 64}

    # ---- building blocks (A8) ----------------------------------------------------------------------------------------------
    @staticmethod
    def _cv(tp, x, pre, **kw):
        return T.conv(tp, x, pre + ".weight", pre + ".bias", **kw)

    def _res(self, tp, x, pre):
        """ResBlock.forward archs/vqgan_arch.py:180-191."""
        h = T.groupnorm(tp, x, pre + ".norm1.weight", pre + ".norm1.bias", swish=True)
        h = self._cv(tp, h, pre + ".conv1")
        h = T.groupnorm(tp, h, pre + ".norm2.weight", pre + ".norm2.bias", swish=True)
        skip = self._cv(tp, x, pre + ".conv_out") if (pre + ".conv_out.weight") in tp.P else x
        return self._cv(tp, h, pre + ".conv2", res=skip)

    def _attn(self, tp, x, pre):
        """AttnBlock archs/vqgan_arch.py:229-253."""
        hn = T.groupnorm(tp, x, pre + ".norm.weight", pre + ".norm.bias", swish=False)
        q, k, v = (self._cv(tp, hn, f"{pre}.{n}") for n in ("q", "k", "v"))
        h = T.attn_core(tp, q, k, v, float(int(x.shape[-1]) ** (-0.5)))
        return self._cv(tp, h, pre + ".proj_out", res=x)

    def _block(self, tp, kind, pre, x):
        if kind == "res":
            return self._res(tp, x, pre)
        if kind == "attn":
            return self._attn(tp, x, pre)
        if kind == "conv":
            return self._cv(tp, x, pre)
        if kind == "down":      # pad (0,1,0,1) + conv3x3 s2 p0  (vqgan_arch.py:144-153)
            return self._cv(tp, x, pre + ".conv", stride=2, pad=(0, 0), out_hw=(x.shape[1] // 2, x.shape[2] // 2))
        if kind == "up":        # nearest x2 + conv3x3 (vqgan_arch.py:156-165)
            return self._cv(tp, x, pre + ".conv", up2=True)
        if kind == "gn":
            return T.groupnorm(tp, x, pre + ".weight", pre + ".bias", swish=False)
        raise ValueError(kind)

    def encode(self, tp, x_nchw, upto=None):
        """encoder blocks on an NCHW image -> (taps after blocks 2, 5, 8 (+11) keyed by size, final x)."""
        x = ops.nchw_to_nhwc(x_nchw)
        tp.stop(x)                                           # the image itself takes no gradient
        feats = {}
        for i, kind in enumerate(self.enc_kinds):
            x = self._block(tp, kind, f"encoder.blocks.{i}", x)
            if i in (2, 5, 8, 11):
                feats[x.shape[1] if i != 11 else "32@11"] = x
            if upto is not None and i == upto:
                break
        return feats, x

    def _generator(self, tp, x, hook=None, last_in=None):
        """last_in: a list that receives the INPUT of the last block (generator.blocks[-1], the image head): the adaptive GAN weight
        needs the gradient of two losses w.r.t. that layer's weight (models/appmotioncomp_model.py:222-228, 334-335)."""
        for i, kind in enumerate(self.gen_kinds):
            if last_in is not None and i == len(self.gen_kinds) - 1:
                last_in.append(x)
            x = self._block(tp, kind, f"generator.blocks.{i}", x)
            if hook is not None:
                x = hook(i, x)
        return x

    # ---- A11 -----------------------------------------------------------------------------------------------------------------
    def _transformer(self, tp, tgt, pre, E, S, pos, ctx, mask=None):
        """TransformerLayer archs/appmotioncodebook_arch.py:88-126; ctx = [K, 2E] projected codebook of this layer."""
        H, dh = self.nhead, E // self.nhead
        ipw, ipb = pre + ".self_attn.in_proj_weight", pre + ".self_attn.in_proj_bias"
        t2, qk_in = T.layernorm(tp, tgt, pre + ".norm1.weight", pre + ".norm1.bias", pos=pos)
        q = T.conv(tp, qk_in, (ipw, slice(0, E)), (ipb, slice(0, E)))
        k = T.conv(tp, qk_in, (ipw, slice(E, 2 * E)), (ipb, slice(E, 2 * E)))
        v = T.conv(tp, t2, (ipw, slice(2 * E, 3 * E)), (ipb, slice(2 * E, 3 * E)))
        o = T.attention(tp, q, k, v, H, dh, 1024, mask=mask)
        tgt = T.conv(tp, T.view(tp, o, tgt.shape), pre + ".self_attn.out_proj.weight", pre + ".self_attn.out_proj.bias", res=tgt)
        ipw, ipb = pre + ".cross_attn.in_proj_weight", pre + ".cross_attn.in_proj_bias"
        t2, q_in = T.layernorm(tp, tgt, pre + ".norm2.weight", pre + ".norm2.bias", pos=pos)
        q = T.conv(tp, q_in, (ipw, slice(0, E)), (ipb, slice(0, E)))
        o = T.attention(tp, q, None, None, H, dh, S, ctx=ctx)
        tgt = T.conv(tp, T.view(tp, o, tgt.shape), pre + ".cross_attn.out_proj.weight", pre + ".cross_attn.out_proj.bias", res=tgt)
        t2, _ = T.layernorm(tp, tgt, pre + ".norm3.weight", pre + ".norm3.bias")
        h = T.act(tp, self._cv(tp, t2, pre + ".conv1"), ACT_GELU)
        return self._cv(tp, h, pre + ".conv2", res=tgt)

    def _ctx(self, tp, pre, cb_name, E):
        """[K | V] = codebook [Wk; Wv]^T + [bk; bv] for ALL rows (a scale's prefix is the first S rows); once per step and layer."""
        key = ("_ctx", pre)
        if key not in tp.packed:
            cb = T.leaf(tp, cb_name)
            K = cb.shape[0]
            ipw, ipb = pre + ".cross_attn.in_proj_weight", pre + ".cross_attn.in_proj_bias"
            kv = T.conv(tp, T.view(tp, cb, (1, K, 1, E)), (ipw, slice(E, 3 * E)), (ipb, slice(E, 3 * E)))     # [1,K,1,2E]
            tp.packed[key] = T.view(tp, kv, (K, 2 * E))
        return tp.packed[key]

    # ---- A12 + :426 ------------------------------------------------------------------------------------------------------------
    def _to_motion(self, tp, zq):
        h = self._cv(tp, zq, "to_motion.0.conv", up2=True)
        h = self._res(tp, h, "to_motion.1")
        h = T.groupnorm(tp, h, "to_motion.2.weight", "to_motion.2.bias", swish=False)
        return self._cv(tp, h, "to_motion.3")

    # ---- A9 ----------------------------------------------------------------------------------------------------------------------
    def _motion_comp(self, tp, flow_res, mq, warp0, s, out):
        Em = self.Em
        # m_feat = motion_emb(m.detach()) (:377): conv3x3 2->E, Downsample, ResBlock
        m1 = self._cv(tp, T.detach(tp, flow_res), "motion_emb.0")
        m2 = self._cv(tp, m1, "motion_emb.1.conv", stride=2, pad=(0, 0), out_hw=(32, 32))
        m_feat = self._res(tp, m2, "motion_emb.2")
        Ks = tp.P["quantize_motion.embedding.weight"].shape[0] // 4 * _SCALE_K[s]
        zq, loss, st = T.quantize(tp, m_feat, "quantize_motion.embedding.weight", Ks, self.beta)
        out["motion_recon"].append(self._to_motion(tp, zq))          # pixel units; /31.5 by the caller (:586)
        out["loss_motion"].append(loss)
        out["stats_motion"].append(st)
        q = self._cv(tp, T.cat(tp, [m_feat, mq]), "motion_query_enc_2")
        pos = T.leaf(tp, "position_emb_motion")
        for l in range(self.nl_m):
            pre = f"motion_block.{l}"
            q = self._transformer(tp, q, pre, Em, Ks, pos, self._ctx(tp, pre, "quantize_motion.embedding.weight", Em))
        motion_f = T.resize(tp, q, 64, 64)
        # BasicMotionEncoder :129-147 (flow_res NOT detached here)
        cor = self._cv(tp, motion_f, "BasicMotionEncoder.convc1", act=ACT_RELU)
        cor = self._cv(tp, cor, "BasicMotionEncoder.convc2", act=ACT_RELU)
        flo = self._cv(tp, flow_res, "BasicMotionEncoder.convf1", act=ACT_RELU)
        flo = self._cv(tp, flo, "BasicMotionEncoder.convf2", act=ACT_RELU)
        mo = self._cv(tp, T.cat(tp, [cor, flo]), "BasicMotionEncoder.conv", act=ACT_RELU)
        m_f = T.cat(tp, [mo, flow_res])                                              # [B,64,64,128]
        wf = self._cv(tp, warp0, f"to_context.{int(math.log2(s)) - 5}", act=ACT_RELU)
        if s != 64:
            wf = T.resize(tp, wf, 64, 64)
        # RefineFlow :150-167
        c = self._cv(tp, wf, "refine.convc1", act=ACT_RELU)
        inp = T.cat(tp, [m_f, c])
        dflow = self._cv(tp, self._cv(tp, inp, "refine.conv1", act=ACT_RELU), "refine.conv2")
        docc = self._cv(tp, self._cv(tp, inp, "refine.convo1", act=ACT_RELU), "refine.convo2")
        return T.cat(tp, [dflow, docc])                                              # [B,64,64,3]

    # ---- A10 -----------------------------------------------------------------------------------------------------------------------
    def _app_embed(self, tp, feat, s):
        if s == 32:
            return self._cv(tp, feat, "app_feat_emb_32")
        return T.conv(tp, feat, f"app_feat_emb_{s}.1.weight", f"app_feat_emb_{s}.1.bias", kind="patch", patch=(s // 32, feat.shape[-1]))

    def _app_unembed(self, tp, q, s, C):
        if s == 32:
            return self._cv(tp, q, "to_app_feat_32")
        return T.conv(tp, q, f"to_app_feat_{s}.0.weight", f"to_app_feat_{s}.0.bias", kind="unpatch", patch=(s // 32, C))

    def _app_comp(self, tp, feat, m_com, s):
        C = feat.shape[-1]
        ign = ops.motion_ignore(m_com)
        q = self._app_embed(tp, feat, s)
        S = tp.P["quantize_app.embedding.weight"].shape[0] // 4 * _SCALE_K[s]
        pos = T.leaf(tp, "position_emb_app")
        for l in range(self.nl_a):
            pre = f"app_block.{l}"
            q = self._transformer(tp, q, pre, self.Ea, S, pos, self._ctx(tp, pre, "quantize_app.embedding.weight", self.Ea),
                                  mask=ign if l == 0 else None)
        return self._app_unembed(tp, q, s, C)

    def _one_scale(self, tp, st, feat, s):
        flow = st["flows"][-1]
        warp0 = T.warp(tp, feat, flow)
        wsrc = warp0 if s == 32 else T.resize(tp, warp0, 32, 32)
        ws = self._cv(tp, wsrc, f"warped_source_enc_{s}", act=ACT_RELU)
        mq = self._cv(tp, T.cat(tp, [ws, st["kp_feat"]]), "motion_query_enc_1")
        r = self._motion_comp(tp, T.flow_to_residual(tp, flow), mq, warp0, s, st["train"])
        m_com, res_norm, occ = T.flow_occ_update(tp, flow, r, st["occ"][-1])
        st["flows"].append(m_com)
        st["res"].append(res_norm)
        st["occ"].append(occ)
        warped = T.warp(tp, feat, m_com, occ)
        st["before"].append(warped)
        comp = self._app_comp(tp, warped, m_com, s)
        st["comp"].append(comp)
        return comp

    # ---- A13 -----------------------------------------------------------------------------------------------------------------------
    def _fuse(self, tp, s, enc, dec, w):
        pre = f"fuse_convs_dict.{s}"
        e = self._res(tp, T.cat(tp, [enc, dec]), pre + ".encode_enc")
        scale = self._cv(tp, self._cv(tp, e, pre + ".scale.0", act=ACT_LRELU02), pre + ".scale.2")
        shift = self._cv(tp, self._cv(tp, e, pre + ".shift.0", act=ACT_LRELU02), pre + ".shift.2")
        x = T.sft_combine(tp, dec, scale, shift, w)
        return self._cv(tp, enc, f"fuse_ms_dict.{s}", res=x)

    # ---- the training-branch forward --------------------------------------------------------------------------------------------------
    def forward(self, tp, x_nchw, deformation, occ64, heat_nhwc, w=1.0, gt_nchw=None):
        """x_nchw [B,3,256,256] sources; deformation [B,64,64,2]; occ64 [B,64,64]; heat [B,64,64,15] (tape tensors: their gradients
        are collected); gt_nchw: the driving frames for app_codebook_loss.  -> state dict (NHWC tensors, tape-tracked)."""
        feats, x = self.encode(tp, x_nchw)
        feats[32] = x
        st = {"flows": [deformation], "occ": [occ64], "res": [], "before": [], "comp": [],
              "train": {"motion_recon": [], "loss_motion": [], "stats_motion": []}}
        st["kp_feat"] = self._cv(tp, T.resize(tp, heat_nhwc, 32, 32), "driving_kp_enc", act=ACT_RELU)
        lq = self._one_scale(tp, st, feats[32], 32)
        st["lq"] = lq

        def fuse(i, t):
            if i in self.fuse_after and w > 0:
                s = self.fuse_after[i]
                enc = self._one_scale(tp, st, feats[s], s)
                t = self._fuse(tp, s, enc, t, w)
            return t
        last_in = []
        st["out"] = self._generator(tp, lq, fuse, last_in)
        st["out_last_in"] = last_in[0]
        st["out_lr"] = self._generator(tp, lq)                   # x_lr_32: every generator block, no fusion (:649-659)
        if gt_nchw is not None:
            st["app"] = self.app_codebook_loss(tp, gt_nchw)
        return st

    def app_codebook_loss(self, tp, gt_nchw):
        """:429-469 on the driving frame: encoder taps -> patch embedding -> quantize_app at the scale's prefix -> un-patchify."""
        feats, _ = self.encode(tp, gt_nchw, upto=11)
        recon, losses, stats = [], [], []
        for s in self.app_sizes:
            feat = feats["32@11"] if s == 32 else feats[s]
            C = feat.shape[-1]
            app_feat = self._app_embed(tp, feat, s)
            zq, loss, stt = T.quantize(tp, app_feat, "quantize_app.embedding.weight",
                                       tp.P["quantize_app.embedding.weight"].shape[0] // 4 * _SCALE_K[s], self.beta)
            recon.append([self._app_unembed(tp, zq, s, C), self._app_unembed(tp, app_feat, s, C), zq, app_feat, feat])
            losses.append(loss)
            stats.append(stt)
        return recon, losses, stats
