"""Differentiable wrappers of the hot-path kernels for the training step: each function launches the forward
kernel(s) through `ops` and records on the `Tape` the closure that launches the backward kernels declared under "TRAINING STEP"
in `include/smx.h`.  fp32 NHWC activations; parameters are referenced by state_dict name (optionally a row range of one:
`("app_block.0.self_attn.in_proj_weight", slice(0, 256))`), their gradients accumulate into `tape.G[name]`.

Reference semantics each op replaces (forward file:line in /root/reference/basicsr, the backward is what torch.autograd derives
from it): conv / Linear `F.conv2d`, `nn.Linear`; `normalize`+`swish` archs/vqgan_arch.py:14-20; `nn.LayerNorm`;
`nn.MultiheadAttention` archs/appmotioncodebook_arch.py:101-115; AttnBlock archs/vqgan_arch.py:229-253; `deform_input` /
`occlude_input` archs/appmotioncodebook_arch.py:349-362; `VectorQuantizer.forward` archs/vqgan_arch.py:33-93;
`Fuse_sft_block` archs/appmotioncodebook_arch.py:28-52; `L1Loss` losses/losses.py.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L
from . import ops
from .lib import ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_SWISH, ACT_GELU, ACT_SIGMOID  # noqa: F401
from .ops import Conv
from .tape import _stream, _pix, _dense, _axpy

F32 = torch.float32


def _empty(shape, like):
    return torch.empty(shape, device=like.device, dtype=F32)


def _zeros(shape, like):
    return torch.zeros(shape, device=like.device, dtype=F32)


# ---- parameters -----------------------------------------------------------------------------------------------------------
def _param(tp, ref):
    """(value view, grad view, cache key) of a parameter reference: a name, or (name, row slice)."""
    if isinstance(ref, tuple):
        name, sl = ref
        return tp.P[name][sl], tp.G[name][sl], (name, sl.start, sl.stop)
    return tp.P[ref], tp.G.get(ref), (ref, None, None)       # no gradient slot: a frozen parameter (the VGG19 of the perceptual loss)


def leaf(tp, name):
    """a parameter used as an ACTIVATION (the codebooks as attention context / quantiser table, position embeddings): returns
    the value tensor; whatever gradient the tape collects for it is added to tape.G[name] after all its consumers ran."""
    cache = tp.packed.setdefault("_leaves", {})
    if name in cache:
        return cache[name]
    t = tp.P[name]
    cache[name] = t

    def bwd():
        g = tp.take(t)
        if g is not None:
            _axpy(tp.lib, g, tp.G[name], 1.0)
    tp.record(bwd)                       # recorded before any consumer -> runs after all of them
    return t


class PackPlan:
    """The weight packings of a training step, remembered across steps.  The step asks for them one by one, lazily, where a layer runs
    (`_packed`, `_packed16`, `_packed_u`: ~660 launches of 6-11 us); a plan attached to the Tape records each request the first time
    (the parameter view, the persistent output buffer, the packing's arguments), and from the next step on ONE `smx_pack_batch`
    launch at tape creation refreshes all of them from the current parameter values and hands them to the tape's cache.  Frozen
    parameters (no gradient slot: the perceptual loss' VGG19) are packed once and never again.  The plan is tied to one parameter
    dict (flat buffers do not move); a different dict resets it."""
    F32, BF16, WINO_U = 0, 1, 2
    ITEM = np.dtype([("w", "<u8"), ("out", "<u8"), ("total", "<i8"), ("cout", "<i4"), ("cin", "<i4"), ("kh", "<i4"), ("kw", "<i4"),
                     ("mode", "<i4"), ("kind", "<i4"), ("first_block", "<i4"), ("reserved", "<i4")])   # == smx_pack_item (include/smx.h)

    def __init__(self):
        self.items = {}          # cache key -> (out tensor, w tensor, total, cout, cin, kh, kw, mode, kind, frozen)
        self._owner = None
        self._table = None
        self._n = self._blocks = 0
        self._dirty = False
        self._ready = {}         # cache key -> out tensor of everything the current table refreshes (+ the frozen ones)

    def add(self, ck, w, out, total, cout, cin, kh, kw, mode, kind, frozen):
        self.items[ck] = (out, w, int(total), cout, cin, kh, kw, mode, kind, bool(frozen))
        self._dirty = True

    def begin(self, tp):
        if self._owner != id(tp.P):
            self.items, self._owner, self._table, self._n, self._dirty, self._ready = {}, id(tp.P), None, 0, False, {}
            return
        if not self.items:
            return
        if self._dirty and not torch.cuda.is_current_stream_capturing():      # (the table upload is a host copy: not inside a capture --
            live = [(ck, it) for ck, it in self.items.items() if not it[9]]   #  requests recorded since the last table keep packing by themselves)
            tab = np.zeros(len(live), dtype=self.ITEM)
            blk = 0
            for r, (_, (out, w, total, cout, cin, kh, kw, mode, kind, _f)) in zip(tab, live):
                r["w"], r["out"], r["total"] = w.data_ptr(), out.data_ptr(), total
                r["cout"], r["cin"], r["kh"], r["kw"], r["mode"], r["kind"], r["first_block"] = cout, cin, kh, kw, mode, kind, blk
                blk += (total + 1023) // 1024
            self._table = torch.from_numpy(tab.view(np.uint8).copy()).to(live[0][1][0].device) if live else None
            self._n, self._blocks, self._dirty = len(live), blk, False
            self._ready = {ck: it[0] for ck, it in self.items.items()}        # in the table, or frozen and packed once
        if self._n:
            L.check(tp.lib.smx_pack_batch(self._table.data_ptr(), self._n, self._blocks, _stream()), "pack_batch")
        tp.packed.update(self._ready)


class ReducePlan:
    """The split reduces of a backward pass, remembered across steps (the weight-gradient counterpart of PackPlan).  Every weight
    gradient is two launches: the TN GEMM that leaves `msplit` raw partial tiles in a workspace and the fixed-order reduce that adds them
    into the parameter's .grad (~460 reduce launches of 5-50 us per step).  The first step that runs with a plan attached to its Tape
    RECORDS: each `_wgrad` gets a persistent workspace, reduces immediately as before and leaves its item (smx_reduce_item) in the
    current segment; `Tape.backward()` closes a segment each time it returns (the step runs its backward in one or two pieces).  From
    the next step on the plan REPLAYS: `_wgrad` finds its workspace by position, launches the GEMM with the reduce deferred, and the end
    of each backward piece finishes the segment's items with ONE `smx_wgrad_reduce_batch` launch per wave (a parameter's second call
    site goes to the next wave: items of one launch must not share an output).  The plan belongs to one gradient dict and one call
    sequence (`key`); a request that does not match the recording raises -- a step variant needs its own plan."""
    ITEM = np.dtype([("ws", "<u8"), ("out", "<u8"), ("bias_ws", "<u8"), ("bias_out", "<u8"), ("msplit", "<i4"), ("Cout", "<i4"), ("K", "<i4"),
                     ("Cin", "<i4"), ("khw", "<i4"), ("layout", "<i4"), ("ldo", "<i4"), ("accumulate", "<i4"), ("alpha", "<f4"), ("kind", "<i4"),
                     ("first_block", "<i4"), ("nblocks", "<i4")])                          # == smx_reduce_item (include/smx.h)

    def __init__(self):
        self._owner = None
        self.segs = []           # recorded: [{"items": [(sig, ws, record)], "waves": [(device table, n items, n blocks)]}]
        self.cur = []
        self.replay = False
        self.complete = False    # the recording step ran to its end (set by the trainer: `finish`); a recording that died after its first flush is not a plan
        self.seg_i = self.item_i = 0

    def finish(self):
        """the step this plan belongs to ended normally."""
        self.complete = True

    def begin(self, tp):
        if self._owner != id(tp.G):
            self._owner, self.segs, self.cur, self.replay, self.complete = id(tp.G), [], [], False, False
        else:
            if self.cur or not self.complete:            # a recording step died half-way (an exception, possibly AFTER its first flush): record again from scratch
                self.segs, self.cur, self.complete = [], [], False
            self.replay = bool(self.segs)
        self.seg_i = self.item_i = 0

    def request(self, tp, sig, n_floats, device):
        """-> (workspace, deferred?)"""
        if not self.replay:
            ws = torch.empty((n_floats,), device=device, dtype=F32)
            self.cur.append([sig, ws, None])
            return ws, False
        seg = self.segs[self.seg_i] if self.seg_i < len(self.segs) else None
        if seg is None or self.item_i >= len(seg["items"]) or seg["items"][self.item_i][0] != sig:
            raise L.SmxError("ReducePlan: this backward asks for a weight gradient the recorded step did not have at this position "
                             f"(segment {self.seg_i}, item {self.item_i}: {sig}) -- a step variant (e.g. the GAN branch) needs its own plan")
        ws = seg["items"][self.item_i][1]
        self.item_i += 1
        return ws, True

    def describe(self, tp, ws, msplit, out, cout, cin, kh, kw, layout, ldo, alpha, bias_out):
        rec = np.zeros(1, dtype=self.ITEM)
        L.check(tp.lib.smx_wgrad_reduce_describe(ws.data_ptr(), msplit, out.data_ptr(), cout, cin, kh, kw, layout, ldo, 1, float(alpha),
                                                 None if bias_out is None else bias_out.data_ptr(), rec.ctypes.data), "wgrad_reduce_describe")
        self.cur[-1][2] = rec

    def flush(self, tp):
        """the end of a piece of the backward: close the recorded segment / finish the deferred items of this one."""
        if not self.replay:
            if torch.cuda.is_current_stream_capturing():
                raise L.SmxError("ReducePlan: the recording step must run eagerly (the table upload is a host copy)")
            items, self.cur = self.cur, []
            waves, seen = {}, {}
            for sig, ws, rec in items:
                wv = seen.get(int(rec["out"][0]), 0)
                seen[int(rec["out"][0])] = wv + 1
                waves.setdefault(wv, []).append(rec)
            tabs = []
            for wv in sorted(waves):
                tab = np.concatenate(waves[wv])
                blk = 0
                for r in tab:
                    r["first_block"] = blk
                    blk += int(r["nblocks"])
                tabs.append((torch.from_numpy(tab.view(np.uint8).copy()).to(items[0][1].device), len(tab), blk))
            self.segs.append({"items": items, "waves": tabs})
            return
        if self.seg_i >= len(self.segs):
            if self.item_i:
                raise L.SmxError("ReducePlan: more backward pieces than recorded")
            return
        seg = self.segs[self.seg_i]
        if self.item_i != len(seg["items"]):
            raise L.SmxError(f"ReducePlan: segment {self.seg_i} ran {self.item_i} of its {len(seg['items'])} recorded weight gradients")
        for tab, n, blk in seg["waves"]:
            L.check(tp.lib.smx_wgrad_reduce_batch(tab.data_ptr(), n, blk, _stream()), "wgrad_reduce_batch")
        self.seg_i += 1
        self.item_i = 0


def _plan_add(tp, ck, w, wc, out, total, cout, cin, kh, kw, mode, kind, frozen):
    if tp.plan is not None and wc is w:                   # a non-contiguous view was copied: nothing stable to point the table at
        tp.plan.add(ck, w, out, total, cout, cin, kh, kw, mode, kind, frozen)


def _packed(tp, ref, mode, cout, cin, kh, kw):
    """per-step cache of a packed weight: mode 0 forward [Cout][(ky,kx,ci)], mode 1 data gradient [Cin][(flipped taps, co)]."""
    w, gslot, key = _param(tp, ref)
    ck = (key, mode)
    if ck not in tp.packed:
        wc = w if w.is_contiguous() else w.contiguous()
        out = torch.empty((cout, kh * kw * cin) if mode == 0 else (cin, kh * kw * cout), device=w.device, dtype=F32)
        L.check(tp.lib.smx_pack_weight_f32(wc.data_ptr(), out.data_ptr(), cout, cin, kh, kw, mode, _stream()), "pack_weight")
        tp.packed[ck] = out
        _plan_add(tp, ck, w, wc, out, out.numel(), cout, cin, kh, kw, mode, PackPlan.F32, gslot is None)
    return tp.packed[ck]


def _packed16(tp, ref, mode, cout, cin, kh, kw):
    """per-step cache of a packed weight rounded to bfloat16 (bf16-compute mode: pack and convert in one launch)."""
    w, gslot, key = _param(tp, ref)
    ck = (key, "bf16", mode)
    if ck not in tp.packed:
        wc = w if w.is_contiguous() else w.contiguous()
        out = torch.empty((cout, kh * kw * cin) if mode == 0 else (cin, kh * kw * cout), device=w.device, dtype=torch.bfloat16)
        L.check(tp.lib.smx_pack_weight_bf16(wc.data_ptr(), out.data_ptr(), cout, cin, kh, kw, mode, _stream()), "pack_weight_bf16")
        tp.packed[ck] = out
        _plan_add(tp, ck, w, wc, out, out.numel(), cout, cin, kh, kw, mode, PackPlan.BF16, gslot is None)
    return tp.packed[ck]


def _conv16(w16, bias, kh, kw, cin, cout):
    """an ops.Conv that only carries the bf16 weight operand (what the mfma16 launch reads)."""
    cv = Conv(None, bias, kh, kw, cin, cout)
    cv._w16 = w16
    return cv


def _packed_u(tp, ref, mode, cout, cin):
    """per-step cache of the Winograd-domain weights of a 3x3 parameter (mode 0 forward, 1 data gradient)."""
    w, gslot, key = _param(tp, ref)
    ck = (key, "u", mode)
    if ck not in tp.packed:
        wc = w if w.is_contiguous() else w.contiguous()
        n, c = (cin, cout) if mode else (cout, cin)
        out = torch.empty((int(tp.lib.smx_winograd_u_floats(n, c)),), device=w.device, dtype=F32)
        L.check(tp.lib.smx_pack_winograd_u_f32(wc.data_ptr(), out.data_ptr(), cout, cin, mode, _stream()), "pack_winograd_u")
        tp.packed[ck] = out
        _plan_add(tp, ck, w, wc, out, out.numel(), cout, cin, 3, 3, mode, PackPlan.WINO_U, gslot is None)
    return tp.packed[ck]


TRAIN_BF3 = bool(int(__import__('os').environ.get('SMX_TRAIN_BF3', '0')))   # the training step's convolutions on the split-bf16 kernels too (default: the exact fp32-MFMA kernels)
WINOGRAD_TRAIN = True       # 3x3 / stride-1 forward and data-gradient convolutions on the fused Winograd kernel (False: implicit GEMM)


# ---- low-level launchers ----------------------------------------------------------------------------------------------------
def _colsum(tp, x2d_ptr, ld, P, Cc, out, accumulate=True, alpha=1.0):
    ws = torch.empty((int(tp.lib.smx_colsum_ws_floats(P, Cc)),), device=out.device, dtype=F32)
    L.check(tp.lib.smx_colsum_f32(x2d_ptr, ld, P, Cc, ws.data_ptr(), out.data_ptr(), int(accumulate), float(alpha), _stream()), "colsum")


def _wgrad(tp, dy, x, out, *, nb=1, M, cout, Hin, Win, cin, Ho, Wo, kh, kw, stride, pt, pl, up2=0, layout=0, ldo=0, accumulate=True,
           alpha=1.0, dy_ld=None, x_ld=None, dy_bs=0, x_bs=0, out_bs=0, bias_out=None, mfma16_ok=False):
    ms = C.c_int(1)
    n = int(tp.lib.smx_wgrad_conv_ws_floats(nb, M, cout, cin, Hin, Win, Ho, Wo, kh, kw, stride, pt, pl, int(up2), C.byref(ms)))
    rp = getattr(tp, "reduce_plan", None)
    planned = rp is not None and nb == 1 and accumulate and out_bs == 0
    deferred = False
    if planned:
        sig = (out.data_ptr(), None if bias_out is None else bias_out.data_ptr(), M, cout, cin, kh, kw, stride, layout, ldo, n, ms.value, float(alpha))
        ws, deferred = rp.request(tp, sig, n, out.device)
    else:
        ws = torch.empty((n,), device=out.device, dtype=F32)
    dyp, ldy = (dy.data_ptr(), dy_ld) if dy_ld is not None else _pix(dy)[:2]
    xp, ldx = (x.data_ptr(), x_ld) if x_ld is not None else _pix(x)[:2]
    bf = bool(tp.mfma16 and mfma16_ok)
    fn = tp.lib.smx_wgrad_mfma16_f32 if bf else tp.lib.smx_wgrad_f32
    K = kh * kw * cin
    meta = {"flops": 2.0 * nb * M * cout * K, "M": M, "N": cout, "K": K, "nb": nb, "k": kh, "bf16": int(bf)} if ops._PROFILE is not None else None
    L.check(ops._timed("wgrad_bf16" if bf else "wgrad", meta, fn, dyp, ldy, dy_bs, xp, ldx, x_bs, nb, M, cout, Hin, Win, cin, Ho, Wo, kh, kw, stride, pt, pl,
                       int(up2), ws.data_ptr(), ms.value, out.data_ptr(), out_bs, layout, ldo, int(accumulate) | (2 if deferred else 0), float(alpha),
                       None if bias_out is None else bias_out.data_ptr(), _stream()), "wgrad")
    if planned and not deferred:
        rp.describe(tp, ws, ms.value, out, cout, cin, kh, kw, layout, ldo, alpha, bias_out)


def act_bwd(tp, g, ref, act):
    out = _empty(g.shape, g)
    gp, ldg, P, Cc = _pix(g)
    rp, ldr, _, _ = _pix(ref)
    L.check(tp.lib.smx_act_bwd_f32(gp, ldg, rp, ldr, out.data_ptr(), Cc, P, Cc, act, _stream()), "act_bwd")
    return out


def scaled(tp, x, alpha):
    """alpha * x as a new dense tensor (no tape)."""
    xd = x if x.is_contiguous() else _dense(tp.lib, x)
    out = _empty(x.shape, x)
    L.check(tp.lib.smx_scale_f32(xd.data_ptr(), out.data_ptr(), xd.numel(), float(alpha), 0, _stream()), "scale")
    return out


def transpose(tp, x, nb, R, Cc):
    """[nb][R][C] -> [nb][C][R] (dense)."""
    y = torch.empty((nb, Cc, R), device=x.device, dtype=F32)
    L.check(tp.lib.smx_transpose_f32(x.data_ptr(), Cc, R * Cc, y.data_ptr(), R, R * Cc, nb, R, Cc, _stream()), "transpose")
    return y


# ---- convolution / Linear -------------------------------------------------------------------------------------------------
def conv(tp, x, w, b=None, *, stride=1, pad=None, up2=False, act=ACT_NONE, res=None, out_hw=None, kind="conv", patch=None, frozen=False):
    """y = act(conv(x) + bias) (+ res).  kind:
      "conv":    parameter [Cout,Cin,kh,kw] (or Linear [out,in] = 1x1); stride 1, or the stride-2 / pad (0,1,0,1) Downsample form;
      "patch":   Linear [Cout, p*p*C] applied as a p x p / stride-p convolution (patchify + Linear, patch=(p, C));
      "unpatch": Linear [p*p*C, Cin] as a 1x1 convolution with the un-patchify (depth-to-space) store (patch=(p, C))."""
    wv, wg, _ = _param(tp, w)
    bv, bg = (None, None) if b is None else _param(tp, b)[:2]
    B, H, W, Cin = x.shape
    if act not in (ACT_NONE, ACT_RELU, ACT_LRELU02):
        raise L.SmxError("train conv: only ReLU / LeakyReLU epilogues are fused (their derivative needs the output only)")
    if act != ACT_NONE and res is not None:
        raise L.SmxError("train conv: activation + residual in one epilogue hides the activation's output (no call site of the path has both)")
    if kind == "conv":
        if wv.dim() == 4:
            cout, cin, kh, kw = wv.shape
        else:
            (cout, cin), kh, kw = wv.shape, 1, 1
        d2s = None
    elif kind == "patch":
        p_, Cc = patch
        cout, cin, kh, kw, stride, pad, d2s = wv.shape[0], Cc, p_, p_, p_, (0, 0), None
    elif kind == "unpatch":
        p_, Cc = patch
        cout, cin, kh, kw, d2s = wv.shape[0], wv.shape[1], 1, 1, (p_, Cc)
    else:
        raise ValueError(kind)
    if cin != Cin:
        raise L.SmxError(f"train conv {w}: input has {Cin} channels, weight expects {cin}")
    if stride not in (1, 2) and kind == "conv":
        raise L.SmxError("train conv: stride 1 or 2")
    m16 = tp.mfma16
    bc = None if bv is None else bv.contiguous()
    if m16 and kind != "patch":
        cv = _conv16(_packed16(tp, w, 0, cout, cin, kh, kw), bc, kh, kw, cin, cout)
    else:
        wp = wv.contiguous() if kind == "patch" else _packed(tp, w, 0, cout, cin, kh, kw)      # the patch Linear is already [Cout][(p1 p2 c)]
        cv = Conv(wp.view(cout, kh * kw * cin), bc, kh, kw, cin, cout)
    pt, pl = (kh // 2, kw // 2) if pad is None else pad
    He, We = (2 * H, 2 * W) if up2 else (H, W)
    # bf16-compute mode: forward and data gradient through the bf16 implicit GEMM (fp32 tensors converted while staging, fp32 out)
    f16 = dict(mfma16=True, out_dtype=F32) if m16 else {}
    # the fused Winograd F(2x2,3x3) kernel takes this step's weights in its own packing (smx_pack_winograd_u_f32)
    wino3 = (not m16) and WINOGRAD_TRAIN and kind == "conv" and (kh, kw, stride, pt, pl) == (3, 3, 1, 1, 1) and He % 8 == 0 and We % 16 == 0
    if wino3 and cin % 32 == 0:
        cv._u = _packed_u(tp, w, 0, cout, cin)
    y = ops.conv(x, cv, stride=stride, pad=(pt, pl), up2=bool(up2), act=act, res=res, out_hw=out_hw, d2s=d2s, direct=cv._u is None, bf3=TRAIN_BF3, **f16)
    Ho, Wo = (y.shape[1], y.shape[2]) if d2s is None else (H, W)

    def bwd():
        g = tp.take(y)
        if g is None:
            return
        if act != ACT_NONE:
            g = act_bwd(tp, g, y, act)
        if res is not None:
            tp.acc(res, g, owned=False)
        lib = tp.lib
        if kind == "unpatch":
            gs = torch.empty((B, H, W, cout), device=g.device, dtype=F32)        # tokens x (p1 p2 c): the store's inverse
            gp_, ldg_, _, _ = _pix(g)
            L.check(lib.smx_space_to_depth_f32(gp_, ldg_, gs.data_ptr(), B, H, W, d2s[1], d2s[0], _stream()), "space_to_depth")
            g2 = gs
        else:
            g2 = g if g.is_contiguous() else _dense(lib, g)
        M = B * Ho * Wo
        # weight gradient (+ the bias gradient out of the same pass over g); frozen: a fixed feature extractor, data gradient only
        if frozen:
            pass
        elif kind == "patch":
            _wgrad(tp, g2, x, wg, M=M, cout=cout, Hin=H, Win=W, cin=cin, Ho=Ho, Wo=Wo, kh=kh, kw=kw, stride=stride, pt=0, pl=0,
                   layout=1, ldo=kh * kw * cin, dy_ld=cout, bias_out=bg, mfma16_ok=True)
        else:
            _wgrad(tp, g2, x, wg, M=M, cout=cout, Hin=H, Win=W, cin=cin, Ho=Ho, Wo=Wo, kh=kh, kw=kw, stride=stride, pt=pt, pl=pl,
                   up2=1 if up2 else 0, layout=0, dy_ld=cout, bias_out=bg, mfma16_ok=True)
        # data gradient: a forward convolution of g with the transposed, tap-flipped weights
        if not tp.needs(x):
            return
        # (the patch Linear is stored [Cout][(p1 p2 c)], not OIHW: its data-gradient operand is the plain transpose)
        pk = _packed16 if (m16 and stride == 1) else _packed        # (the stride-2 zero-insert gather has no bf16 form: fp32 operand)
        mk = (lambda t, *dims: _conv16(t, None, *dims)) if (m16 and stride == 1) else (lambda t, *dims: Conv(t, None, *dims))
        wt = pk(tp, w, 1, cout, kh * kw * cin, 1, 1) if kind == "patch" else pk(tp, w, 1, cout, cin, kh, kw)
        if kind == "patch":
            dx = ops.conv(g2, mk(wt.view(kh * kw * cin, cout), 1, 1, cout, kh * kw * cin), d2s=(kh, cin), direct=True, **f16)
        elif kind == "unpatch":
            dx = ops.conv(g2, mk(wt.view(cin, cout), 1, 1, cout, cin), direct=True, **f16)
        elif stride == 1:
            dcv = mk(wt.view(cin, kh * kw * cout), kh, kw, cout, cin)
            if wino3 and cout % 32 == 0:
                dcv._u = _packed_u(tp, w, 1, cout, cin)
            dx = ops.conv(g2, dcv, pad=(kh - 1 - pt, kw - 1 - pl), out_hw=(He, We), direct=dcv._u is None, bf3=TRAIN_BF3, **f16)
            if up2:             # adjoint of nearest x2: sum of each 2x2 block
                dx = scaled(tp, ops.avgpool2(dx), 4.0)
        else:                   # stride 2: zero-insert gather (smx_gemm_conv_f32 up2 = 2)
            dcv = Conv(wt.view(cin, kh * kw * cout), None, kh, kw, cout, cin)
            dx = ops.conv(g2, dcv, up2=2, pad=(kh - 1 - pt, kw - 1 - pl), out_hw=(H, W), direct=True)
        tp.acc(x, dx)
    tp.record(bwd)
    return y


def act(tp, x, kind):
    """y = act(x) as its own pass (GELU / swish / sigmoid: the derivative needs the pre-activation, which stays alive here)."""
    y = _empty(x.shape, x)
    xp, ldx, P, Cc = _pix(x)
    L.check(tp.lib.smx_act_f32(xp, ldx, y.data_ptr(), Cc, P, Cc, kind, _stream()), "act")

    def bwd():
        g = tp.take(y)
        if g is not None and tp.needs(x):
            tp.acc(x, act_bwd(tp, g, y if kind in (ACT_RELU, ACT_LRELU02, ACT_SIGMOID) else x, kind))
    tp.record(bwd)
    return y


# ---- normalisation ---------------------------------------------------------------------------------------------------------
def groupnorm(tp, x, gamma, beta, swish=True, groups=32, eps=1e-6):
    gv, gg, _ = _param(tp, gamma)
    bv, bg, _ = _param(tp, beta)
    B, H, W, Cc = x.shape
    lib = tp.lib
    xp, ldx = ops._pix(x, "groupnorm input")
    ss = torch.empty((B, Cc, 2), device=x.device, dtype=F32)
    mr = torch.empty((B, groups, 2), device=x.device, dtype=F32)
    ws = torch.empty((int(lib.smx_groupnorm_ws_floats(B, H * W, Cc)) + 2 * B * groups,), device=x.device, dtype=F32)
    L.check(lib.smx_groupnorm_stats_train_f32(xp, ldx, gv.data_ptr(), bv.data_ptr(), ss.data_ptr(), mr.data_ptr(), B, H * W, Cc, groups, eps,
                                              ws.data_ptr(), _stream()), "groupnorm_stats_train")
    y = ops.groupnorm_apply(x, ss, swish)

    def bwd():
        g = tp.take(y)
        if g is None:
            return
        g = g if g.is_contiguous() else _dense(lib, g)
        dx = _empty((B, H, W, Cc), x)
        L.check(lib.smx_groupnorm_bwd_f32(xp, ldx, g.data_ptr(), Cc, ss.data_ptr(), mr.data_ptr(), gv.data_ptr(), dx.data_ptr(), Cc,
                                          gg.data_ptr(), bg.data_ptr(), B, H * W, Cc, groups, int(swish), ws.data_ptr(), _stream()), "groupnorm_bwd")
        tp.acc(x, dx)
    tp.record(bwd)
    return y


def layernorm(tp, x, gamma, beta, pos=None, eps=1e-5):
    """-> (LN(x), LN(x) + pos or None); pos: a `leaf` tensor [npos, E] (its gradient is the batch sum of d(LN(x)+pos))."""
    gv, gg, _ = _param(tp, gamma)
    bv, bg, _ = _param(tp, beta)
    xc = x if x.is_contiguous() else _dense(tp.lib, x)
    y, yp = ops.layernorm(xc, gv, bv, pos=pos, eps=eps)
    E = x.shape[-1]
    T = x.numel() // E

    def bwd():
        gy, gyp = tp.take(y), (tp.take(yp) if yp is not None else None)
        if gy is None and gyp is None:
            return
        lib = tp.lib
        gy = gy if gy is None or gy.is_contiguous() else _dense(lib, gy)
        gyp = gyp if gyp is None or gyp.is_contiguous() else _dense(lib, gyp)
        dx = _empty(x.shape, x)
        ws = torch.empty((int(lib.smx_layernorm_bwd_ws_floats(T, E)),), device=x.device, dtype=F32)
        dpos = None
        if gyp is not None and pos is not None and tp.needs(pos):
            dpos = _zeros(pos.shape, pos)
        L.check(lib.smx_layernorm_bwd_f32(xc.data_ptr(), gv.data_ptr(), None if gy is None else gy.data_ptr(),
                                          None if gyp is None else gyp.data_ptr(), dx.data_ptr(), gg.data_ptr(), bg.data_ptr(),
                                          None if dpos is None else dpos.data_ptr(), T, E, 0 if pos is None else pos.shape[0], eps,
                                          ws.data_ptr(), _stream()), "layernorm_bwd")
        if dpos is not None:
            tp.acc(pos, dpos)
        tp.acc(x, dx)
    tp.record(bwd)
    return y, yp


# ---- attention ---------------------------------------------------------------------------------------------------------------
def attention(tp, q, k, v, nhead, dh, S, *, mask=None, ctx=None):
    """o = softmax(q k^T / sqrt(dh) [+ key mask]) v per head.  Self attention: q, k, v dense [B,L,E].  Cross attention against a
    projected codebook: ctx = the [K, 2E] = [K | V] tensor (k, v ignored), first S rows used, shared by the batch."""
    B = q.shape[0]
    E = nhead * dh
    Lq = q.numel() // B // E
    if ctx is not None:
        k, v = ctx[:, :E], ctx[:, E:]
    o = ops.attention(q, k, v, nhead, dh, S, k_shared=ctx is not None, mask=mask)

    def bwd():
        g = tp.take(o)
        if g is None:
            return
        lib = tp.lib
        g = g if g.is_contiguous() else _dense(lib, g)
        shared = ctx is not None
        dq = _empty((B, Lq, E), q)
        dk = _empty((B, S, E), q)
        dv = _empty((B, S, E), q)
        stats = _empty((B * nhead * Lq * 3,), q)
        qp, ldq = ops._pix(q, "attention q")
        kp, ldk = ops._pix(k, "attention k")
        vp, ldv = ops._pix(v, "attention v")
        kbs = 0 if shared else S * ldk
        vbs = 0 if shared else S * ldv
        L.check(lib.smx_attention_bwd_f32(qp, ldq, Lq * ldq, kp, ldk, kbs, vp, ldv, vbs, o.data_ptr(), g.data_ptr(),
                                          None if mask is None else mask.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                          stats.data_ptr(), B, nhead, Lq, S, dh, dh ** -0.5, _stream()), "attention_bwd")
        tp.acc(q, dq.view(q.shape))
        if shared:                       # per-sample dK / dV -> their batch sums (fixed order) -> rows :S of d ctx = [dK | dV]
            sk, sv = _zeros((S, E), q), _zeros((S, E), q)
            L.check(lib.smx_batch_sum_f32(dk.data_ptr(), sk.data_ptr(), B, S * E, _stream()), "batch_sum")
            L.check(lib.smx_batch_sum_f32(dv.data_ptr(), sv.data_ptr(), B, S * E, _stream()), "batch_sum")
            dctx = _zeros(ctx.shape, ctx)
            _axpy(lib, sk, dctx[:S, :E], 1.0)
            _axpy(lib, sv, dctx[:S, E:], 1.0)
            tp.acc(ctx, dctx)
        else:
            tp.acc(k, dk.view(k.shape))
            tp.acc(v, dv.view(v.shape))
    tp.record(bwd)
    return o


def attn_core(tp, q, k, v, scale):
    """AttnBlock core (archs/vqgan_arch.py:236-249): h = softmax(scale q k^T) v, one head of width C; q, k, v dense [B,H,W,C].
    The [B,N,N] probabilities are kept for the backward (N = 1024: 4 MB per sample)."""
    B, H, W, Cc = q.shape
    N = H * W
    s = torch.empty((B, N, N), device=q.device, dtype=F32)
    ops.gemm_nt(q, k, s, M=N, N=N, K=Cc, lda=Cc, ldb=Cc, ldc=N, nb0=B, a_bs=(N * Cc, 0), bt_bs=(N * Cc, 0), c_bs=(N * N, 0))
    ops.softmax_rows(s, N, float(scale))
    vt = transpose(tp, v, B, N, Cc)
    h = torch.empty((B, H, W, Cc), device=q.device, dtype=F32)
    ops.gemm_nt(s, vt, h, M=N, N=Cc, K=N, lda=N, ldb=N, ldc=Cc, nb0=B, a_bs=(N * N, 0), bt_bs=(Cc * N, 0), c_bs=(N * Cc, 0))

    def bwd():
        g = tp.take(h)
        if g is None:
            return
        lib = tp.lib
        g = g if g.is_contiguous() else _dense(lib, g)
        dP = torch.empty((B, N, N), device=q.device, dtype=F32)
        ops.gemm_nt(g, v, dP, M=N, N=N, K=Cc, lda=Cc, ldb=Cc, ldc=N, nb0=B, a_bs=(N * Cc, 0), bt_bs=(N * Cc, 0), c_bs=(N * N, 0))
        dv = _empty((B, H, W, Cc), q)        # dV[j][c] = sum_i P[i][j] dH[i][c]
        _wgrad(tp, s, g, dv, nb=B, M=N, cout=N, Hin=N, Win=1, cin=Cc, Ho=N, Wo=1, kh=1, kw=1, stride=1, pt=0, pl=0, layout=1, ldo=Cc,
               accumulate=False, dy_ld=N, x_ld=Cc, dy_bs=N * N, x_bs=N * Cc, out_bs=N * Cc)
        L.check(lib.smx_softmax_rows_bwd_f32(s.data_ptr(), dP.data_ptr(), B * N, N, float(scale), _stream()), "softmax_bwd")
        kt = transpose(tp, k, B, N, Cc)
        dq = _empty((B, H, W, Cc), q)        # dQ = dS K
        ops.gemm_nt(dP, kt, dq, M=N, N=Cc, K=N, lda=N, ldb=N, ldc=Cc, nb0=B, a_bs=(N * N, 0), bt_bs=(Cc * N, 0), c_bs=(N * Cc, 0))
        dk = _empty((B, H, W, Cc), q)        # dK[j][c] = sum_i dS[i][j] Q[i][c]
        _wgrad(tp, dP, q, dk, nb=B, M=N, cout=N, Hin=N, Win=1, cin=Cc, Ho=N, Wo=1, kh=1, kw=1, stride=1, pt=0, pl=0, layout=1, ldo=Cc,
               accumulate=False, dy_ld=N, x_ld=Cc, dy_bs=N * N, x_bs=N * Cc, out_bs=N * Cc)
        tp.acc(q, dq)
        tp.acc(k, dk)
        tp.acc(v, dv)
    tp.record(bwd)
    return h


# ---- warp / resize -----------------------------------------------------------------------------------------------------------
def warp(tp, feat, flow, occ=None):
    out = ops.warp(feat, flow, occ)
    B, Hf, Wf, _ = flow.shape
    _, H, W, Cc = feat.shape

    def bwd():
        g = tp.take(out)
        if g is None:
            return
        lib = tp.lib
        g = g if g.is_contiguous() else _dense(lib, g)
        need_f = tp.needs(feat)
        need_m = tp.needs(flow) or (occ is not None and tp.needs(occ))
        dfeat = _zeros(feat.shape, feat) if need_f else None
        gsm = _empty((B, H, W, 3), feat) if need_m else None
        L.check(lib.smx_warp_bwd_f32(feat.data_ptr(), feat.shape[0], flow.data_ptr(), None if occ is None else occ.data_ptr(), g.data_ptr(),
                                     None if dfeat is None else dfeat.data_ptr(), None if gsm is None else gsm.data_ptr(),
                                     B, H, W, Cc, Hf, Wf, _stream()), "warp_bwd")
        if need_f:
            tp.acc(feat, dfeat)
        if need_m:
            if (Hf, Wf) != (H, W):
                coarse = _zeros((B, Hf, Wf, 3), feat)
                L.check(lib.smx_resize_ac_bwd_f32(gsm.data_ptr(), 3, coarse.data_ptr(), 3, B, Hf, Wf, H, W, 3, _stream()), "resize_ac_bwd")
            else:
                coarse = gsm
            tp.acc(flow, _dense(lib, coarse[..., :2]))
            if occ is not None:
                tp.acc(occ, _dense(lib, coarse[..., 2:3]).view(occ.shape))
    tp.record(bwd)
    return out


def resize(tp, x, Ho, Wo):
    y = ops.resize(x, Ho, Wo)
    B, H, W, Cc = x.shape

    def bwd():
        g = tp.take(y)
        if g is None or not tp.needs(x):
            return
        dx = _zeros(x.shape, x)
        gp, ldg, _, _ = _pix(g)
        L.check(tp.lib.smx_resize_ac_bwd_f32(gp, ldg, dx.data_ptr(), Cc, B, H, W, Ho, Wo, Cc, _stream()), "resize_ac_bwd")
        tp.acc(x, dx)
    tp.record(bwd)
    return y


# ---- vector quantiser -----------------------------------------------------------------------------------------------------------
def quantize(tp, z, cb_name, Ks, beta):
    """VectorQuantizer.forward on NHWC tokens -> (z_q NHWC, loss [1] device scalar, stats).  Straight-through: d z_q flows to z;
    the codebook loss moves z (beta term) and the selected codebook rows."""
    B, H, W, D = z.shape
    zc = z if z.is_contiguous() else _dense(tp.lib, z)
    cb = tp.P[cb_name]
    idx, zq, dmin, sq = ops.vq_nearest(zc.view(-1, D), cb, Ks)
    loss = scaled(tp, sq, (1.0 + beta) / float(zc.numel()))
    zq = zq.view(B, H, W, D)

    def bwd():
        gz, gl = tp.take(zq), tp.take(loss)
        if gz is None and gl is None:
            return
        lib = tp.lib
        gz = gz if gz is None or gz.is_contiguous() else _dense(lib, gz)
        dz = _empty(zc.shape, zc)
        L.check(lib.smx_vq_bwd_f32(zc.data_ptr(), cb.data_ptr(), idx.data_ptr(), None if gz is None else gz.data_ptr(),
                                   None if gl is None else gl.data_ptr(), float(beta), dz.data_ptr(), tp.G[cb_name].data_ptr(),
                                   zc.numel() // D, D, _stream()), "vq_bwd")
        tp.acc(z, dz.view(z.shape))
    tp.record(bwd)
    return zq, loss, {"min_encoding_indices": idx.view(-1, 1), "min_distance": dmin}


# ---- flow / occlusion / SFT / plumbing --------------------------------------------------------------------------------------------
def flow_to_residual(tp, flow):
    res = ops.flow_to_residual(flow)
    hs = (flow.shape[1] - 1) / 2.0

    def bwd():
        g = tp.take(res)
        if g is not None and tp.needs(flow):
            tp.acc(flow, scaled(tp, g, hs))
    tp.record(bwd)
    return res


def flow_occ_update(tp, flow, r, occ_prev):
    m_com, res_norm, occ = ops.flow_occ_update(flow, r, occ_prev)
    B, H, W, _ = flow.shape

    def bwd():
        gm, go = tp.take(m_com), tp.take(occ)
        tp.take(res_norm)
        if gm is None and go is None:
            return
        lib = tp.lib
        gm = gm if gm is None or gm.is_contiguous() else _dense(lib, gm)
        go = go if go is None or go.is_contiguous() else _dense(lib, go)
        d_flow, d_r, d_occ = _empty(flow.shape, flow), _empty(r.shape, r), _empty(occ_prev.shape, occ_prev)
        L.check(lib.smx_flow_occ_update_bwd_f32(None if gm is None else gm.data_ptr(), None if go is None else go.data_ptr(), occ.data_ptr(),
                                                d_flow.data_ptr(), d_r.data_ptr(), d_occ.data_ptr(), B, H, W, _stream()), "flow_occ_update_bwd")
        tp.acc(flow, d_flow)
        tp.acc(r, d_r)
        tp.acc(occ_prev, d_occ)
    tp.record(bwd)
    return m_com, res_norm, occ


def sft_combine(tp, dec, scale, shift, w):
    out = ops.sft_combine(dec, scale, shift, w)
    Cc = dec.shape[-1]

    def bwd():
        g = tp.take(out)
        if g is None:
            return
        lib = tp.lib
        g = g if g.is_contiguous() else _dense(lib, g)
        dp, ldd = ops._pix(dec, "dec")
        d_dec, d_sc, d_sh = _empty(g.shape, g), _empty(g.shape, g), _empty(g.shape, g)
        L.check(lib.smx_sft_combine_bwd_f32(g.data_ptr(), dp, ldd, scale.data_ptr(), d_dec.data_ptr(), d_sc.data_ptr(), d_sh.data_ptr(),
                                            float(w), g.numel() // Cc, Cc, _stream()), "sft_combine_bwd")
        tp.acc(dec, d_dec)
        tp.acc(scale, d_sc)
        tp.acc(shift, d_sh)
    tp.record(bwd)
    return out


def scale(tp, x, alpha):
    y = scaled(tp, x, alpha)

    def bwd():
        g = tp.take(y)
        if g is not None and tp.needs(x):
            tp.acc(x, scaled(tp, g, alpha))
    tp.record(bwd)
    return y


def cat(tp, parts):
    """channel concatenation into a fresh buffer (copy kernels); gradient = dense copies of the slices."""
    Cs = [p.shape[-1] for p in parts]
    out = _empty(parts[0].shape[:-1] + (sum(Cs),), parts[0])
    o = 0
    for p, c in zip(parts, Cs):
        ops.copy_slice(p, out[..., o:o + c])
        o += c

    def bwd():
        g = tp.take(out)
        if g is None:
            return
        o2 = 0
        for p, c in zip(parts, Cs):
            if tp.needs(p):
                tp.acc(p, _dense(tp.lib, g[..., o2:o2 + c]))
            o2 += c
    tp.record(bwd)
    return out


def slice_ch(tp, x, a, b):
    """x[..., a:b] as a view usable by every launcher; its gradient lands in a zero-initialised full-width buffer."""
    y = x[..., a:b]

    def bwd():
        g = tp.take(y)
        if g is None or not tp.needs(x):
            return
        full = _zeros(x.shape, x)
        _axpy(tp.lib, g, full[..., a:b], 1.0)
        tp.acc(x, full)
    tp.record(bwd)
    return y


def detach(tp, x):
    """x.detach(): the same memory under a new tensor object that takes no gradient."""
    return tp.stop(x.view(x.shape))


def view(tp, x, shape):
    """a reshaping view (dense tensors) -- tokens [B,1024,E] <-> maps [B,32,32,E]."""
    y = x.view(shape)

    def bwd():
        g = tp.take(y)
        if g is not None and tp.needs(x):
            tp.acc(x, (g if g.is_contiguous() else _dense(tp.lib, g)).view(x.shape), owned=False)
    tp.record(bwd)
    return y


def nchw_to_nhwc(tp, x):
    y = ops.nchw_to_nhwc(x)

    def bwd():
        g = tp.take(y)
        if g is not None and tp.needs(x):
            tp.acc(x, ops.nhwc_to_nchw(g))
    tp.record(bwd)
    return y


def nhwc_to_nchw(tp, x):
    y = ops.nhwc_to_nchw(x)

    def bwd():
        g = tp.take(y)
        if g is not None and tp.needs(x):
            tp.acc(x, ops.nchw_to_nhwc(g.contiguous()))
    tp.record(bwd)
    return y


# ---- losses ---------------------------------------------------------------------------------------------------------------------
def l1_loss(tp, a, target, weight=1.0):
    """weight * mean |a - target| -> [1] device scalar; target takes no gradient (losses/losses.py L1Loss, reduction='mean')."""
    ac = a if a.is_contiguous() else _dense(tp.lib, a)
    tc = target if target.is_contiguous() else _dense(tp.lib, target)
    if ac.numel() != tc.numel():
        raise L.SmxError("l1_loss: operand sizes differ")
    out = _empty((1,), a)
    part = _empty((1024,), a)
    L.check(tp.lib.smx_l1_loss_f32(ac.data_ptr(), tc.data_ptr(), ac.numel(), float(weight), part.data_ptr(), out.data_ptr(), _stream()), "l1_loss")

    def bwd():
        g = tp.take(out)
        if g is None or not tp.needs(a):
            return
        da = _empty(ac.shape, ac)
        L.check(tp.lib.smx_l1_loss_bwd_f32(ac.data_ptr(), tc.data_ptr(), g.data_ptr(), ac.numel(), float(weight), da.data_ptr(), 0, _stream()), "l1_loss_bwd")
        tp.acc(a, da.view(a.shape))
    tp.record(bwd)
    return out


def weighted_sum(tp, terms):
    """sum_i w_i * t_i over [1] device scalars -> [1]."""
    out = _zeros((1,), terms[0][0])
    for t, w in terms:
        L.check(tp.lib.smx_scale_f32(t.data_ptr(), out.data_ptr(), 1, float(w), 1, _stream()), "scale")

    def bwd():
        g = tp.take(out)
        if g is None:
            return
        for t, w in terms:
            if tp.needs(t):
                tp.acc(t, scaled(tp, g, w))
    tp.record(bwd)
    return out


# ---- motion estimator (training mode) ---------------------------------------------------------------------------------------------
def bn_relu(tp, x, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, relu=True):
    """relu(BatchNorm2d(x)) with BATCH statistics (utils/motion_estimator_util.py:214-231, 363-380 under .train()); the running
    buffers (tensors, updated in place like F.batch_norm(training=True)) are optional.  relu=False: the plain BatchNorm2d of the
    discriminator (archs/vqgan_arch.py:546-559; its LeakyReLU is a separate `act`)."""
    gv, gg, _ = _param(tp, gamma)
    bv, bg, _ = _param(tp, beta)
    lib = tp.lib
    xp, ldx, P, Cc = _pix(x)
    y = _empty(x.shape, x)
    mr = _empty((Cc, 2), x)
    ws = _empty((int(lib.smx_batchnorm_ws_floats(P, Cc)),), x)
    L.check(lib.smx_batchnorm_train_f32(xp, ldx, y.data_ptr(), Cc, gv.data_ptr(), bv.data_ptr(), mr.data_ptr(),
                                        None if running_mean is None else running_mean.data_ptr(),
                                        None if running_var is None else running_var.data_ptr(), P, Cc, eps, momentum, int(relu), ws.data_ptr(),
                                        _stream()), "batchnorm_train")

    def bwd():
        g = tp.take(y)
        if g is None:
            return
        g = g if g.is_contiguous() else _dense(lib, g)
        dx = _empty(x.shape, x)
        L.check(lib.smx_batchnorm_train_bwd_f32(xp, ldx, g.data_ptr(), Cc, y.data_ptr() if relu else None, Cc, mr.data_ptr(), gv.data_ptr(), dx.data_ptr(), Cc,
                                                gg.data_ptr(), bg.data_ptr(), P, Cc, ws.data_ptr(), _stream()), "batchnorm_train_bwd")
        tp.acc(x, dx)
    tp.record(bwd)
    return y


def avgpool2(tp, x):
    y = ops.avgpool2(x)
    B, H, W, Cc = x.shape

    def bwd():
        g = tp.take(y)
        if g is None or not tp.needs(x):
            return
        dx = _empty(x.shape, x)
        gp, ldg, _, _ = _pix(g)
        L.check(tp.lib.smx_avgpool2_bwd_f32(gp, ldg, dx.data_ptr(), Cc, B, H, W, Cc, _stream()), "avgpool2_bwd")
        tp.acc(x, dx)
    tp.record(bwd)
    return y


def kp_head(tp, logits, jmaps, K, temperature):
    """gaussian2kp + heatmap-weighted jacobians (archs/keypoint_detector_arch.py:48-86): logits [B,58,58,K], jmaps [B,58,58,4K]
    -> (value [B,K,2], jacobian [B,K,2,2])."""
    value, jac = ops.kp_head(logits, jmaps, K, temperature)
    B, H, W, _ = logits.shape

    def bwd():
        gv, gj = tp.take(value), tp.take(jac)
        if gv is None and gj is None:
            return
        lib = tp.lib
        gv = gv if gv is None or gv.is_contiguous() else gv.contiguous()
        gj = gj if gj is None or gj.is_contiguous() else gj.contiguous()
        dl, dj = _empty((B, H, W, K), logits), _empty((B, H, W, 4 * K), logits)
        lp, ldl = ops._pix(logits, "kp logits")
        jp, ldj = ops._pix(jmaps, "kp jacobian maps")
        L.check(lib.smx_kp_head_bwd_f32(lp, ldl, jp, ldj, None if gv is None else gv.data_ptr(), None if gj is None else gj.data_ptr(),
                                        dl.data_ptr(), K, dj.data_ptr(), 4 * K, B, H, W, K, float(temperature), _stream()), "kp_head_bwd")
        tp.acc(logits, dl)
        tp.acc(jmaps, dj)
    tp.record(bwd)
    return value, jac


def sparse_motion(tp, src64, kpd_value, kpd_jac, kps_value, kps_jac, K=15, var=0.01):
    """heatmaps + sparse motions + 16 sparse warps (archs/dense_motion_arch.py:65-116) -> (hg_in [B,64,64,4(K+1)], sparse
    [B,K+1,64,64,2], drv_heat [B,64,64,K]); gradients reach the four keypoint tensors."""
    B = kpd_value.shape[0]
    hg_in = _empty((B, 64, 64, 4 * (K + 1)), src64)
    dj, sj = kpd_jac.reshape(B, K, 4), kps_jac.reshape(B, K, 4)
    sparse, heat = ops.sparse_motion(src64, kpd_value, dj, kps_value, sj, hg_in, B, K, var)

    def bwd():
        gh, gs, gd = tp.take(hg_in), tp.take(sparse), tp.take(heat)
        if gh is None and gs is None and gd is None:
            return
        lib = tp.lib
        if gh is None:
            gh = _zeros(hg_in.shape, hg_in)
        gh = gh if gh.is_contiguous() else _dense(lib, gh)
        gs = gs if gs is None or gs.is_contiguous() else gs.contiguous()
        gd = gd if gd is None or gd.is_contiguous() else _dense(lib, gd)
        d = [_empty((B, K, 2), src64), _empty((B, K, 4), src64), _empty((B, K, 2), src64), _empty((B, K, 4), src64)]
        live = [t if t.is_contiguous() else t.contiguous() for t in (kpd_value, dj, kps_value, sj)]       # referenced until the launch is queued (ops._live)
        L.check(lib.smx_sparse_motion_bwd_f32(src64.data_ptr(), live[0].data_ptr(), live[1].data_ptr(),
                                              live[2].data_ptr(), live[3].data_ptr(), gh.data_ptr(), 4 * (K + 1),
                                              None if gs is None else gs.data_ptr(), None if gd is None else gd.data_ptr(),
                                              d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), B, 64, 64, K, float(var), _stream()),
                "sparse_motion_bwd")
        tp.acc(kpd_value, d[0])
        tp.acc(kpd_jac, d[1].view(kpd_jac.shape))
        tp.acc(kps_value, d[2])
        tp.acc(kps_jac, d[3].view(kps_jac.shape))
    tp.record(bwd)
    return hg_in, sparse, heat


def mask_deformation(tp, mlog, sparse, K1):
    """mask softmax + flow blend + occlusion sigmoid (archs/dense_motion_arch.py:140-158): mlog [B,64,64,K1+1] -> (deformation
    [B,64,64,2], occlusion [B,64,64])."""
    deform, _, occ = ops.mask_deformation(mlog, sparse, want_mask=False, K1=K1, fused_occ=True)
    B, H, W, Cl = mlog.shape

    def bwd():
        gd, go = tp.take(deform), tp.take(occ)
        if gd is None and go is None:
            return
        lib = tp.lib
        gd = gd if gd is None or gd.is_contiguous() else _dense(lib, gd)
        go = go if go is None or go.is_contiguous() else go.contiguous()
        dl = _zeros(mlog.shape, mlog)
        ds = _empty(sparse.shape, sparse)
        mp, ldm = ops._pix(mlog, "mask logits")
        L.check(lib.smx_mask_deformation_bwd_f32(mp, ldm, sparse.data_ptr(), None if gd is None else gd.data_ptr(),
                                                 None if go is None else go.data_ptr(), dl.data_ptr(), Cl, ds.data_ptr(), B, H, W, K1, _stream()),
                "mask_deformation_bwd")
        tp.acc(mlog, dl)
        tp.acc(sparse, ds)
    tp.record(bwd)
    return deform, occ


# ---- perceptual-loss pieces (csrc/train_percep.hip; reference losses/losses.py:293-387, archs/vgg_arch.py:167-210) -----------------------
def antialias(tp, x, w, step):
    """AntiAliasInterpolation2d on NHWC: zero pad K/2, depthwise Gaussian w [K,K] (device tensor, the same for every channel), every
    `step`-th output."""
    B, H, W, Cc = x.shape
    K = w.shape[-1]
    Ho, Wo = (H + step - 1) // step, (W + step - 1) // step
    y = _empty((B, Ho, Wo, Cc), x)
    xp, ldx, _, _ = _pix(x)
    L.check(tp.lib.smx_antialias_nhwc_f32(xp, ldx, w.data_ptr(), y.data_ptr(), Cc, B, H, W, Cc, K, step, _stream()), "antialias_nhwc")

    def bwd():
        g = tp.take(y)
        if g is None or not tp.needs(x):
            return
        g = g if g.is_contiguous() else _dense(tp.lib, g)
        dx = _empty((B, H, W, Cc), x)
        L.check(tp.lib.smx_antialias_nhwc_bwd_f32(g.data_ptr(), Cc, w.data_ptr(), dx.data_ptr(), Cc, B, H, W, Cc, K, step, _stream()), "antialias_nhwc_bwd")
        tp.acc(x, dx)
    tp.record(bwd)
    return y


def maxpool2(tp, x):
    """2 x 2 / stride-2 max pooling (nn.MaxPool2d(2, 2) of the VGG19 feature stack)."""
    B, H, W, Cc = x.shape
    xd = x if x.is_contiguous() else _dense(tp.lib, x)
    y = _empty((B, H // 2, W // 2, Cc), x)
    L.check(tp.lib.smx_maxpool2_f32(xd.data_ptr(), Cc, y.data_ptr(), Cc, B, H, W, Cc, _stream()), "maxpool2")

    def bwd():
        g = tp.take(y)
        if g is None or not tp.needs(x):
            return
        g = g if g.is_contiguous() else _dense(tp.lib, g)
        dx = _empty((B, H, W, Cc), x)
        L.check(tp.lib.smx_maxpool2_bwd_f32(xd.data_ptr(), Cc, g.data_ptr(), Cc, dx.data_ptr(), Cc, B, H, W, Cc, _stream()), "maxpool2_bwd")
        tp.acc(x, dx)
    tp.record(bwd)
    return y


def chan_affine(tp, x, scale, shift):
    """y = x * scale[c] + shift[c] (fixed per-channel constants: the VGG input normalisation)."""
    B, H, W, Cc = x.shape
    y = _empty((B, H, W, Cc), x)
    xp, ldx, P, _ = _pix(x)
    L.check(tp.lib.smx_chan_affine_f32(xp, ldx, scale.data_ptr(), shift.data_ptr(), y.data_ptr(), Cc, P, Cc, _stream()), "chan_affine")

    def bwd():
        g = tp.take(y)
        if g is None or not tp.needs(x):
            return
        g = g if g.is_contiguous() else _dense(tp.lib, g)
        dx = _empty((B, H, W, Cc), x)
        L.check(tp.lib.smx_chan_affine_f32(g.data_ptr(), Cc, scale.data_ptr(), None, dx.data_ptr(), Cc, P, Cc, _stream()), "chan_affine_bwd")
        tp.acc(x, dx)
    tp.record(bwd)
    return y


def torch_scalar(tp, x, fn):
    """a scalar loss of a SMALL tensor written with torch ops (the hinge terms over a B x 30 x 30 discriminator map): value [1] on the tape,
    gradient by torch.autograd over that one expression (a few elementwise kernels on the launch stream; capturable in the step's hipGraph)."""
    leaf = x.detach().clone().requires_grad_()
    with torch.enable_grad():
        val = fn(leaf).reshape(1)
    out = val.detach()

    def bwd():
        g = tp.take(out)
        if g is None or not tp.needs(x):
            return
        (gx,) = torch.autograd.grad(val, leaf, g.reshape(1))
        tp.acc(x, gx.contiguous())
    tp.record(bwd)
    return out
