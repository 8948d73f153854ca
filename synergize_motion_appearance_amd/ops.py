"""Host-side launchers: torch device tensors -> libsmx C-ABI calls (include/smx.h).

PyTorch is used for device memory and the current HIP stream only.  Activations are NHWC
fp32 tensors `[B,H,W,C]`; a channel slice `buf[..., a:b]` is a legal operand (its pixel
stride is the channel count of `buf`), which is how concatenations are built in place.
Every launcher raises on CPU tensors: there is no fallback path."""
import ctypes as C

import torch

from . import lib as L
from .lib import ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_SWISH, ACT_GELU, ACT_SIGMOID  # noqa: F401


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- optional per-launch timing with HIP events on the launch stream (bench.py roofline leg) ----
_PROFILE = None


class profile:
    """with ops.profile() as rec: ...  -> rec.rows = [(kernel, meta, ms)] after the block."""

    def __enter__(self):
        global _PROFILE
        self._ev = []
        self.rows = []
        _PROFILE = self._ev
        return self

    def __exit__(self, *exc):
        global _PROFILE
        _PROFILE = None
        torch.cuda.synchronize()
        self.rows = [(n, m, e0.elapsed_time(e1)) for n, m, e0, e1 in self._ev]
        return False


def _timed(name, meta, fn, *args):
    if _PROFILE is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    _PROFILE.append((name, meta, e0, e1))
    return rc


BF16 = torch.bfloat16


def _dev(t, name="tensor", dtypes=(torch.float32,)):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise L.SmxError(f"{name}: the HIP path needs a device tensor (got {type(t).__name__} on "
                         f"{getattr(t, 'device', '?')}); there is no CPU fallback")
    if t.dtype not in dtypes:
        raise L.SmxError(f"{name}: {' / '.join(str(d) for d in dtypes)} expected, got {t.dtype}")
    return t


_ANY = (torch.float32, torch.bfloat16)


def _sfx(t):
    """entry-point suffix for the storage type of an activation (configs[1] fp32 / configs[2] bf16)."""
    return "_bf16" if t.dtype == BF16 else "_f32"


def _fn(base, t):
    return getattr(L.load(), base + _sfx(t))


def _same(name, *ts):
    d = ts[0].dtype
    for t in ts[1:]:
        if t is not None and t.dtype != d:
            raise L.SmxError(f"{name}: operands must share one storage type ({d} vs {t.dtype})")


def _live(*ts):
    """dense versions of the operands, to be kept in a local until the launch that reads them has been queued.  `t.contiguous().data_ptr()`
    inside a call expression frees a dense COPY the moment its address has been taken; with two such operands in one call the second
    copy can be handed the block the first one just gave back -- and the kernel reads the second operand through the first pointer."""
    return [t if t.is_contiguous() else t.contiguous() for t in ts]


def _pix(t, name="tensor"):
    """(ptr, ld) of an NHWC tensor or channel-slice view (fp32 or bf16): last stride 1, dense pixels."""
    _dev(t, name, _ANY)
    if t.dim() < 2 or t.stride(-1) != 1:
        raise L.SmxError(f"{name}: innermost (channel) stride must be 1")
    ld = t.stride(-2) if t.shape[-2] > 1 else max(t.shape[-1], t.stride(-2))
    # pixels must be dense at stride ld
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] > 1 and t.stride(d) != exp:
            raise L.SmxError(f"{name}: non-dense pixel layout {tuple(t.shape)} / {t.stride()}")
        exp *= t.shape[d]
    return t.data_ptr(), int(ld)


import os as _os


def set_tuning(name, value):
    """launch-selection knob of libsmx (include/smx.h: smx_set_tuning); returns the previous value."""
    old = C.c_int(0)
    L.check(L.load().smx_get_tuning(name.encode(), C.byref(old)), f"smx_get_tuning({name})")
    L.check(L.load().smx_set_tuning(name.encode(), int(value)), f"smx_set_tuning({name})")
    return old.value


def _knob(name, default):
    """kernel-selection switches are a TOOLS facility (bisection, A/B timing): they are read from the environment only when SMX_TOOLS is set
    (tools/*.sh export it).  A product run never consults them -- which kernel renders a frame does not depend on a stray variable;
    tests change the module attributes below (or `set_tuning`) explicitly."""
    return int(_os.environ.get(name, default)) if _os.environ.get("SMX_TOOLS") else int(default)


SMALLN = bool(_knob("SMX_SMALLN", 1))
CONV7_F16 = _knob("SMX_CONV7_F16", 1)               # fp32 configuration: the big launches of the 7x7 heads in the f16x3 arithmetic (conv7_bf16x3_kernel<NT, true>); 0 = conv7_f32 / implicit GEMM
CONV7_F32 = _knob("SMX_CONV7_F32", 1)               # fp32 configuration: the 7x7 heads on conv7_f32_kernel (0 = implicit GEMM)
CONV7_C2 = _knob("SMX_CONV7_C2", 1)                 # bf16 configuration: BasicMotionEncoder.convf1 on csrc/conv7_c2_bf16.hip (0 = implicit GEMM)
SMALLN_MFMA_MIN_BLOCKS = 512                                          # 8 x 32-pixel tiles; below: the VALU kernel (tests lower it)
SMALLN_MFMA = _knob("SMX_SMALLN_MFMA", 1)           # bf16 storage: C_out <= 4 3x3 layers on the bf16 MFMA (csrc/conv3x3_smalln_mfma16.hip); 0 = the VALU kernel

WINOGRAD = bool(_knob("SMX_WINOGRAD", 1))


class Conv:
    """A packed convolution / linear layer: weights [Cout][kh][kw][Cin] (k contiguous), bias."""
    __slots__ = ("w", "b", "kh", "kw", "cin", "cout", "_u", "_w16", "_u43", "_w16t", "_w16rp", "_wrp", "_w7x3", "_wsn16", "_w7c2", "_w7c2f", "_w7f", "_u3", "_wrp3", "_u3h", "_wrp3h", "_w7h")

    def __init__(self, w, b, kh, kw, cin, cout):
        self.w, self.b, self.kh, self.kw, self.cin, self.cout = w, b, kh, kw, cin, cout
        self._u = None
        self._w16 = None
        self._u43 = None
        self._u3 = None
        self._u3h = None
        self._wrp3h = None
        self._w7h = None
        self._wrp3 = None
        self._w16t = None
        self._w16rp = None
        self._wrp = None
        self._w7x3 = None
        self._wsn16 = None
        self._w7c2 = None
        self._w7c2f = None
        self._w7f = None

    @property
    def w7_f32(self):
        """fp32 fragment-ordered pack of a 7x7 head for conv7_f32_kernel (csrc/conv7_bf16x3.hip), built once per layer."""
        if self._w7f is None:
            n = int(L.load().smx_conv7_bf16x3_pack_elems(self.cin, self.cout))
            if n <= 0 or self.kh != 7 or self.kw != 7:
                raise L.SmxError(f"w7_f32: not a 7x7 layer with N <= 96 ({self.kh}x{self.kw}, N {self.cout})")
            wp = torch.empty(n // 2, device=self.w.device, dtype=torch.float32)
            L.check(L.load().smx_conv7_f32_pack(_dev(self.w).data_ptr(), wp.data_ptr(), self.cin, self.cout, _stream()), "smx_conv7_f32_pack")
            self._w7f = wp
        return self._w7f

    @property
    def w7_c2f(self):
        """fp32 fragment-ordered pack of a 7x7, C_in = 2 layer for conv7_c2_f32_kernel, built once per layer."""
        if self._w7c2f is None:
            wp = torch.empty((self.cout // 32) * 49 * 64, device=self.w.device, dtype=torch.float32)
            L.check(L.load().smx_conv7_c2_f32_pack(_dev(self.w).data_ptr(), wp.data_ptr(), self.cout, _stream()), "smx_conv7_c2_f32_pack")
            self._w7c2f = wp
        return self._w7c2f

    @property
    def w7_c2(self):
        """bf16 fragment-ordered pack of a 7x7, C_in = 2 layer for csrc/conv7_c2_bf16.hip, built once per layer."""
        if self._w7c2 is None:
            wp = torch.empty((self.cout // 32) * 7 * 512, device=self.w.device, dtype=BF16)
            L.check(L.load().smx_conv7_c2_bf16_pack(_dev(self.w).data_ptr(), wp.data_ptr(), self.cout, _stream()), "smx_conv7_c2_bf16_pack")
            self._w7c2 = wp
        return self._w7c2

    @property
    def w_sn16(self):
        """bf16 fragment-ordered pack of a C_out <= 4 3x3 layer for csrc/conv3x3_smalln_mfma16.hip, built once per layer."""
        if self._wsn16 is None:
            n = int(L.load().smx_conv3x3_smalln_mfma_pack_elems(self.cin, self.cout))
            if n <= 0 or self.kh != 3 or self.kw != 3:
                raise L.SmxError(f"w_sn16: not a 3x3 layer with Cin % 64 == 0 and C_out <= 4 (Cin {self.cin}, C_out {self.cout})")
            wp = torch.empty(n, device=self.w.device, dtype=BF16)
            L.check(L.load().smx_conv3x3_smalln_mfma_pack(_dev(self.w).data_ptr(), wp.data_ptr(), self.cin, self.cout, _stream()), "smx_conv3x3_smalln_mfma_pack")
            self._wsn16 = wp
        return self._wsn16

    @property
    def w7_x3(self):
        """hi / lo bf16 split of a 7x7 layer in the fragment order of csrc/conv7_bf16x3.hip, built once per layer."""
        if self._w7x3 is None:
            n = int(L.load().smx_conv7_bf16x3_pack_elems(self.cin, self.cout))
            if n <= 0 or self.kh != 7 or self.kw != 7:
                raise L.SmxError(f"w7_x3: not a 7x7 layer with N <= 96 ({self.kh}x{self.kw}, N {self.cout})")
            wp = torch.empty(n, device=self.w.device, dtype=BF16)
            L.check(L.load().smx_conv7_bf16x3_pack(self.w.data_ptr(), wp.data_ptr(), self.cin, self.cout, _stream()), "smx_conv7_bf16x3_pack")
            self._w7x3 = wp
        return self._w7x3

    @property
    def w7_f16(self):
        """a 7x7 layer's weights scaled by a power of two and split into two IEEE-half levels in the fragment order of csrc/conv7_bf16x3.hip (smx_conv7_f16_pack)."""
        if self._w7h is None:
            n = int(L.load().smx_conv7_bf16x3_pack_elems(self.cin, self.cout))
            if n <= 0 or self.kh != 7 or self.kw != 7:
                raise L.SmxError(f"w7_f16: not a 7x7 layer with N <= 96 ({self.kh}x{self.kw}, N {self.cout})")
            wp = torch.empty(2 * n + 16, device=self.w.device, dtype=torch.uint8)
            L.check(L.load().smx_conv7_f16_pack(self.w.data_ptr(), wp.data_ptr(), self.cin, self.cout, _stream()), "smx_conv7_f16_pack")
            self._w7h = wp
        return self._w7h

    @property
    def w_rp(self):
        """fragment-ordered fp32 pack of a 1x1 layer for the fp32 row-panel kernel (smx_gemm_rp_f32_pack), built once per layer."""
        if self._wrp is None:
            wp = torch.empty((self.cout // 32) * (self.cin // 8) * 256, device=self.w.device, dtype=torch.float32)
            L.check(L.load().smx_gemm_rp_f32_pack(self.w.data_ptr(), self.w.shape[1], wp.data_ptr(), self.cout, self.cin, _stream()), "smx_gemm_rp_f32_pack")
            self._wrp = wp
        return self._wrp

    @property
    def w_rp3(self):
        """the three bf16 levels of a 1x1 layer's fp32 weights in MFMA fragment order for the split-bf16 row-panel kernel (smx_gemm_rp_bf3_pack), built once per layer."""
        if self._wrp3 is None:
            lib = L.load()
            wp = torch.empty(int(lib.smx_gemm_rp_bf3_pack_bytes(self.cout, self.cin)), device=self.w.device, dtype=torch.uint8)
            L.check(lib.smx_gemm_rp_bf3_pack(_dev(self.w).data_ptr(), self.w.shape[1], wp.data_ptr(), self.cout, self.cin, _stream()), "smx_gemm_rp_bf3_pack")
            self._wrp3 = wp
        return self._wrp3

    @property
    def w_rp3h(self):
        """a 1x1 layer's weights scaled by a power of two and split into two IEEE-half levels for the f16x3 row-panel kernel (smx_gemm_rp_f16_pack), built once."""
        if self._wrp3h is None:
            lib = L.load()
            wp = torch.empty(int(lib.smx_gemm_rp_f16_pack_bytes(self.cout, self.cin)), device=self.w.device, dtype=torch.uint8)
            L.check(lib.smx_gemm_rp_f16_pack(_dev(self.w).data_ptr(), self.w.shape[1], wp.data_ptr(), self.cout, self.cin, _stream()), "smx_gemm_rp_f16_pack")
            self._wrp3h = wp
        return self._wrp3h

    @property
    def w16_rp(self):
        """fragment-ordered bf16 pack of a 1x1 layer for the row-panel kernel (smx_gemm_rp_bf16_pack), built once per layer."""
        if self._w16rp is None:
            w16 = self.w16
            wp = torch.empty((self.cout // 32) * (self.cin // 16) * 512, device=w16.device, dtype=BF16)
            L.check(L.load().smx_gemm_rp_bf16_pack(w16.data_ptr(), w16.shape[1], wp.data_ptr(), self.cout, self.cin, _stream()), "smx_gemm_rp_bf16_pack")
            self._w16rp = wp
        return self._w16rp

    @property
    def w16_t32(self):
        """fragment-ordered bf16 pack of a 3x3 layer for the 16x32-tile region kernel (smx_conv3x3_bf16_t32_pack), built once per layer."""
        if self._w16t is None:
            lib = L.load()
            n = int(lib.smx_conv3x3_bf16_t32_pack_elems(self.cin, self.cout))
            if n <= 0 or self.kh != 3 or self.kw != 3:
                raise L.SmxError(f"w16_t32: not a 3x3 layer with Cin % 16 == 0 ({self.kh}x{self.kw}, Cin {self.cin})")
            wp = torch.empty(n, device=self.w.device, dtype=BF16)
            w16 = self.w16
            L.check(lib.smx_conv3x3_bf16_t32_pack(w16.data_ptr(), w16.shape[1], wp.data_ptr(), self.cin, self.cout, _stream()), "smx_conv3x3_bf16_t32_pack")
            self._w16t = wp
        return self._w16t

    @property
    def w16(self):
        """bf16 copy of the packed weights (round-to-nearest-even), built once per layer for the bf16 MFMA path."""
        if self._w16 is None:
            self._w16 = self.w.to(BF16).contiguous()
        return self._w16

    def winograd_u(self):
        """U = G g G^T for F(2x2,3x3), fragment-ordered [16][ceil(Cout/32)][Cin/8][64 lanes][4]
        (lane l <-> row n = 32*nt + (l&31), channels 8s + 4*(l>>5) + 0..3); built once per layer."""
        if self._u is None:
            lib = L.load()
            if self.kh != 3 or self.kw != 3 or self.cin % 8:
                raise L.SmxError(f"winograd_u: not a 3x3 layer with Cin % 8 == 0 ({self.kh}x{self.kw}, Cin {self.cin})")
            # the library's own pack kernel (fp64 products, one rounding: smx_pack_winograd_u_f32, the training step's) on the OIHW view of the weights
            w_oihw = _dev(self.w).view(self.cout, 3, 3, self.cin).permute(0, 3, 1, 2).contiguous()
            u = torch.empty(int(lib.smx_winograd_u_floats(self.cout, self.cin)), device=self.w.device, dtype=torch.float32)
            L.check(lib.smx_pack_winograd_u_f32(w_oihw.data_ptr(), u.data_ptr(), self.cout, self.cin, 0, _stream()), "smx_pack_winograd_u_f32")
            self._u = u
        return self._u

    def winograd_bf3_u(self):
        """the three bf16 planes of `winograd_u` in bf16-MFMA fragment order (csrc/winograd_bf3.hip: smx_winograd_bf3_pack), built once per layer."""
        if self._u3 is None:
            lib = L.load()
            n = int(lib.smx_winograd_bf3_u_bytes(self.cout, self.cin))
            if n <= 0 or self.kh != 3 or self.kw != 3:
                raise L.SmxError(f"winograd_bf3_u: not a 3x3 layer with Cout % 32 == 0 and Cin % 16 == 0 ({self.kh}x{self.kw}, {self.cin} -> {self.cout})")
            u3 = torch.empty(n, device=self.w.device, dtype=torch.uint8)
            L.check(lib.smx_winograd_bf3_pack(self.winograd_u().data_ptr(), u3.data_ptr(), self.cout, self.cin, _stream()), "smx_winograd_bf3_pack")
            self._u3 = u3
        return self._u3

    def winograd_f16_u(self):
        """U scaled by a power of two and split into two IEEE-half levels for the f16x3 form of the split Winograd kernel (smx_winograd_f16_pack), built once."""
        if self._u3h is None:
            lib = L.load()
            up = torch.empty(int(lib.smx_winograd_f16_u_bytes(self.cout, self.cin)), device=self.w.device, dtype=torch.uint8)
            L.check(lib.smx_winograd_f16_pack(self.winograd_u().data_ptr(), up.data_ptr(), self.cout, self.cin, _stream()), "smx_winograd_f16_pack")
            self._u3h = up
        return self._u3h

    def winograd43_u(self):
        """U = G g G^T for F(4x4,3x3) (G 6x3: Lavin & Gray), fragment-ordered [36][Cout/32][Cin/8][64 lanes][4] like `winograd_u`;
        built once per layer (fp64 products, rounded once)."""
        if self._u43 is None:
            g = self.w.view(self.cout, 3, 3, self.cin).double()
            G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                              [0, 0, 1]], dtype=torch.float64, device=g.device)
            U = torch.einsum("ia,nabc,jb->ijnc", G, g, G).float()                     # [6,6,Cout,Cin]
            n32 = (self.cout + 31) // 32
            Up = torch.zeros((36, n32 * 32, self.cin), device=g.device, dtype=torch.float32)
            Up[:, :self.cout] = U.reshape(36, self.cout, self.cin)
            Up = Up.view(36, n32, 32, self.cin // 8, 2, 4).permute(0, 1, 3, 4, 2, 5).contiguous()
            self._u43 = torch.cat([Up.view(-1), torch.zeros(1024, device=g.device, dtype=torch.float32)])
        return self._u43

    @staticmethod
    def from_torch(weight, bias):
        """weight: conv [Cout,Cin,kh,kw] or linear [out,in] (device tensors)."""
        if weight.dim() == 4:
            co, ci, kh, kw = weight.shape
            w = weight.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).contiguous()
        else:
            co, ci = weight.shape
            kh = kw = 1
            w = weight.contiguous()
        return Conv(w.float(), None if bias is None else bias.contiguous().float(), kh, kw, ci, co)

    @staticmethod
    def cat(convs):
        """stack output channels of convs that share their input."""
        c0 = convs[0]
        return Conv(torch.cat([c.w for c in convs], 0).contiguous(), torch.cat([c.b for c in convs], 0).contiguous(),
                    c0.kh, c0.kw, c0.cin, sum(c.cout for c in convs))


def gemm_raw(**kw):
    d = L.GemmDesc()
    for k, v in kw.items():
        setattr(d, k, v)
    meta = {"flops": 2.0 * d.M * d.N * d.K * d.nb0 * d.nb1, "M": d.M, "N": d.N, "K": d.K, "nb": d.nb0 * d.nb1,
            "k": d.kh} if _PROFILE is not None else None
    L.check(_timed("gemm_conv", meta, L.load().smx_gemm_conv_f32, C.byref(d), _stream()), "smx_gemm_conv_f32")


def gemm16_raw(**kw):
    d = L.Gemm16Desc()
    for k, v in kw.items():
        setattr(d, k, v)
    meta = {"flops": 2.0 * d.M * d.N * d.K * d.nb0 * d.nb1, "M": d.M, "N": d.N, "K": d.K, "nb": d.nb0 * d.nb1,
            "k": d.kh, "bf16": 1} if _PROFILE is not None else None
    L.check(_timed("gemm_bf16", meta, L.load().smx_gemm_conv_bf16, C.byref(d), _stream()), "smx_gemm_conv_bf16")


REGION3X3 = bool(_knob("SMX_REGION3X3", 1))
CONV16_TILE_H = _knob("SMX_CONV16_TILE_H", 0)      # 0 = auto, 8 | 16 = forced (tools / tests)


CONV16_F32_REGION = _knob("SMX_CONV16_F32_REGION", 1)   # fp32-storage form of the region kernel (bf16-compute training); 0 = implicit GEMM


GEMM_RP = _knob("SMX_GEMM_RP", 1)          # fp32 row-panel kernel (csrc/gemm_rp_f32.hip); 0 = implicit GEMM
GEMM_RP_F16 = _knob("SMX_GEMM_RP_F16", 1)  # the split row-panel launches in the f16x3 form (two half levels, three products; per-row input scale); 0 = the six-product bf16 form
GEMM_RP_BF3 = _knob("SMX_GEMM_RP_BF3", 1)  # the same launches on the bf16 MFMA with three-way split fp32 operands, six products (csrc/gemm_rp_bf3.hip); 0 = the fp32-MFMA kernel
GEMM16_RP = _knob("SMX_GEMM16_RP", 1)      # row-panel kernel (csrc/gemm_rp_bf16.hip) for the K = 128 / 256 1x1 layers; 0 = implicit GEMM
GEMM16_RP_MIN_ROWS = 16384                                           # below: too few 32-row tiles to fill the persistent blocks (tests lower it)


def _rows_dense(t):
    """all leading dims collapse to ONE row index at a common stride (what a [M][ld] view needs)."""
    ld = t.stride(-2) if t.dim() > 1 else t.shape[-1]
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] > 1 and t.stride(d) != exp:
            return False
        exp *= t.shape[d]
    return True


CONV16_T32 = _knob("SMX_CONV16_T32", 1)           # 16x32-tile kernel (csrc/conv3x3_bf16_t32.hip) for the big launches; 0 = off
CONV16_T32_MIN_BLOCKS = 1024                                         # two resident rounds of the chip's 512 block slots; tests lower it


def _conv16_t32_ok(B, Ho, Wo, Cin, cout, in_ss=None):
    """the 16x32-tile kernel's own limits: its fused GroupNorm loader keeps the per-channel (scale, shift) table in 4 KB of LDS (Cin <= 512);
    a GroupNorm-fed layer above that falls through to the 16x16 region kernel / the implicit GEMM."""
    return (CONV16_T32 and Ho % 16 == 0 and Wo % 32 == 0 and Cin % 16 == 0 and (in_ss is None or Cin <= 512) and
            B * (Ho // 16) * (Wo // 32) * ((cout + 63) // 64) >= CONV16_T32_MIN_BLOCKS)


def _conv16_tile_h(Cin, Ho, nblk16=0):
    """output tile height of the region-direct bf16 3x3 kernel (measured per shape, tools/conv16_bench.py): 16x16 tiles (2 workgroups
    per CU) once the launch has >= 1024 of them -- 5-15 % over 8x16 tiles at every B=60 shape since the loader normalises at store
    time -- and 8x16 tiles (3 per CU, twice the workgroups) for the small launches."""
    if CONV16_TILE_H in (8, 16) and Ho % CONV16_TILE_H == 0:
        return CONV16_TILE_H
    return 16 if (Ho % 16 == 0 and (nblk16 >= 1024 or Cin >= 512)) else 8


def _conv16(x, cv, out, stride, pt, pl, up2, act, res, Ho, Wo, d2s, tile, in_ss, in_swish, out_dtype, want_stats=False):
    """bf16-MFMA form of conv(): x bf16 (or fp32, converted while staging), weights bf16, fp32 accumulate."""
    B, H, W, Cin = x.shape
    if out is None:
        od = out_dtype or BF16
        shape = (B, Ho * d2s[0], Wo * d2s[0], d2s[1]) if d2s else (B, Ho, Wo, cv.cout)
        out = torch.empty(shape, device=x.device, dtype=od)
    a_ptr, lda = _pix(x, "conv input")
    c_ptr, ldc = _pix(out, "conv output")
    r_ptr, ldr = (None, 0) if res is None else _pix(res, "conv residual")
    if (CONV7_C2 and tile == 0 and x.dtype == torch.float32 and out.dtype == BF16 and cv.kh == 7 and cv.kw == 7 and Cin == 2 and stride == 1
            and (pt, pl) == (3, 3) and not d2s and not up2 and res is None and in_ss is None and cv.w is not None and cv.cout % 128 == 0
            and (Ho, Wo) == (H, W) and H % 8 == 0 and W % 32 == 0 and lda == 2 and ldc % 8 == 0 and a_ptr % 8 == 0 and c_ptr % 16 == 0
            and act in (ACT_NONE, ACT_RELU, ACT_LRELU02) and B * (H // 8) * (W // 32) >= 256):
        # BasicMotionEncoder.convf1: the lane builds its MFMA operand from the 2-channel region in LDS instead of a scalar implicit-GEMM gather
        meta = {"flops": 2.0 * B * Ho * Wo * cv.cout * 98, "M": B * Ho * Wo, "N": cv.cout, "K": 98, "nb": 1, "k": 7, "bf16": 1} if _PROFILE is not None else None
        L.check(_timed("gemm_bf16", meta, L.load().smx_conv7_c2_bf16, a_ptr, cv.w7_c2.data_ptr(), None if cv.b is None else cv.b.data_ptr(),
                       c_ptr, ldc, B, H, W, cv.cout, act, _stream()), "smx_conv7_c2_bf16")
        return out
    if (SMALLN and SMALLN_MFMA and tile == 0 and x.dtype == BF16 and out.dtype == torch.float32 and cv.kh == 3 and cv.kw == 3 and stride == 1
            and (pt, pl) == (1, 1) and not d2s and not up2 and res is None and cv.cout <= 4 and Cin % 64 == 0 and (Ho, Wo) == (H, W)
            and H % 8 == 0 and W % 32 == 0 and lda % 8 == 0 and a_ptr % 16 == 0 and B * (H // 8) * (W // 32) >= SMALLN_MFMA_MIN_BLOCKS
            and act in (ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_SIGMOID)):
        # the same layers on the bf16 MFMA (one 32-wide N tile, region-direct): the image head at 256^2 was 2.85 ms on the vector ALUs
        meta = {"flops": 2.0 * B * Ho * Wo * cv.cout * 9 * Cin, "mfma_flops": 2.0 * B * Ho * Wo * 32 * 9 * Cin, "M": B * Ho * Wo, "N": cv.cout, "K": 9 * Cin,
                "nb": 1, "k": 3, "bf16": 1, "bytes": B * Ho * Wo * (2.0 * Cin + 4.0 * cv.cout)} if _PROFILE is not None else None
        L.check(_timed("conv_small_n", meta, L.load().smx_conv3x3_smalln_mfma_bf16, a_ptr, lda, cv.w_sn16.data_ptr(),
                       None if cv.b is None else cv.b.data_ptr(), c_ptr, ldc, B, H, W, Cin, cv.cout, act,
                       None if in_ss is None else in_ss.data_ptr(), int(in_swish), _stream()), "smx_conv3x3_smalln_mfma_bf16")
        return out
    if (SMALLN and tile == 0 and x.dtype == BF16 and out.dtype == torch.float32 and cv.kh == 3 and cv.kw == 3 and stride == 1
            and (pt, pl) == (1, 1) and not d2s and not up2 and res is None and cv.cout <= 4 and Cin in (64, 128, 256)
            and (Ho, Wo) == (H, W) and W % 4 == 0 and lda % 4 == 0 and a_ptr % 8 == 0):
        meta = {"flops": 2.0 * B * Ho * Wo * cv.cout * 9 * Cin, "mfma_flops": 0.0, "M": B * Ho * Wo, "N": cv.cout, "K": 9 * Cin,
                "nb": 1, "k": 3, "bytes": B * Ho * Wo * (2.0 * Cin + 4.0 * cv.cout)} if _PROFILE is not None else None
        L.check(_timed("conv_small_n", meta, L.load().smx_conv3x3_smalln_bf16, a_ptr, lda, _dev(cv.w).data_ptr(),
                       None if cv.b is None else cv.b.data_ptr(), c_ptr, ldc, B, H, W, Cin, cv.cout, act,
                       None if in_ss is None else in_ss.data_ptr(), int(in_swish), _stream()), "smx_conv3x3_smalln_bf16")
        return out
    if getattr(out, "_gn_part", None) is not None:
        out._gn_part = None
    if (REGION3X3 and tile == 0 and x.dtype == BF16 and out.dtype == BF16 and cv.kh == 3 and cv.kw == 3 and stride == 1 and (pt, pl) == (1, 1)
            and not d2s and (Ho, Wo) == ((2 * H, 2 * W) if up2 else (H, W)) and lda % 8 == 0 and a_ptr % 16 == 0
            and _conv16_t32_ok(B, Ho, Wo, Cin, cv.cout, in_ss)):
        # the big launches: 16x32-pixel tiles, 16-channel slices through a double-buffered stage, weights by LDS-DMA from the fragment-ordered pack
        meta = {"flops": 2.0 * B * Ho * Wo * cv.cout * 9 * Cin, "M": B * Ho * Wo, "N": cv.cout, "K": 9 * Cin, "nb": 1, "k": 3, "bf16": 1, "t32": 1,
                "bytes": 2.0 * B * Ho * Wo * (Cin / (4.0 if up2 else 1.0) + cv.cout * (2 if res is not None else 1))} if _PROFILE is not None else None
        part = torch.empty((B, (Ho // 16) * (Wo // 32), cv.cout, 2), device=x.device, dtype=torch.float32) if want_stats else None
        L.check(_timed("conv3x3_bf16", meta, L.load().smx_conv3x3_bf16_t32, a_ptr, lda, cv.w16_t32.data_ptr(),
                       None if cv.b is None else cv.b.data_ptr(), r_ptr, int(res is not None and res.dtype == torch.float32), ldr,
                       c_ptr, ldc, B, Ho, Wo, Cin, cv.cout, int(up2), act, None if in_ss is None else in_ss.data_ptr(), int(in_swish),
                       None if part is None else part.data_ptr(), _stream()), "smx_conv3x3_bf16_t32")
        if part is not None:
            out._gn_part = part
        return out
    if (REGION3X3 and tile == 0 and x.dtype == BF16 and out.dtype == BF16 and cv.kh == 3 and cv.kw == 3 and stride == 1 and (pt, pl) == (1, 1)
            and not d2s and Cin % 64 == 0 and Ho % 16 == 0 and Wo % 16 == 0 and (Ho, Wo) == ((2 * H, 2 * W) if up2 else (H, W))
            and lda % 8 == 0 and a_ptr % 16 == 0):
        # region-direct 3x3 kernel: the input region is staged once per 64-channel slice and all nine taps read it from LDS
        meta = {"flops": 2.0 * B * Ho * Wo * cv.cout * 9 * Cin, "M": B * Ho * Wo, "N": cv.cout, "K": 9 * Cin, "nb": 1, "k": 3, "bf16": 1,
                "bytes": 2.0 * B * Ho * Wo * (Cin / (4.0 if up2 else 1.0) + cv.cout * (2 if res is not None else 1))} if _PROFILE is not None else None
        th = _conv16_tile_h(Cin, Ho, B * (Ho // 16) * (Wo // 16) * ((cv.cout + 63) // 64))
        part = torch.empty((B, (Ho // th) * (Wo // 16), cv.cout, 2), device=x.device, dtype=torch.float32) if want_stats else None
        L.check(_timed("conv3x3_bf16", meta, L.load().smx_conv3x3_bf16, a_ptr, lda, cv.w16.data_ptr(), cv.w16.shape[1],
                       None if cv.b is None else cv.b.data_ptr(), r_ptr, int(res is not None and res.dtype == torch.float32), ldr,
                       c_ptr, ldc, B, Ho, Wo, Cin, cv.cout, int(up2), act, None if in_ss is None else in_ss.data_ptr(), int(in_swish),
                       None if part is None else part.data_ptr(), th, _stream()), "smx_conv3x3_bf16")
        if part is not None:
            out._gn_part = part
        return out
    if (REGION3X3 and CONV16_F32_REGION and tile == 0 and x.dtype == torch.float32 and out.dtype == torch.float32 and cv.kh == 3 and cv.kw == 3
            and stride == 1 and (pt, pl) == (1, 1) and not d2s and in_ss is None and not want_stats and (res is None or res.dtype == torch.float32)
            and Cin % 64 == 0 and Ho % 8 == 0 and Wo % 16 == 0 and (Ho, Wo) == ((2 * H, 2 * W) if up2 else (H, W)) and lda % 4 == 0
            and a_ptr % 16 == 0 and cv.w16.data_ptr() % 16 == 0):
        # bf16-compute training mode (fp32 storage): the region-direct kernel instead of the implicit GEMM (9 shifted copies through LDS)
        meta = {"flops": 2.0 * B * Ho * Wo * cv.cout * 9 * Cin, "M": B * Ho * Wo, "N": cv.cout, "K": 9 * Cin, "nb": 1, "k": 3, "bf16": 1,
                "bytes": 4.0 * B * Ho * Wo * (Cin / (4.0 if up2 else 1.0) + cv.cout * (2 if res is not None else 1))} if _PROFILE is not None else None
        th = _conv16_tile_h(Cin, Ho, B * (Ho // 16) * (Wo // 16) * ((cv.cout + 63) // 64))
        L.check(_timed("conv3x3_mfma16", meta, L.load().smx_conv3x3_mfma16_f32, a_ptr, lda, cv.w16.data_ptr(), cv.w16.shape[1],
                       None if cv.b is None else cv.b.data_ptr(), r_ptr, ldr, c_ptr, ldc, B, Ho, Wo, Cin, cv.cout, int(up2), act, th, _stream()),
                "smx_conv3x3_mfma16_f32")
        return out
    M, K = B * Ho * Wo, cv.kh * cv.kw * Cin
    if (GEMM16_RP and tile == 0 and x.dtype == BF16 and out.dtype == BF16 and cv.kh == 1 and cv.kw == 1 and stride == 1 and (pt, pl) == (0, 0)
            and not up2 and not d2s and in_ss is None and (Ho, Wo) == (H, W) and (res is None or res.dtype == BF16) and M >= GEMM16_RP_MIN_ROWS
            and cv.w is not None and L.load().smx_gemm_rp_bf16_ok(M, cv.cout, K) and lda % 8 == 0 and ldc % 8 == 0 and a_ptr % 16 == 0
            and c_ptr % 16 == 0 and (res is None or (ldr % 8 == 0 and r_ptr % 16 == 0)) and _rows_dense(x) and _rows_dense(out)
            and (res is None or _rows_dense(res))):
        # short-K token Linears: persistent row-panel kernel (weights in registers, rows by LDS-DMA)
        meta = {"flops": 2.0 * M * cv.cout * K, "M": M, "N": cv.cout, "K": K, "nb": 1, "k": 1, "bf16": 1, "rp": 1} if _PROFILE is not None else None
        L.check(_timed("gemm_bf16", meta, L.load().smx_gemm_rp_bf16, a_ptr, lda, cv.w16_rp.data_ptr(), None if cv.b is None else cv.b.data_ptr(),
                       r_ptr, ldr, c_ptr, ldc, M, cv.cout, K, act, _stream()), "smx_gemm_rp_bf16")
        return out
    if (GEMM16_RP and d2s and tile == 0 and x.dtype == BF16 and out.dtype == BF16 and cv.kh == 1 and cv.kw == 1 and stride == 1 and (pt, pl) == (0, 0)
            and not up2 and in_ss is None and res is None and (Ho, Wo) == (H, W) and M >= GEMM16_RP_MIN_ROWS and d2s[1] % 16 == 0
            and L.load().smx_gemm_rp_bf16_ok(M, cv.cout, K) and lda % 8 == 0 and ldc % 8 == 0 and a_ptr % 16 == 0 and c_ptr % 16 == 0):
        meta = {"flops": 2.0 * M * cv.cout * K, "M": M, "N": cv.cout, "K": K, "nb": 1, "k": 1, "bf16": 1, "rp": 1} if _PROFILE is not None else None
        L.check(_timed("gemm_bf16", meta, L.load().smx_gemm_rp_d2s_bf16, a_ptr, lda, cv.w16_rp.data_ptr(), None if cv.b is None else cv.b.data_ptr(),
                       c_ptr, ldc, M, cv.cout, K, act, d2s[0], d2s[1], Ho, Wo, _stream()), "smx_gemm_rp_d2s_bf16")
        return out
    ksplit, ws = 1, None
    if not d2s and K >= 2048:
        blocks = ((M + 63) // 64) * ((cv.cout + 63) // 64)
        if blocks < 256:
            ksplit = max(1, min(K // 64 // 8, (1024 + blocks - 1) // blocks, 32))
            if ksplit > 1:
                ws = torch.empty((ksplit, M, cv.cout), device=x.device, dtype=torch.float32)
    gemm16_raw(a=a_ptr, bt=cv.w16.data_ptr(), c=c_ptr, bias=None if cv.b is None else cv.b.data_ptr(),
               res=r_ptr, in_ss=None if in_ss is None else in_ss.data_ptr(), in_swish=int(in_swish), nb0=1, nb1=1,
               M=M, N=cv.cout, K=K, lda=lda, ldb=cv.w16.shape[1], ldc=ldc, ldres=ldr, Hin=H, Win=W, Cin=Cin, Ho=Ho, Wo=Wo,
               kh=cv.kh, kw=cv.kw, stride=stride, pad_t=pt, pad_l=pl, up2=int(up2), act=act, alpha=1.0,
               d2s_p=d2s[0] if d2s else 0, d2s_c=d2s[1] if d2s else 0, tile=tile, ksplit=ksplit,
               ws=None if ws is None else ws.data_ptr(), a_f32=int(x.dtype == torch.float32), c_f32=int(out.dtype == torch.float32),
               res_f32=int(res is not None and res.dtype == torch.float32))
    return out


# F(4x4,3x3) (csrc/winograd43.hip): parity-tested, selectable, OFF by default -- measured 0.81-0.93x the speed of the F(2x2,3x3) wide
# kernel on the B=60 shapes (executed-MFMA fraction 0.23-0.29 vs 0.45-0.61; profiles/r03_wino43_*.txt, DESIGN section 4 "Round 3")
WINO43 = _knob("SMX_WINO43", 0)
WINO43_MIN_BLOCKS = 512                                 # tests force the kernel on small inputs by lowering this


def _wino43_ok(B, H, W, cin, cout, up2, lda, ldc, ldres):
    """launches that go to the F(4x4,3x3) kernel: whole 16x32-pixel blocks, 32-channel N blocks, and enough of them to fill the chip
    (one 768-thread block per CU): the B=60 launches of the big layers."""
    return (WINO43 and not up2 and H % 16 == 0 and W % 32 == 0 and cin % 16 == 0 and cout % 32 == 0 and lda % 4 == 0 and ldc % 4 == 0
            and ldres % 4 == 0 and B * (H // 16) * (W // 32) * (cout // 32) >= WINO43_MIN_BLOCKS)


# fp32 3x3 convolutions on the BF16 matrix pipe with three-way split operands (csrc/winograd_bf3.hip): 6 = the fp32-grade six-product form (default:
# error against fp64 below the fp32-MFMA kernel's, 1.08-1.40x its speed per layer at B = 300, +11 % on the configs[1] step), 0 = off (the fp32-MFMA
# kernels everywhere), 3 = the two-way split (comparison only, never the fp32 configuration).  WINO_BF3_MIN_BLOCKS: 16x16-pixel x 64-channel blocks a launch must have (one block per CU).
WINO_BF3 = _knob("SMX_WINO_BF3", 6)
WINO_BF3_MIN_BLOCKS = 512
WINO_F16 = _knob("SMX_WINO_F16", 2)          # the split kernel's f16x3 form (two half levels, three products): 1 = launches fed through the fused GroupNorm (+ swish) loader, 2 = every launch (raw inputs get a per-block power-of-two scale)
GEMM_RP_BF3_MIN_ROWS = 32 * 512               # the split row-panel GEMM: persistent blocks, one per CU -- at least two 32-row tiles each


def _wino_bf3_ok(B, H, W, cin, cout, lda, ldc, ldres, ldmul, *ptrs, ragged=False):
    """mirror of smx_winograd_bf3_shape_ok + the size threshold: blocks are 8x16 pixels x 128 channels when C_out % 128 == 0, else 16x16 x 64.
    ragged: the f16x3 form takes any C_out (its pack pads U to the 64-channel block width, the epilogue masks the ragged quad)."""
    if WINO_BF3 not in (3, 6) or (cout % 64 and not (ragged and cout > 64)) or cin % 32 or cin > 512 or W % 16:     # (C_out < 64 padded to 64 loses to the fp32 kernel: measured)
        return False
    cout = 64 * ((cout + 63) // 64)
    mt = 1 if cout % 128 == 0 else 2
    return (H % (8 * mt) == 0 and lda % 4 == 0 and ldc % 4 == 0 and ldres % 4 == 0 and ldmul % 4 == 0 and all((q or 0) % 16 == 0 for q in ptrs)
            and B * (H // (8 * mt)) * (W // 16) * (cout // (128 if mt == 1 else 64)) >= WINO_BF3_MIN_BLOCKS)


def _wino_wide(B, H, W, cout):
    """mirror of winograd_launch's block-shape rule (csrc/winograd.hip): 64-channel 'wide' blocks once >= 1024 of them exist --
    in this pipeline exactly the B=60 launches of the big layers; used to label profile rows only."""
    return int(cout % 64 == 0 and B * (H // 8) * (W // 16) * (cout // 64) >= 1024)


def conv(x, cv, out=None, *, stride=1, pad=None, up2=False, act=ACT_NONE, res=None, out_hw=None,
         d2s=None, tile=0, in_ss=None, in_swish=False, want_stats=False, mfma16=False, out_dtype=None, direct=False, bf3=True):
    """y = act(conv(x) + bias) [+ res].  x [B,H,W,Cin] (slice ok) -> out [B,Ho,Wo,Cout] (slice ok).
    pad: (top, left) (default (kh//2, kw//2)); out_hw for asymmetric pads / strides.
    d2s=(p, C): un-patchify store, out is [B,Ho*p,Wo*p,C].
    up2: True = nearest x2 folded into the gather; 2 = ZERO-INSERT x2 (the data gradient of a stride-2 convolution as a
    stride-1 launch; implicit-GEMM kernel only).
    direct: always the implicit-GEMM kernel (training: weights change every step, so the Winograd-domain / fragment-ordered
    packings of the inference engines are not built).
    bf3=False: never the split-bf16 Winograd kernel (training: its three-plane weight pack is built once per layer, not per step).
    want_stats: the output feeds a GroupNorm -- on the fused Winograd path the epilogue also emits the
    per-block {sum, sum^2} partials and tags the returned tensor with them (`_gn_part`), so that
    `groupnorm_stats(y, ...)` is a finalize over a few KB instead of a read of y."""
    B, H, W, Cin = x.shape
    if Cin != cv.cin:
        raise L.SmxError(f"conv: input has {Cin} channels, layer expects {cv.cin}")
    He, We = (2 * H, 2 * W) if up2 else (H, W)
    pt, pl = (cv.kh // 2, cv.kw // 2) if pad is None else pad
    if out_hw is None:
        Ho = (He + 2 * pt - cv.kh) // stride + 1
        Wo = (We + 2 * pl - cv.kw) // stride + 1
    else:
        Ho, Wo = out_hw
    if x.dtype == BF16 or mfma16 or (out is not None and out.dtype == BF16) or (res is not None and res.dtype == BF16):
        # configs[2]: bf16 storage and/or bf16 MFMA (mfma16: fp32-stored input converted while staging)
        return _conv16(x, cv, out, stride, pt, pl, up2, act, res, Ho, Wo, d2s, tile, in_ss, in_swish, out_dtype, want_stats)
    if out is None:
        if d2s:
            out = torch.empty((B, Ho * d2s[0], Wo * d2s[0], d2s[1]), device=x.device, dtype=torch.float32)
        else:
            out = torch.empty((B, Ho, Wo, cv.cout), device=x.device, dtype=torch.float32)
    a_ptr, lda = _pix(x, "conv input")
    c_ptr, ldc = _pix(out, "conv output")
    r_ptr, ldr = (None, 0) if res is None else _pix(res, "conv residual")
    if getattr(out, "_gn_part", None) is not None:      # a caller-provided buffer tagged by an earlier producer
        out._gn_part = None
    if (SMALLN and not direct and tile == 0 and cv.kh == 3 and cv.kw == 3 and stride == 1 and (pt, pl) == (1, 1) and not d2s and not up2
            and res is None and cv.cout <= 4 and Cin in (64, 128, 256) and (Ho, Wo) == (H, W) and W % 4 == 0 and lda % 4 == 0 and a_ptr % 16 == 0):
        # N <= 4: HBM-shaped, runs on the vector ALUs (the matrix cores would pad N to 32)
        meta = {"flops": 2.0 * B * Ho * Wo * cv.cout * 9 * Cin, "mfma_flops": 0.0, "M": B * Ho * Wo, "N": cv.cout, "K": 9 * Cin,
                "nb": 1, "k": 3, "bytes": 4.0 * B * Ho * Wo * (Cin + cv.cout)} if _PROFILE is not None else None
        L.check(_timed("conv_small_n", meta, L.load().smx_conv3x3_smalln_f32, a_ptr, lda, _dev(cv.w).data_ptr(),
                       None if cv.b is None else cv.b.data_ptr(), c_ptr, ldc, B, H, W, Cin, cv.cout, act,
                       None if in_ss is None else in_ss.data_ptr(), int(in_swish), _stream()), "smx_conv3x3_smalln_f32")
        return out
    if (WINOGRAD and not direct and up2 != 2 and tile == 0 and cv.kh == 3 and cv.kw == 3 and stride == 1 and (pt, pl) == (1, 1) and not d2s
            and (Ho, Wo) == (He, We) and Cin % 32 == 0 and He % 8 == 0 and We % 16 == 0
            and lda % 4 == 0 and a_ptr % 16 == 0):
        meta = {"flops": 2.0 * B * Ho * Wo * cv.cout * 9 * Cin, "mfma_flops": 2.0 * B * Ho * Wo * cv.cout * 4 * Cin,
                "M": B * Ho * Wo, "N": cv.cout, "K": 9 * Cin, "nb": 1, "k": 3, "wino": 1,
                "wide": _wino_wide(B, He, We, cv.cout)} if _PROFILE is not None else None
        if (_wino43_ok(B, He, We, Cin, cv.cout, up2, lda, ldc, ldr) and c_ptr % 16 == 0 and (r_ptr or 0) % 16 == 0
                and (cv.b is None or cv.b.data_ptr() % 16 == 0) and (in_ss is None or in_ss.data_ptr() % 16 == 0)):
            if meta is not None:
                meta.update(mfma_flops=2.0 * B * Ho * Wo * cv.cout * 2.25 * Cin, wide=0, w43=1)
            part = torch.empty((B, (He // 16) * (We // 32), cv.cout, 2), device=x.device, dtype=torch.float32) if want_stats else None
            L.check(_timed("gemm_conv", meta, L.load().smx_winograd43_conv3x3_f32, a_ptr, lda, cv.winograd43_u().data_ptr(),
                           None if cv.b is None else cv.b.data_ptr(), r_ptr, ldr, c_ptr, ldc, B, He, We, Cin, cv.cout, act,
                           None if in_ss is None else in_ss.data_ptr(), int(in_swish), None if part is None else part.data_ptr(), _stream()),
                    "smx_winograd43_conv3x3_f32")
            if part is not None:
                out._gn_part = part
            return out
        part = torch.empty((B, (He // 8) * (We // 16), cv.cout, 2), device=x.device, dtype=torch.float32) if want_stats else None
        if bf3 and _wino_bf3_ok(B, He, We, Cin, cv.cout, lda, ldc, ldr, 0, a_ptr, c_ptr, r_ptr, None if cv.b is None else cv.b.data_ptr(),
                        None if in_ss is None else in_ss.data_ptr(), ragged=(WINO_BF3 == 6 and (WINO_F16 == 2 or (WINO_F16 == 1 and in_ss is not None)))):
            # normalised input (O(1) by construction) -> the half-precision form: 3 products instead of 6 at the same measured error (tests/test_gpu_wino_bf3.py)
            npr = 4 if (WINO_BF3 == 6 and (WINO_F16 == 2 or (WINO_F16 == 1 and in_ss is not None))) else WINO_BF3
            if meta is not None:
                meta.update(mfma_flops=2.0 * B * Ho * Wo * cv.cout * 4 * Cin * (3 if npr == 4 else npr), bf3=npr)
            L.check(_timed("gemm_conv", meta, L.load().smx_winograd_bf3_conv3x3_f32, a_ptr, lda, (cv.winograd_f16_u() if npr == 4 else cv.winograd_bf3_u()).data_ptr(),
                           None if cv.b is None else cv.b.data_ptr(), r_ptr, ldr, c_ptr, ldc, B, He, We, Cin, cv.cout,
                           int(up2), act, None if in_ss is None else in_ss.data_ptr(), int(in_swish),
                           None if part is None else part.data_ptr(), npr, _stream()),
                    "smx_winograd_bf3_conv3x3_f32")
            if part is not None:
                out._gn_part = part
            return out
        L.check(_timed("gemm_conv", meta, L.load().smx_winograd_conv3x3_f32, a_ptr, lda, cv.winograd_u().data_ptr(),
                       None if cv.b is None else cv.b.data_ptr(), r_ptr, ldr, c_ptr, ldc, B, He, We, Cin, cv.cout,
                       int(up2), act, None if in_ss is None else in_ss.data_ptr(), int(in_swish),
                       None if part is None else part.data_ptr(), _stream()),
                "smx_winograd_conv3x3_f32")
        if part is not None:
            out._gn_part = part
        return out
    if (CONV7_F16 and bf3 and not direct and tile == 0 and cv.kh == 7 and cv.kw == 7 and stride == 1 and (pt, pl) in ((3, 3), (0, 0)) and not d2s and not up2
            and res is None and in_ss is None and Cin % 4 == 0 and cv.cout <= 96 and (Ho, Wo) == (H + 2 * pt - 6, W + 2 * pl - 6)
            and lda % 4 == 0 and a_ptr % 16 == 0 and ldc == cv.cout and act in (ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_SIGMOID)
            and B * ((Ho + 7) // 8) * ((Wo + 31) // 32) >= 256):
        # the motion estimator's 7x7 heads, big launches: region-direct on the 16-bit MFMA in the f16x3 arithmetic (fp32-grade; 2.5x conv7_f32 / the implicit GEMM)
        meta = {"flops": 2.0 * B * Ho * Wo * cv.cout * 49 * Cin, "mfma_flops": 6.0 * B * Ho * Wo * 32 * ((cv.cout + 31) // 32) * 49 * 16 * ((Cin + 15) // 16),
                "M": B * Ho * Wo, "N": cv.cout, "K": 49 * Cin, "nb": 1, "k": 7, "bf3": 4} if _PROFILE is not None else None
        L.check(_timed("gemm_conv", meta, L.load().smx_conv7_f16_f32, a_ptr, lda, cv.w7_f16.data_ptr(), None if cv.b is None else cv.b.data_ptr(),
                       c_ptr, ldc, B, H, W, Cin, cv.cout, pt, act, _stream()), "smx_conv7_f16_f32")
        return out
    if (CONV7_F32 and not direct and tile == 0 and cv.kh == 7 and cv.kw == 7 and stride == 1 and (pt, pl) in ((3, 3), (0, 0)) and not d2s and not up2
            and res is None and in_ss is None and Cin % 16 == 0 and cv.cout <= 96 and (Ho, Wo) == (H + 2 * pt - 6, W + 2 * pl - 6)
            and lda % 4 == 0 and a_ptr % 16 == 0 and ldc == cv.cout and act in (ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_SIGMOID)
            and B * ((Ho + 7) // 8) * ((Wo + 31) // 32) >= 256):
        # the mask / occlusion head (128 -> 17): region-direct on the fp32 MFMA (the implicit GEMM staged every input pixel 49 times): 6.0 -> 3.5 ms.
        # C_in % 16 == 0 only: the keypoint head's 36 channels pad to 48 here and lose to the implicit GEMM (4.4 against 4.1 ms)
        meta = {"flops": 2.0 * B * Ho * Wo * cv.cout * 49 * Cin, "mfma_flops": 2.0 * B * Ho * Wo * 32 * ((cv.cout + 31) // 32) * 49 * 16 * ((Cin + 15) // 16),
                "M": B * Ho * Wo, "N": cv.cout, "K": 49 * Cin, "nb": 1, "k": 7} if _PROFILE is not None else None
        L.check(_timed("gemm_conv", meta, L.load().smx_conv7_f32, a_ptr, lda, cv.w7_f32.data_ptr(), None if cv.b is None else cv.b.data_ptr(),
                       c_ptr, ldc, B, H, W, Cin, cv.cout, pt, act, _stream()), "smx_conv7_f32")
        return out
    if (CONV7_C2 and not direct and tile == 0 and cv.kh == 7 and cv.kw == 7 and Cin == 2 and stride == 1 and (pt, pl) == (3, 3) and not d2s and not up2
            and res is None and in_ss is None and cv.cout % 128 == 0 and (Ho, Wo) == (H, W) and H % 8 == 0 and W % 32 == 0 and lda == 2
            and ldc % 4 == 0 and a_ptr % 8 == 0 and c_ptr % 16 == 0 and act in (ACT_NONE, ACT_RELU, ACT_LRELU02) and B * (H // 8) * (W // 32) >= 256):
        # BasicMotionEncoder.convf1 (7x7, 2 channels): one MFMA per tap, the k pair = the channel pair, operands straight from a 4 KB LDS region
        meta = {"flops": 2.0 * B * Ho * Wo * cv.cout * 98, "M": B * Ho * Wo, "N": cv.cout, "K": 98, "nb": 1, "k": 7} if _PROFILE is not None else None
        L.check(_timed("gemm_conv", meta, L.load().smx_conv7_c2_f32, a_ptr, cv.w7_c2f.data_ptr(), None if cv.b is None else cv.b.data_ptr(),
                       c_ptr, ldc, B, H, W, cv.cout, act, _stream()), "smx_conv7_c2_f32")
        return out
    if in_ss is not None:      # layer not eligible for the fused loader: normalise in its own pass, then convolve
        x = groupnorm_apply(x, in_ss, in_swish)
        a_ptr, lda = _pix(x, "conv input")
    M, K = B * Ho * Wo, cv.kh * cv.kw * Cin
    rp3 = (GEMM_RP and GEMM_RP_BF3 and bf3 and not direct and tile == 0 and cv.kh == 1 and cv.kw == 1 and stride == 1 and (pt, pl) == (0, 0) and not up2
           and (Ho, Wo) == (H, W) and M >= GEMM_RP_BF3_MIN_ROWS and L.load().smx_gemm_rp_bf3_ok(M, cv.cout, K) and lda % 4 == 0 and ldc % 4 == 0
           and a_ptr % 16 == 0 and c_ptr % 16 == 0 and (cv.b is None or cv.b.data_ptr() % 16 == 0))
    if rp3 and not d2s and (res is None or (ldr % 4 == 0 and r_ptr % 16 == 0)):
        # the short-K token Linears / 1x1 convolutions of the fp32 configuration: fp32-grade products on the bf16 matrix pipe (three-way split operands)
        npr = 4 if GEMM_RP_F16 else 6
        meta = {"flops": 2.0 * M * cv.cout * K, "mfma_flops": (6.0 if npr == 4 else 12.0) * M * cv.cout * K, "M": M, "N": cv.cout, "K": K, "nb": 1, "k": 1, "rp": 1, "bf3": npr} if _PROFILE is not None else None
        L.check(_timed("gemm_conv", meta, L.load().smx_gemm_rp_f16 if npr == 4 else L.load().smx_gemm_rp_bf3, a_ptr, lda, (cv.w_rp3h if npr == 4 else cv.w_rp3).data_ptr(),
                       None if cv.b is None else cv.b.data_ptr(), r_ptr, ldr, c_ptr, ldc, M, cv.cout, K, act, _stream()), "smx_gemm_rp_bf3")
        return out
    if rp3 and d2s and res is None and d2s[1] % 16 == 0:
        npr = 4 if GEMM_RP_F16 else 6
        meta = {"flops": 2.0 * M * cv.cout * K, "mfma_flops": (6.0 if npr == 4 else 12.0) * M * cv.cout * K, "M": M, "N": cv.cout, "K": K, "nb": 1, "k": 1, "rp": 1, "bf3": npr} if _PROFILE is not None else None
        L.check(_timed("gemm_conv", meta, L.load().smx_gemm_rp_d2s_f16 if npr == 4 else L.load().smx_gemm_rp_d2s_bf3, a_ptr, lda, (cv.w_rp3h if npr == 4 else cv.w_rp3).data_ptr(),
                       None if cv.b is None else cv.b.data_ptr(), c_ptr, ldc, M, cv.cout, K, act, d2s[0], d2s[1], Ho, Wo, _stream()), "smx_gemm_rp_d2s_bf3")
        return out
    if (GEMM_RP and not direct and tile == 0 and cv.kh == 1 and cv.kw == 1 and stride == 1 and (pt, pl) == (0, 0) and not up2 and not d2s
            and (Ho, Wo) == (H, W) and M >= GEMM16_RP_MIN_ROWS and L.load().smx_gemm_rp_f32_ok(M, cv.cout, K) and lda % 4 == 0 and ldc % 4 == 0
            and a_ptr % 16 == 0 and c_ptr % 16 == 0 and (res is None or (ldr % 4 == 0 and r_ptr % 16 == 0))):
        # short-K token Linears: persistent row-panel kernel (weights in registers, rows by LDS-DMA)
        meta = {"flops": 2.0 * M * cv.cout * K, "M": M, "N": cv.cout, "K": K, "nb": 1, "k": 1, "rp": 1} if _PROFILE is not None else None
        L.check(_timed("gemm_conv", meta, L.load().smx_gemm_rp_f32, a_ptr, lda, cv.w_rp.data_ptr(), None if cv.b is None else cv.b.data_ptr(),
                       r_ptr, ldr, c_ptr, ldc, M, cv.cout, K, act, _stream()), "smx_gemm_rp_f32")
        return out
    if (GEMM_RP and d2s and not direct and tile == 0 and cv.kh == 1 and cv.kw == 1 and stride == 1 and (pt, pl) == (0, 0) and not up2 and res is None
            and (Ho, Wo) == (H, W) and M >= GEMM16_RP_MIN_ROWS and d2s[1] % 16 == 0 and cv.cout % 256 == 0 and L.load().smx_gemm_rp_f32_ok(M, cv.cout, K)
            and lda % 4 == 0 and ldc % 4 == 0 and a_ptr % 16 == 0 and c_ptr % 16 == 0):
        meta = {"flops": 2.0 * M * cv.cout * K, "M": M, "N": cv.cout, "K": K, "nb": 1, "k": 1, "rp": 1} if _PROFILE is not None else None
        L.check(_timed("gemm_conv", meta, L.load().smx_gemm_rp_d2s_f32, a_ptr, lda, cv.w_rp.data_ptr(), None if cv.b is None else cv.b.data_ptr(),
                       c_ptr, ldc, M, cv.cout, K, act, d2s[0], d2s[1], Ho, Wo, _stream()), "smx_gemm_rp_d2s_f32")
        return out
    ksplit, ws = 1, None
    if not d2s and K >= 1024:
        # weight-streaming layers (deep hourglass): few output tiles, long K -> split K over blocks
        blocks = ((M + 63) // 64) * ((cv.cout + 63) // 64)
        if blocks < 256:
            ksplit = max(1, min(K // 32 // 8, (1024 + blocks - 1) // blocks, 32))
            if ksplit > 1:
                ws = torch.empty((ksplit, M, cv.cout), device=x.device, dtype=torch.float32)
    gemm_raw(a=a_ptr, bt=_dev(cv.w).data_ptr(), c=c_ptr, bias=None if cv.b is None else cv.b.data_ptr(),
             res=r_ptr, nb0=1, nb1=1, M=B * Ho * Wo, N=cv.cout, K=cv.kh * cv.kw * Cin,
             lda=lda, ldb=cv.w.shape[1], ldc=ldc, ldres=ldr, Hin=H, Win=W, Cin=Cin, Ho=Ho, Wo=Wo,
             kh=cv.kh, kw=cv.kw, stride=stride, pad_t=pt, pad_l=pl, up2=int(up2), act=act, alpha=1.0,
             d2s_p=d2s[0] if d2s else 0, d2s_c=d2s[1] if d2s else 0, tile=tile, ksplit=ksplit,
             ws=None if ws is None else ws.data_ptr())
    return out


def conv7_x3(x, cv, pad=3, act=ACT_NONE):
    """7x7 / stride 1 head on fp32 storage in bf16x3 arithmetic (csrc/conv7_bf16x3.hip): x [B,H,W,Cin] fp32 -> [B,H+2pad-6,W+2pad-6,N] fp32."""
    B, H, W, Cin = x.shape
    a_ptr, lda = _pix(x, "conv7 input")
    out = torch.empty((B, H + 2 * pad - 6, W + 2 * pad - 6, cv.cout), device=x.device, dtype=torch.float32)
    M = B * out.shape[1] * out.shape[2]
    meta = {"flops": 2.0 * M * cv.cout * 49 * Cin, "mfma_flops": 6.0 * M * 32 * ((cv.cout + 31) // 32) * 49 * 16 * ((Cin + 15) // 16),
            "M": M, "N": cv.cout, "K": 49 * Cin, "nb": 1, "k": 7, "bf16": 1} if _PROFILE is not None else None
    L.check(_timed("conv7_x3", meta, L.load().smx_conv7_bf16x3_f32, a_ptr, lda, cv.w7_x3.data_ptr(), None if cv.b is None else cv.b.data_ptr(),
                   out.data_ptr(), cv.cout, B, H, W, Cin, cv.cout, pad, act, _stream()), "smx_conv7_bf16x3_f32")
    return out


def gemm_nt(a, bt, c, *, M, N, K, lda, ldb, ldc, nb0=1, nb1=1, a_bs=(0, 0), bt_bs=(0, 0), c_bs=(0, 0),
            bias=None, bias_per_row=False, alpha=1.0, act=ACT_NONE, res=None, ldres=0, res_bs=(0, 0),
            a_off=0, bt_off=0, c_off=0, tile=0):
    """batched C[g] = act(alpha * A[g] @ Bt[g]^T + bias) (+res); offsets/strides in elements."""
    if a.dtype == BF16 or bt.dtype == BF16:
        if bt.dtype != BF16:
            raise L.SmxError("gemm_nt: the bf16 MFMA path needs bf16 Bt")
        _dev(a, "gemm A", _ANY), _dev(c, "gemm C", _ANY)
        asz, csz = a.element_size(), c.element_size()
        gemm16_raw(a=a.data_ptr() + asz * a_off, a_bs0=a_bs[0], a_bs1=a_bs[1], bt=bt.data_ptr() + 2 * bt_off, bt_bs0=bt_bs[0], bt_bs1=bt_bs[1],
                   c=c.data_ptr() + csz * c_off, c_bs0=c_bs[0], c_bs1=c_bs[1], bias=None if bias is None else _dev(bias).data_ptr(),
                   res=None if res is None else _dev(res, "gemm res", _ANY).data_ptr(), res_bs0=res_bs[0], res_bs1=res_bs[1],
                   in_ss=None, in_swish=0, nb0=nb0, nb1=nb1, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, ldres=ldres,
                   Hin=M, Win=1, Cin=K, Ho=M, Wo=1, kh=1, kw=1, stride=1, pad_t=0, pad_l=0, up2=0, act=act, alpha=alpha,
                   bias_per_row=int(bias_per_row), d2s_p=0, d2s_c=0, tile=tile, ksplit=1, ws=None,
                   a_f32=int(a.dtype == torch.float32), c_f32=int(c.dtype == torch.float32),
                   res_f32=int(res is not None and res.dtype == torch.float32))
        return c
    gemm_raw(a=_dev(a).data_ptr() + 4 * a_off, a_bs0=a_bs[0], a_bs1=a_bs[1],
             bt=_dev(bt).data_ptr() + 4 * bt_off, bt_bs0=bt_bs[0], bt_bs1=bt_bs[1],
             c=_dev(c).data_ptr() + 4 * c_off, c_bs0=c_bs[0], c_bs1=c_bs[1],
             bias=None if bias is None else bias.data_ptr(),
             res=None if res is None else res.data_ptr(), res_bs0=res_bs[0], res_bs1=res_bs[1],
             nb0=nb0, nb1=nb1, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, ldres=ldres,
             Hin=M, Win=1, Cin=K, Ho=M, Wo=1, kh=1, kw=1, stride=1, pad_t=0, pad_l=0, up2=0,
             act=act, alpha=alpha, bias_per_row=int(bias_per_row), d2s_p=0, d2s_c=0, tile=tile)
    return c


def groupnorm(x, gamma, beta, swish=True, out=None, groups=32, eps=1e-6):
    B, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((B, H, W, Cc), device=x.device, dtype=x.dtype)
    _same("groupnorm", x, out)
    xp, ldx = _pix(x, "groupnorm input")
    yp, ldy = _pix(out, "groupnorm output")
    lib = L.load()
    ws = torch.empty(int(lib.smx_groupnorm_ws_floats(B, H * W, Cc)), device=x.device, dtype=torch.float32)
    L.check(_timed("groupnorm", {"bytes": 2.0 * x.element_size() * B * H * W * Cc}, _fn("smx_groupnorm_swish_nhwc", x), xp, ldx, _dev(gamma).data_ptr(),
                   _dev(beta).data_ptr(), yp, ldy, B, H * W, Cc, groups, eps, int(swish), ws.data_ptr(), _stream()), "groupnorm")
    return out


EPILOGUE_STATS = True      # groupnorm_stats consumes the Welford partials a producing convolution left on its output (False: always the two-pass kernel; tests)


def groupnorm_stats(x, gamma, beta, groups=32, eps=1e-6):
    """per-(b,c) {scale, shift} of GroupNorm for x [B,H,W,C] (fp32 or bf16 storage) -> ss [B,C,2] fp32 (consumed by conv(in_ss=...))."""
    B, H, W, Cc = x.shape
    xp, ldx = _pix(x, "groupnorm input")
    lib = L.load()
    ss = torch.empty((B, Cc, 2), device=x.device, dtype=torch.float32)
    part = getattr(x, "_gn_part", None)
    if part is not None and part.shape[0] == B and part.shape[2] == Cc and EPILOGUE_STATS:
        # the producing conv already reduced its output per block: finalize only (reads B*nch*C*8 bytes)
        L.check(_timed("groupnorm", {"bytes": 8.0 * part.numel() / 2}, lib.smx_groupnorm_finalize_f32, part.data_ptr(),
                       _dev(gamma).data_ptr(), _dev(beta).data_ptr(), ss.data_ptr(), B, H * W, Cc, groups, part.shape[1], eps,
                       _stream()), "groupnorm_finalize")
        return ss
    ws = torch.empty(int(lib.smx_groupnorm_ws_floats(B, H * W, Cc)), device=x.device, dtype=torch.float32)
    L.check(_timed("groupnorm", {"bytes": 1.0 * x.element_size() * B * H * W * Cc}, _fn("smx_groupnorm_stats", x), xp, ldx, _dev(gamma).data_ptr(),
                   _dev(beta).data_ptr(), ss.data_ptr(), B, H * W, Cc, groups, eps, ws.data_ptr(), _stream()), "groupnorm_stats")
    return ss


def groupnorm_apply(x, ss, swish=True, out=None):
    B, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((B, H, W, Cc), device=x.device, dtype=x.dtype)
    _same("groupnorm_apply", x, out)
    xp, ldx = _pix(x, "groupnorm input")
    yp, ldy = _pix(out, "groupnorm output")
    L.check(_timed("groupnorm", {"bytes": 2.0 * x.element_size() * B * H * W * Cc}, _fn("smx_groupnorm_apply", x), xp, ldx, ss.data_ptr(), yp, ldy,
                   B, H * W, Cc, int(swish), _stream()), "groupnorm_apply")
    return out


def layernorm(x, gamma, beta, pos=None, eps=1e-5):
    """x tokens [..., E] contiguous (fp32 or bf16 storage; gamma / beta / pos fp32) -> (LN(x), LN(x)+pos or None)."""
    _dev(x, "layernorm input", _ANY)
    if not x.is_contiguous():
        raise L.SmxError("layernorm: contiguous tokens expected")
    E = x.shape[-1]
    T = x.numel() // E
    y = torch.empty_like(x)
    yp = torch.empty_like(x) if pos is not None else None
    L.check(_timed("layernorm", {"bytes": x.element_size() * (2.0 + (pos is not None)) * x.numel()}, _fn("smx_layernorm_pos", x), x.data_ptr(),
                   _dev(gamma).data_ptr(), _dev(beta).data_ptr(), None if pos is None else _dev(pos).data_ptr(), y.data_ptr(),
                   None if yp is None else yp.data_ptr(), T, E, 0 if pos is None else pos.shape[0], eps, _stream()), "layernorm")
    return y, yp


def softmax_rows(s, S, scale=1.0, mask=None, rows_per_mask=0):
    """in-place softmax over the last dim (S) of contiguous s."""
    _dev(s, "softmax", _ANY)
    R = s.numel() // S
    L.check(_timed("softmax", {"bytes": 2.0 * s.element_size() * R * S}, _fn("smx_softmax_rows", s), s.data_ptr(), S, R, S, scale,
                   None if mask is None else mask.data_ptr(), rows_per_mask, _stream()), "softmax_rows")
    return s


def warp(feat, flow, occ=None, out=None):
    """deform_input (+occlude_input): feat [1|B,H,W,C] (fp32 or bf16 storage), flow [B,Hf,Wf,2] fp32, occ [B,Hf,Wf(,1)] fp32."""
    _dev(feat, "warp features", _ANY), _dev(flow)
    B, Hf, Wf, _ = flow.shape
    Bf, H, W, Cc = feat.shape
    if out is None:
        out = torch.empty((B, H, W, Cc), device=feat.device, dtype=feat.dtype)
    _same("warp", feat, out)
    if not (feat.is_contiguous() and flow.is_contiguous() and out.is_contiguous() and (occ is None or occ.is_contiguous())):
        raise L.SmxError("warp: contiguous operands expected")
    # algorithmic bytes: features read + written, flow, occlusion
    meta = {"bytes": feat.element_size() * 2.0 * B * H * W * Cc + 4.0 * B * Hf * Wf * (2 + (0 if occ is None else 1)), "s": H, "C": Cc}
    L.check(_timed("warp", meta, _fn("smx_warp_nhwc", feat), feat.data_ptr(), Bf, flow.data_ptr(),
                   None if occ is None else _dev(occ).data_ptr(), out.data_ptr(), B, H, W, Cc, Hf, Wf, _stream()), "warp")
    return out


def resize(x, Ho, Wo, out=None):
    B, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=x.dtype)
    _same("resize", x, out)
    xp, ldx = _pix(x, "resize input")
    yp, ldy = _pix(out, "resize output")
    L.check(_timed("resize", None, _fn("smx_resize_bilinear_ac_nhwc", x), xp, ldx, yp, ldy, B, H, W, Ho, Wo, Cc, _stream()), "resize")
    return out


def resize_taps_gather(x, Ho, Wo):
    """x [B,H,W,C] -> the 4 bilinear (align_corners=True) taps of every output pixel, [B,Ho,Wo*4,C] (a dense NHWC
    map 4x as wide: row-compatible with a 1x1 conv)."""
    B, H, W, Cc = x.shape
    xp, ldx = _pix(x, "resize_taps input")
    t = torch.empty((B, Ho, Wo * 4, Cc), device=x.device, dtype=x.dtype)
    L.check(_fn("smx_resize_taps_gather", x)(xp, ldx, t.data_ptr(), B, H, W, Ho, Wo, Cc, _stream()), "resize_taps_gather")
    return t


def resize_taps_combine(t, Hin, Win, out=None):
    """t [B,Ho,Wo*4,C] (a per-pixel op applied to `resize_taps_gather`) -> the bilinear blend [B,Ho,Wo,C]: equals
    resize(op(x), Ho, Wo) for the [B,Hin,Win,*] map the taps were gathered from."""
    B, Ho, W4, Cc = t.shape
    if not t.is_contiguous() or W4 % 4:
        raise L.SmxError("resize_taps_combine: dense [B,Ho,Wo*4,C] expected")
    if out is None:
        out = torch.empty((B, Ho, W4 // 4, Cc), device=t.device, dtype=t.dtype)
    _same("resize_taps_combine", t, out)
    yp, ldy = _pix(out, "resize_taps output")
    L.check(_fn("smx_resize_taps_combine", t)(_dev(t, "taps", _ANY).data_ptr(), yp, ldy, B, Hin, Win, Ho, W4 // 4, Cc, _stream()), "resize_taps_combine")
    return out


def avgpool2(x, out=None):
    B, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((B, H // 2, W // 2, Cc), device=x.device, dtype=torch.float32)
    xp, ldx = _pix(x, "avgpool input")
    yp, ldy = _pix(out, "avgpool output")
    L.check(L.load().smx_avgpool2_nhwc_f32(xp, ldx, yp, ldy, B, H, W, Cc, _stream()), "avgpool2")
    return out


def antialias_down(img_nchw, w, out=None, step=4):
    _dev(img_nchw)
    B, Cc, H, W = img_nchw.shape
    K = w.shape[-1]
    if out is None:
        out = torch.empty((B, (H + step - 1) // step, (W + step - 1) // step, Cc), device=img_nchw.device, dtype=torch.float32)
    yp, ldy = _pix(out, "antialias output")
    xc, wc = img_nchw.contiguous(), _dev(w).contiguous()     # dense copies (if any) stay referenced until the launch is queued: see _live
    L.check(L.load().smx_antialias_down_f32(xc.data_ptr(), wc.data_ptr(), yp, ldy, B, Cc, H, W, K, step, _stream()), "antialias_down")
    return out


def kp_head(logits, jmaps, K, temperature):
    B, H, W, _ = logits.shape
    lp, ldl = _pix(logits, "kp logits")
    jp, ldj = _pix(jmaps, "kp jacobian maps")
    value = torch.empty((B, K, 2), device=logits.device, dtype=torch.float32)
    jac = torch.empty((B, K, 2, 2), device=logits.device, dtype=torch.float32)
    L.check(L.load().smx_kp_head_f32(lp, ldl, jp, ldj, value.data_ptr(), jac.data_ptr(), B, H, W, K, temperature, _stream()), "kp_head")
    return value, jac


def normalize_kp(kp_d, kp_0, kp_s, scale, rel_move, rel_jac):
    """device version of demo.py:24-44 for a batch of driving keypoints against one source / initial frame.
    scale: host float, or a one-element device tensor (the hull ratio as it arrives inside a broadcast source state:
    read by the kernel, never synchronised to the host; NaN there means "no adaptation")."""
    v, j = _dev(kp_d["value"]).contiguous(), _dev(kp_d["jacobian"]).contiguous()
    B, K = v.shape[0], v.shape[1]
    ov, oj = torch.empty_like(v), torch.empty_like(j)
    live = []                                                # see _live

    def ptr(d, k):
        if d is None:
            return None
        live.append(_dev(d[k]).contiguous())
        return live[-1].data_ptr()
    if torch.is_tensor(scale):
        _dev(scale, "normalize_kp scale")
        L.check(L.load().smx_normalize_kp_dscale_f32(v.data_ptr(), j.data_ptr(), ptr(kp_0, "value"), ptr(kp_0, "jacobian"), ptr(kp_s, "value"),
                                                     ptr(kp_s, "jacobian"), ov.data_ptr(), oj.data_ptr(), B, K, scale.data_ptr(), int(rel_move),
                                                     int(rel_jac), _stream()), "normalize_kp")
    else:
        L.check(L.load().smx_normalize_kp_f32(v.data_ptr(), j.data_ptr(), ptr(kp_0, "value"), ptr(kp_0, "jacobian"), ptr(kp_s, "value"),
                                              ptr(kp_s, "jacobian"), ov.data_ptr(), oj.data_ptr(), B, K, float(scale), int(rel_move),
                                              int(rel_jac), _stream()), "normalize_kp")
    return {"value": ov, "jacobian": oj}


def sparse_motion(src64, kpd_value, kpd_jac, kps_value, kps_jac, hg_in, B, K=15, var=0.01):
    """-> (sparse [B,K+1,H,W,2], drv_heat [B,H,W,K]); hg_in slice [B,H,W,4(K+1)] is written."""
    Bs, H, W, _ = src64.shape
    hp, ldh = _pix(hg_in, "hourglass input")
    sparse = torch.empty((B, K + 1, H, W, 2), device=src64.device, dtype=torch.float32)
    heat = torch.empty((B, H, W, K), device=src64.device, dtype=torch.float32)
    for t in (kpd_value, kpd_jac, kps_value, kps_jac):
        _dev(t)
    live = _live(_dev(src64), kpd_value, kpd_jac, kps_value, kps_jac)
    L.check(L.load().smx_sparse_motion_f32(live[0].data_ptr(), Bs, live[1].data_ptr(), live[2].data_ptr(), live[3].data_ptr(),
                                           live[4].data_ptr(), kps_value.shape[0], hp, ldh, sparse.data_ptr(),
                                           heat.data_ptr(), B, H, W, K, var, _stream()), "sparse_motion")
    return sparse, heat


def mask_deformation(mask_logits, sparse, want_mask=False, K1=None, fused_occ=False):
    """mask_logits [B,H,W,>=K1]; fused_occ: channel K1 is the occlusion logit -> also returns sigmoid."""
    B, H, W, Cl = mask_logits.shape
    K1 = Cl if K1 is None else K1
    mp, ldm = _pix(mask_logits, "mask logits")
    deform = torch.empty((B, H, W, 2), device=mask_logits.device, dtype=torch.float32)
    mask = torch.empty((B, H, W, K1), device=mask_logits.device, dtype=torch.float32) if want_mask else None
    occ = torch.empty((B, H, W), device=mask_logits.device, dtype=torch.float32) if fused_occ else None
    L.check(L.load().smx_mask_deformation_f32(mp, ldm, sparse.data_ptr(), deform.data_ptr(),
                                              None if mask is None else mask.data_ptr(),
                                              None if occ is None else occ.data_ptr(), B, H, W, K1, _stream()), "mask_deformation")
    return deform, mask, occ


def flow_to_residual(flow):
    B, H, W, _ = flow.shape
    res = torch.empty_like(flow)
    L.check(L.load().smx_flow_to_residual_f32(_dev(flow).data_ptr(), res.data_ptr(), B, H, W, _stream()), "flow_to_residual")
    return res


def flow_occ_update(flow, r, occ_prev):
    B, H, W, _ = flow.shape
    m_com, res_norm = torch.empty_like(flow), torch.empty_like(flow)
    occ = torch.empty((B, H, W), device=flow.device, dtype=torch.float32)
    L.check(L.load().smx_flow_occ_update_f32(_dev(flow).data_ptr(), _dev(r).data_ptr(), _dev(occ_prev).data_ptr(), m_com.data_ptr(),
                                             res_norm.data_ptr(), occ.data_ptr(), B, H, W, _stream()), "flow_occ_update")
    return m_com, res_norm, occ


def motion_ignore(flow, Ht=32, Wt=32):
    B, Hf, Wf, _ = flow.shape
    ign = torch.empty((B, Ht * Wt), device=flow.device, dtype=torch.uint8)
    L.check(L.load().smx_motion_ignore_f32(_dev(flow).data_ptr(), ign.data_ptr(), B, Hf, Wf, Ht, Wt, _stream()), "motion_ignore")
    return ign


def sft_combine(dec, scale, shift, w=1.0):
    """dec may be a channel-slice view (the dec half of the [enc|dec] buffer); scale / shift dense."""
    dp, ldd = _pix(dec, "dec")
    _same("sft_combine", dec, scale, shift)
    Cc = dec.shape[-1]
    if not (scale.is_contiguous() and shift.is_contiguous()):
        raise L.SmxError("sft_combine: dense scale / shift expected")
    out = torch.empty(dec.shape, device=dec.device, dtype=dec.dtype)
    L.check(_timed("sft_combine", {"bytes": 4.0 * dec.element_size() * dec.numel()}, _fn("smx_sft_combine", dec), dp, ldd, scale.data_ptr(),
                   shift.data_ptr(), out.data_ptr(), float(w), dec.numel() // Cc, Cc, _stream()), "sft_combine")
    return out


def conv_sft(x, cv, dec, scale, w=1.0):
    """Fuse_sft_block's modulation with the shift branch's last conv folded in (archs/appmotioncodebook_arch.py:49-51):
    dec + w * (dec * scale + conv3x3(x)).  fp32 + Winograd-eligible shapes run it as the conv's epilogue
    (smx_winograd_conv3x3_sft_f32); anything else is the conv followed by sft_combine."""
    B, H, W, Cin = x.shape
    fused = (WINOGRAD and x.dtype == torch.float32 and dec.dtype == torch.float32 and scale.dtype == torch.float32
             and cv.kh == 3 and cv.kw == 3 and Cin == cv.cin and Cin % 32 == 0 and H % 8 == 0 and W % 16 == 0 and cv.cout % 4 == 0
             and tuple(dec.shape) == (B, H, W, cv.cout) and tuple(scale.shape) == (B, H, W, cv.cout))
    if fused:
        a_ptr, lda = _pix(x, "conv_sft input")
        d_ptr, ldd = _pix(dec, "conv_sft dec")
        s_ptr, lds_ = _pix(scale, "conv_sft scale")
        fused = (lda % 4 == 0 and ldd % 4 == 0 and lds_ % 4 == 0 and (a_ptr | d_ptr | s_ptr) % 16 == 0
                 and (cv.b is None or cv.b.data_ptr() % 16 == 0))
    if (not fused and REGION3X3 and x.dtype == BF16 and dec.dtype == BF16 and scale.dtype == BF16 and cv.kh == 3 and cv.kw == 3
            and Cin == cv.cin and Cin % 64 == 0 and H % 8 == 0 and W % 16 == 0 and cv.cout % 8 == 0
            and tuple(dec.shape) == (B, H, W, cv.cout) and tuple(scale.shape) == (B, H, W, cv.cout)):
        a_ptr, lda = _pix(x, "conv_sft input")
        d_ptr, ldd = _pix(dec, "conv_sft dec")
        s_ptr, lds_ = _pix(scale, "conv_sft scale")
        if lda % 8 == 0 and ldd % 8 == 0 and lds_ % 8 == 0 and (a_ptr | d_ptr | s_ptr) % 16 == 0:
            out = torch.empty((B, H, W, cv.cout), device=x.device, dtype=BF16)
            if _conv16_t32_ok(B, H, W, Cin, cv.cout):
                meta = {"flops": 2.0 * B * H * W * cv.cout * 9 * Cin, "M": B * H * W, "N": cv.cout, "K": 9 * Cin, "nb": 1, "k": 3, "bf16": 1, "t32": 1,
                        "bytes": 2.0 * B * H * W * (Cin + 3 * cv.cout)} if _PROFILE is not None else None
                L.check(_timed("conv3x3_bf16", meta, L.load().smx_conv3x3_sft_bf16_t32, a_ptr, lda, cv.w16_t32.data_ptr(),
                               None if cv.b is None else cv.b.data_ptr(), d_ptr, ldd, s_ptr, lds_, float(w), out.data_ptr(), cv.cout,
                               B, H, W, Cin, cv.cout, _stream()), "smx_conv3x3_sft_bf16_t32")
                return out
            th = _conv16_tile_h(Cin, H, B * (H // 16) * (W // 16) * ((cv.cout + 63) // 64))
            meta = {"flops": 2.0 * B * H * W * cv.cout * 9 * Cin, "M": B * H * W, "N": cv.cout, "K": 9 * Cin, "nb": 1, "k": 3, "bf16": 1,
                    "bytes": 2.0 * B * H * W * (Cin + 3 * cv.cout)} if _PROFILE is not None else None
            L.check(_timed("conv3x3_bf16", meta, L.load().smx_conv3x3_sft_bf16, a_ptr, lda, cv.w16.data_ptr(), cv.w16.shape[1],
                           None if cv.b is None else cv.b.data_ptr(), d_ptr, ldd, s_ptr, lds_, float(w), out.data_ptr(), cv.cout,
                           B, H, W, Cin, cv.cout, th, _stream()), "smx_conv3x3_sft_bf16")
            return out
    if not fused:
        return sft_combine(dec, scale, conv(x, cv), w)
    out = torch.empty((B, H, W, cv.cout), device=x.device, dtype=torch.float32)
    meta = {"flops": 2.0 * B * H * W * cv.cout * 9 * Cin, "mfma_flops": 2.0 * B * H * W * cv.cout * 4 * Cin,
            "M": B * H * W, "N": cv.cout, "K": 9 * Cin, "nb": 1, "k": 3, "wino": 1, "wide": _wino_wide(B, H, W, cv.cout)} if _PROFILE is not None else None
    if _wino_bf3_ok(B, H, W, Cin, cv.cout, lda, cv.cout, ldd, lds_, a_ptr, d_ptr, s_ptr, out.data_ptr(), None if cv.b is None else cv.b.data_ptr()):
        npr = 4 if (WINO_BF3 == 6 and WINO_F16 == 2) else WINO_BF3
        if meta is not None:
            meta.update(mfma_flops=2.0 * B * H * W * cv.cout * 4 * Cin * (3 if npr == 4 else npr), bf3=npr)
        L.check(_timed("gemm_conv", meta, L.load().smx_winograd_bf3_conv3x3_sft_f32, a_ptr, lda, (cv.winograd_f16_u() if npr == 4 else cv.winograd_bf3_u()).data_ptr(),
                       None if cv.b is None else cv.b.data_ptr(), d_ptr, ldd, s_ptr, lds_, float(w), out.data_ptr(), cv.cout,
                       B, H, W, Cin, cv.cout, None, npr, _stream()), "smx_winograd_bf3_conv3x3_sft_f32")
        return out
    L.check(_timed("gemm_conv", meta, L.load().smx_winograd_conv3x3_sft_f32, a_ptr, lda, cv.winograd_u().data_ptr(),
                   None if cv.b is None else cv.b.data_ptr(), d_ptr, ldd, s_ptr, lds_, float(w), out.data_ptr(), cv.cout,
                   B, H, W, Cin, cv.cout, None, _stream()), "smx_winograd_conv3x3_sft_f32")
    return out


def fingerprint(x):
    """two 64-bit content hashes (raw bit patterns) of a small device tensor as a host tuple -- a content key for the source
    caches (one tiny launch + a 16-byte D2H copy)."""
    xc = _dev(x).contiguous()
    out = torch.empty((2,), device=x.device, dtype=torch.int64)
    L.check(L.load().smx_fingerprint_f32(xc.data_ptr(), xc.numel(), out.data_ptr(), _stream()), "fingerprint")
    return tuple(out.tolist())


def add(a, b):
    _same("add", a, b)
    y = torch.empty_like(a)
    L.check(_fn("smx_add", a)(_dev(a, "add", _ANY).data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream()), "add")
    return y


def copy_slice(x, out):
    """out[..., :C] = x (channel-slice views ok); converts between fp32 and bf16 storage when the dtypes differ."""
    xp, ldx = _pix(x, "copy input")
    yp, ldy = _pix(out, "copy output")
    Cc = x.shape[-1]
    L.check(L.load().smx_convert_slice(xp, int(x.dtype == BF16), ldx, yp, int(out.dtype == BF16), ldy, x.numel() // Cc, Cc, _stream()), "copy_slice")
    return out


def nchw_to_nhwc(x, out=None, dtype=torch.float32):
    """fp32 NCHW -> NHWC in `dtype` storage (fp32 | bf16)."""
    B, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((B, H, W, Cc), device=x.device, dtype=dtype)
    yp, ldy = _pix(out, "nhwc output")
    L.check(_fn("smx_nchw_to_nhwc", out)(_dev(x).contiguous().data_ptr(), yp, ldy, B, Cc, H, W, _stream()), "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x):
    """NHWC (fp32 or bf16 storage, channel-slice ok) -> fp32 NCHW."""
    B, H, W, Cc = x.shape
    xp, ldx = _pix(x, "nhwc input")
    y = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32)
    L.check(_fn("smx_nhwc_to_nchw", x)(xp, ldx, y.data_ptr(), B, Cc, H, W, _stream()), "nhwc_to_nchw")
    return y


def frames_u8_to_nchw(frames_u8, out_hw=(256, 256), swap_rb=False, mean=0.5, std=0.5, out=None):
    """uint8 HWC frames [B,H,W,3] on the device -> normalised fp32 NCHW [B,3,Ho,Wo] (demo.py:177-185 on the device)."""
    if not (torch.is_tensor(frames_u8) and frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[-1] == 3):
        raise L.SmxError("frames_u8_to_nchw: a uint8 device tensor [B,H,W,3] is expected")
    B, H, W, _ = frames_u8.shape
    if out is None:
        out = torch.empty((B, 3, out_hw[0], out_hw[1]), device=frames_u8.device, dtype=torch.float32)
    L.check(L.load().smx_frames_u8_to_nchw_f32(frames_u8.contiguous().data_ptr(), _dev(out).data_ptr(), B, H, W, out_hw[0], out_hw[1],
                                               int(swap_rb), float(mean), float(std), _stream()), "frames_u8_to_nchw")
    return out


def to_uint8(x, lo=-1.0, hi=1.0):
    y = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    L.check(L.load().smx_to_uint8_f32(_dev(x).contiguous().data_ptr(), y.data_ptr(), x.numel(), lo, hi, _stream()), "to_uint8")
    return y


def vq_nearest(z_tokens, codebook, Ks, want_zq=True):
    """z_tokens [N,D], codebook [K,D] -> (idx int64 [N], zq [N,D], dmin [N], sum (zq-z)^2)."""
    _dev(z_tokens), _dev(codebook)
    N, D = z_tokens.shape
    idx = torch.empty((N,), device=z_tokens.device, dtype=torch.int64)
    zq = torch.empty_like(z_tokens) if want_zq else None
    dmin = torch.empty((N,), device=z_tokens.device, dtype=torch.float32)
    sq = torch.empty((1,), device=z_tokens.device, dtype=torch.float32)
    ws = torch.empty((int(L.load().smx_vq_ws_floats(N)),), device=z_tokens.device, dtype=torch.float32)   # code norms + per-block loss partials (no atomics)
    meta = {"bytes": 8.0 * N * D + 4.0 * Ks * D + 8.0 * N, "flops": 2.0 * N * Ks * D}
    live = _live(z_tokens, codebook)
    L.check(_timed("vq", meta, L.load().smx_vq_nearest_f32, live[0].data_ptr(), live[1].data_ptr(),
                   idx.data_ptr(), None if zq is None else zq.data_ptr(), dmin.data_ptr(), sq.data_ptr(), ws.data_ptr(), N, D, Ks,
                   _stream()), "vq_nearest")
    return idx, zq, dmin, sq


def attention(q, k, v, nhead, dh, S, *, k_shared=False, mask=None, k_off=0, scale=None):
    """o = softmax(scale q k^T [+mask]) v per head.  q [B,1024,ldq] (view ok), k/v rows [S] of
    width ld (head h at column h*dh); k_shared: codebook K/V common to the batch (batch stride 0)."""
    B = q.shape[0]
    Lq = q.numel() // B // q.shape[-1]
    _same("attention", q, k, v)
    qp, ldq = _pix(q, "attention q")
    kp, ldk = _pix(k, "attention k")
    vp, ldv = _pix(v, "attention v")
    E = nhead * dh
    es = q.element_size()
    o = torch.empty((B, Lq, E), device=q.device, dtype=q.dtype)
    kbs = 0 if k_shared else (k.numel() // B // k.shape[-1]) * ldk
    vbs = 0 if k_shared else (v.numel() // B // v.shape[-1]) * ldv
    meta = {"flops": 4.0 * B * nhead * Lq * S * dh, "bytes": es * (2.0 * B * Lq * E + 2 * (1 if k_shared else B) * S * E)}
    if _PROFILE is not None and q.dtype == torch.float32:
        np_ = int(L.load().smx_attention_f32_uses_bf3(B, nhead, Lq, S, dh))          # split-bf16 arithmetic: 6 (+ 6 | 5) bf16 products per multiply
        if np_:
            meta.update(bf3=np_, mfma_flops=(6 if np_ == 4 else 6 + (6 if np_ == 3 else 5)) * 2.0 * B * nhead * Lq * S * dh)    # products per multiply, S and P V together
    L.check(_timed(f"attention_d{dh}", meta, _fn("smx_attention", q), qp, ldq, Lq * ldq, kp + es * k_off, ldk, kbs, vp, ldv, vbs,
                   o.data_ptr(), E, Lq * E, None if mask is None else mask.data_ptr(), B, nhead, Lq, S, dh,
                   dh ** -0.5 if scale is None else scale, _stream()), "attention")
    return o
