/*
 * smx.h -- C ABI of libsmx.so: the MI355X (gfx950) kernels of the per-frame reenactment
 * hot path of ShaelynZ/synergize-motion-appearance.
 *
 * The reference has NO native boundary on this path: its hot path is
 * stock ATen calls made from `basicsr/archs/*.py`.  The drop-in boundary is therefore the
 * Python plugin surface (ARCH_REGISTRY names / options/test.yml kwargs / checkpoint keys),
 * mirrored by `synergize_motion_appearance_amd/archs`, and THIS header is what that host
 * layer binds (ctypes).  Each entry point names the reference call
 * sites (file:line under /root/reference/basicsr) whose ATen op sequence it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated; activations are NHWC
 *     ([B][H][W][C], C fastest) -- tokens [B][1024][E] are the same layout at 32x32;
 *   - no allocation, no synchronisation, no retained pointers; kernels are enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return 0 on success, negative SMX_E* on a rejected argument / launch failure;
 *   - channel-sliced views: (ptr, ld = channels of the underlying buffer) -- a producer
 *     can write straight into a slice of a concat buffer, so torch.cat never materialises.
 */
#ifndef SMX_H
#define SMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMX_OK 0
#define SMX_EINVAL (-22)
#define SMX_ELAUNCH (-5)

/* activation codes for fused epilogues */
enum { SMX_ACT_NONE = 0, SMX_ACT_RELU = 1, SMX_ACT_LRELU02 = 2, SMX_ACT_SWISH = 3,
       SMX_ACT_GELU = 4, SMX_ACT_SIGMOID = 5 };

/* library identification: returns a static string "smx <version> gfx950" */
const char* smx_version(void);

/* Launch-selection knobs (tests / tuning tools; the defaults are the device-tuned choices).  Names: "wino_nw"
 * (1|2 N tiles per Winograd block, -1 auto), "wino_wide" (64-channel blocks: 1 = 4 waves x 64 n, 0 = 8 waves), "wino_ablate" /
 * "wino_nt" (tools), "gemm_variant", "gemm_xcd_swizzle", "warp_rows", "warp_reorder", "attn16" (bf16 storage, d_head 32: 1 = bf16
 * MFMA kernel, 0 = fp32 MFMA kernel), "attn4_mfma" (d_head 4: 1 = 4x4x1 MFMA kernel, 0 = VALU kernel).  Initialised once from the
 * SMX_* environment; the launch paths read the table, never the environment.  Returns SMX_EINVAL for an unknown name. */
int smx_set_tuning(const char* name, int value);
int smx_get_tuning(const char* name, int* value);

/* ---------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / batched NT-GEMM on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *   C[g][m][n] = epi( alpha * sum_k A[g][m][k] * Bt[g][n][k] )
 * A is gathered on the fly from an NHWC tensor (im2col never materialises):
 *   m -> (img, oy, ox), k -> (ky, kx, c); zero padding; optional virtual nearest x2
 *   upsampling of the input (`up2`) and asymmetric pads (pad_t/pad_l + implied bottom/right).
 * Bt is [N][K] row-major (k fastest): conv weights repacked [Cout][kh][kw][Cin], or an
 * activation (attention K / V^T).  g = g0*nb1+g1 is a two-level batch (frame, head).
 * epi: + bias (per n, or per m when bias_per_row), activation, + residual, then an
 * optional depth-to-space scatter (un-patchify: n = (p1*p+p2)*dc + c).
 *
 * Replaces: every `aten::convolution` / `addmm` / `bmm` on the path -- ResBlock, Upsample,
 * Downsample, AttnBlock q/k/v/proj + both bmm (archs/vqgan_arch.py:144-253), Hourglass
 * blocks with BN folded (utils/motion_estimator_util.py:214-230,363-380), 7x7 heads
 * (archs/keypoint_detector_arch.py:65,74; archs/dense_motion_arch.py:134,158),
 * nn.MultiheadAttention projections + QK^T + PV and the conv-FFN
 * (archs/appmotioncodebook_arch.py:69-74,101-121), patch (un)embedding Linear layers
 * (:222-240), BasicMotionEncoder/RefineFlow/to_context (:129-167,:296-301), SFT fusion
 * (:28-52,:259).
 * ------------------------------------------------------------------------------------- */
typedef struct smx_gemm_desc {
  const float* a;    int64_t a_bs0, a_bs1;     /* batch strides in elements (0 = shared) */
  const float* bt;   int64_t bt_bs0, bt_bs1;
  float* c;          int64_t c_bs0, c_bs1;
  const float* bias;                            /* [N] (or [M] if bias_per_row) or NULL */
  const float* res;  int64_t res_bs0, res_bs1; /* residual, same indexing as c, or NULL */
  int32_t nb0, nb1;
  int32_t M, N, K;                              /* per batch; K = kh*kw*Cin */
  int32_t lda, ldb, ldc, ldres;                 /* A pixel stride (channels of the buffer), Bt row stride, C row stride */
  int32_t Hin, Win, Cin, Ho, Wo;                /* im2col geometry; images per batch = M/(Ho*Wo) */
  int32_t kh, kw, stride, pad_t, pad_l, up2;
  int32_t act; float alpha;
  int32_t bias_per_row;
  int32_t d2s_p, d2s_c;                         /* 0 = plain store */
  int32_t tile;                                 /* 0 = auto; else force a tile config id (tests/tuning) */
  int32_t ksplit;                               /* >1: split K over blockIdx.z (needs ws, nb0=nb1=1, no d2s) */
  float* ws;                                    /* split-K workspace: ksplit*M*N floats */
} smx_gemm_desc;

int smx_gemm_conv_f32(const smx_gemm_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * BASELINE configs[2] form of the same contraction: bf16 storage, v_mfma_f32_32x32x16_bf16 with fp32 accumulate.
 * A: NHWC activation in bf16 (raw 16-bit) or, with a_f32, in fp32 (converted while staging: the fp32 keypoint / flow
 * maps enter the bf16 path here); Bt: bf16 [N][K]; C: bf16 or (c_f32) fp32; bias fp32; residual bf16 or (res_f32) fp32.
 * Strides are in elements of the operand's own type.  in_ss != NULL (nb0 = nb1 = 1): GroupNorm(+swish) of the producer
 * applied to A while staging, x*in_ss[img][c][0] + in_ss[img][c][1] (padding stays 0), as in smx_winograd_conv3x3_f32.
 * Same call sites as smx_gemm_conv_f32.
 * ------------------------------------------------------------------------------------- */
typedef struct smx_gemm16_desc {
  const void* a;     int64_t a_bs0, a_bs1;
  const void* bt;    int64_t bt_bs0, bt_bs1;
  void* c;           int64_t c_bs0, c_bs1;
  const float* bias;
  const void* res;   int64_t res_bs0, res_bs1;
  const float* in_ss;
  int32_t nb0, nb1;
  int32_t M, N, K;
  int32_t lda, ldb, ldc, ldres;
  int32_t Hin, Win, Cin, Ho, Wo;
  int32_t kh, kw, stride, pad_t, pad_l, up2;
  int32_t act; float alpha;
  int32_t bias_per_row;
  int32_t d2s_p, d2s_c;
  int32_t tile;
  int32_t ksplit;
  float* ws;
  int32_t a_f32, c_f32, res_f32, in_swish;
} smx_gemm16_desc;

int smx_gemm_conv_bf16(const smx_gemm16_desc* d, void* stream);
/* Short-K Linear / 1x1 convolution on bf16 storage, row-panel form (csrc/gemm_rp_bf16.hip): c[M][N] = act(a[M][K] w^T + bias) (+ res),
 * K = 128 | 256, N % 128 == 0, M % 32 == 0 (smx_gemm_rp_bf16_ok).  Persistent blocks stream 32-row tiles of `a` through LDS by LDS-DMA
 * while every wave keeps its weight fragments in registers; `wp` = the [N/32][K/16][64 lanes][8] fragment-ordered pack built once per
 * layer by smx_gemm_rp_bf16_pack from the [N][ldw >= K] bf16 layout smx_gemm_conv_bf16 takes ((N/32)*(K/16)*512 elements).
 * Replaces smx_gemm_conv_bf16 at the token Linears of the transformer layers (archs/appmotioncodebook_arch.py:69-70,101-115). */
/* 7x7 / stride 1 heads of the motion estimator on fp32 storage in "bf16x3" arithmetic (csrc/conv7_bf16x3.hip; configs[2] only): every
 * fp32 operand is split hi + lo (bf16 each) and a product is three bf16 MFMAs (hi hi + hi lo + lo hi), fp32 accumulate: ~2^-17 relative
 * per product, so keypoints / jacobians / masks keep fp32-grade accuracy at 3/16 of the fp32 pipe's time; region-direct (the input is
 * staged once per 16-channel slice, not once per tap).  w [N][7][7][Cin] fp32 -> wp by smx_conv7_bf16x3_pack
 * (smx_conv7_bf16x3_pack_elems bf16 elements); pad 0 (valid: archs/keypoint_detector_arch.py:60-86) or 3 (archs/dense_motion_arch.py:
 * 118-161); Cin % 4 == 0, N <= 96; y [B][H+2pad-6][W+2pad-6][ldc] fp32. */
/* BasicMotionEncoder.convf1 in the bf16 configuration (csrc/conv7_c2_bf16.hip): 7x7 / pad 3 convolution of a 2-channel fp32 map (dense,
 * [B][H][W][2]) to N % 128 == 0 bf16 channels; wp = [N/32][7][64 lanes][8] bf16 from smx_conv7_c2_bf16_pack (w [N][7][7][2] fp32, K padded
 * 98 -> 112); H % 8 == 0, W % 32 == 0. */
int smx_conv7_c2_bf16_pack(const float* w, void* wp, int N, void* stream);
int smx_conv7_c2_bf16(const float* x, const void* wp, const float* bias, void* y, int ldc, int B, int H, int W, int N, int act, void* stream);
/* the fp32 configuration's form (v_mfma_f32_32x32x2_f32: the MFMA's k pair is the channel pair; wp = [N/32][49][64 lanes] floats), fp32 output */
int smx_conv7_c2_f32_pack(const float* w, float* wp, int N, void* stream);
int smx_conv7_c2_f32(const float* x, const float* wp, const float* bias, float* y, int ldc, int B, int H, int W, int N, int act, void* stream);
long long smx_conv7_bf16x3_pack_elems(int Cin, int N);
int smx_conv7_bf16x3_pack(const float* w, void* wp, int Cin, int N, void* stream);
int smx_conv7_bf16x3_f32(const float* x, int lda, const void* wp, const float* bias, float* y, int ldc, int B, int H, int W, int Cin,
                         int N, int pad, int act, void* stream);
/* smx_conv7_bf16x3_f32 in the "f16x3" arithmetic (csrc/conv7_bf16x3.hip, F16 = true): two IEEE-half levels per operand, three v_mfma_f32_32x32x16_f16 products --
 * fp32-grade (error against fp64 below smx_conv7_f32's: tests/test_gpu_kernels.py), so the fp32 configuration's 7x7 heads (archs/keypoint_detector_arch.py:60-86,
 * archs/dense_motion_arch.py:118-161) run here for big launches.  wp = smx_conv7_f16_pack(w [N][7][7][Cin]): 16 header bytes + smx_conv7_bf16x3_pack's layout
 * (2 * smx_conv7_bf16x3_pack_elems(Cin, N) + 16 bytes); weights scaled by a power of two on the device, the input by the block.  Arguments as smx_conv7_bf16x3_f32. */
int smx_conv7_f16_pack(const float* w, void* wp, int Cin, int N, void* stream);
int smx_conv7_f16_f32(const float* x, int lda, const void* wp, const float* bias, float* y, int ldc, int B, int H, int W, int Cin,
                      int N, int pad, int act, void* stream);
/* the fp32 configuration's form of the same two heads: exact fp32 products (v_mfma_f32_32x32x2_f32), the same region-direct staging;
 * wp from smx_conv7_f32_pack (smx_conv7_bf16x3_pack_elems(Cin, N) / 2 floats) */
int smx_conv7_f32_pack(const float* w, float* wp, int Cin, int N, void* stream);
int smx_conv7_f32(const float* x, int lda, const float* wp, const float* bias, float* y, int ldc, int B, int H, int W, int Cin,
                  int N, int pad, int act, void* stream);
int smx_gemm_rp_f32_ok(long long M, int N, int K);      /* the fp32 form (csrc/gemm_rp_f32.hip): fp32 MFMA, weights [N/32][K/8][64 lanes][4] */
int smx_gemm_rp_f32_pack(const float* w, int ldw, float* wp, int N, int K, void* stream);
int smx_gemm_rp_f32(const float* a, int lda, const float* wp, const float* bias, const float* res, int ldres, float* c, int ldc,
                    long long M, int N, int K, int act, void* stream);
int smx_gemm_rp_d2s_f32(const float* a, int lda, const float* wp, const float* bias, float* c, int ldc, long long M, int N, int K, int act,
                        int d2s_p, int d2s_c, int Ho, int Wo, void* stream);      /* the un-patchify store, as smx_gemm_rp_d2s_bf16; N % 256 == 0 */
int smx_gemm_rp_bf16_ok(long long M, int N, int K);
int smx_gemm_rp_bf16_pack(const void* w, int ldw, void* wp, int N, int K, void* stream);
int smx_gemm_rp_bf16(const void* a, int lda, const void* wp, const float* bias, const void* res, int ldres, void* c, int ldc,
                     long long M, int N, int K, int act, void* stream);
/* with the un-patchify (depth-to-space) store of the patch Linears (n = (p1 p + p2) d2s_c + ch -> pixel (oy p + p1, ox p + p2), channel ch;
 * c is [B][Ho p][Wo p][ldc >= d2s_c], M = B Ho Wo tokens, d2s_c % 16 == 0): smx_gemm_conv_bf16's d2s_p / d2s_c on the row-panel kernel */
int smx_gemm_rp_d2s_bf16(const void* a, int lda, const void* wp, const float* bias, void* c, int ldc, long long M, int N, int K, int act,
                         int d2s_p, int d2s_c, int Ho, int Wo, void* stream);

/* Fused Winograd F(2x2,3x3) for 3x3 / stride 1 / pad 1 convolutions (same call sites as above for
 * the eligible layers: ResBlock / Upsample / SFT / FFN / RefineFlow 3x3 convs): 2.25x fewer MFMA
 * passes, fp32, input + output transforms fused (nothing transformed touches HBM).
 * x NHWC [B][H][W][lda] (or [B][H/2][W/2][lda] with up2: virtual nearest x2 upsampling),
 * u_packed = G g G^T in fragment order [16][ceil(Cout/32)][Cin/8][64][4] (host: ops.Conv.winograd_u),
 * y NHWC [B][H][W][ldc]; act/bias/res as in smx_gemm_conv_f32.  H%8==0, W%16==0, Cin%32==0. */
int smx_winograd_conv3x3_f32(const float* x, int lda, const float* u_packed, const float* bias,
                             const float* res, int ldres, float* y, int ldc, int B, int H, int W,
                             int Cin, int Cout, int up2, int act, const float* in_ss, int in_swish,
                             float* stats_part, void* stream);
/* The same convolution with the SFT modulation of Fuse_sft_block (archs/appmotioncodebook_arch.py:49-51,
 * `out = dec_feat + w * (dec_feat * scale + shift)`) as its epilogue: x is the shift branch's hidden
 * activation, the convolution IS `shift`, and y = dec + w * (dec * scale + conv(x)) -- the separate
 * smx_sft_combine_f32 pass (4 tensor streams) disappears.  dec / scale NHWC with row strides lddec /
 * ldscale; 16 B-aligned rows, Cout % 4 == 0. */
int smx_winograd_conv3x3_sft_f32(const float* x, int lda, const float* u_packed, const float* bias,
                                 const float* dec, int lddec, const float* scale, int ldscale, float w,
                                 float* y, int ldc, int B, int H, int W, int Cin, int Cout,
                                 float* stats_part, void* stream);
/* smx_gemm_rp_f32 / smx_gemm_rp_d2s_f32 on the BF16 matrix pipe with fp32-grade arithmetic (csrc/gemm_rp_bf3.hip): both operands split exactly three ways
 * into bf16, six MFMA products per multiply (down to 2^-24), fp32 accumulation -- the arithmetic of smx_winograd_bf3_conv3x3_f32; same call sites (the token
 * Linears of /root/reference/basicsr/archs/appmotioncodebook_arch.py:69-70, 101-115 and the other K = 128 / 256 1x1 convolutions), same arguments.
 * wp = smx_gemm_rp_bf3_pack(w [N][ldw] fp32): [N/32][K/16][3 levels][64 lanes][8] bf16, smx_gemm_rp_bf3_pack_bytes bytes.  Shapes (smx_gemm_rp_bf3_ok):
 * M % 32 == 0; K == 256 with N % 128 == 0, or K == 128 with N % 256 == 0; all pointers (bias too) 16 B-aligned, row strides % 4 == 0. */
int smx_gemm_rp_bf3_ok(long long M, int N, int K);
int64_t smx_gemm_rp_bf3_pack_bytes(int N, int K);
int smx_gemm_rp_bf3_pack(const float* w, int ldw, void* wp, int N, int K, void* stream);
int smx_gemm_rp_bf3(const float* a, int lda, const void* wp, const float* bias, const float* res, int ldres, float* c, int ldc,
                    long long M, int N, int K, int act, void* stream);
int smx_gemm_rp_d2s_bf3(const float* a, int lda, const void* wp, const float* bias, float* c, int ldc, long long M, int N, int K, int act,
                        int d2s_p, int d2s_c, int Ho, int Wo, void* stream);
/* smx_gemm_rp_bf3 / smx_gemm_rp_d2s_bf3 in the "f16x3" arithmetic (csrc/gemm_rp_bf3.hip, F16 = true): every operand as two IEEE-half levels, three
 * v_mfma_f32_32x32x16_f16 products per multiply.  The weights are scaled by a power of two at pack time (chosen on the device from max |W|; 16 header bytes in
 * front of smx_gemm_rp_bf3_pack's record layout), every A row by its own power of two inside the kernel: fp32-grade products for inputs of any magnitude
 * (tests/test_gpu_gemm_bf3.py: error against fp64 below the fp32-MFMA kernel's).  Same shapes, arguments and call sites as the bf3 entry points. */
int64_t smx_gemm_rp_f16_pack_bytes(int N, int K);
int smx_gemm_rp_f16_pack(const float* w, int ldw, void* wp, int N, int K, void* stream);
int smx_gemm_rp_f16(const float* a, int lda, const void* wp, const float* bias, const float* res, int ldres, float* c, int ldc,
                    long long M, int N, int K, int act, void* stream);
int smx_gemm_rp_d2s_f16(const float* a, int lda, const void* wp, const float* bias, float* c, int ldc, long long M, int N, int K, int act,
                        int d2s_p, int d2s_c, int Ho, int Wo, void* stream);
/* Host-side (no device work): PNG scanline reconstruction, filter types 0-4 of RFC 2083, 8 bits per sample -- the inner loop of the frame reader either
 * side of the animation loop (reference basicsr/demo.py:166-185 reads the clip; the in-tree codec is synergize_motion_appearance_amd/png.py).  raw: h rows of
 * (1 filter byte + stride bytes) = the inflated IDAT stream; out: h rows of stride bytes; bpp = bytes per pixel.  Called through ctypes it runs
 * without the interpreter lock (driver.LazyFrames decodes on a thread pool). */
int smx_png_unfilter_u8(const uint8_t* raw, int h, int stride, int bpp, uint8_t* out);
/* The same two convolutions on the BF16 matrix pipe with fp32-grade arithmetic ("bf16x6", csrc/winograd_bf3.hip): every fp32 operand is
 * split exactly into three bf16 values (U at pack time, the transformed input in registers) and a product is the six bf16 MFMA products
 * down to 2^-24 -- same call sites (/root/reference/basicsr/archs/vqgan_arch.py:168-191, appmotioncodebook_arch.py:49-51), same
 * arguments and epilogues as smx_winograd_conv3x3_f32 / _sft_f32.  u3 = smx_winograd_bf3_pack(u_packed of the fp32 kernel):
 * [16][Cout/32][Cin/16][3 splits][64 lanes][8] bf16, smx_winograd_bf3_u_bytes bytes.  nprod: 6 (fp32-grade) or 3 (two-way split,
 * ~2^-17, for comparison only).  Shapes: smx_winograd_bf3_shape_ok returns 0 (not eligible) or the M tiles per block the launcher
 * will use -- 1: 8 x 16 pixels x 128 channels (Cout % 128 == 0, H % 8 == 0), 2: 16 x 16 pixels x 64 channels (Cout % 64 == 0, H % 16 == 0);
 * always W % 16 == 0, Cin % 32 == 0, Cin <= 512, every row stride % 4 == 0, all pointers 16 B-aligned.  One 8-wave block per CU: meant
 * for launches of >= 2 x 256 blocks (the host layer keeps smaller launches on smx_winograd_conv3x3_f32).  stats_part chunks are the
 * fp32 kernel's (8 x 16 pixels). */
int64_t smx_winograd_bf3_u_bytes(int Cout, int Cin);
int smx_winograd_bf3_pack(const float* u_packed, void* u3, int Cout, int Cin, void* stream);
/* nprod = 4 of smx_winograd_bf3_conv3x3_f32 ("f16x3"): the same kernel with IEEE-half levels -- every fp32 operand as TWO halves (11 + 11 significand bits), three
 * v_mfma_f32_32x32x16_f16 products per multiply (h1 g1 + h1 g2 + h2 g1; the dropped terms are 2^-22 relative).  U is scaled by a power of two chosen on the device
 * from max |U| so that its second level stays a normal half (the epilogue divides it out, exactly); the transformed input is taken as it is, so the form is
 * normalised, raw inputs get a per-block power-of-two scale (exact rescale of region and accumulators if a later channel slice outgrows it).  ANY C_out: the pack
 * pads U to the 64-channel block width with zero columns, the epilogue masks the ragged quad (C_out % 64 != 0 is an nprod = 4 privilege).  Pack:
 * smx_winograd_f16_pack(u_f32 = ceil(C_out / 32) tiles as smx_pack_winograd_u_f32 writes them) -> 16 header bytes + the record layout of smx_winograd_bf3_pack at the
 * padded width, smx_winograd_f16_u_bytes bytes; pass it as `u3` with nprod = 4. */
int64_t smx_winograd_f16_u_bytes(int Cout, int Cin);
int smx_winograd_f16_pack(const float* u_f32, void* up, int Cout, int Cin, void* stream);
int smx_winograd_bf3_shape_ok(int B, int H, int W, int Cin, int Cout, int lda, int ldc, int ldres, int ldmul);
int smx_winograd_bf3_conv3x3_f32(const float* x, int lda, const void* u3, const float* bias, const float* res, int ldres, float* y, int ldc,
                                 int B, int H, int W, int Cin, int Cout, int up2, int act, const float* in_ss, int in_swish,
                                 float* stats_part, int nprod, void* stream);
int smx_winograd_bf3_conv3x3_sft_f32(const float* x, int lda, const void* u3, const float* bias, const float* dec, int lddec,
                                     const float* scale, int ldscale, float w, float* y, int ldc, int B, int H, int W, int Cin, int Cout,
                                     float* stats_part, int nprod, void* stream);
/* in_ss != NULL: the GroupNorm(+swish, if in_swish) that precedes the conv in ResBlock
 * (archs/vqgan_arch.py:183-188) is applied by the region loader: x*in_ss[b][c][0] + in_ss[b][c][1],
 * with in_ss from smx_groupnorm_stats_f32 -- the separate normalise read+write pass disappears.
 * stats_part != NULL: the epilogue also emits, per block of 8x16 output pixels, the per-channel
 * {mean, M2 = sum (v - mean)^2} (Welford form, 128 values) of the values it stores: [B][(H/8)*(W/16)][Cout][2] floats -- the statistics pass
 * of the NEXT GroupNorm (ResBlock norm2 / the following block's norm1) reduces to
 * smx_groupnorm_finalize_f32 over these partials, the activation is not re-read. */

/* ---------------------------------------------------------------------------------------
 * GroupNorm(32 groups, eps) [+ swish] on NHWC.   normalize/swish archs/vqgan_arch.py:14-20.
 * Two launches inside: per-(b,chunk,c) partial moments -> per-(b,c) scale/shift -> apply.
 * ws: workspace of smx_groupnorm_ws_floats(B,HW,C) floats.
 * ------------------------------------------------------------------------------------- */
int64_t smx_groupnorm_ws_floats(int B, int HW, int C);
int smx_groupnorm_swish_nhwc_f32(const float* x, int ldx, const float* gamma, const float* beta,
                                 float* y, int ldy, int B, int HW, int C, int groups, float eps,
                                 int swish, float* ws, void* stream);

/* the two halves of the above: per-(b,c) {scale, shift} = {rstd*gamma, beta - mean*rstd*gamma} into
 * ss[B][C][2] (ws: B*nchunks*C*2 floats, smx_groupnorm_ws_floats is enough), and the apply pass */
int smx_groupnorm_stats_f32(const float* x, int ldx, const float* gamma, const float* beta, float* ss,
                            int B, int HW, int C, int groups, float eps, float* ws, void* stream);
/* 3x3 / stride 1 / pad 1 convolution with Cout <= 4 on the vector ALUs (Generator conv_out 64->3,
 * archs/vqgan_arch.py:339-342; RefineFlow conv2|convo2 256->3, archs/appmotioncodebook_arch.py:150-167):
 * HBM-shaped layers that would waste 10x MFMA passes on N padding.  x NHWC [B][H][W][lda], w [Cout][3][3][Cin],
 * Cin in {64,128,256}; y NHWC [B][H][W][ldc]; in_ss / in_swish as in smx_winograd_conv3x3_f32. */
int smx_conv3x3_smalln_f32(const float* x, int lda, const float* w, const float* bias, float* y, int ldc,
                           int B, int H, int W, int Cin, int Cout, int act, const float* in_ss, int in_swish,
                           void* stream);
/* partials [B][nch][C][2] ({mean, M2} per channel over nch equal disjoint pixel chunks covering the image; from
 * smx_winograd_conv3x3_f32(stats_part) or the first pass of smx_groupnorm_stats_f32) -> ss [B][C][2] */
int smx_groupnorm_finalize_f32(const float* part, const float* gamma, const float* beta, float* ss,
                               int B, int HW, int C, int groups, int nch, float eps, void* stream);
int smx_groupnorm_apply_f32(const float* x, int ldx, const float* ss, float* y, int ldy, int B, int HW, int C,
                            int swish, void* stream);

/* LayerNorm over E (eps) on tokens [T][E]; also writes y_pos = LN(x) + pos[t % npos] when
 * y_pos != NULL.  TransformerLayer.norm1/2/3 + with_pos_embed (appmotioncodebook_arch.py:97-119). */
int smx_layernorm_pos_f32(const float* x, const float* gamma, const float* beta, const float* pos,
                          float* y, float* y_pos, int T, int E, int npos, float eps, void* stream);

/* Fused multi-head attention o = softmax(scale * q k^T [+ key mask]) v, online softmax, no score
 * tensor in HBM.  nn.MultiheadAttention core as used by TransformerLayer
 * (archs/appmotioncodebook_arch.py:101-115).  q rows [L] at stride ldq, head h at column h*dh;
 * k, v rows [S] likewise; *_bs = batch stride in elements (0 = shared codebook K/V);
 * key_mask uint8 [B][S] (1 = masked) or NULL.  dh = 32/64: fp32 MFMA; dh = 4: VALU. */
int smx_attention_f32(const float* q, int ldq, int64_t q_bs, const float* k, int ldk, int64_t k_bs,
                      const float* v, int ldv, int64_t v_bs, float* o, int ldo, int64_t o_bs,
                      const uint8_t* key_mask, int B, int H, int L, int S, int dh, float scale, void* stream);
/* Which arithmetic smx_attention_f32 takes for a launch: 0 = exact fp32 products on v_mfma_f32_32x32x2_f32 (attn_mfma_kernel), 3 / 2 = fp32-grade products on
 * the bf16 matrix pipe (attn_bf3_kernel: q, k, v and the probabilities split exactly into three bf16 levels, six MFMA products per multiply; 2 = the
 * probabilities on two levels, five products for P V).  d_head 32, S % 64 == 0, >= 512 blocks of 128 queries; tuning knob "attn_bf3" (3 | 2, + 16 = any
 * launch size, 0 = off).  Same call site: nn.MultiheadAttention of /root/reference/basicsr/archs/appmotioncodebook_arch.py:69-70, 101-115. */
int smx_attention_f32_uses_bf3(int B, int H, int L, int S, int dh);

/* Row softmax in place over [R][S] (ld = row stride): softmax(scale * s + mask) with an
 * optional key-padding mask uint8 [R / rows_per_mask][S] (1 = -inf).  F.softmax at
 * vqgan_arch.py:243 and inside nn.MultiheadAttention (appmotioncodebook_arch.py:101-115). */
int smx_softmax_rows_f32(float* s, int ld, int R, int S, float scale, const uint8_t* mask,
                         int rows_per_mask, void* stream);

/* ---------------------------------------------------------------------------------------
 * A7: fused backward warp.  deform_input + occlude_input, appmotioncodebook_arch.py:349-362:
 * flow [B][Hf][Wf][2] bilinear-resized (align_corners=True) to HxW, grid_sample bilinear /
 * zeros / align_corners=True of feat [Bf][H][W][C] (Bf = 1 broadcasts the cached source
 * features over B driving frames), times occ [B][Hf][Wf] resized likewise (NULL = no occlusion).
 * ------------------------------------------------------------------------------------- */
int smx_warp_nhwc_f32(const float* feat, int feat_batch, const float* flow, const float* occ,
                      float* out, int B, int H, int W, int C, int Hf, int Wf, void* stream);

/* bilinear resize, align_corners=True, NHWC slices (F.interpolate at :354,:360,:390,:414,:418,:488,:571,:671) */
int smx_resize_bilinear_ac_nhwc_f32(const float* x, int ldx, float* y, int ldy, int B, int Hin, int Win,
                                    int Hout, int Wout, int C, void* stream);

/* A per-pixel op followed by a bilinear (align_corners=True) down-sampling only has to be evaluated at the 4 taps
 * of each OUTPUT pixel -- exact, the op is per pixel (to_context: relu(conv1x1(warp)) on 256x256 kept at 64x64,
 * appmotioncodebook_arch.py:416-418: 1/4 of the pixels).  gather: taps[b][oy][ox][tap][C] (dense) <- x at
 * (y0,x0) (y0,x1) (y1,x0) (y1,x1); the caller runs the per-pixel op on the B*Hout*Wout*4 rows; combine: the bilinear
 * blend of the 4 rows of each pixel, in the association of smx_resize_bilinear_ac_nhwc_f32 (equal up to FMA contraction, <= 1 ulp). */
int smx_resize_taps_gather_f32(const float* x, int ldx, float* taps, int B, int Hin, int Win, int Hout, int Wout,
                               int C, void* stream);
int smx_resize_taps_combine_f32(const float* taps, float* y, int ldy, int B, int Hin, int Win, int Hout, int Wout,
                                int C, void* stream);

/* avg_pool2d 2x2 NHWC (DownBlock2d.pool, utils/motion_estimator_util.py:374) */
int smx_avgpool2_nhwc_f32(const float* x, int ldx, float* y, int ldy, int B, int Hin, int Win, int C, void* stream);

/* A1: AntiAliasInterpolation2d (utils/motion_estimator_util.py:636-645): depthwise KxK (zero pad K/2)
 * evaluated only at every `step`-th pixel.  img NCHW [B][C][H][W] -> out NHWC slice. w: [C][K][K]. */
int smx_antialias_down_f32(const float* img_nchw, const float* w, float* out, int ldo, int B, int C,
                           int H, int W, int K, int step, void* stream);

/* A3: KP head.  softmax(logits/T) over HW positions per (b,k), E[grid] and heatmap-weighted
 * jacobian sums (archs/keypoint_detector_arch.py:48-58,66-85).  logits NHWC [B][H][W][ldl]
 * (channels 0..K-1), jac maps NHWC [B][H][W][ldj] (channel 4k+j).  value [B][K][2], jac [B][K][4]. */
int smx_kp_head_f32(const float* logits, int ldl, const float* jmaps, int ldj, float* value, float* jac,
                    int B, int H, int W, int K, float temperature, void* stream);

/* A0: normalize_kp (demo.py:24-44) for B driving frames against ONE source / initial frame:
 * value = (kp_d - kp_d0)*scale + kp_s ; jac = J_d inv(J_d0) J_s  (rel_move / rel_jac as in the reference's
 * use_relative_movement / use_relative_jacobian; scale = sqrt(hull(kp_s)/hull(kp_d0)) from the host, 1 if off).
 * kpd_* [B][K][2|4]; kp0_*, kps_* [K][2|4]. */
int smx_normalize_kp_f32(const float* kpd_value, const float* kpd_jac, const float* kp0_value, const float* kp0_jac,
                         const float* kps_value, const float* kps_jac, float* out_value, float* out_jac,
                         int B, int K, float scale, int rel_move, int rel_jac, void* stream);
/* same, with the hull ratio read from device memory (one float; NaN = no adaptation, i.e. 1): the N>1 path receives it inside
 * the broadcast source state and never brings it to the host. */
int smx_normalize_kp_dscale_f32(const float* kpd_value, const float* kpd_jac, const float* kp0_value, const float* kp0_jac,
                                const float* kps_value, const float* kps_jac, float* out_value, float* out_jac,
                                int B, int K, const float* scale_dev, int rel_move, int rel_jac, void* stream);

/* A4-A6: heatmaps + sparse motions + 16 sparse warps fused (archs/dense_motion_arch.py:65-116).
 * src NHWC [Bs][H][W][3] (Bs = 1 broadcasts); kp value [B][K][2], jacobian [B][K][4].
 * hg_in: NHWC [B][H][W][ldh], channel 4k+0 = heatmap_k (k=0 background = 0), 4k+1..3 = deformed rgb.
 * sparse: [B][K+1][H][W][2]; drv_heat: NHWC [B][H][W][K] (driving gaussians). */
int smx_sparse_motion_f32(const float* src, int src_batch, const float* kpd_value, const float* kpd_jac,
                          const float* kps_value, const float* kps_jac, int kps_batch,
                          float* hg_in, int ldh, float* sparse, float* drv_heat,
                          int B, int H, int W, int K, float kp_variance, void* stream);

/* A6b: softmax over the K+1 mask logits and deformation = sum_k mask_k * T_k
 * (archs/dense_motion_arch.py:134-140). mask_logits NHWC [B][H][W][ldm]; deformation [B][H][W][2];
 * mask_out NHWC [B][H][W][K+1] or NULL.  occ_out != NULL: channel K1 of the logits buffer holds the
 * occlusion logit (mask and occlusion heads run as one stacked 7x7 conv) and
 * occ_out[B][H][W] = sigmoid(logit)  (dense_motion_arch.py:158). */
int smx_mask_deformation_f32(const float* mask_logits, int ldm, const float* sparse, float* deformation,
                             float* mask_out, float* occ_out, int B, int H, int W, int K1, void* stream);

/* ---------------------------------------------------------------------------------------
 * small fused elementwise stages of AppMotionCompFormer.forward
 * ------------------------------------------------------------------------------------- */
/* flow_res = (flow - grid) * (H-1)/2 ; grid = make_coordinate_grid (appmotioncodebook_arch.py:562-577) */
int smx_flow_to_residual_f32(const float* flow, float* res, int B, int H, int W, void* stream);
/* m_com = flow + r[...,0:2]/half ; res_norm = r[...,0:2]/half ; occ_out = sigmoid(occ_prev + r[...,2])
 * (:590-601, :689-710).  r: NHWC [B][H][W][3]. */
int smx_flow_occ_update_f32(const float* flow, const float* r, const float* occ_prev, float* m_com,
                            float* res_norm, float* occ_out, int B, int H, int W, void* stream);
/* motion_ignore[b][n] = any(|flow32|>1) with flow bilinear-resized (ac=True) to 32x32 (:487-492) */
int smx_motion_ignore_f32(const float* flow, uint8_t* ignore, int B, int Hf, int Wf, int Ht, int Wt, void* stream);
/* out = dec + w*(dec*scale + shift)  (Fuse_sft_block.forward :50-51); P pixels x C channels, dec may be a
 * channel slice (row stride ld_dec floats) of the [enc|dec] concat buffer, the others are dense */
int smx_sft_combine_f32(const float* dec, int ld_dec, const float* scale, const float* shift, float* out, float w,
                        int64_t P, int C, void* stream);
/* content fingerprint of n floats -> out16 = two uint64 (8-byte aligned): order-independent sums of per-element 64-bit
 * avalanche hashes of (raw bit pattern, index), so ANY bit change of any element changes the key.
 * The host layer keys its frame-invariant source caches on it (the reference recomputes the source encoding every
 * frame, demo.py:130; a pointer-based key would miss raw-pointer rewrites of a reused buffer) */
int smx_fingerprint_f32(const float* x, int64_t n, void* out16, void* stream);
/* y = a + b (n elements) */
int smx_add_f32(const float* a, const float* b, float* y, int64_t n, void* stream);
/* copy a channel slice: y[.., 0:C] (ld ldy) = x[.., 0:C] (ld ldx) over P pixels */
int smx_copy_slice_f32(const float* x, int ldx, float* y, int ldy, int64_t P, int C, void* stream);

/* layout + A14 */
int smx_nchw_to_nhwc_f32(const float* x, float* y, int ldy, int B, int C, int H, int W, void* stream);
int smx_nhwc_to_nchw_f32(const float* x, int ldx, float* y, int B, int C, int H, int W, void* stream);
/* N3, the input side of demo.py's loop on the device (demo.py:177-185): uint8 HWC frames [B][Hin][Win][3] -> normalised fp32
 * NCHW [B][3][Hout][Wout] = ((resize(frame) / 255) - mean) / std; resize = cv2.INTER_LINEAR geometry with its uint8 rounding
 * (identity when Hin x Win == Hout x Wout); swap_rb = img2tensor's bgr2rgb.  H2D traffic is 1 byte per sample. */
int smx_frames_u8_to_nchw_f32(const uint8_t* x, float* y, int B, int Hin, int Win, int Hout, int Wout, int swap_rb, float mean,
                              float stdv, void* stream);
/* tensor2img (utils/img_util.py:70,93): clamp[lo,hi] -> (x-lo)/(hi-lo)*255 -> round-half-even -> uint8, HWC */
int smx_to_uint8_f32(const float* x, uint8_t* y, int64_t n, float lo, float hi, void* stream);

/* F(4x4,3x3) form of smx_winograd_conv3x3_f32 for the big launches (csrc/winograd43.hip): 36 multiplies per 4x4 output tile and channel
 * (2.25 per output; F(2x2,3x3): 4, direct: 9).  H % 16 == 0, W % 32 == 0, Cin % 16 == 0, Cout % 32 == 0, rows 16-B aligned.
 * u43 = G g G^T (6x6 frequencies) packed [36 f][Cout/32][Cin/8][64 lanes][4] (lane l <-> output channel 32 nt + (l & 31), input channels
 * 8 s + 4 (l >> 5) + 0..3), + 1024 floats of prefetch pad.  in_ss / in_swish: the producing GroupNorm(+swish) folded into the staging pass.
 * stats_part (optional): [B][(H/16)*(W/32)][Cout][2] = {mean, M2} of each stored 16x32-pixel block (smx_groupnorm_finalize_f32, nch = that). */
int smx_winograd43_conv3x3_f32(const float* x, int lda, const float* u43, const float* bias, const float* res, int ldres, float* y, int ldc,
                               int B, int H, int W, int Cin, int Cout, int act, const float* in_ss, int in_swish, float* stats_part, void* stream);

/* ---------------------------------------------------------------------------------------
 * A12: VectorQuantizer.forward (archs/vqgan_arch.py:33-93), fused: d = |z|^2 + |e|^2 - 2 z.e
 * over the first Ks rows, first-minimum argmin, gather, z_q = z + (e - z).
 * z tokens [N][D]; codebook [Ks..][D]; idx int64 [N]; zq [N][D]; dmin [N] (min distance, optional); Ks <= 4096.
 * sq_ws: REQUIRED workspace of smx_vq_ws_floats(N) floats, 16-byte aligned: the code norms |e|^2 (computed once per call by a small
 * kernel of their own, then read by every block), the per-block partials of sqerr, and -- for launches of few tokens (fewer 128-token
 * blocks than half the CUs), where the codebook sweep is split over up to 8 shares per token block and a second kernel folds the
 * shares' (distance, index) pairs and does the gather (tuning knob "vq_split"; same indices / distances / z_q bit for bit) -- the shares.
 * sqerr: one float, = sum (zq-z)^2 (optional; written, not accumulated), computed WITHOUT atomics: the per-block partials are
 * summed in a fixed order -> the codebook loss is bit-reproducible.  z is read from HBM once (the gather epilogue reuses the
 * MFMA operand fragments).  z, codebook, zq 16-byte aligned.
 * ------------------------------------------------------------------------------------- */
int64_t smx_vq_ws_floats(int N);
int smx_vq_nearest_f32(const float* z, const float* codebook, int64_t* idx, float* zq, float* dmin,
                       float* sqerr, float* sq_ws, int N, int D, int Ks, void* stream);

/* ---------------------------------------------------------------------------------------
 * BASELINE configs[2]: bf16 STORAGE variants (activation pointers are raw 16-bit bfloat16, void* here; every
 * statistic, coordinate, weight of a normalisation and accumulator stays fp32).  Same semantics, argument meaning
 * and reference call sites as the _f32 entry point of the same name.
 * ------------------------------------------------------------------------------------- */
/* 3x3 / stride 1 / pad 1 convolution on the bf16 MFMA, region-direct (the configs[2] counterpart of smx_winograd_conv3x3_f32, same
 * call sites): x bf16 NHWC [B][H][W][lda] ([B][H/2][W/2][lda] with up2), w bf16 [Cout][ldw >= 9*Cin] (k = (ky*3+kx)*Cin + c),
 * y bf16 NHWC; bias fp32; res bf16 or (res_f32) fp32; in_ss / in_swish / stats_part as in the Winograd entry point, with
 * stats_part = {mean, M2} of the STORED (bf16-rounded) values per tile_h x 16-pixel tile: [B][(H/tile_h)*(W/16)][Cout][2].
 * tile_h = 16 (2 workgroups per CU) or 8 (3 per CU: more loads in flight for the small-C_in layers); H % tile_h == 0,
 * W % 16 == 0, Cin % 64 == 0. */
int smx_conv3x3_bf16(const void* x, int lda, const void* w, int ldw, const float* bias, const void* res, int res_f32, int ldres,
                     void* y, int ldc, int B, int H, int W, int Cin, int Cout, int up2, int act, const float* in_ss, int in_swish,
                     float* stats_part, int tile_h, void* stream);
/* The big-launch form of smx_conv3x3_bf16 (csrc/conv3x3_bf16_t32.hip; same call sites, same semantics): 16 x 32-pixel output tiles x 64
 * channels per workgroup, 16-channel slices through a double-buffered LDS stage, weights by LDS-DMA from a FRAGMENT-ORDERED pack
 * `wp` = [ceil(Cout/64)][Cin/16][9 taps][2][64 lanes][8] bf16 built once per layer by smx_conv3x3_bf16_t32_pack from the
 * [Cout][ldw >= 9*Cin] layout smx_conv3x3_bf16 takes (smx_conv3x3_bf16_t32_pack_elems = its element count, -1 on bad arguments).
 * H % 16 == 0, W % 32 == 0, Cin % 16 == 0; stats_part = {mean, M2} per 16 x 32-pixel tile: [B][(H/16)*(W/32)][Cout][2]. */
long long smx_conv3x3_bf16_t32_pack_elems(int Cin, int Cout);
int smx_conv3x3_bf16_t32_pack(const void* w, int ldw, void* wp, int Cin, int Cout, void* stream);
int smx_conv3x3_bf16_t32(const void* x, int lda, const void* wp, const float* bias, const void* res, int res_f32, int ldres, void* y, int ldc,
                         int B, int H, int W, int Cin, int Cout, int up2, int act, const float* in_ss, int in_swish, float* stats_part,
                         void* stream);
int smx_conv3x3_sft_bf16_t32(const void* x, int lda, const void* wp, const float* bias, const void* dec, int lddec, const void* scale,
                             int ldscale, float sft_w, void* y, int ldc, int B, int H, int W, int Cin, int Cout, void* stream);
/* The bf16 counterpart of smx_winograd_conv3x3_sft_f32: y = dec + w * (dec * scale + conv3x3(x)) as the convolution's epilogue
 * (Fuse_sft_block, archs/appmotioncodebook_arch.py:49-51); dec / scale / y bf16 NHWC with 16 B-aligned rows, Cout % 8 == 0. */
int smx_conv3x3_sft_bf16(const void* x, int lda, const void* w, int ldw, const float* bias, const void* dec, int lddec,
                         const void* scale, int ldscale, float sft_w, void* y, int ldc, int B, int H, int W, int Cin, int Cout,
                         int tile_h, void* stream);
/* the same region-direct kernel on fp32 STORAGE (x, res, y fp32; w the bf16 [Cout][9*Cin] pack): inputs rounded to bf16 (RNE) while
 * staged, fp32 accumulate and output -- torch.autocast(bfloat16)'s arithmetic for F.conv2d; the forward and data gradient of the
 * 3x3 / stride 1 / pad 1 convolutions in the bf16-compute training mode (BASELINE configs[4]).  Cin % 64 == 0, H % tile_h == 0
 * (tile_h 8 | 16), W % 16 == 0; no fused GroupNorm loader / statistics in this form. */
int smx_conv3x3_mfma16_f32(const float* x, int lda, const void* w, int ldw, const float* bias, const float* res, int ldres,
                           float* y, int ldc, int B, int H, int W, int Cin, int Cout, int up2, int act, int tile_h, void* stream);
int smx_groupnorm_swish_nhwc_bf16(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, int B, int HW,
                                  int C, int groups, float eps, int swish, float* ws, void* stream);
int smx_groupnorm_stats_bf16(const void* x, int ldx, const float* gamma, const float* beta, float* ss, int B, int HW, int C,
                             int groups, float eps, float* ws, void* stream);
int smx_groupnorm_apply_bf16(const void* x, int ldx, const float* ss, void* y, int ldy, int B, int HW, int C, int swish, void* stream);
int smx_layernorm_pos_bf16(const void* x, const float* gamma, const float* beta, const float* pos, void* y, void* y_pos, int T, int E,
                           int npos, float eps, void* stream);
int smx_attention_bf16(const void* q, int ldq, int64_t q_bs, const void* k, int ldk, int64_t k_bs, const void* v, int ldv, int64_t v_bs,
                       void* o, int ldo, int64_t o_bs, const uint8_t* key_mask, int B, int H, int L, int S, int dh, float scale, void* stream);
/* The vqgan AttnBlock core (archs/vqgan_arch.py:229-253) on bf16 storage as ONE kernel: o[b][l][:] = softmax_s(q[b][l].k[b][s] * scale) v[b][s],
 * one head of d = 256; `vt` is V TRANSPOSED ([d][ldvt >= S] per image, what the value projection writes with its per-row bias).  The
 * [B][L][S] score tensor of the three-launch form (QK^T GEMM, softmax_rows, PV GEMM) never exists.  L % 128 == 0, S % 32 == 0. */
int smx_attnblock_bf16(const void* q, int ldq, int64_t q_bs, const void* k, int ldk, int64_t k_bs, const void* vt, int ldvt, int64_t vt_bs,
                       void* o, int ldo, int64_t o_bs, int B, int L, int S, int d, float scale, void* stream);
/* The same AttnBlock core in the fp32 configuration (archs/vqgan_arch.py:229-253 in the reference's own arithmetic): fp32 storage, exact fp32
 * products, one kernel -- replaces the QK^T GEMM + softmax_rows + PV GEMM launches and their [B][L][S] fp32 score tensor.  Same contract as
 * the bf16 entry (d = 256, `vt` = V transposed, L % 128 == 0, S % 32 == 0), rows 16-B aligned. */
int smx_attnblock_f32(const float* q, int ldq, int64_t q_bs, const float* k, int ldk, int64_t k_bs, const float* vt, int ldvt, int64_t vt_bs,
                      float* o, int ldo, int64_t o_bs, int B, int L, int S, int d, float scale, void* stream);
int smx_softmax_rows_bf16(void* s, int ld, int R, int S, float scale, const uint8_t* mask, int rows_per_mask, void* stream);
int smx_warp_nhwc_bf16(const void* feat, int feat_batch, const float* flow, const float* occ, void* out, int B, int H, int W, int C,
                       int Hf, int Wf, void* stream);
int smx_resize_bilinear_ac_nhwc_bf16(const void* x, int ldx, void* y, int ldy, int B, int Hin, int Win, int Hout, int Wout, int C, void* stream);
int smx_resize_taps_gather_bf16(const void* x, int ldx, void* taps, int B, int Hin, int Win, int Hout, int Wout, int C, void* stream);
int smx_resize_taps_combine_bf16(const void* taps, void* y, int ldy, int B, int Hin, int Win, int Hout, int Wout, int C, void* stream);
int smx_conv3x3_smalln_bf16(const void* x, int lda, const float* w, const float* bias, float* y, int ldc, int B, int H, int W, int Cin,
                            int Cout, int act, const float* in_ss, int in_swish, void* stream);
/* the same layers (C_out <= 4, 3x3 / s1 / p1, bf16 input, fp32 output, fused GroupNorm(+swish) loader) on the bf16 MFMA with one 32-wide N
 * tile (csrc/conv3x3_smalln_mfma16.hip): region-direct, weights bf16 in the fragment-ordered pack of smx_conv3x3_smalln_mfma_pack
 * (w [N][3][3][Cin] fp32 -> smx_conv3x3_smalln_mfma_pack_elems bf16 elements).  Cin % 64 == 0, H % 8 == 0, W % 32 == 0. */
long long smx_conv3x3_smalln_mfma_pack_elems(int Cin, int N);
int smx_conv3x3_smalln_mfma_pack(const float* w, void* wp, int Cin, int N, void* stream);
int smx_conv3x3_smalln_mfma_bf16(const void* x, int lda, const void* wp, const float* bias, float* y, int ldc, int B, int H, int W, int Cin,
                                 int Cout, int act, const float* in_ss, int in_swish, void* stream);   /* fp32 weights / bias / output */
int smx_sft_combine_bf16(const void* dec, int ld_dec, const void* scale, const void* shift, void* out, float w, int64_t P, int C, void* stream);
int smx_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);
/* copy / convert a channel slice between storage types (dtype codes: 0 = fp32, 1 = bf16) */
int smx_convert_slice(const void* x, int x_dtype, int ldx, void* y, int y_dtype, int ldy, int64_t P, int C, void* stream);
int smx_nchw_to_nhwc_bf16(const float* x, void* y, int ldy, int B, int C, int H, int W, void* stream);   /* fp32 NCHW -> bf16 NHWC */
int smx_nhwc_to_nchw_bf16(const void* x, int ldx, float* y, int B, int C, int H, int W, void* stream);   /* bf16 NHWC -> fp32 NCHW */

/* =======================================================================================
 * TRAINING STEP (BASELINE configs[4]): backward kernels of the hot path and the optimiser side.
 * Reference: models/appmotioncomp_model.py:294-434 (optimize_parameters -> l_g_total.backward() + Adam steps + EMA);
 * what torch.autograd dispatches there (convolution_backward, native_group_norm_backward, native_layer_norm_backward,
 * grid_sampler_2d_backward, upsample_bilinear2d_backward, _softmax_backward_data, bmm, index / scatter of the quantiser)
 * is replaced by the entry points below.  All fp32; pointers are device pointers; no allocation, no sync.
 * ===================================================================================== */

/* ---- convolution / Linear / batched GEMM (csrc/train_gemm.hip) ----
 * The DATA gradient needs no kernel of its own: it is a forward convolution of dY with the packed, tap-flipped, transposed
 * weights (smx_pack_weight_f32 mode 1) through smx_gemm_conv_f32 -- `up2 = 2` in the descriptor zero-inserts the input x2,
 * which makes the data gradient of the stride-2 Downsample (archs/vqgan_arch.py:144-153) a stride-1 launch.
 * WEIGHT gradient: dW[co][(ky,kx,ci)] = sum_m dY[m][co] X~[m][(ky,kx,ci)], m over the nb x (B*Ho*Wo) output pixels, X~ the implicit
 * im2col of x (zero pad, stride, up2 = nearest x2).  ws: smx_wgrad_ws_floats(...) floats (also returns the pixel split to pass).
 * out layout 0: OIHW parameter; 1: row-major out[co*ldo + k] (Linear / batched C); 2: transposed out[k*ldo + co].
 * accumulate: out += alpha * dW (a .grad that already holds another call site's contribution), else out = alpha * dW.
 * bias_out (nullable): the BIAS gradient sum_m dY[m][co] comes out of the same pass (the dY tiles stream through the k-tile-0 blocks anyway;
 * their column sums are finished in a fixed order) -- same accumulate / alpha. */
int64_t smx_wgrad_ws_floats(int nb, int M, int Cout, int K, int* msplit_out);
/* workspace / pixel split for a CONVOLUTION's weight gradient: as smx_wgrad_ws_floats, but told the geometry, so that 3x3 / stride 1 /
 * pad 1 layers with 64-multiple channel counts (and 32-multiple output widths) get the split of the region kernel (one block holds all
 * nine taps of a 64 x 64 (co, ci) tile and walks down a 32-pixel strip: csrc/train_wgrad_region.hip).  smx_wgrad_*_f32 pick the kernel
 * from the same geometry and the msplit passed. */
int64_t smx_wgrad_conv_ws_floats(int nb, int M, int Cout, int Cin, int Hin, int Win, int Ho, int Wo, int kh, int kw, int stride,
                                 int pad_t, int pad_l, int up2, int* msplit_out);
int smx_wgrad_f32(const float* dy, int ldy, int64_t dy_bs, const float* x, int ldx, int64_t x_bs, int nb, int M, int Cout,
                  int Hin, int Win, int Cin, int Ho, int Wo, int kh, int kw, int stride, int pad_t, int pad_l, int up2,
                  float* ws, int msplit, float* out, int64_t out_bs, int layout, int ldo, int accumulate, float alpha,
                  float* bias_out, void* stream);
/* the same contraction on v_mfma_f32_32x32x16_bf16: both operands rounded to bfloat16 (RNE) on the way to the matrix cores, fp32
 * accumulate, fp32 result -- torch.autocast(bfloat16)'s arithmetic for the weight gradient of F.conv2d / F.linear (the bf16-compute
 * training mode; the bias gradient still sums the unrounded dY). */
int smx_wgrad_mfma16_f32(const float* dy, int ldy, int64_t dy_bs, const float* x, int ldx, int64_t x_bs, int nb, int M, int Cout,
                         int Hin, int Win, int Cin, int Ho, int Wo, int kh, int kw, int stride, int pad_t, int pad_l, int up2,
                         float* ws, int msplit, float* out, int64_t out_bs, int layout, int ldo, int accumulate, float alpha,
                         float* bias_out, void* stream);
/* out[c] (+)= alpha * sum_p x[p*ld + c]  (bias gradients); ws: smx_colsum_ws_floats(P, C) floats; two fixed-order stages */
/* DEFERRED split reduce.  `accumulate | 2` in smx_wgrad_f32 / smx_wgrad_mfma16_f32 (nb == 1) leaves the raw partials in `ws` and skips
 * the reduce launch; smx_wgrad_reduce_describe fills the item the skipped launch would have been (host side, no launch; `first_block`
 * is the caller's: the running sum of `nblocks` in table order), and ONE smx_wgrad_reduce_batch launch over a device table of such
 * items finishes all of them (bit-identical to the per-layer launches).  Items of one launch must have distinct outputs.  The
 * workspaces must stay untouched between the weight-gradient launch and the batch. */
typedef struct smx_reduce_item {
  const float* ws; float* out; const float* bias_ws; float* bias_out;
  int32_t msplit, Cout, K, Cin, khw, layout, ldo, accumulate;
  float alpha; int32_t kind, first_block, nblocks;
} smx_reduce_item;
int smx_wgrad_reduce_describe(const float* ws, int msplit, float* out, int Cout, int Cin, int kh, int kw, int layout, int ldo,
                              int accumulate, float alpha, float* bias_out, smx_reduce_item* item);
int smx_wgrad_reduce_batch(const smx_reduce_item* items_dev, int n_items, int n_blocks, void* stream);
int64_t smx_colsum_ws_floats(int64_t P, int C);
int smx_colsum_f32(const float* x, int ld, int64_t P, int C, float* ws, float* out, int accumulate, float alpha, void* stream);
int smx_partial_reduce_f32(const float* part, int nchunk, int C, float* out, int accumulate, float alpha, void* stream);
/* parameter (OIHW, or Linear [out][in] with kh = kw = 1) -> mode 0: [Cout][(ky,kx,ci)] (forward operand);
 * mode 1: [Cin][(kh-1-ky, kw-1-kx, co)] (data-gradient operand) */
int smx_pack_weight_f32(const float* w_oihw, float* packed, int Cout, int Cin, int kh, int kw, int mode, void* stream);
/* the same packing, rounded to bfloat16 on the way out (the weight operand of the bf16-compute training mode) */
int smx_pack_weight_bf16(const float* w_oihw, void* packed, int Cout, int Cin, int kh, int kw, int mode, void* stream);
/* OIHW 3x3 parameter -> the fragment-ordered Winograd-domain weights smx_winograd_conv3x3_f32 reads (u: smx_winograd_u_floats(N, C)
 * floats incl. the prefetch pad): mode 0 forward (N = Cout, C = Cin), mode 1 data gradient (N = Cin, C = Cout, taps flipped) --
 * the training step's 3x3 forward and data-gradient convolutions run on the fused Winograd kernel with the current weights */
int64_t smx_winograd_u_floats(int N, int C);
int smx_pack_winograd_u_f32(const float* w_oihw, float* u, int Cout, int Cin, int mode, void* stream);
/* All of a step's packings in one launch.  items_dev: DEVICE array of n_items descriptors sorted by first_block; item i owns blocks
 * [first_block, first_block + ceil(total / 1024)), n_blocks = the sum.  kind: which of the three packings above (same arguments, same
 * results bit for bit); total = output elements (Cout*Cin*kh*kw, or smx_winograd_u_floats(N, C) for SMX_PACK_WINOGRAD_U). */
enum { SMX_PACK_F32 = 0, SMX_PACK_BF16 = 1, SMX_PACK_WINOGRAD_U = 2 };
typedef struct smx_pack_item {
  const float* w;      /* OIHW parameter */
  void* out;
  int64_t total;
  int32_t cout, cin, kh, kw, mode, kind, first_block, reserved;
} smx_pack_item;
int smx_pack_batch(const smx_pack_item* items_dev, int n_items, int n_blocks, void* stream);
/* batched y[g][c][r] = x[g][r][c] */
int smx_transpose_f32(const float* x, int ldx, int64_t x_bs, float* y, int ldy, int64_t y_bs, int nb, int R, int C, void* stream);
/* y = act(x) as its own pass (training keeps the pre-activation of GELU / swish / sigmoid for the backward) */
int smx_act_f32(const float* x, int ldx, float* y, int ldy, int64_t P, int C, int act, void* stream);
/* gx = g * act'(.): ref = the activation's OUTPUT for RELU / LRELU02 / SIGMOID, its INPUT for SWISH / GELU; P x C with row strides */
int smx_act_bwd_f32(const float* g, int ldg, const float* ref, int ldr, float* gx, int ldo, int64_t P, int C, int act, void* stream);
/* y[.., :C] += alpha * x[.., :C] over P pixels (gradient accumulation into channel slices) */
int smx_axpy_slice_f32(const float* x, int ldx, float* y, int ldy, int64_t P, int C, float alpha, void* stream);

/* ---- normalisation / softmax (csrc/norm_softmax.hip) ---- */
/* GroupNorm statistics for training: ss as smx_groupnorm_stats_f32 plus mr [B][groups][2] = {mean, rstd} saved for the backward */
int smx_groupnorm_stats_train_f32(const float* x, int ldx, const float* gamma, const float* beta, float* ss, float* mr,
                                  int B, int HW, int C, int groups, float eps, float* ws, void* stream);
/* z = act(GN(x)) backward (native_group_norm_backward + the swish): dx; dgamma / dbeta ACCUMULATED.
 * ws: smx_groupnorm_ws_floats(B, HW, C) + 2*B*groups floats */
int smx_groupnorm_bwd_f32(const float* x, int ldx, const float* dz, int ldz, const float* ss, const float* mr, const float* gamma,
                          float* dx, int ldo, float* dgamma, float* dbeta, int B, int HW, int C, int groups, int swish, float* ws, void* stream);
/* LayerNorm(+pos) backward: gy / gypos = gradients of LN(x) and LN(x)+pos (either may be null); dgamma / dbeta / dpos ACCUMULATED */
int64_t smx_layernorm_bwd_ws_floats(int T, int E);
int smx_layernorm_bwd_f32(const float* x, const float* gamma, const float* gy, const float* gypos, float* dx, float* dgamma, float* dbeta,
                          float* dpos, int T, int E, int npos, float eps, float* ws, void* stream);
/* out[i] += sum_b g[b*per + i] */
int smx_batch_sum_f32(const float* g, float* out, int B, int64_t per, void* stream);
/* in place on dP: dS = scale * P * (dP - rowsum(dP * P))  (AttnBlock, archs/vqgan_arch.py:242-245) */
int smx_softmax_rows_bwd_f32(const float* P, float* dP, int64_t R, int S, float scale, void* stream);

/* ---- multi-head attention core backward (csrc/train_attn.hip), flash style: P recomputed from q, k and row statistics ----
 * layouts as smx_attention_f32; o / d_o [B][L][H*dh] dense; k_bs = v_bs = 0: context shared by the batch; dq [B][L][H*dh];
 * dk, dv [B][S][H*dh] per sample (shared context: sum them over the batch with smx_batch_sum_f32); stats: B*H*L*3 floats scratch;
 * dh in {4, 32} */
int smx_attention_bwd_f32(const float* q, int ldq, int64_t q_bs, const float* k, int ldk, int64_t k_bs, const float* v, int ldv, int64_t v_bs,
                          const float* o, const float* d_o, const uint8_t* key_mask, float* dq, float* dk, float* dv, float* stats,
                          int B, int H, int L, int S, int dh, float scale, void* stream);

/* ---- HBM-bound stages + optimiser (csrc/train_misc.hip) ---- */
/* A7 backward (grid_sampler_2d_backward fused with the flow / occlusion resize): dfeat += (atomics; pre-zeroed; may be null),
 * gsm [B][H][W][3] = {d gx, d gy, d occ_s} at the feature resolution (may be null; bring to the flow grid with smx_resize_ac_bwd_f32) */
int smx_warp_bwd_f32(const float* feat, int feat_batch, const float* flow, const float* occ, const float* g, float* dfeat, float* gsm,
                     int B, int H, int W, int C, int Hf, int Wf, void* stream);
/* adjoint of smx_resize_bilinear_ac_nhwc_f32: dx [B][Hin][Win][ldx] += (atomics) from g [B][Hout][Wout][ldg] */
int smx_resize_ac_bwd_f32(const float* g, int ldg, float* dx, int ldx, int B, int Hin, int Win, int Hout, int Wout, int C, void* stream);
/* y[b][h][w][(p1*p+p2)*C + c] = x[b][h*p+p1][w*p+p2][c]: the un-patchify store's inverse (x [B][Ho*p][Wo*p][ldx], y dense [B][Ho][Wo][p*p*C]) */
int smx_space_to_depth_f32(const float* x, int ldx, float* y, int B, int Ho, int Wo, int C, int p, void* stream);
/* A12 backward (archs/vqgan_arch.py:60-76): dz = g_zq + g_loss*2*beta/numel*(z-e); dcodebook[idx] += g_loss*2/numel*(e-z) (atomics);
 * g_zq may be null; g_loss: one device float (d total / d this quantiser's loss) or null */
int smx_vq_bwd_f32(const float* z, const float* codebook, const int64_t* idx, const float* g_zq, const float* g_loss, float beta,
                   float* dz, float* dcodebook, int64_t N, int D, void* stream);
/* smx_flow_occ_update_f32 backward: d_flow = g_mcom; d_r = {g_mcom / ((H-1)/2), g_occ*occ*(1-occ)}; d_occ_prev = g_occ*occ*(1-occ) */
int smx_flow_occ_update_bwd_f32(const float* g_mcom, const float* g_occ, const float* occ, float* d_flow, float* d_r, float* d_occ_prev,
                                int B, int H, int W, void* stream);
int smx_sft_combine_bwd_f32(const float* g, const float* dec, int ld_dec, const float* scale, float* d_dec, float* d_scale, float* d_shift,
                            float w, int64_t P, int C, void* stream);
/* out[0] = weight * mean|a - b| (losses/losses.py L1Loss, reduction mean); part: 1024 floats scratch; fixed-order reduction */
int smx_l1_loss_f32(const float* a, const float* b, int64_t n, float weight, float* part, float* out, void* stream);
int smx_l1_loss_bwd_f32(const float* a, const float* b, const float* g_loss, int64_t n, float weight, float* da, int accumulate, void* stream);
/* y = alpha * x (+ y) */
int smx_scale_f32(const float* x, float* y, int64_t n, float alpha, int accumulate, void* stream);
/* one torch.optim.Adam step (t = 1, 2, ...; no amsgrad) over n contiguous fp32 parameters; the gradient is multiplied by gscale first */
int smx_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int t, float gscale, void* stream);
/* ema = decay * ema + (1 - decay) * p (models/sr_model.py model_ema) */
int smx_ema_f32(float* ema, const float* p, int64_t n, float decay, void* stream);

/* ---- motion estimator in training mode (csrc/train_motion.hip) ---- */
/* y = relu?(BatchNorm(x)) with BATCH statistics over the P rows (F.batch_norm(training=True), sync_batchnorm/batchnorm.py:48-53);
 * mr [C][2] = {mean, rstd} kept for the backward; running_mean / running_var (may be null) updated with `momentum`, unbiased variance.
 * ws: smx_batchnorm_ws_floats(P, C) floats */
int64_t smx_batchnorm_ws_floats(int64_t P, int C);
int smx_batchnorm_train_f32(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta, float* mr, float* running_mean,
                            float* running_var, int64_t P, int C, float eps, float momentum, int relu, float* ws, void* stream);
/* backward of y = relu(BN_train(x)): dx; dgamma / dbeta ACCUMULATED */
int smx_batchnorm_train_bwd_f32(const float* x, int ldx, const float* g, int ldg, const float* y, int ldy, const float* mr, const float* gamma,
                                float* dx, int ldo, float* dgamma, float* dbeta, int64_t P, int C, float* ws, void* stream);
int smx_avgpool2_bwd_f32(const float* g, int ldg, float* dx, int ldx, int B, int H, int W, int C, void* stream);
/* A3 backward (archs/keypoint_detector_arch.py:48-86): dvalue [B][K][2], djac [B][K][4] (either may be null)
 * -> dlogits [B][H][W][lddl] (channels 0..K-1), djmaps [B][H][W][lddj] (channel 4k+j) */
int smx_kp_head_bwd_f32(const float* logits, int ldl, const float* jmaps, int ldj, const float* dvalue, const float* djac, float* dlogits, int lddl,
                        float* djmaps, int lddj, int B, int H, int W, int K, float temperature, void* stream);
/* A4-A6 backward (archs/dense_motion_arch.py:65-116): gradients of the hourglass input (g_hg NHWC, ld ldh), the sparse motions (g_sparse
 * [B][K+1][H][W][2], may be null) and the driving heatmaps (g_dheat [B][H][W][K], may be null) -> keypoint gradients (written) */
int smx_sparse_motion_bwd_f32(const float* src, const float* kpd_value, const float* kpd_jac, const float* kps_value, const float* kps_jac,
                              const float* g_hg, int ldh, const float* g_sparse, const float* g_dheat, float* d_kpd_value, float* d_kpd_jac,
                              float* d_kps_value, float* d_kps_jac, int B, int H, int W, int K, float kp_variance, void* stream);
/* A6b backward (archs/dense_motion_arch.py:140-158): g_def [B][H][W][2], g_occ [B][H][W] (either may be null) -> d_logits [B][H][W][lddm]
 * (K1 mask logits + the occlusion logit at channel K1), d_sparse [B][K1][H][W][2] */
int smx_mask_deformation_bwd_f32(const float* mask_logits, int ldm, const float* sparse, const float* g_def, const float* g_occ, float* d_logits,
                                 int lddm, float* d_sparse, int B, int H, int W, int K1, void* stream);
/* Transform.transform_frame (models/appmotioncomp_model.py:50-72): random affine + thin-plate warp of a frame, sampled with reflection
 * padding; x, y NCHW; theta [B][2][3]; control_points [ncp][2], control_params [B][ncp] (both null = affine only) */
int smx_tps_transform_frame_f32(const float* x, float* y, const float* theta, const float* control_points, const float* control_params,
                                int ncp, int B, int C, int H, int W, void* stream);

/* ---- perceptual loss pieces (csrc/train_percep.hip): MultiScalePyramidPerceptualLoss, losses/losses.py:293-387 ----
 * AntiAliasInterpolation2d (:341-387) on NHWC: zero pad K/2, depthwise K x K Gaussian (one kernel w [K][K] for every channel), every
 * `step`-th output kept (Ho = ceil(H / step)); and its adjoint. */
int smx_antialias_nhwc_f32(const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W, int C, int K, int step, void* stream);
int smx_antialias_nhwc_bwd_f32(const float* gy, int ldg, const float* w, float* dx, int ldx, int B, int H, int W, int C, int K, int step, void* stream);
/* 2 x 2 / stride-2 max pooling of the VGG19 feature stack (torchvision cfg "E"; archs/vgg_arch.py:167-210) and its backward: the gradient
 * goes to the first maximum of the window in scan order; x is the pooling INPUT. */
int smx_maxpool2_f32(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, void* stream);
int smx_maxpool2_bwd_f32(const float* x, int ldx, const float* gy, int ldg, float* dx, int ldo, int B, int H, int W, int C, void* stream);
/* y[p][c] = x[p][c] * scale[c] + shift[c] (shift may be NULL): the (x - mean) / std input normalisation (vgg_arch.py:203) and its backward */
int smx_chan_affine_f32(const float* x, int ldx, const float* scale, const float* shift, float* y, int ldy, int64_t P, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMX_H */
