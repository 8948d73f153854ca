// Fused Winograd F(2x2,3x3) convolution with fp32 operands on the BF16 matrix pipe of gfx950 ("bf16x6": three-way split operands).
//
// Why: on gfx950 the fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate and shares its lanes with the VALU
// (profiles/r05_winograd_valu_vs_mfma.txt); v_mfma_f32_32x32x16_bf16 is 16x faster per multiply and co-issues with VALU work.  An fp32
// value splits EXACTLY into three bf16 values x = x1 + x2 + x3 (8 + 8 + 8 significand bits, round-to-nearest at every level), and
//     u * v  =  u1 v1 + (u1 v2 + u2 v1) + (u1 v3 + u2 v2 + u3 v1)  +  O(2^-24 |u v|)
// -- six bf16 products, each exact in the MFMA's fp32 accumulation -- carries the same 2^-24 relative error bound as one rounded fp32
// product: 6 x 32 = 192 pipe cycles per 32x32x16 block of multiplies against 8 x 64 = 512 on the fp32 MFMA (tests/test_gpu_wino_bf3.py
// measures both against fp64: the split form is not less accurate than the fp32-MFMA kernel).  NPROD = 3 keeps only the first three
// products (two-way split, error ~2^-17: reported next to the six-product form, never selected for the fp32 configuration).
//
// The same reference call sites as smx_winograd_conv3x3_f32 (3x3 / s1 / p1 layers of /root/reference/basicsr/archs/vqgan_arch.py:168-191
// and appmotioncodebook_arch.py's Fuse_sft_block), same epilogues (bias / activation / residual / SFT / GroupNorm partials), same fused
// GroupNorm(+swish) loader; U = G g G^T is the fp32 kernel's, split at pack time.
//
// Shape of a block (one per CU: 8 waves, 256 registers each):
//   * 16 x 16 output pixels (8 x 8 Winograd tiles = two MFMA M tiles) x 64 output channels x all 16 frequencies.  What fixes it:
//     the accumulators (16 frequencies x M x N fp32) must fit the register file, U bytes per multiply fall with M (every CU streams U from
//     L2 through its 64 B/clk vector-memory path: 2048 / M B/clk at full matrix rate -> M = 64 tiles = 32 B/clk), and the V split (VALU, per
//     element of V) amortises over N -- so M = N = 64, and a wave owns BOTH M tiles and BOTH N tiles of two frequencies:
//     wave w = (frequency row fi = w >> 1, column pair jh = w & 1), 2 x 2 x 2 accumulator tiles = 128 registers;
//   * per 32-channel slice the raw 18 x 18-pixel region goes global -> LDS by LDS-DMA (no staging registers), is normalised (fused
//     GroupNorm + swish) once per element on its way into the transform layout: [8 channel quads][18 rows][x parity][9] float4, so that the
//     16 lanes of a ds_read_b128 group (tile columns 0-3 or 4-7 x four tile rows) hit 16 distinct bank groups;
//   * a "phase" = one frequency x 16 channels: input transform of the lane's two tiles x 8 channels (its MFMA B fragment) in fp32,
//     three-way split (v_cvt_pk_bf16_f32), then 24 MFMAs (6 products x 2 M tiles x 2 N tiles) on U fragments that came straight from L2 into
//     registers one phase ahead (hand-counted vmcnt: the region DMA of the next slice is in flight under them);
//   * two waves share a SIMD: one transforms / splits while the other multiplies;
//   * epilogue per M tile: every wave writes its two partial column sums to the exchange buffer, one barrier, each thread finishes one
//     2x2 output tile x 4 channels (bias / activation / residual or SFT, float4 stores, Welford partials in the fp32 kernel's chunk format).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>
#include <mutex>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

#ifndef B3_ABL
#define B3_ABL 0      // tools builds only (tools/wino_bf3_{trace,ablate}.sh; WRONG results by design): timing-only ablation mask -- 1 no transform / split, 2 no MFMAs, 4 no region loads,
#endif                // 8 no U requests, 16 no epilogue; 64: s_memtime stamps of the block's phases written over the GroupNorm partials

// Two block shapes, MT x NT = 4 MFMA tiles per frequency and wave either way:
//   MT = 2: 16 x 16 pixels x  64 channels (any C_out % 64 == 0)                         -- region 18 x 18 pixels
//   MT = 1:  8 x 16 pixels x 128 channels (C_out % 128 == 0): the input transform + split, the GroupNorm of the staging and the patch reads are
//            per V element, and here a V element meets twice the output channels -- half the VALU work per MFMA (the split form is VALU-bound:
//            profiles/r06_wino_bf3.txt); the price is twice the U bytes per MFMA from L2
template <int MT> struct B3G {
  static constexpr int RR = MT == 2 ? 18 : 10;           // region rows
  static constexpr int PLANE = RR * 2 * 9 + 2;           // float4 slots per channel-quad plane: RR rows x 2 x-parities x 9 (+2: planes 6 bank groups apart)
  static constexpr int REGION_F = 8 * PLANE * 4;         // floats per region buffer (32 channels)
  static constexpr int REGION_B = REGION_F * 4;          // 41,728 B | 23,296 B
  static constexpr int NT = 4 / MT;
  static constexpr int NSET = MT == 2 ? 3 : 2;           // two-item staging sets per slice (rows 3k + r3, k = 2 set + q)
};
constexpr int B3_SS_B = 512 * 2 * 4;                   // GroupNorm scale / shift of the image (C_in <= 512)
constexpr int B3_XCH_F = 2 * 8 * 32 * 68;              // exchange [2 q][8 waves][32 tiles][64 n, pitch 68]
constexpr int B3_RED_F = 32 * 64 * 2;
constexpr int B3_MAIN_B = 2 * B3G<2>::REGION_B + B3_SS_B;
constexpr int B3_LDS = B3_XCH_F * 4 + B3_RED_F * 4;    // 155,648 B (the main loop's 87,552 B live inside the exchange's bytes)
static_assert(B3_MAIN_B <= B3_XCH_F * 4, "main-loop LDS must fit under the exchange buffer");

struct B3P {
  const float* x; const unsigned char* u; const float* bias; const float* res; float* y; float* stats;
  const float* in_ss; int in_swish;
  int lda, ldc, ldres;
  int B, H, W, Cin, Cout, up2, act;
  int ty, tx;                       // 16 x 16-pixel blocks per image
  int n32, nsteps;                  // Cout / 32, Cin / 16
  int xcd_group;
  const float* mul; int ldmul; float sft_w;
  unsigned u_bytes;
  const float* u_hdr;               // NPROD = 4: {max |U| bits, 1 / (the power of two U was scaled by)} in front of the packed planes
};

__device__ __forceinline__ float b3_act(float v, int act) {
  switch (act) {
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    case SMX_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case SMX_ACT_SWISH: return v / (1.f + expf(-v));
    case SMX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case SMX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// ---- vector memory: plain buffer loads the compiler sees (it counts vmcnt itself: every request of the loop is unconditional and in a fixed order,
// so the wait in front of a phase's MFMAs is exactly "everything up to my fragments", with the two region items requested behind them still in
// flight).  Two earlier forms were measured wrong on the device: region by LDS-DMA (its requests do not retire in order with register loads), and
// hand-counted inline-asm requests with a conditional claim (hipcc copied the not-yet-landed registers in front of the wait on one branch).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 b3_load16(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned unit_bytes) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, unit_bytes, 0));
}

// x -> (hi, mid, lo) bf16, round-to-nearest-even at every level; eight values = one MFMA B fragment per level
template <int NS>
__device__ __forceinline__ void b3_split8(const float (&f)[8], uint4& hi, uint4& mid, uint4& lo) {
  hi = pack8(f);
  float h[8]; unpack8(hi, h);
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = f[e] - h[e];
  mid = pack8(r);
  if (NS == 3) {
    float m[8]; unpack8(mid, m);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = r[e] - m[e];
    lo = pack8(s);
  }
}

// the same split on four values: one v_cvt_pk_bf16_f32 per PAIR and level (cvt2_bf16: a vector conversion -- two scalar ones convert the even element a second
// time to form its fp32 image) -- 22 VALU instructions per four values
__device__ __forceinline__ unsigned b3_cvt2(float a, float b) { return cvt2_bf16(a, b); }
template <int NS, bool F16 = false>
__device__ __forceinline__ void b3_split4(const float (&v)[4], unsigned (&hi)[2], unsigned (&mid)[2], unsigned (&lo)[2]) {
  float r[4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (F16) {                                  // two IEEE-half levels (11 + 11 significand bits): 12 VALU instructions per four values
      hi[q] = cvt2_f16(v[2 * q], v[2 * q + 1]);
      mid[q] = cvt2_f16(v[2 * q] - f16lo(hi[q]), v[2 * q + 1] - f16hi(hi[q]));
      lo[q] = 0u;
      continue;
    }
    hi[q] = b3_cvt2(v[2 * q], v[2 * q + 1]);
    r[2 * q] = v[2 * q] - __uint_as_float(hi[q] << 16);
    r[2 * q + 1] = v[2 * q + 1] - __uint_as_float(hi[q] & 0xffff0000u);
    mid[q] = b3_cvt2(r[2 * q], r[2 * q + 1]);
    if (NS == 3) {
      const float s0 = r[2 * q] - __uint_as_float(mid[q] << 16), s1 = r[2 * q + 1] - __uint_as_float(mid[q] & 0xffff0000u);
      lo[q] = b3_cvt2(s0, s1);
    } else lo[q] = 0u;
  }
}

__device__ __forceinline__ bf16x8 b3_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 b3_frag(f32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f16x8_t b3_fragh(uint4 v) { return __builtin_bit_cast(f16x8_t, v); }
__device__ __forceinline__ f16x8_t b3_fragh(f32x4 v) { return __builtin_bit_cast(f16x8_t, v); }

// tile index (MFMA column 0..31 of an M tile) -> (tile row 0..3, tile column 0..7): columns 0-3 / 4-7 follow the lane groups a ds_read_b128 is
// served in ({0-3,12-15,20-27} and {4-11,16-19,28-31}), so that every group reads four tile rows x four adjacent tile columns
__device__ __forceinline__ int b3_tile_col(int t) { return 4 * (((t >> 2) ^ (t >> 3) ^ (t >> 4)) & 1) + (t & 3); }

template <int NPROD, int MT>
__global__ __launch_bounds__(512, 2) void winograd_bf3_kernel(B3P p) {
  constexpr int NS = NPROD == 6 ? 3 : 2;
  constexpr bool F16 = NPROD == 4;             // NPROD = 4: IEEE-half levels (two of them, three products; U arrives scaled by a power of two: p.u_scale_inv)
  constexpr int NT = B3G<MT>::NT, RR = B3G<MT>::RR, NSET = B3G<MT>::NSET;
  constexpr int B3_PLANE = B3G<MT>::PLANE, B3_REGION_F = B3G<MT>::REGION_F, B3_REGION_B = B3G<MT>::REGION_B;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* smem_b = reinterpret_cast<unsigned char*>(smem);
  float* ss_lds = reinterpret_cast<float*>(smem_b + 2 * B3_REGION_B);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long tstamp[5] = {0, 0, 0, 0, 0};
  auto mark = [&](int k) __attribute__((always_inline)) { if (B3_ABL & 64) tstamp[k] = __builtin_readcyclecounter(); };
  mark(0);
  const int fi = wave >> 1, jh = wave & 1;
  const int hh = lane >> 5, t = lane & 31;
  const int trl = t >> 3, tc = b3_tile_col(t);

  // block -> (spatial block, output block); with xcd_group the output blocks of one spatial block are 8 block ids apart (same XCD, resident together)
  int bid = blockIdx.x, nblk = blockIdx.y;
  if (p.xcd_group) {
    const unsigned L = blockIdx.y * gridDim.x + blockIdx.x, per = 8u * gridDim.y, grp = L / per, r = L - grp * per;
    bid = (int)(grp * 8u + (r & 7u)); nblk = (int)(r >> 3);
  }
  const int bx = bid % p.tx; bid /= p.tx;
  const int by = bid % p.ty; const int img = bid / p.ty;
  const int y0 = by * (8 * MT), x0 = bx * 16;
  const int Hs = p.up2 ? p.H >> 1 : p.H, Ws = p.up2 ? p.W >> 1 : p.W;
  const float* __restrict__ X = p.x + (long long)img * Hs * Ws * p.lda;
  const __amdgpu_buffer_rsrc_t RX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, Hs * Ws * p.lda * 4, 0x00020000);

  // ---- staging: 432 threads = 3 region rows x 18 pixels x 8 channel quads; item k is the row 3k + r3 ----------------------------------
  // The thread's staging geometry is RE-DERIVED from the lane id where it is used (once per slice, ~20 instructions): held in registers
  // across the phases it would be spilled, and a scratch reload is a vector-memory request in the middle of the hand-counted ones.
  const bool interior = y0 >= 1 && y0 + RR - 1 <= p.H && x0 >= 1 && x0 + 17 <= p.W;
  const int loader = p.in_ss ? (p.in_swish ? 2 : 1) : 0;
  // NPROD = 4, raw (not normalised) input: the block scales its input by a power of two sv so that the largest value of the first 32-channel slice lands in (2, 4]
  // -- the second half level of every value within 2^-5 of that maximum stays a normal half, and 12 bits of headroom remain before a half overflows.  A later slice that
  // outgrows the headroom raises a flag (ss_lds[8]); the block then rescales region, accumulators and sv by an exact power of two (dyn_rescale below).
  float sv = 1.f, omax = 0.f;
  bool unscaled = false;                       // the first slice was all zeros: the first slice that is not sets the scale (same path as the rescale)
  const bool dyn = F16 && loader == 0;
  auto stage_geo = [&](int& l, int& r3, int& srx, int& sc4) __attribute__((always_inline)) {
    l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));                                                            // opaque: computed here, not carried
    const int st = wave * 64 + l;
    r3 = (st * 57) >> 13;                                                                   // st / 144 for st < 512; 3 = not a staging thread
    const int rem = st - r3 * 144;
    srx = rem >> 3; sc4 = rem & 7;
  };
  // two items per phase: rows 3k + r3 for k = 2 ph, 2 ph + 1 (ph = 0..2 covers the 18 rows)
  f32x4 sreg[2];
  auto issue_items = [&](f32x4 (&rg)[2], int ph, int c0) __attribute__((always_inline)) {
    int l, r3, srx, sc4; stage_geo(l, r3, srx, sc4);
    r3 = min(r3, 2);                                                                        // a non-staging lane of a staging wave requests a legal address and drops it
    int sxc = min(max(x0 - 1 + srx, 0), p.W - 1);
    if (p.up2) sxc >>= 1;
    const int gcol = sxc * p.lda + sc4 * 4 + c0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int iy = min(max(y0 - 1 + 3 * (2 * ph + q) + r3, 0), p.H - 1);
      if (p.up2) iy >>= 1;
      if (B3_ABL & 4) rg[q] = f32x4{1.f, 2.f, 3.f, 4.f}; else rg[q] = b3_load16(RX, (unsigned)(iy * Ws * p.lda + gcol) * 4u, 0u);
    }
  };
  auto store_items = [&](const f32x4 (&rg)[2], int ph, float* rb, int c0, auto mode, auto inside) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode)::value;
    constexpr bool INSIDE = decltype(inside)::value;
    int l, r3, srx, sc4; stage_geo(l, r3, srx, sc4);
    if (r3 >= 3) return;
    const int woff = (sc4 * B3_PLANE + (r3 * 2 + (srx & 1)) * 9 + (srx >> 1)) * 4;               // + k * 216 floats
    const int six = x0 - 1 + srx;
    const bool x_ok = six >= 0 && six < p.W;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (MODE >= 1) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ss_lds + (c0 + sc4 * 4) * 2), b = *reinterpret_cast<const f32x4*>(ss_lds + (c0 + sc4 * 4) * 2 + 4);
      sc = f32x4{a.x, a.z, b.x, b.z}; sh = f32x4{a.y, a.w, b.y, b.w};
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = 2 * ph + q;
      f32x4 v = rg[q];
      if (F16 && MODE == 0) {
        v *= sv;
        omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
      if (MODE >= 1) {
        v = __builtin_elementwise_fma(v, sc, sh);
        if (MODE == 2) {
          constexpr float L2E = 1.44269504088896340736f;
          f32x4 e = v * (-L2E);
          e = f32x4{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y), __builtin_amdgcn_exp2f(e.z), __builtin_amdgcn_exp2f(e.w)} + 1.f;
          v *= f32x4{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y), __builtin_amdgcn_rcpf(e.z), __builtin_amdgcn_rcpf(e.w)};
        }
      }
      if (!INSIDE) {
        const int iy = y0 - 1 + 3 * k + r3;
        const bool ok = x_ok && iy >= 0 && iy < p.H;
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;                // the convolution's zero padding
      }
      if (3 * k + r3 < RR) *reinterpret_cast<f32x4*>(rb + woff + k * 216) = v;                 // (MT = 1: ten rows -- the fourth item exists for r3 = 0 only)
    }
  };
  auto store_region = [&](const f32x4 (&rg)[2], int ph, int buf, int c0) __attribute__((always_inline)) {
    float* rb = smem + buf * B3_REGION_F;
    if (interior) {
      if (loader == 2) store_items(rg, ph, rb, c0, std::integral_constant<int, 2>{}, std::true_type{});
      else if (loader == 1) store_items(rg, ph, rb, c0, std::integral_constant<int, 1>{}, std::true_type{});
      else store_items(rg, ph, rb, c0, std::integral_constant<int, 0>{}, std::true_type{});
    } else {
      if (loader == 2) store_items(rg, ph, rb, c0, std::integral_constant<int, 2>{}, std::false_type{});
      else if (loader == 1) store_items(rg, ph, rb, c0, std::integral_constant<int, 1>{}, std::false_type{});
      else store_items(rg, ph, rb, c0, std::integral_constant<int, 0>{}, std::false_type{});
    }
  };

  // ---- transform geometry: frequency row fi needs patch rows (ra, rb) and sign sr; column j needs (ca, cb) and sign sc of the same table --
  //   0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3
  const int ra = (fi == 0) ? 0 : ((fi == 2) ? 2 : 1);
  const int rb_ = (fi == 0) ? 2 : ((fi == 1) ? 2 : ((fi == 2) ? 1 : 3));
  const float sr = (fi == 1) ? 1.f : -1.f;
  // lane part of a patch read (floats): channel quad 2 hh of the 16-channel step, tile row trl, tile column tc
  const int lbase = (2 * hh * B3_PLANE + (2 * trl * 2) * 9 + tc) * 4;
  auto col_off = [&](int r, int b) { return ((r * 2 + (b & 1)) * 9 + (b >> 1)) * 4; };     // patch pixel (row r, column b) relative to the tile's first pixel
  int oa[2][2], ob[2][2];                                                                    // [frequency of the wave][row a | b]: WAVE-UNIFORM float offsets (SGPRs)
  float scj[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int j = 2 * jh + f;
    const int ca = (j == 0) ? 0 : ((j == 2) ? 2 : 1);
    const int cb = (j == 0) ? 2 : ((j == 1) ? 2 : ((j == 2) ? 1 : 3));
    scj[f] = (j == 1) ? 1.f : -1.f;
    oa[f][0] = col_off(ra, ca); oa[f][1] = col_off(rb_, ca);
    ob[f][0] = col_off(ra, cb); ob[f][1] = col_off(rb_, cb);
  }

  // ---- U: [16 f][n32][Cin/16][3 splits][64 lanes][8 bf16] ------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t RU = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.u), 0, (int)p.u_bytes, 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned ustep_b = 3u * 1024u;                                                       // bytes per (frequency, n tile, step)
  unsigned ubase[2][NT];                                                                     // [frequency of the wave][n tile]: byte offset of step 0
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int n = 0; n < NT; ++n)
      ubase[f][n] = (unsigned)(((fi * 4 + 2 * jh + f) * p.n32 + nblk * NT + n) * p.nsteps) * ustep_b;
  // ONE set of fragment registers (the register file is the limit: 128 accumulators + 24 + the transform's values): a phase's fragments are
  // requested right after the previous phase's MFMAs were issued and land under this phase's transform + split (and the partner wave's MFMAs)
  f32x4 ur[NT][3];                                                                           // [n tile][split]
  auto request_u = [&](int f, int step) __attribute__((always_inline)) {
    const unsigned so = (unsigned)step * ustep_b;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3)
        if (s3 < NS) { if (B3_ABL & 8) ur[n][s3] = f32x4{1.f, 1.f, 1.f, 1.f}; else ur[n][s3] = b3_load16(RU, lane16, ubase[f][n] + so + (unsigned)s3 * 1024u); }
        else ur[n][s3] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  f32x16 acc[2][MT][NT];                                                                     // [frequency][M tile][N tile]
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][m][n][r] = 0.f;

  // one phase: frequency f of the wave, 16-channel step `sub` of the slice in region buffer rb
  auto phase = [&](int buf, int f, int sub, auto next) __attribute__((always_inline)) {
    unsigned vh[MT][4], vm[MT][4], vl[MT][4];                                                // [M tile][4 dwords = 8 bf16: the lane's MFMA B fragment]
    // four lane addresses per phase (lane part + the wave's row / column offsets + the region buffer), everything else is an immediate
    const float* paa = smem + (lbase + (buf * B3_REGION_F + oa[f][0]));
    const float* pba = smem + (lbase + (buf * B3_REGION_F + oa[f][1]));
    const float* pab = smem + (lbase + (buf * B3_REGION_F + ob[f][0]));
    const float* pbb = smem + (lbase + (buf * B3_REGION_F + ob[f][1]));
    // one channel quad at a time (four patch reads -> four values of V -> their three bf16 levels), with the NEXT quad's reads issued before this quad's
    // arithmetic (an in-order wave that reads and then waits pays the LDS round trip once per quad: four times a phase); fenced, because left alone
    // the scheduler hoists all sixteen reads of the phase to its top (64 registers the kernel does not have)
    constexpr int NQ = 2 * MT;
    f32x4 rd[2][4];
    auto quad_reads = [&](int q, f32x4 (&d)[4]) __attribute__((always_inline)) {
      const int m = q >> 1, e2 = q & 1;
      const int o = (sub * 4 + e2) * B3_PLANE * 4 + m * (4 * 2 * 2 * 9 * 4);
      d[0] = *reinterpret_cast<const f32x4*>(paa + o); d[1] = *reinterpret_cast<const f32x4*>(pba + o);
      d[2] = *reinterpret_cast<const f32x4*>(pab + o); d[3] = *reinterpret_cast<const f32x4*>(pbb + o);
    };
    if (!(B3_ABL & 1)) quad_reads(0, rd[0]);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int m = q >> 1, e2 = q & 1;
      if (B3_ABL & 1) { vh[m][2 * e2] = 0x3f803f80u + m; vh[m][2 * e2 + 1] = 0x3f803f80u + q; vm[m][2 * e2] = 0x3c003c00u; vm[m][2 * e2 + 1] = 0x3c003c00u + q; vl[m][2 * e2] = 0x38003800u; vl[m][2 * e2 + 1] = 0x38003800u + m; continue; }
      if (q + 1 < NQ) quad_reads(q + 1, rd[(q + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 (&d)[4] = rd[q & 1];
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ta = fmaf(sr, d[1][e], d[0][e]), tb = fmaf(sr, d[3][e], d[2][e]);
        v[e] = fmaf(scj[f], tb, ta);
      }
      unsigned h2[2], m2[2], l2[2];
      b3_split4<NS, F16>(v, h2, m2, l2);
      vh[m][2 * e2] = h2[0]; vh[m][2 * e2 + 1] = h2[1]; vm[m][2 * e2] = m2[0]; vm[m][2 * e2 + 1] = m2[1]; vl[m][2 * e2] = l2[0]; vl[m][2 * e2 + 1] = l2[1];
      __builtin_amdgcn_sched_barrier(0);
    }
#ifndef B3_PRIO
#define B3_PRIO 1     // 1: MFMA block at priority 1 (shipped), 0: no priority changes, 2: the transform at priority 1 instead
#endif
    if (B3_PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if (B3_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    // smallest products first; consecutive MFMAs go to different accumulators
    auto prod = [&](const unsigned (&vv)[MT][4], int us) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          if (B3_ABL & 2) acc[f][m][n][0] += __uint_as_float(vv[m][0]) * ur[n][us].x;
          else if (F16) acc[f][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b3_fragh(ur[n][us]), b3_fragh(make_uint4(vv[m][0], vv[m][1], vv[m][2], vv[m][3])), acc[f][m][n], 0, 0, 0);
          else acc[f][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b3_frag(ur[n][us]), b3_frag(make_uint4(vv[m][0], vv[m][1], vv[m][2], vv[m][3])), acc[f][m][n], 0, 0, 0);
    };
    if (NPROD == 6) { prod(vl, 0); prod(vh, 2); prod(vm, 1); }
    prod(vm, 0); prod(vh, 1); prod(vh, 0);
    if (B3_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (B3_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    next();                                                                                 // the next phase's fragments (the MFMAs above have read theirs)
  };

  // ---- prologue ------------------------------------------------------------------------------------------------------------------------
  const int nsl = p.Cin >> 5;
  {
    // one memory round trip, not three: the first region, the first fragments and the GroupNorm scale / shift table are all requested before anything waits
    f32x4 pr[3][2];
#pragma unroll
    for (int ph = 0; ph < NSET; ++ph) issue_items(pr[ph], ph, 0);
    request_u(0, 0);
    if (loader) {
      const float* sp = p.in_ss + (long long)img * p.Cin * 2;
      for (int i = tid; i < p.Cin / 2; i += 512) *reinterpret_cast<float4*>(ss_lds + i * 4) = *reinterpret_cast<const float4*>(sp + i * 4);
    }
    if (dyn) {                                                                              // largest |value| of the first slice's region (every thread holds six items of it)
      float tm = 0.f;
#pragma unroll
      for (int ph = 0; ph < NSET; ++ph)
#pragma unroll
        for (int q = 0; q < 2; ++q) tm = fmaxf(tm, fmaxf(fmaxf(fabsf(pr[ph][q].x), fabsf(pr[ph][q].y)), fmaxf(fabsf(pr[ph][q].z), fabsf(pr[ph][q].w))));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tm = fmaxf(tm, __shfl_xor(tm, o, 64));
      if (lane == 0) ss_lds[wave] = tm;
      if (tid == 0) ss_lds[8] = 0.f;
    }
    __syncthreads();                                                                        // ss_lds complete
    if (dyn) {
      float bm = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) bm = fmaxf(bm, ss_lds[w]);
      sv = (bm > 0.f && bm < 3.0e38f) ? exp2f(2.f - ceilf(log2f(bm))) : 1.f;
      unscaled = !(bm > 0.f);
    }
#pragma unroll
    for (int ph = 0; ph < NSET; ++ph) store_region(pr[ph], ph, 0, 0);
  }
  __syncthreads();

  mark(1);
  // Vector-memory program of a phase p -- every request unconditional, in this order:
  //   transform(p) | MFMAs(p) (wait: fragments of p; the two region items requested behind them may still fly) | request U(p+1) |
  //   normalise + store the region items of phase p-1 (wait: those two; six fragment requests younger) | request the items of phase p.
  // The next slice's region thus trickles in two items per phase (phases 0-2), each with a full phase of latency cover, in eight registers.
  // The last slice requests its own region again (dropped) so that the program, and with it every vmcnt the compiler derives, has ONE shape.
#pragma unroll 1
  for (int s = 0; s < nsl; ++s) {
    const int rb = s & 1;
    const bool more = s + 1 < nsl;
    const int cn = (more ? s + 1 : s) * 32;
#ifdef B3_SKEW
    if (wave >= 4) __builtin_amdgcn_s_sleep(B3_SKEW);                                       // experiment: the second wave of every SIMD half a phase behind the first
#endif
    phase(rb, 0, 0, [&]() __attribute__((always_inline)) { request_u(1, 2 * s); __builtin_amdgcn_sched_barrier(0); issue_items(sreg, 0, cn); });
    phase(rb, 1, 0, [&]() __attribute__((always_inline)) { request_u(0, 2 * s + 1); __builtin_amdgcn_sched_barrier(0);
                                                          if (more) store_region(sreg, 0, rb ^ 1, cn); else asm volatile("" :: "v"(sreg[0]), "v"(sreg[1]));
                                                          __builtin_amdgcn_sched_barrier(0); issue_items(sreg, 1, cn); });
    phase(rb, 0, 1, [&]() __attribute__((always_inline)) { request_u(1, 2 * s + 1); __builtin_amdgcn_sched_barrier(0);
                                                          if (more) store_region(sreg, 1, rb ^ 1, cn); else asm volatile("" :: "v"(sreg[0]), "v"(sreg[1]));
                                                          __builtin_amdgcn_sched_barrier(0); if (NSET > 2) issue_items(sreg, 2, cn); });
    phase(rb, 1, 1, [&]() __attribute__((always_inline)) { request_u(0, more ? 2 * s + 2 : 2 * s + 1); __builtin_amdgcn_sched_barrier(0);
                                                          if (NSET > 2) { if (more) store_region(sreg, 2, rb ^ 1, cn); else asm volatile("" :: "v"(sreg[0]), "v"(sreg[1])); } });
    if (dyn && (omax > 4096.f || (unscaled && omax > 0.f))) ss_lds[8] = 1.f;                                              // (4 x 4096 is still a quarter of the largest half)
    __syncthreads();
    if (dyn && ss_lds[8] != 0.f) {
      // rare: the next slice outgrew the headroom.  Exact power-of-two rescale of everything that carries sv: the region just staged, the accumulators, sv itself
      float tm = omax;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tm = fmaxf(tm, __shfl_xor(tm, o, 64));
      if (lane == 0) ss_lds[wave] = tm;
      __syncthreads();
      float bm = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) bm = fmaxf(bm, ss_lds[w]);
      const float t = (bm > 0.f && bm < 3.0e38f) ? exp2f(2.f - ceilf(log2f(bm))) : 1.f;
      unscaled = unscaled && !(bm > 0.f);
      float* nbuf = smem + (rb ^ 1) * B3_REGION_F;
      for (int i = tid * 4; i < B3_REGION_F; i += 2048) { f32x4 v = *reinterpret_cast<f32x4*>(nbuf + i); v *= t; *reinterpret_cast<f32x4*>(nbuf + i) = v; }
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[f][m][n] *= t;
      sv *= t; omax *= t;
      __syncthreads();
      if (tid == 0) ss_lds[8] = 0.f;
    }
  }
  mark(2);
#pragma unroll
  for (int n = 0; n < NT; ++n) asm volatile("" :: "v"(ur[n][0]), "v"(ur[n][1]), "v"(ur[n][2]));   // the request past the last phase
  if (B3_ABL & 16) { float t0 = 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) t0 += acc[f][m][n][f + 2 * m + 4 * n];
    if (t0 == 123.456f) p.y[0] = 1.f; return; }

  // ---- epilogue: two passes of 32 tiles x 64 channels: pass = M tile (MT = 2) | 64-channel half of the block's 128 (MT = 1) ------------------------------------------------------------------------------------------------------
  float* zb = smem;
  float* red = smem + B3_XCH_F;
  float* __restrict__ Yi = p.y + (long long)img * p.H * p.W * p.ldc;
  const float* __restrict__ Ri = p.res ? p.res + (long long)img * p.H * p.W * p.ldres : nullptr;
  const float* __restrict__ Mi = p.mul ? p.mul + (long long)img * p.H * p.W * p.ldmul : nullptr;
  const int tl = tid >> 4, n4q = (tid & 15) * 4;
  const int tl_row = tl >> 3, tl_col = b3_tile_col(tl);
  const float uinv = F16 ? p.u_hdr[1] / sv : 1.f;                                              // undo the power-of-two scales of U (pack) and of the input (sv): exact                                                  // undo the power-of-two scale of the half-precision U planes (exact)
  const int amode = p.act == SMX_ACT_NONE ? 0 : ((p.act == SMX_ACT_RELU || p.act == SMX_ACT_LRELU02) ? 1 : 2);
  const float slope = p.act == SMX_ACT_RELU ? 0.f : 0.2f;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    constexpr int dummy_ = 0; (void)dummy_;
    const int m = MT == 2 ? pass : 0, nb = MT == 2 ? 0 : 2 * pass;                            // M tile and first N tile of the pass
    const int nq = nblk * (32 * NT) + nb * 32 + n4q;
    // ragged C_out (the pack pads U with zero columns up to the block width): this thread's quad holds nv = 0 .. 4 real channels
    const int nv = min(max(p.Cout - nq, 0), 4);
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) {
      if (nv == 4) bq = *reinterpret_cast<const float4*>(p.bias + nq);
      else { if (nv > 0) bq.x = p.bias[nq]; if (nv > 1) bq.y = p.bias[nq + 1]; if (nv > 2) bq.z = p.bias[nq + 2]; }
    }
    const float bn[4] = {bq.x, bq.y, bq.z, bq.w};
    // residual (and SFT scale) of this thread's 2x2 pixels: requested before the exchange, consumed after the barrier
    const int pix = (y0 + 2 * (4 * m + tl_row)) * p.W + x0 + 2 * tl_col;
    float4 rr[2][2], mm[2][2];
    if (Ri) {
#pragma unroll
      for (int yy = 0; yy < 2; ++yy)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (nv == 4) {
            rr[yy][q] = *reinterpret_cast<const float4*>(Ri + (pix + yy * p.W + q) * p.ldres + nq);
            if (Mi) mm[yy][q] = *reinterpret_cast<const float4*>(Mi + (pix + yy * p.W + q) * p.ldmul + nq);
          } else {
            const float* rp_ = Ri + (pix + yy * p.W + q) * p.ldres + nq;
            rr[yy][q] = make_float4(nv > 0 ? rp_[0] : 0.f, nv > 1 ? rp_[1] : 0.f, nv > 2 ? rp_[2] : 0.f, 0.f);
            if (Mi) { const float* mp_ = Mi + (pix + yy * p.W + q) * p.ldmul + nq; mm[yy][q] = make_float4(nv > 0 ? mp_[0] : 0.f, nv > 1 ? mp_[1] : 0.f, nv > 2 ? mp_[2] : 0.f, 0.f); }
          }
        }
    }
    if (pass > 0) __syncthreads();                                                          // the previous pass's exchange reads are done
    // partial column sums of this wave's two frequencies: Z[0] = M0 + M1 + M2, Z[1] = M1 - M2 - M3
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 q0, q1;
        const f32x16& a0 = acc[0][m][nb + n]; const f32x16& a1 = acc[1][m][nb + n];
        if (jh == 0) {
          q0 = make_float4(a0[4 * g] + a1[4 * g], a0[4 * g + 1] + a1[4 * g + 1], a0[4 * g + 2] + a1[4 * g + 2], a0[4 * g + 3] + a1[4 * g + 3]);
          q1 = make_float4(a1[4 * g], a1[4 * g + 1], a1[4 * g + 2], a1[4 * g + 3]);
        } else {
          q0 = make_float4(a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3]);
          q1 = make_float4(-(a0[4 * g] + a1[4 * g]), -(a0[4 * g + 1] + a1[4 * g + 1]), -(a0[4 * g + 2] + a1[4 * g + 2]), -(a0[4 * g + 3] + a1[4 * g + 3]));
        }
        *reinterpret_cast<float4*>(zb + ((0 * 8 + wave) * 32 + t) * 68 + n * 32 + 8 * g + 4 * hh) = q0;
        *reinterpret_cast<float4*>(zb + ((1 * 8 + wave) * 32 + t) * 68 + n * 32 + 8 * g + 4 * hh) = q1;
      }
    __syncthreads();
    float4 o[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float4 z[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) z[w] = *reinterpret_cast<const float4*>(zb + ((q * 8 + w) * 32 + tl) * 68 + n4q);
      const float zz[4][4] = {{z[0].x + z[1].x, z[0].y + z[1].y, z[0].z + z[1].z, z[0].w + z[1].w}, {z[2].x + z[3].x, z[2].y + z[3].y, z[2].z + z[3].z, z[2].w + z[3].w},
                              {z[4].x + z[5].x, z[4].y + z[5].y, z[4].z + z[5].z, z[4].w + z[5].w}, {z[6].x + z[7].x, z[6].y + z[7].y, z[6].z + z[7].z, z[6].w + z[7].w}};
      float a0[4], a1[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0[e] = F16 ? fmaf(zz[0][e] + zz[1][e] + zz[2][e], uinv, bn[e]) : zz[0][e] + zz[1][e] + zz[2][e] + bn[e];
        a1[e] = F16 ? fmaf(zz[1][e] - zz[2][e] - zz[3][e], uinv, bn[e]) : zz[1][e] - zz[2][e] - zz[3][e] + bn[e];
        if (amode == 1) { a0[e] = fmaxf(a0[e], 0.f) + slope * fminf(a0[e], 0.f); a1[e] = fmaxf(a1[e], 0.f) + slope * fminf(a1[e], 0.f); }
        else if (amode == 2) { a0[e] = b3_act(a0[e], p.act); a1[e] = b3_act(a1[e], p.act); }
      }
      o[0][q] = make_float4(a0[0], a0[1], a0[2], a0[3]);
      o[1][q] = make_float4(a1[0], a1[1], a1[2], a1[3]);
    }
#pragma unroll
    for (int yy = 0; yy < 2; ++yy)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (Mi) {
          const float4 r4 = rr[yy][q], m4 = mm[yy][q];
          o[yy][q] = make_float4(r4.x + p.sft_w * (r4.x * m4.x + o[yy][q].x), r4.y + p.sft_w * (r4.y * m4.y + o[yy][q].y),
                                 r4.z + p.sft_w * (r4.z * m4.z + o[yy][q].z), r4.w + p.sft_w * (r4.w * m4.w + o[yy][q].w));
        } else if (Ri) { o[yy][q].x += rr[yy][q].x; o[yy][q].y += rr[yy][q].y; o[yy][q].z += rr[yy][q].z; o[yy][q].w += rr[yy][q].w; }
        float* yp_ = Yi + (pix + yy * p.W + q) * p.ldc + nq;
        if (nv == 4) *reinterpret_cast<float4*>(yp_) = o[yy][q];
        else { if (nv > 0) yp_[0] = o[yy][q].x; if (nv > 1) yp_[1] = o[yy][q].y; if (nv > 2) yp_[2] = o[yy][q].z; }
      }
    if (p.stats && !(B3_ABL & 64)) {
      // GroupNorm partials of the consumer in the fp32 kernel's chunk format: one {mean, M2} per 8 x 16-pixel chunk (= this M tile) and channel
      const float va4[4][4] = {{o[0][0].x, o[0][0].y, o[0][0].z, o[0][0].w}, {o[1][0].x, o[1][0].y, o[1][0].z, o[1][0].w},
                               {o[0][1].x, o[0][1].y, o[0][1].z, o[0][1].w}, {o[1][1].x, o[1][1].y, o[1][1].z, o[1][1].w}};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s0 = va4[0][e] + va4[1][e], d0 = va4[0][e] - va4[1][e], s1 = va4[2][e] + va4[3][e], d1 = va4[2][e] - va4[3][e], ds = s0 - s1;
        red[(tl * 64 + n4q + e) * 2] = 0.25f * (s0 + s1);
        red[(tl * 64 + n4q + e) * 2 + 1] = 0.5f * (d0 * d0 + d1 * d1) + 0.25f * ds * ds;
      }
      __syncthreads();
      // 32 tile entries {mean, M2; 4 values each} per channel -> one chunk entry, in two levels (Chan's merge with equal counts): every wave folds four
      // entries of each of the 64 channels (all 512 threads busy: the one-level form kept 64 threads in a 64-read serial loop), then 64 threads fold the eight
      float* red2 = zb;                                                   // (the exchange planes are free: every thread read its share before the barrier above)
      {
        float mk[4], m2 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float2 e2 = *reinterpret_cast<const float2*>(red + ((4 * wave + k) * 64 + lane) * 2); mk[k] = e2.x; m2 += e2.y; a4 += e2.x; }
        a4 *= 0.25f;
        float c2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float d = mk[k] - a4; c2 += d * d; }
        *reinterpret_cast<float2*>(red2 + (wave * 64 + lane) * 2) = make_float2(a4, m2 + 4.f * c2);      // 16 values
      }
      __syncthreads();
      if (tid < 64) {
        float mk[8], a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float2 e2 = *reinterpret_cast<const float2*>(red2 + (k * 64 + tid) * 2); mk[k] = e2.x; a += e2.x; b += e2.y; }
        a *= 0.125f;
        float c2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = mk[k] - a; c2 += d * d; }
        b += 16.f * c2;
        const long long chunk = ((long long)img * (p.ty * MT) + by * MT + m) * p.tx + bx;
        const int nch = nblk * (32 * NT) + nb * 32 + tid;
        if (nch < p.Cout) { float* o2 = p.stats + (chunk * p.Cout + nch) * 2; o2[0] = a; o2[1] = b; }
      }
    }
  }
  if ((B3_ABL & 64) && p.stats && lane == 0) {
    mark(3);
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.stats) + ((long long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = tstamp[k];
  }
}

// fp32 fragment-ordered U ([16][n32][Cin/8][64][4], ops.Conv.winograd_u / smx_pack_winograd_u_f32) -> three bf16 planes in the order the
// bf16 MFMA reads them: [16][n32][Cin/16][3][64 lanes][8]; lane l <-> row n = 32 nt + (l & 31), channels 16 s + 8 (l >> 5) + 0..7
__global__ void winograd_bf3_pack_kernel(const float* __restrict__ u32, unsigned char* __restrict__ u3, int n32, int Cin, long long frags) {
  const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);                        // (f, nt, step)
  if (g >= frags) return;
  const int lane = threadIdx.x & 63, row = lane & 31, hh = lane >> 5;
  const int nsteps = Cin / 16;
  const int step = (int)(g % nsteps);
  const long long fn = g / nsteps;                                                           // f * n32 + nt
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int s8 = 2 * step + hh, half = e >> 2, el = e & 3;
    v[e] = u32[((fn * (Cin / 8) + s8) * 64 + row + 32 * half) * 4 + el];
  }
  uint4 hi, mid, lo;
  b3_split8<3>(v, hi, mid, lo);
  unsigned char* o = u3 + g * 3072 + lane * 16;
  *reinterpret_cast<uint4*>(o) = hi; *reinterpret_cast<uint4*>(o + 1024) = mid; *reinterpret_cast<uint4*>(o + 2048) = lo;
  (void)n32;
}

// NPROD = 4 pack: max |U| (bits, by atomicMax) -> the power of two that brings it into [2^11, 2^12) -> two IEEE-half levels of U x that scale in planes 0 / 1 of
// the same record layout; header {max bits, 1 / scale} in the 16 bytes in front of the records
__global__ void winograd_f16_absmax_kernel(const float* __restrict__ u32, long long n, unsigned* __restrict__ hdr) {
  unsigned m = 0u;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = max(m, __float_as_uint(u32[i]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(hdr, m);
}
// n32s = ceil(C_out / 32) tiles in the source, n32 = tiles in the pack (C_out rounded up to the 64-channel block width): the extra tiles are zero
__global__ void winograd_f16_pack_kernel(const float* __restrict__ u32, unsigned char* __restrict__ up, int n32, int n32s, int Cin, long long frags) {
  const float umax = __uint_as_float(reinterpret_cast<const unsigned*>(up)[0]);
  const float su = (umax > 0.f && umax < 3.0e38f) ? exp2f(11.f - floorf(log2f(umax))) : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(up)[1] = 1.f / su;
  const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);                        // (f, nt, step)
  if (g >= frags) return;
  const int lane = threadIdx.x & 63, row = lane & 31, hh = lane >> 5;
  const int nsteps = Cin / 16;
  const int step = (int)(g % nsteps);
  const long long fn = g / nsteps;                                                           // f * n32 + nt (pack)
  const int nt = (int)(fn % n32);
  const long long fns = (fn / n32) * n32s + nt;                                              // f * n32s + nt (source)
  unsigned h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = 2 * q + j, s8 = 2 * step + hh, half = e >> 2, el = e & 3;
      v[j] = nt < n32s ? u32[((fns * (Cin / 8) + s8) * 64 + row + 32 * half) * 4 + el] * su : 0.f;
    }
    h[q] = cvt2_f16(v[0], v[1]);
    l[q] = cvt2_f16(v[0] - f16lo(h[q]), v[1] - f16hi(h[q]));
  }
  unsigned char* o = up + 16 + g * 3072 + lane * 16;
  *reinterpret_cast<uint4*>(o) = make_uint4(h[0], h[1], h[2], h[3]); *reinterpret_cast<uint4*>(o + 1024) = make_uint4(l[0], l[1], l[2], l[3]);
  *reinterpret_cast<uint4*>(o + 2048) = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace

/* The half-precision pack of smx_winograd_bf3_conv3x3_f32's nprod = 4 arithmetic ("f16x3"): U scaled by a power of two (chosen on the device from max |U|) and split
 * into two IEEE-half levels; 16 header bytes + the record layout of smx_winograd_bf3_pack. */
/* (any C_out: the pack pads it to the 64-channel block width with zero columns; u_f32 holds ceil(C_out / 32) tiles as smx_pack_winograd_u_f32 writes them) */
extern "C" int64_t smx_winograd_f16_u_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || Cin % 16) return 0;
  return 16 + 16LL * (2 * ((Cout + 63) / 64)) * (Cin / 16) * 3072;
}

extern "C" int smx_winograd_f16_pack(const float* u_f32, void* up, int Cout, int Cin, void* stream) {
  if (!u_f32 || !up || Cout <= 0 || Cin <= 0 || Cin % 16 || ((uintptr_t)up & 15)) return SMX_EINVAL;
  const int n32s = (Cout + 31) / 32, n32 = 2 * ((Cout + 63) / 64);
  const long long frags = 16LL * n32 * (Cin / 16), n = 16LL * n32s * 32 * Cin;
  SMX_HIP(hipMemsetAsync(up, 0, 16, (hipStream_t)stream));
  int gb = (int)((n + 255) / 256); if (gb > 2048) gb = 2048;
  SMX_LAUNCH(winograd_f16_absmax_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, u_f32, n, (unsigned*)up);
  SMX_LAUNCH(winograd_f16_pack_kernel, dim3((unsigned)((frags + 3) / 4)), dim3(256), 0, (hipStream_t)stream, u_f32, (unsigned char*)up, n32, n32s, Cin, frags);
  return smx_launch_status();
}

extern "C" int64_t smx_winograd_bf3_u_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || Cout % 32 || Cin % 16) return 0;
  return 16LL * (Cout / 32) * (Cin / 16) * 3072;
}

extern "C" int smx_winograd_bf3_pack(const float* u_f32, void* u3, int Cout, int Cin, void* stream) {
  if (!u_f32 || !u3 || Cout <= 0 || Cin <= 0 || Cout % 32 || Cin % 16) return SMX_EINVAL;
  const long long frags = 16LL * (Cout / 32) * (Cin / 16);
  SMX_LAUNCH(winograd_bf3_pack_kernel, dim3((unsigned)((frags + 3) / 4)), dim3(256), 0, (hipStream_t)stream, u_f32, (unsigned char*)u3, Cout / 32, Cin, frags);
  return smx_launch_status();
}

// 0: not eligible; otherwise the M tiles per block the launcher uses: 1 (8 x 16 pixels x 128 channels, C_out % 128 == 0) | 2 (16 x 16 pixels x 64 channels)
// (Cout < 0: -Cout output channels of the f16x3 form, ANY count -- its pack pads U to the 64-channel block width and the epilogue masks the ragged quad)
extern "C" int smx_winograd_bf3_shape_ok(int B, int H, int W, int Cin, int Cout, int lda, int ldc, int ldres, int ldmul) {
  const bool ragged_ok = Cout < 0;
  if (ragged_ok) Cout = 64 * ((-Cout + 63) / 64);
  if (B <= 0 || H % 8 || W % 16 || Cin % 32 || Cin > 512 || Cout <= 0 || Cout % 64 || lda % 4 || ldc % 4 || ldres % 4 || ldmul % 4) return 0;
  const int shape = smx_tune(SMX_TUNE_WINO_BF3_SHAPE);
  const int mt = (Cout % 128 == 0 && shape != 2) ? 1 : 2;
  if (mt == 2 && H % 16) return 0;
  if ((long long)B * (H / 8) * (W / 16) > 2147483647LL) return 0;
  if ((long long)H * W * lda * 4 > 2147483647LL || (long long)H * W * ldc > 2147483647LL || (long long)H * W * ldres > 2147483647LL || (long long)H * W * ldmul > 2147483647LL) return 0;
  if (smx_winograd_bf3_u_bytes(Cout, Cin) > 2147483647LL) return 0;
  return mt;
}

static int winograd_bf3_launch(const float* x, int lda, const void* u3, const float* bias, const float* res, int ldres, const float* mul, int ldmul, float sft_w,
                               float* y, int ldc, int B, int H, int W, int Cin, int Cout, int up2, int act, const float* in_ss, int in_swish,
                               float* stats_part, int nprod, void* stream) {
  if (!x || !u3 || !y || (nprod != 6 && nprod != 3 && nprod != 4)) return SMX_EINVAL;
  const int cpad = nprod == 4 ? 64 * ((Cout + 63) / 64) : Cout;                              // the f16x3 form takes any C_out (its pack is padded to the block width)
  const int mt = smx_winograd_bf3_shape_ok(B, H, W, Cin, nprod == 4 ? -Cout : Cout, lda, ldc, res ? ldres : 0, mul ? ldmul : 0);
  if (!mt) return SMX_EINVAL;
  if (lda < Cin || ldc < Cout || (res && ldres < Cout) || (mul && (!res || ldmul < Cout || act != SMX_ACT_NONE))) return SMX_EINVAL;
  if ((((uintptr_t)x) | ((uintptr_t)u3) | ((uintptr_t)y) | ((uintptr_t)res) | ((uintptr_t)mul) | ((uintptr_t)bias) | ((uintptr_t)in_ss)) & 15) return SMX_EINVAL;
  B3P p;
  p.x = x; p.u = (const unsigned char*)u3; p.bias = bias; p.res = res; p.y = y; p.stats = stats_part; p.in_ss = in_ss; p.in_swish = in_swish;
  p.lda = lda; p.ldc = ldc; p.ldres = res ? ldres : 0; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.up2 = up2 ? 1 : 0; p.act = act;
  p.ty = H / (8 * mt); p.tx = W / 16; p.n32 = cpad / 32; p.nsteps = Cin / 16; p.mul = mul; p.ldmul = mul ? ldmul : 0; p.sft_w = sft_w;
  p.u_bytes = (unsigned)smx_winograd_bf3_u_bytes(cpad, Cin);
  p.u_hdr = nullptr;
  if (nprod == 4) { p.u_hdr = (const float*)u3; p.u = (const unsigned char*)u3 + 16; }       // the half-precision pack carries a 16-byte header
  const long long blocks = (long long)B * p.ty * p.tx;
  const int nby = cpad / (mt == 1 ? 128 : 64);                                             // output blocks per spatial block
  p.xcd_group = (smx_tune(SMX_TUNE_WINO_XCD) != 0 && blocks % 8 == 0 && nby > 1 && blocks * nby <= 0x7fffffffLL) ? 1 : 0;
  {
    const void* ks[6] = {(const void*)(winograd_bf3_kernel<6, 2>), (const void*)(winograd_bf3_kernel<3, 2>), (const void*)(winograd_bf3_kernel<4, 2>),
                         (const void*)(winograd_bf3_kernel<6, 1>), (const void*)(winograd_bf3_kernel<3, 1>), (const void*)(winograd_bf3_kernel<4, 1>)};
    SMX_HIP(smx_max_dynamic_lds(ks[(mt == 1 ? 3 : 0) + (nprod == 6 ? 0 : nprod == 3 ? 1 : 2)], B3_LDS));
  }
  dim3 grid((unsigned)blocks, nby);
  if (mt == 2) {
    if (nprod == 6) SMX_LAUNCH((winograd_bf3_kernel<6, 2>), grid, dim3(512), B3_LDS, (hipStream_t)stream, p);
    else if (nprod == 4) SMX_LAUNCH((winograd_bf3_kernel<4, 2>), grid, dim3(512), B3_LDS, (hipStream_t)stream, p);
    else SMX_LAUNCH((winograd_bf3_kernel<3, 2>), grid, dim3(512), B3_LDS, (hipStream_t)stream, p);
  } else {
    if (nprod == 6) SMX_LAUNCH((winograd_bf3_kernel<6, 1>), grid, dim3(512), B3_LDS, (hipStream_t)stream, p);
    else if (nprod == 4) SMX_LAUNCH((winograd_bf3_kernel<4, 1>), grid, dim3(512), B3_LDS, (hipStream_t)stream, p);
    else SMX_LAUNCH((winograd_bf3_kernel<3, 1>), grid, dim3(512), B3_LDS, (hipStream_t)stream, p);
  }
  return smx_launch_status();
}

extern "C" int smx_winograd_bf3_conv3x3_f32(const float* x, int lda, const void* u3, const float* bias, const float* res, int ldres, float* y, int ldc,
                                            int B, int H, int W, int Cin, int Cout, int up2, int act, const float* in_ss, int in_swish,
                                            float* stats_part, int nprod, void* stream) {
  return winograd_bf3_launch(x, lda, u3, bias, res, ldres, nullptr, 0, 0.f, y, ldc, B, H, W, Cin, Cout, up2, act, in_ss, in_swish, stats_part, nprod, stream);
}

extern "C" int smx_winograd_bf3_conv3x3_sft_f32(const float* x, int lda, const void* u3, const float* bias, const float* dec, int lddec,
                                                const float* scale, int ldscale, float w, float* y, int ldc, int B, int H, int W, int Cin, int Cout,
                                                float* stats_part, int nprod, void* stream) {
  if (!dec || !scale) return SMX_EINVAL;
  return winograd_bf3_launch(x, lda, u3, bias, dec, lddec, scale, ldscale, w, y, ldc, B, H, W, Cin, Cout, 0, SMX_ACT_NONE, nullptr, 0, stats_part, nprod, stream);
}
