// Short-K Linear / 1x1 convolution of the fp32 configuration on the BF16 matrix pipe with fp32-grade arithmetic ("bf16x6"): the row-panel form of
// gemm_rp_f32.hip with both operands split EXACTLY three ways into bf16 (x = x1 + x2 + x3, each level the rounded remainder of the one before) and a product
// taken as the six bf16 MFMA products down to 2^-24 -- the arithmetic of csrc/winograd_bf3.hip (measured there and in tests/test_gpu_gemm_bf3.py: error
// against fp64 not above the fp32-MFMA kernel's), 6 x 32 = 192 matrix-pipe cycles per 32x32x16 block of multiplies against 8 x 64 = 512 on the fp32 MFMA.
// Same call sites as smx_gemm_rp_f32: the token Linears of the transformer layers (nn.MultiheadAttention in / out projections and FFN,
// /root/reference/basicsr/archs/appmotioncodebook_arch.py:69-70, 101-115: M = B x 1024 tokens, K = 256, N = 256 / 512 / 2048 / 4096) and the other K = 128 / 256
// 1x1 convolutions.
//   * a block is PERSISTENT over 32-row tiles of A and owns 128 output columns: 8 waves = 4 column tiles x 2 halves of K (K = 256) or 8 column tiles (K = 128:
//     256 columns); a wave keeps the three bf16 levels of ITS weights -- 32 columns x 128 k -- in 96 registers for its whole life (pack: smx_gemm_rp_bf3_pack);
//   * the A tile is split ONCE per block: global -> registers one tile ahead (every thread K / 128 groups of 8 floats), split there, and stored as three bf16
//     level planes in LDS ([level][row][K], 16-B chunks XOR-swizzled by row), double buffered, under the MFMAs of the tile before -- one barrier per tile.  A
//     wave's step is 3 ds_read_b128 (fetched one step ahead) + 6 MFMAs on two accumulator chains (a filler between two MFMAs on ONE accumulator stalls the
//     pipe).  Measured against splitting in every wave straight from an fp32 LDS tile (the first form of this kernel: 44 VALU instructions per step and wave,
//     redundant across the column tiles): 1.27x / 1.05x became 1.96x / 1.49x over the fp32-MFMA kernel on 256 -> 256 / 256 -> 1024;
//     Also measured and kept out: an alternating schedule (two barriers per tile, the two waves of a SIMD never multiplying at the same time -- one runs its 48
//     MFMAs while the other finishes the previous tile and splits its share of the next): 0.85x on K = 256, 1.15x on K = 128 -- one wave alone does not keep
//     the pipe fed through its ds_read waits, two interleaved ones do;
//   * K = 256: the two K halves of a column tile are the two waves of one SIMD and meet in the epilogue (4.5 KB LDS hand-over, one per tile parity); bias /
//     activation / residual, 16-B stores of whole 128-B lines, optional un-patchify (depth-to-space) store as in gemm_rp_f32.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TM = 32;                         // rows per tile
constexpr int EXP = 36;                        // epilogue exchange pitch (floats)
constexpr int EX_F = 32 * EXP;
constexpr int KSW = 8;                         // 16-deep steps per wave (128 k)

struct RPB {
  const float* a; const unsigned char* wp; const float* bias; const float* res; float* c;
  int lda, ldres, ldc, M, N, K, act, tiles;
  int d2s_p, d2s_c, Ho, Wo;
};

__device__ __forceinline__ float rpb_act(float v, int act) {
  switch (act) {
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    case SMX_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case SMX_ACT_SWISH: return v / (1.f + expf(-v));
    case SMX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case SMX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

__device__ __forceinline__ unsigned rpb_cvt2(float a, float b) { return cvt2_bf16(a, b); }   // one v_cvt_pk_bf16_f32 per pair
// eight fp32 values -> their three bf16 levels (round to nearest even at every level), as three MFMA fragments
__device__ __forceinline__ void rpb_split8(const float (&v)[8], uint4& hi, uint4& mid, uint4& lo) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h[q] = rpb_cvt2(v[2 * q], v[2 * q + 1]);
    const float r0 = v[2 * q] - __uint_as_float(h[q] << 16), r1 = v[2 * q + 1] - __uint_as_float(h[q] & 0xffff0000u);
    m[q] = rpb_cvt2(r0, r1);
    l[q] = rpb_cvt2(r0 - __uint_as_float(m[q] << 16), r1 - __uint_as_float(m[q] & 0xffff0000u));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]); mid = make_uint4(m[0], m[1], m[2], m[3]); lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ bf16x8 rpb_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f16x8_t rpb_fragh(uint4 v) { return __builtin_bit_cast(f16x8_t, v); }

// KSPLIT = K / 128 (1 | 2): waves per column tile; 8 waves per block = 8 / KSPLIT column tiles
// F16: the "f16x3" arithmetic -- two IEEE-half levels per operand, three products (h1 g1 + h1 g2 + h2 g1).  The weights arrive scaled by a power of two (pack header),
// every A row is scaled by its own power of two (its largest |value| into [2, 4): found by the half-wave that stages the row, kept in LDS for the epilogue)
template <int KSPLIT, bool D2S = false, bool F16 = false>
__global__ __launch_bounds__(512, 2) void gemm_rp_bf3_kernel(RPB p) {
  constexpr int NW = 8, NTB = NW / KSPLIT;
  constexpr int K = 128 * KSPLIT;
  constexpr int CPR = K / 8;                   // 16-B chunks (8 bf16) per level row
  constexpr int LROW = K * 2;
  constexpr int LEVEL = TM * LROW;             // 8 KB | 16 KB
  constexpr int LBUF = 3 * LEVEL;
  constexpr int GPT = KSPLIT;                  // 8-float groups per thread and tile (32 x K / 8 / 512)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Ex = reinterpret_cast<float*>(smem + 2 * LBUF);
  float* Rs = Ex + NTB * 2 * EX_F;             // F16: 1 / (row scale), [2 buffers][32 rows]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = wave % NTB, kh = wave / NTB;  // the two K halves of a column tile are waves w and w + 4: one SIMD, so every SIMD carries one finishing wave
  const int n0 = (blockIdx.y * NTB + nt) * 32;

  uint4 wf[KSW][3];
  {
    const uint4* wp = reinterpret_cast<const uint4*>(p.wp + (F16 ? 16 : 0)) + ((long long)(n0 / 32) * (K / 16) + kh * KSW) * 3 * 64 + lane;
#pragma unroll
    for (int i = 0; i < KSW; ++i)
#pragma unroll
      for (int s = 0; s < 3; ++s) wf[i][s] = (!F16 || s < 2) ? wp[(i * 3 + s) * 64] : make_uint4(0u, 0u, 0u, 0u);
  }
  const float winv = F16 ? reinterpret_cast<const float*>(p.wp)[1] : 1.f;
  // staging: group g = tid + 512 j -> row g / CPR, chunk g % CPR
  int gsrc[GPT], gdst[GPT];
#pragma unroll
  for (int j = 0; j < GPT; ++j) {
    const int g = tid + 512 * j, row = g / CPR, c = g % CPR;
    gsrc[j] = row * p.lda + c * 8;
    gdst[j] = row * LROW + ((c ^ (row & 15)) << 4);
  }
  float4 raw[GPT][2];
  auto fetch = [&](int t) {
    const float* base = p.a + (long long)t * TM * p.lda;
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
      raw[j][0] = *reinterpret_cast<const float4*>(base + gsrc[j]);
      raw[j][1] = *reinterpret_cast<const float4*>(base + gsrc[j] + 4);
    }
  };
  auto split_store = [&](int j, int buf) {
    float v[8] = {raw[j][0].x, raw[j][0].y, raw[j][0].z, raw[j][0].w, raw[j][1].x, raw[j][1].y, raw[j][1].z, raw[j][1].w};
    unsigned char* d = smem + buf * LBUF + gdst[j];
    if (F16) {
      // the row's largest |value|: its CPR groups sit in CPR consecutive lanes
      float m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
#pragma unroll
      for (int o = CPR / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      const int e = __builtin_amdgcn_frexp_expf(m);                      // m = f 2^e, f in [0.5, 1) (0 for m = 0): m x 2^(2 - e) in [2, 4)
      const float sc = __builtin_amdgcn_ldexpf(1.f, 2 - e);
      unsigned h[4], l[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a0 = v[2 * q] * sc, a1 = v[2 * q + 1] * sc;
        h[q] = cvt2_f16(a0, a1);
        l[q] = cvt2_f16(a0 - f16lo(h[q]), a1 - f16hi(h[q]));
      }
      *reinterpret_cast<uint4*>(d) = make_uint4(h[0], h[1], h[2], h[3]); *reinterpret_cast<uint4*>(d + LEVEL) = make_uint4(l[0], l[1], l[2], l[3]);
      const int g = tid + 512 * j;
      if (g % CPR == 0) Rs[buf * 32 + g / CPR] = __builtin_amdgcn_ldexpf(1.f, e - 2);
      return;
    }
    uint4 h, m, l;
    rpb_split8(v, h, m, l);
    *reinterpret_cast<uint4*>(d) = h; *reinterpret_cast<uint4*>(d + LEVEL) = m; *reinterpret_cast<uint4*>(d + 2 * LEVEL) = l;
  };
  const int arow = lane & 31, ahalf = lane >> 5;
  const int erow = lane >> 1, eh = lane & 1;
  int d_p1 = 0, d_p2 = 0, d_oc = 0;
  if (D2S) {
    const int nc = n0 + 16 * eh, dq = nc / p.d2s_c;
    d_oc = nc - dq * p.d2s_c; d_p1 = dq / p.d2s_p; d_p2 = dq - d_p1 * p.d2s_p;
  }
  float4 bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + 16 * eh + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int gx = gridDim.x;
  int t = blockIdx.x, it = 0;
#pragma unroll
  for (int j = 0; j < GPT; ++j) raw[j][0] = raw[j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < p.tiles) {
    fetch(t);
#pragma unroll
    for (int j = 0; j < GPT; ++j) split_store(j, 0);
    if (t + gx < p.tiles) fetch(t + gx);
  }
  __syncthreads();
  for (; t < p.tiles; t += gx, ++it) {
    const int buf = it & 1;
    float* ex = Ex + (nt * 2 + buf) * EX_F;
    f32x16 acc, acb;                           // two chains: consecutive MFMAs never share an accumulator (a filler between same-accumulator MFMAs stalls the pipe)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acb[r] = 0.f;
    const unsigned char* ab = smem + buf * LBUF + arow * LROW;
    uint4 xf[2][3];                            // the A fragments one step ahead of their MFMAs
    auto frag = [&](int i, uint4 (&x)[3]) {
      const unsigned char* q = ab + (((2 * (kh * KSW + i) + ahalf) ^ (arow & 15)) << 4);
      x[0] = *reinterpret_cast<const uint4*>(q); x[1] = *reinterpret_cast<const uint4*>(q + LEVEL);
      if (!F16) x[2] = *reinterpret_cast<const uint4*>(q + 2 * LEVEL);
    };
    frag(0, xf[0]);
#pragma unroll
    for (int i = 0; i < KSW; ++i) {
      if (i + 1 < KSW) frag(i + 1, xf[(i + 1) & 1]);
      const uint4 xh = xf[i & 1][0], xm = xf[i & 1][1], xl = xf[i & 1][2];
      if (F16) {                               // three products, the chains alternating from step to step so that no two consecutive MFMAs share an accumulator
        f32x16& c0 = (i & 1) ? acc : acb; f32x16& c1 = (i & 1) ? acb : acc;
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rpb_fragh(wf[i][0]), rpb_fragh(xm), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rpb_fragh(wf[i][1]), rpb_fragh(xh), c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rpb_fragh(wf[i][0]), rpb_fragh(xh), c0, 0, 0, 0);
      } else {
        acb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rpb_frag(wf[i][0]), rpb_frag(xl), acb, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rpb_frag(wf[i][1]), rpb_frag(xh), acc, 0, 0, 0);
        acb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rpb_frag(wf[i][2]), rpb_frag(xh), acb, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rpb_frag(wf[i][0]), rpb_frag(xm), acc, 0, 0, 0);
        acb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rpb_frag(wf[i][1]), rpb_frag(xm), acb, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rpb_frag(wf[i][0]), rpb_frag(xh), acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);       // steps stay in this order: reads of step i + 1 ahead of the MFMAs of step i
      if (GPT == 2 ? (i == 2 || i == 5) : (i == 3)) {
        split_store(GPT == 2 ? (i == 5) : 0, buf ^ 1);      // one block between two steps (which steps does not matter: measured)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const float rsc = F16 ? Rs[buf * 32 + erow] * winv : 1.f;             // this tile's row scale (written a tile ago), read BEFORE the barrier: the next tile's overwrites it after
    if (t + 2 * gx < p.tiles) fetch(t + 2 * gx);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acb[r];                     // the 2^-16-class chain into the leading one
    if (KSPLIT == 2 && kh == 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(ex + arow * EXP + 8 * g + 4 * ahalf) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
    __syncthreads();                           // levels[buf ^ 1] complete, levels[buf] free, hand-over visible
    if (KSPLIT == 2) {
      if (kh == 1) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 o = *reinterpret_cast<const float4*>(ex + arow * EXP + 8 * g + 4 * ahalf);
        acc[4 * g] += o.x; acc[4 * g + 1] += o.y; acc[4 * g + 2] += o.z; acc[4 * g + 3] += o.w;
      }
    }
    const long long grow = (long long)t * TM + erow;
    const int nc = n0 + 16 * eh;
    float4 rq[4];
    if (p.res) {
      const float* rp = p.res + grow * p.ldres + nc;
#pragma unroll
      for (int q = 0; q < 4; ++q) rq[q] = *reinterpret_cast<const float4*>(rp + 4 * q);
    }
    // wave-private transpose through this parity's exchange tile (the hand-over in it has been read by this very wave; its next writer is two barriers away)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(ex + arow * EXP + 8 * g + 4 * ahalf) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    float* cp = p.c + grow * p.ldc + nc;
    if (D2S) {
      const int hw = p.Ho * p.Wo;
      const int img = (int)(grow / hw), rem = (int)(grow - (long long)img * hw);
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const long long opix = ((long long)img * (p.Ho * p.d2s_p) + oy * p.d2s_p + d_p1) * (p.Wo * p.d2s_p) + ox * p.d2s_p + d_p2;
      cp = p.c + opix * p.ldc + d_oc;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 x = *reinterpret_cast<const float4*>(ex + erow * EXP + 16 * eh + 4 * q);
      if (F16) { x.x = fmaf(x.x, rsc, bq[q].x); x.y = fmaf(x.y, rsc, bq[q].y); x.z = fmaf(x.z, rsc, bq[q].z); x.w = fmaf(x.w, rsc, bq[q].w); }
      else { x.x += bq[q].x; x.y += bq[q].y; x.z += bq[q].z; x.w += bq[q].w; }
      if (p.act != SMX_ACT_NONE) { x.x = rpb_act(x.x, p.act); x.y = rpb_act(x.y, p.act); x.z = rpb_act(x.z, p.act); x.w = rpb_act(x.w, p.act); }
      if (p.res) { x.x += rq[q].x; x.y += rq[q].y; x.z += rq[q].z; x.w += rq[q].w; }
      *reinterpret_cast<float4*>(cp + 4 * q) = x;
    }
  }
}

// W [N][ldw] (row n: K contiguous floats) -> [N/32][K/16][3 levels][64 lanes][8 bf16]: lane l of (n-tile, step i) holds row 32 nt + (l & 31),
// k = 16 i + 8 (l >> 5) + 0..7
__global__ __launch_bounds__(256) void gemm_rp_bf3_pack_kernel(const float* __restrict__ w, int ldw, uint4* __restrict__ wp, int N, int K) {
  const int KS = K / 16;
  const long long total = (long long)(N / 32) * KS * 64;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    const long long f = i >> 6;
    const int g = (int)(f % KS), nt = (int)(f / KS);
    const float* src = w + (long long)(nt * 32 + (lane & 31)) * ldw + g * 16 + (lane >> 5) * 8;
    const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
    const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    uint4 hi, mid, lo;
    rpb_split8(v, hi, mid, lo);
    uint4* o = wp + (f * 3) * 64 + lane;
    o[0] = hi; o[64] = mid; o[128] = lo;
  }
}

// the f16x3 pack: max |W| (bits, atomicMax) -> the power of two that brings it into [2^11, 2^12) -> two IEEE-half levels of W x that scale in levels 0 / 1 of the
// same record layout; 16 header bytes {max bits, 1 / scale} in front
__global__ void gemm_rp_f16_absmax_kernel(const float* __restrict__ w, int ldw, int N, int K, unsigned* __restrict__ hdr) {
  unsigned m = 0u;
  const long long n = (long long)N * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = max(m, __float_as_uint(w[(i / K) * ldw + i % K]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(hdr, m);
}
__global__ __launch_bounds__(256) void gemm_rp_f16_pack_kernel(const float* __restrict__ w, int ldw, unsigned char* __restrict__ wpb, int N, int K) {
  const float wmax = __uint_as_float(reinterpret_cast<const unsigned*>(wpb)[0]);
  const float su = (wmax > 0.f && wmax < 3.0e38f) ? __builtin_amdgcn_ldexpf(1.f, 12 - __builtin_amdgcn_frexp_expf(wmax)) : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(wpb)[1] = 1.f / su;
  uint4* wp = reinterpret_cast<uint4*>(wpb + 16);
  const int KS = K / 16;
  const long long total = (long long)(N / 32) * KS * 64;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    const long long f = i >> 6;
    const int g = (int)(f % KS), nt = (int)(f / KS);
    const float* src = w + (long long)(nt * 32 + (lane & 31)) * ldw + g * 16 + (lane >> 5) * 8;
    const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
    const float v[8] = {a0.x * su, a0.y * su, a0.z * su, a0.w * su, a1.x * su, a1.y * su, a1.z * su, a1.w * su};
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { h[q] = cvt2_f16(v[2 * q], v[2 * q + 1]); l[q] = cvt2_f16(v[2 * q] - f16lo(h[q]), v[2 * q + 1] - f16hi(h[q])); }
    uint4* o = wp + (f * 3) * 64 + lane;
    o[0] = make_uint4(h[0], h[1], h[2], h[3]); o[64] = make_uint4(l[0], l[1], l[2], l[3]); o[128] = make_uint4(0u, 0u, 0u, 0u);
  }
}

template <int KSPLIT, bool D2S = false, bool F16 = false>
int rpb_launch(const RPB& p, hipStream_t st) {
  constexpr int LDS = 2 * 3 * TM * 128 * KSPLIT * 2 + (8 / KSPLIT) * 2 * EX_F * 4 + 256;  // 133 KB (K = 256) | 122 KB (K = 128): one block per CU
  SMX_HIP(smx_max_dynamic_lds((const void*)gemm_rp_bf3_kernel<KSPLIT, D2S, F16>, LDS));
  const int ny = p.N / (32 * (8 / KSPLIT));
  int gx = 256 / ny; if (gx < 1) gx = 1; if (gx > p.tiles) gx = p.tiles;
  SMX_LAUNCH((gemm_rp_bf3_kernel<KSPLIT, D2S, F16>), dim3(gx, ny), dim3(512), LDS, st, p);
  return smx_launch_status();
}

}  // namespace

extern "C" int smx_gemm_rp_bf3_ok(long long M, int N, int K) {
  return (M > 0 && M % TM == 0 && M <= 2147483647LL && ((K == 128 && N % 256 == 0) || (K == 256 && N % 128 == 0)) && N > 0) ? 1 : 0;
}

extern "C" int64_t smx_gemm_rp_bf3_pack_bytes(int N, int K) { return (N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0) ? (int64_t)(N / 32) * (K / 16) * 3 * 1024 : 0; }

extern "C" int smx_gemm_rp_bf3_pack(const float* w, int ldw, void* wp, int N, int K, void* stream) {
  if (!w || !wp || N <= 0 || N % 32 || K <= 0 || K % 16 || ldw < K || ldw % 4 || ((uintptr_t)w & 15) || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  const long long total = (long long)(N / 32) * (K / 16) * 64;
  int g = smx_cdiv(total, 256); if (g > 4096) g = 4096;
  SMX_LAUNCH(gemm_rp_bf3_pack_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w, ldw, (uint4*)wp, N, K);
  return smx_launch_status();
}

static int rp_bf3_launch(const float* a, int lda, const void* wp, const float* bias, const float* res, int ldres, float* c, int ldc,
                         long long M, int N, int K, int act, int d2s_p, int d2s_c, int Ho, int Wo, void* stream, bool f16 = false) {
  if (!a || !wp || !c || !smx_gemm_rp_bf3_ok(M, N, K)) return SMX_EINVAL;
  if (d2s_p) {
    if (d2s_p < 1 || d2s_c <= 0 || d2s_c % 16 || N != d2s_p * d2s_p * d2s_c || ldc < d2s_c || res || Ho <= 0 || Wo <= 0 || M % ((long long)Ho * Wo)) return SMX_EINVAL;
  } else if (ldc < N) return SMX_EINVAL;
  if (lda < K || lda % 4 || ldc % 4 || (res && (ldres < N || ldres % 4))) return SMX_EINVAL;
  if (((uintptr_t)a | (uintptr_t)wp | (uintptr_t)c | (uintptr_t)res | (uintptr_t)bias) & 15) return SMX_EINVAL;
  if ((long long)TM * lda > 2147483647LL) return SMX_EINVAL;
  RPB p;
  p.a = a; p.wp = (const unsigned char*)wp; p.bias = bias; p.res = res; p.c = c;
  p.lda = lda; p.ldres = res ? ldres : 0; p.ldc = ldc; p.M = (int)M; p.N = N; p.K = K; p.act = act; p.tiles = (int)(M / TM);
  p.d2s_p = d2s_p; p.d2s_c = d2s_c; p.Ho = Ho; p.Wo = Wo;
  hipStream_t st = (hipStream_t)stream;
  if (f16) {
    if (d2s_p) return K == 256 ? rpb_launch<2, true, true>(p, st) : rpb_launch<1, true, true>(p, st);
    return K == 256 ? rpb_launch<2, false, true>(p, st) : rpb_launch<1, false, true>(p, st);
  }
  if (d2s_p) return K == 256 ? rpb_launch<2, true>(p, st) : rpb_launch<1, true>(p, st);
  return K == 256 ? rpb_launch<2>(p, st) : rpb_launch<1>(p, st);
}

extern "C" int smx_gemm_rp_bf3(const float* a, int lda, const void* wp, const float* bias, const float* res, int ldres, float* c, int ldc,
                               long long M, int N, int K, int act, void* stream) {
  return rp_bf3_launch(a, lda, wp, bias, res, ldres, c, ldc, M, N, K, act, 0, 0, 0, 0, stream);
}

extern "C" int smx_gemm_rp_d2s_bf3(const float* a, int lda, const void* wp, const float* bias, float* c, int ldc, long long M, int N, int K, int act,
                                   int d2s_p, int d2s_c, int Ho, int Wo, void* stream) {
  if (d2s_p < 1) return SMX_EINVAL;
  return rp_bf3_launch(a, lda, wp, bias, nullptr, 0, c, ldc, M, N, K, act, d2s_p, d2s_c, Ho, Wo, stream);
}

/* The same launches in the "f16x3" arithmetic: two IEEE-half levels per operand, three v_mfma_f32_32x32x16_f16 products per multiply.  wp = smx_gemm_rp_f16_pack
 * (16 header bytes + the record layout of smx_gemm_rp_bf3_pack; W scaled by a power of two chosen on the device); every A row is scaled by its own power of two
 * inside the kernel, so inputs of any magnitude keep fp32-grade products. */
extern "C" int64_t smx_gemm_rp_f16_pack_bytes(int N, int K) { const int64_t b = smx_gemm_rp_bf3_pack_bytes(N, K); return b ? b + 16 : 0; }

extern "C" int smx_gemm_rp_f16_pack(const float* w, int ldw, void* wp, int N, int K, void* stream) {
  if (!w || !wp || N <= 0 || N % 32 || K <= 0 || K % 16 || ldw < K || ldw % 4 || ((uintptr_t)w & 15) || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  SMX_HIP(hipMemsetAsync(wp, 0, 16, (hipStream_t)stream));
  const long long n = (long long)N * K, total = (long long)(N / 32) * (K / 16) * 64;
  int gb = smx_cdiv(n, 256); if (gb > 2048) gb = 2048;
  SMX_LAUNCH(gemm_rp_f16_absmax_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, w, ldw, N, K, (unsigned*)wp);
  int g = smx_cdiv(total, 256); if (g > 4096) g = 4096;
  SMX_LAUNCH(gemm_rp_f16_pack_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w, ldw, (unsigned char*)wp, N, K);
  return smx_launch_status();
}

extern "C" int smx_gemm_rp_f16(const float* a, int lda, const void* wp, const float* bias, const float* res, int ldres, float* c, int ldc,
                               long long M, int N, int K, int act, void* stream) {
  return rp_bf3_launch(a, lda, wp, bias, res, ldres, c, ldc, M, N, K, act, 0, 0, 0, 0, stream, true);
}

extern "C" int smx_gemm_rp_d2s_f16(const float* a, int lda, const void* wp, const float* bias, float* c, int ldc, long long M, int N, int K, int act,
                                   int d2s_p, int d2s_c, int Ho, int Wo, void* stream) {
  if (d2s_p < 1) return SMX_EINVAL;
  return rp_bf3_launch(a, lda, wp, bias, nullptr, 0, c, ldc, M, N, K, act, d2s_p, d2s_c, Ho, Wo, stream, true);
}

