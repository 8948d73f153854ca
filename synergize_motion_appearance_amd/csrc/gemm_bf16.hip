// Implicit-GEMM convolution / batched NT-GEMM for gfx950 on the bf16 MFMA (v_mfma_f32_32x32x16_bf16,
// 2.5 PFLOP/s dense, fp32 accumulate) -- the configs[2] ("bf16 storage / MFMA") form of gemm_conv.hip.
//
//   C[g][m][n] = epi(alpha * sum_k A[g][m][k] * Bt[g][n][k])
//
// A rows are gathered on the fly from an NHWC tensor stored in bf16 (or fp32, converted while staging: the fp32
// keypoint / flow maps enter the bf16 path here); Bt is [N][K] bf16, k contiguous.  At 16x the fp32 matrix rate
// these layers stop being MFMA-bound: the design target is HBM / L2 traffic, so
//   * a K slice is 64 deep: an LDS row is 128 B + 16 B pad (144 B = 36 dwords: the conflict-free pitch of the
//     fp32 kernel), staged global -> registers -> LDS in 16-B chunks (8 bf16), double buffered, one barrier per
//     slice = per 4 MFMAs of every (32x32) sub-tile;
//   * a lane's MFMA operand (8 consecutive k of one row) is ONE ds_read_b128: lane l <-> row l&31, k = 8*(l>>5)..+7;
//   * GroupNorm(+swish) of the producer is applied while staging (in_ss), so the normalised activation never exists
//     in HBM;
//   * the epilogue transposes the block's accumulators through LDS and stores full 16-B chunks of 8 channels per
//     lane (rows of BN*2 bytes contiguous), with bias / activation / residual / depth-to-space fused.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 64;                 // k elements per slice
constexpr int ROWB = 144;              // bytes per LDS row (128 + 16 pad)

struct GB {
  const void* a; const void* bt; void* c; const float* bias; const void* res;
  const float* in_ss; int in_swish;
  long long a_bs0, a_bs1, bt_bs0, bt_bs1, c_bs0, c_bs1, res_bs0, res_bs1;
  int nb1;
  int M, N, K;
  int lda, ldb, ldc, ldres;
  int Hin, Win, Cin, Ho, Wo, kh, kw, stride, pad_t, pad_l, up2;
  int cin_bk;
  int act; float alpha; int bias_per_row; int d2s_p, d2s_c;
  int c_f32, res_f32;
  int tiles_n; int is1x1;
  int xcd_swizzle;
  int ksplit; float* ws;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    case SMX_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case SMX_ACT_SWISH: return v / (1.f + expf(-v));
    case SMX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case SMX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// A_F32: A is stored in fp32 (converted to bf16 while staging).  VEC: 8-element chunks (Cin % 8 == 0, aligned).
template <int BM, int BN, int WGM, int WGN, bool A_F32, bool VEC, bool SHORTK = false>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16_kernel(GB p) {
  constexpr int NT = 64 * WGM * WGN, RPP = NT / 8;       // 8 chunks of 16 B per row
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int RA = (BM + RPP - 1) / RPP, RB = (BN + RPP - 1) / RPP;
  constexpr bool GA = (BM % RPP) != 0, GBN = (BN % RPP) != 0;
  static_assert(TM >= 1 && TN >= 1, "tile / wave grid mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                              // [2][BM][ROWB]
  unsigned char* Bs = smem + 2 * BM * ROWB;              // [2][BN][ROWB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  int logical = blockIdx.x;
  if (p.xcd_swizzle) {     // XCD b%8 gets a contiguous range of logical tiles (see gemm_conv.hip)
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = logical % p.tiles_n, tile_m = logical / p.tiles_n;
  const int g = blockIdx.y, g0 = g / p.nb1, g1 = g - g0 * p.nb1;
  const long long a_goff = g0 * p.a_bs0 + g1 * p.a_bs1;
  const bf16_t* __restrict__ A16 = reinterpret_cast<const bf16_t*>(p.a) + a_goff;
  const float* __restrict__ A32 = reinterpret_cast<const float*>(p.a) + a_goff;
  const bf16_t* __restrict__ Bt = reinterpret_cast<const bf16_t*>(p.bt) + g0 * p.bt_bs0 + g1 * p.bt_bs1;

  const int c8 = tid & 7, r0 = tid >> 3;
  long long a_base[RA]; int a_iy0[RA], a_ix0[RA], a_img[RA]; bool a_ok[RA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = tile_m * BM + r0 + RPP * i;
    a_ok[i] = m < p.M && (!GA || r0 + RPP * i < BM);
    const int img = m / HoWo;
    a_img[i] = img;
    if (p.is1x1) {
      a_base[i] = (long long)m * p.lda; a_iy0[i] = 0; a_ix0[i] = 0;
    } else {
      const int rem = m - img * HoWo, oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_base[i] = (long long)img * p.Hin * p.Win * p.lda;
      a_iy0[i] = oy * p.stride - p.pad_t; a_ix0[i] = ox * p.stride - p.pad_l;
    }
  }
  long long b_base[RB]; bool b_ok[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = tile_n * BN + r0 + RPP * i;
    b_ok[i] = n < p.N && (!GBN || r0 + RPP * i < BN); b_base[i] = (long long)n * p.ldb;
  }
  const int Hlim = p.up2 ? 2 * p.Hin : p.Hin, Wlim = p.up2 ? 2 * p.Win : p.Win;

  const int nslices_all = (p.K + BK - 1) / BK;
  int s_begin = 0, s_end = nslices_all;
  if (p.ksplit > 1) {
    const int per = (nslices_all + p.ksplit - 1) / p.ksplit;
    s_begin = blockIdx.z * per; s_end = min(nslices_all, s_begin + per);
  }
  int tap_c0 = 0, tap_ky = 0, tap_kx = 0;
  if (VEC && s_begin > 0) {
    const int k0 = s_begin * BK, tap = k0 / p.Cin;
    tap_c0 = k0 - tap * p.Cin; tap_ky = tap / p.kw; tap_kx = tap - tap_ky * p.kw;
  }

  // fused GroupNorm(+swish) of 8 consecutive channels of image `img`
  auto norm8 = [&](float (&v)[8], int img, int c) {
    const float* sp = p.in_ss + ((long long)img * p.Cin + c) * 2;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const float4 s = *reinterpret_cast<const float4*>(sp + 2 * e);
      v[e] = fmaf(v[e], s.x, s.y); v[e + 1] = fmaf(v[e + 1], s.z, s.w);
    }
    if (p.in_swish) {
      constexpr float L2E = 1.44269504088896340736f;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * v[e]));
    }
  };

  auto load_into = [&](uint4 (&areg)[RA], uint4 (&breg)[RB], int k0) __attribute__((always_inline)) {
    if (VEC) {
      int l_ky = tap_ky, l_kx = tap_kx, l_c = tap_c0 + c8 * 8; bool kin = true;
      if (!p.cin_bk) {
        const int k = k0 + c8 * 8; kin = k < p.K;
        const int tap = k / p.Cin; l_c = k - tap * p.Cin; l_ky = tap / p.kw; l_kx = tap - l_ky * p.kw;
      }
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (a_ok[i] && kin) {
          long long off = -1; int cc = l_c;
          if (p.is1x1) { off = a_base[i] + k0 + c8 * 8; cc = k0 + c8 * 8; }
          else {
            int iy = a_iy0[i] + l_ky, ix = a_ix0[i] + l_kx;
            if (iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim) {
              if (p.up2) { iy >>= 1; ix >>= 1; }
              off = a_base[i] + ((long long)iy * p.Win + ix) * p.lda + l_c;
            }
          }
          if (off >= 0) {
            if (A_F32) {
              const float4 f0 = *reinterpret_cast<const float4*>(A32 + off), f1 = *reinterpret_cast<const float4*>(A32 + off + 4);
              float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
              if (p.in_ss) norm8(f, a_img[i], cc);
              v = pack8(f);
            } else {
              v = *reinterpret_cast<const uint4*>(A16 + off);
              if (p.in_ss) { float f[8]; unpack8(v, f); norm8(f, a_img[i], cc); v = pack8(f); }
            }
          }
        }
        areg[i] = v;
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (b_ok[i] && kin) v = *reinterpret_cast<const uint4*>(Bt + b_base[i] + k0 + c8 * 8);
        breg[i] = v;
      }
      tap_c0 += BK;
      if (tap_c0 >= p.Cin) { tap_c0 = 0; if (++tap_kx == p.kw) { tap_kx = 0; ++tap_ky; } }
    } else {
      // generic path: any Cin / K / alignment (the 2-, 3-, 15-channel fp32 inputs): per-element index math
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = k0 + c8 * 8 + e; float x = 0.f;
          if (a_ok[i] && k < p.K) {
            const int tap = k / p.Cin, cc = k - tap * p.Cin;
            const int ky = tap / p.kw, kx = tap - ky * p.kw;
            int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
            if (p.is1x1) { iy = 0; ix = 0; }
            if (iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim) {
              if (p.up2) { iy >>= 1; ix >>= 1; }
              const long long off = a_base[i] + (p.is1x1 ? 0LL : ((long long)iy * p.Win + ix) * p.lda) + cc;
              x = A_F32 ? A32[off] : bf2f(A16[off]);
              if (p.in_ss) {
                const float* sp = p.in_ss + ((long long)a_img[i] * p.Cin + cc) * 2;
                x = fmaf(x, sp[0], sp[1]);
                if (p.in_swish) x = x / (1.f + expf(-x));
              }
            }
          }
          f[e] = x;
        }
        areg[i] = pack8(f);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = k0 + c8 * 8 + e;
          f[e] = (b_ok[i] && k < p.K) ? bf2f(Bt[b_base[i] + k]) : 0.f;
        }
        breg[i] = pack8(f);
      }
    }
  };
  auto store_from = [&](const uint4 (&areg)[RA], const uint4 (&breg)[RB], int buf) __attribute__((always_inline)) {
    unsigned char* as = As + buf * BM * ROWB; unsigned char* bs = Bs + buf * BN * ROWB;
#pragma unroll
    for (int i = 0; i < RA; ++i) if (!GA || r0 + RPP * i < BM) *reinterpret_cast<uint4*>(as + (r0 + RPP * i) * ROWB + c8 * 16) = areg[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) if (!GBN || r0 + RPP * i < BN) *reinterpret_cast<uint4*>(bs + (r0 + RPP * i) * ROWB + c8 * 16) = breg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nslices = s_end - s_begin;
  const int frag_off = (lane & 31) * ROWB + (lane >> 5) * 16;
  auto compute = [&](int buf) {
    const unsigned char* as = As + buf * BM * ROWB + (wm * WTM) * ROWB + frag_off;
    const unsigned char* bs = Bs + buf * BN * ROWB + (wn * WTN) * ROWB + frag_off;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(as + i * 32 * ROWB + kk * 32));
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bs + j * 32 * ROWB + kk * 32));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };
  if (SHORTK && nslices <= 4) {
    // short K (<= 256: the 1x1 convolutions, projections, patch GEMMs -- most launches of the step): the whole K range of the
    // tile is requested up front into four register sets, so the tile pays ONE memory round trip instead of one per 64-deep
    // slice (these launches are latency-bound: 2-4 slices per tile, nothing to amortise a serialised pipeline over)
    uint4 a0[RA], b0[RB], a1[RA], b1[RB], a2[RA], b2[RB], a3[RA], b3[RB];
    load_into(a0, b0, s_begin * BK);
    if (nslices > 1) load_into(a1, b1, (s_begin + 1) * BK);
    if (nslices > 2) load_into(a2, b2, (s_begin + 2) * BK);
    if (nslices > 3) load_into(a3, b3, (s_begin + 3) * BK);
    store_from(a0, b0, 0);
    __syncthreads();
    if (nslices > 1) store_from(a1, b1, 1);
    compute(0);
    __syncthreads();
    if (nslices > 1) {
      if (nslices > 2) store_from(a2, b2, 0);
      compute(1);
      __syncthreads();
    }
    if (nslices > 2) {
      if (nslices > 3) store_from(a3, b3, 1);
      compute(0);
      __syncthreads();
    }
    if (nslices > 3) { compute(1); __syncthreads(); }
  } else {
    // write-late pipeline: the loads of slice s+1 are issued before the MFMAs of slice s and written after them
    uint4 areg[RA], breg[RB];
    if (nslices > 0) { load_into(areg, breg, s_begin * BK); store_from(areg, breg, 0); }
    __syncthreads();
    for (int s = 0; s < nslices; ++s) {
      const int buf = s & 1;
      if (s + 1 < nslices) load_into(areg, breg, (s_begin + s + 1) * BK);
      compute(buf);
      if (s + 1 < nslices) store_from(areg, breg, buf ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue ----------------------------------------------------------------------
  if (p.ksplit > 1) {                                  // raw partial sums; the reduce kernel finishes
    float* __restrict__ Wp = p.ws + (long long)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = tile_n * BN + wn * WTN + j * 32 + (lane & 31);
      if (n >= p.N) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = tile_m * BM + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < p.M) Wp[(long long)m * p.N + n] = acc[i][j][r];
        }
    }
    return;
  }
  // block transpose through LDS: Cs[BM][BN + 4] fp32 (the main-loop buffers are free after the last barrier)
  constexpr int CLD = BN + 4;
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        Cs[row * CLD + wn * WTN + j * 32 + (lane & 31)] = acc[i][j][r];
      }
  __syncthreads();
  const long long c_goff = g0 * p.c_bs0 + g1 * p.c_bs1, r_goff = g0 * p.res_bs0 + g1 * p.res_bs1;
  bf16_t* __restrict__ C16 = reinterpret_cast<bf16_t*>(p.c) + c_goff;
  float* __restrict__ C32 = reinterpret_cast<float*>(p.c) + c_goff;
  const bf16_t* __restrict__ R16 = reinterpret_cast<const bf16_t*>(p.res) + r_goff;
  const float* __restrict__ R32 = reinterpret_cast<const float*>(p.res) + r_goff;
  constexpr int CPR = BN / 8;                            // 8-channel chunks per tile row
  const bool vec_c = p.d2s_p ? (p.d2s_c % 8 == 0) : true;
  const bool al_c = (p.ldc % 8 == 0) && ((((uintptr_t)p.c) & 15) == 0) && (c_goff % 8 == 0);
  const bool al_r = !p.res || ((p.ldres % 8 == 0) && ((((uintptr_t)p.res) & 15) == 0) && (r_goff % 8 == 0));
  // Fast path (plain NHWC store, aligned rows, tile inside N -- the engines' layers): this thread's chunk column is the same
  // for all its rows (NT % CPR == 0), so the bias vector is loaded once, the activation is resolved once per block, and a bf16
  // residual is requested for ALL of the thread's rows before the first is used (the rolled loop below issued one dependent
  // HBM round trip per row, 8 in sequence for a 128-row tile).
  static_assert(NT % CPR == 0 && (BM * CPR) % NT == 0, "epilogue mapping");
  constexpr int ITER = BM * CPR / NT, RSTEP = NT / CPR;
  if (!p.d2s_p && al_c && al_r && tile_n * BN + BN <= p.N && !p.bias_per_row) {
    const int cq = tid % CPR, row0 = tid / CPR;
    const int n0 = tile_n * BN + cq * 8;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = p.bias ? p.bias[n0 + e] : 0.f;
    const bool pre = p.res && !p.res_f32;
    uint4 rq[ITER];
    if (pre) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int m = tile_m * BM + row0 + RSTEP * it;
        rq[it] = m < p.M ? *reinterpret_cast<const uint4*>(R16 + (long long)m * p.ldres + n0) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    auto rows = [&](auto mode) __attribute__((always_inline)) {
      constexpr int MODE = decltype(mode)::value;           // 0 identity, 1 relu / leaky relu (slope form), 2 generic, 3 gelu
      const float slope = p.act == SMX_ACT_RELU ? 0.f : 0.2f;
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int row = row0 + RSTEP * it, m = tile_m * BM + row;
        if (m >= p.M) continue;
        const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * CLD + cq * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * CLD + cq * 8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = p.alpha * v[e] + bv[e];
          if (MODE == 1) v[e] = fmaxf(v[e], 0.f) + slope * fminf(v[e], 0.f);
          if (MODE == 2) v[e] = apply_act(v[e], p.act);
          if (MODE == 3) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752440f));
        }
        if (pre) {
          float q[8]; unpack8(rq[it], q);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += q[e];
        } else if (p.res) {
          const float4 q0 = *reinterpret_cast<const float4*>(R32 + (long long)m * p.ldres + n0), q1 = *reinterpret_cast<const float4*>(R32 + (long long)m * p.ldres + n0 + 4);
          v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w; v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
        }
        if (p.c_f32) {
          *reinterpret_cast<float4*>(C32 + (long long)m * p.ldc + n0) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(C32 + (long long)m * p.ldc + n0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          *reinterpret_cast<uint4*>(C16 + (long long)m * p.ldc + n0) = pack8(v);
        }
      }
    };
    if (p.act == SMX_ACT_NONE) rows(std::integral_constant<int, 0>{});
    else if (p.act == SMX_ACT_RELU || p.act == SMX_ACT_LRELU02) rows(std::integral_constant<int, 1>{});
    else if (p.act == SMX_ACT_GELU) rows(std::integral_constant<int, 3>{});
    else rows(std::integral_constant<int, 2>{});
    return;
  }
  // un-patchify (depth-to-space) store, aligned and inside N, no residual (the engines' to_app_feat launches): the thread's chunk column --
  // hence its (p1, p2) sub-pixel and channel offset -- is fixed, only the token's (image, oy, ox) changes per row: the general loop below
  // re-derived all of it per 16-B chunk with five integer divisions (0.18 of the launch's byte bound)
  if (p.d2s_p && vec_c && al_c && !p.res && tile_n * BN + BN <= p.N && !p.bias_per_row) {
    const int cq = tid % CPR, row0 = tid / CPR;
    const int n0 = tile_n * BN + cq * 8;
    const int dq = n0 / p.d2s_c, oc = n0 - dq * p.d2s_c, p1 = dq / p.d2s_p, p2 = dq - p1 * p.d2s_p;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = p.bias ? p.bias[n0 + e] : 0.f;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int row = row0 + RSTEP * it, m = tile_m * BM + row;
      if (m >= p.M) continue;
      const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * CLD + cq * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * CLD + cq * 8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = apply_act(p.alpha * v[e] + bv[e], p.act);
      const int img = m / HoWo, rem = m - img * HoWo, oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const long long opix = ((long long)img * (p.Ho * p.d2s_p) + oy * p.d2s_p + p1) * (p.Wo * p.d2s_p) + ox * p.d2s_p + p2;
      if (p.c_f32) {
        *reinterpret_cast<float4*>(C32 + opix * p.ldc + oc) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(C32 + opix * p.ldc + oc + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        *reinterpret_cast<uint4*>(C16 + opix * p.ldc + oc) = pack8(v);
      }
    }
    return;
  }
  for (int ch = tid; ch < BM * CPR; ch += NT) {
    const int row = ch / CPR, cq = ch - row * CPR;
    const int m = tile_m * BM + row, n0 = tile_n * BN + cq * 8;
    if (m >= p.M || n0 >= p.N) continue;
    const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * CLD + cq * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * CLD + cq * 8 + 4);
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    const float rowb = (p.bias && p.bias_per_row) ? p.bias[m] : 0.f;
    const bool full = n0 + 7 < p.N;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float b = p.bias ? (p.bias_per_row ? rowb : (n0 + e < p.N ? p.bias[n0 + e] : 0.f)) : 0.f;
      v[e] = apply_act(p.alpha * v[e] + b, p.act);
    }
    long long opix; int oc;                               // output pixel row and first channel of this chunk
    if (p.d2s_p) {                                        // un-patchify: n = (p1*p + p2)*dc + c
      const int dq = n0 / p.d2s_c; oc = n0 - dq * p.d2s_c;
      const int p1 = dq / p.d2s_p, p2 = dq - p1 * p.d2s_p;
      const int img = m / HoWo, rem = m - img * HoWo, oy = rem / p.Wo, ox = rem - oy * p.Wo;
      opix = ((long long)img * (p.Ho * p.d2s_p) + oy * p.d2s_p + p1) * (p.Wo * p.d2s_p) + ox * p.d2s_p + p2;
    } else { opix = m; oc = n0; }
    if (full && vec_c && al_c && al_r) {
      if (p.res) {
        if (p.res_f32) {
          const float4 q0 = *reinterpret_cast<const float4*>(R32 + opix * p.ldres + oc), q1 = *reinterpret_cast<const float4*>(R32 + opix * p.ldres + oc + 4);
          v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w; v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
        } else {
          float q[8]; unpack8(*reinterpret_cast<const uint4*>(R16 + opix * p.ldres + oc), q);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += q[e];
        }
      }
      if (p.c_f32) {
        *reinterpret_cast<float4*>(C32 + opix * p.ldc + oc) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(C32 + opix * p.ldc + oc + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        *reinterpret_cast<uint4*>(C16 + opix * p.ldc + oc) = pack8(v);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (n0 + e >= p.N) break;
        long long px = opix; int c = oc + e;
        if (p.d2s_p && !vec_c) {                            // chunk may straddle a (p1,p2) boundary: per element
          const int n = n0 + e, dq = n / p.d2s_c; c = n - dq * p.d2s_c;
          const int p1 = dq / p.d2s_p, p2 = dq - p1 * p.d2s_p;
          const int img = m / HoWo, rem = m - img * HoWo, oy = rem / p.Wo, ox = rem - oy * p.Wo;
          px = ((long long)img * (p.Ho * p.d2s_p) + oy * p.d2s_p + p1) * (p.Wo * p.d2s_p) + ox * p.d2s_p + p2;
        }
        float o = v[e];
        if (p.res) o += p.res_f32 ? R32[px * p.ldres + c] : bf2f(R16[px * p.ldres + c]);
        if (p.c_f32) C32[px * p.ldc + c] = o; else C16[px * p.ldc + c] = f2bf(o);
      }
    }
  }
}

__global__ __launch_bounds__(256) void splitk_reduce16_kernel(GB p) {
  const long long total = (long long)p.M * p.N;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / p.N), n = (int)(i - (long long)m * p.N);
    float acc = 0.f;
    for (int z = 0; z < p.ksplit; ++z) acc += p.ws[(long long)z * total + i];
    float v = p.alpha * acc;
    if (p.bias) v += p.bias[p.bias_per_row ? m : n];
    v = apply_act(v, p.act);
    if (p.res) v += p.res_f32 ? reinterpret_cast<const float*>(p.res)[(long long)m * p.ldres + n]
                              : bf2f(reinterpret_cast<const bf16_t*>(p.res)[(long long)m * p.ldres + n]);
    if (p.c_f32) reinterpret_cast<float*>(p.c)[(long long)m * p.ldc + n] = v;
    else reinterpret_cast<bf16_t*>(p.c)[(long long)m * p.ldc + n] = f2bf(v);
  }
}

template <int BM, int BN, int WGM, int WGN>
int launch16(const GB& p, int nb, bool a_f32, bool vec, hipStream_t st) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  GB q = p; q.tiles_n = tiles_n;
  dim3 grid(tiles_m * tiles_n, nb, p.ksplit > 1 ? p.ksplit : 1), block(64 * WGM * WGN);
  size_t lds = (size_t)2 * (BM + BN) * ROWB, lds_c = (size_t)BM * (BN + 4) * sizeof(float);
  if (lds_c > lds) lds = lds_c;
#define SMX_L16(AF, VC)                                                                                             \
  do {                                                                                                              \
    auto k = gemm_bf16_kernel<BM, BN, WGM, WGN, AF, VC>;                                                            \
    if (lds > 64 * 1024) SMX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    SMX_LAUNCH(k, grid, block, lds, st, q);                                                                         \
  } while (0)
  const bool shortk = vec && !a_f32 && p.ksplit <= 1 && p.K <= 4 * BK;
  if (shortk) {
    auto k = gemm_bf16_kernel<BM, BN, WGM, WGN, false, true, true>;
    if (lds > 64 * 1024) SMX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SMX_LAUNCH(k, grid, block, lds, st, q);
  }
  else if (a_f32) { if (vec) SMX_L16(true, true); else SMX_L16(true, false); }
  else { if (vec) SMX_L16(false, true); else SMX_L16(false, false); }
#undef SMX_L16
  if (p.ksplit > 1) {
    int blocks = (int)(((long long)p.M * p.N + 255) / 256); if (blocks > 4096) blocks = 4096;
    SMX_LAUNCH(splitk_reduce16_kernel, dim3(blocks), dim3(256), 0, st, q);
  }
  return smx_launch_status();
}

}  // namespace

extern "C" int smx_gemm_conv_bf16(const smx_gemm16_desc* d, void* stream) {
  if (!d || !d->a || !d->bt || !d->c) return SMX_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->nb0 <= 0 || d->nb1 <= 0) return SMX_EINVAL;
  if (d->kh <= 0 || d->kw <= 0 || d->Cin <= 0 || d->K != d->kh * d->kw * d->Cin) return SMX_EINVAL;
  if (d->Ho <= 0 || d->Wo <= 0 || d->M % (d->Ho * d->Wo) != 0) return SMX_EINVAL;
  if (d->stride <= 0 || d->lda < d->Cin || d->ldb < d->K) return SMX_EINVAL;
  if (d->d2s_p ? (d->d2s_c <= 0 || d->N != d->d2s_p * d->d2s_p * d->d2s_c || d->ldc < d->d2s_c) : d->ldc < d->N) return SMX_EINVAL;
  if (d->in_ss && (((uintptr_t)d->in_ss) & 15)) return SMX_EINVAL;
  const long long nb = (long long)d->nb0 * d->nb1;
  if (nb > 65535) return SMX_EINVAL;
  if (d->in_ss && nb != 1) return SMX_EINVAL;
  GB p;
  p.a = d->a; p.bt = d->bt; p.c = d->c; p.bias = d->bias; p.res = d->res; p.in_ss = d->in_ss; p.in_swish = d->in_swish;
  p.a_bs0 = d->a_bs0; p.a_bs1 = d->a_bs1; p.bt_bs0 = d->bt_bs0; p.bt_bs1 = d->bt_bs1;
  p.c_bs0 = d->c_bs0; p.c_bs1 = d->c_bs1; p.res_bs0 = d->res_bs0; p.res_bs1 = d->res_bs1;
  p.nb1 = d->nb1; p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldres = d->res ? d->ldres : 0;
  p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo;
  p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.up2 = d->up2;
  p.act = d->act; p.alpha = d->alpha; p.bias_per_row = d->bias_per_row; p.d2s_p = d->d2s_p; p.d2s_c = d->d2s_c;
  p.c_f32 = d->c_f32; p.res_f32 = d->res_f32;
  p.tiles_n = 1;
  p.xcd_swizzle = smx_tune(SMX_TUNE_GEMM_XCD_SWIZZLE) ? 1 : 0;
  p.ksplit = d->ksplit > 1 ? d->ksplit : 1; p.ws = d->ws;
  if (p.ksplit > 1 && (!d->ws || nb != 1 || d->d2s_p || p.ksplit > (d->K + BK - 1) / BK)) return SMX_EINVAL;
  p.is1x1 = (d->kh == 1 && d->kw == 1 && d->stride == 1 && !d->up2 && d->pad_t == 0 && d->pad_l == 0 &&
             d->Hin == d->Ho && d->Win == d->Wo) ? 1 : 0;
  p.cin_bk = d->Cin % BK == 0 ? 1 : 0;
  const int asz = d->a_f32 ? 4 : 2;
  const bool vec = (d->Cin % 8 == 0) && ((long long)d->lda * asz % 16 == 0) && (d->ldb % 8 == 0) &&
                   (((uintptr_t)d->a & 15) == 0) && (((uintptr_t)d->bt & 15) == 0) &&
                   (d->a_bs0 * asz % 16 == 0) && (d->a_bs1 * asz % 16 == 0) && (d->bt_bs0 % 8 == 0) && (d->bt_bs1 % 8 == 0);
  hipStream_t st = (hipStream_t)stream;
  int tile = d->tile;
  if (tile == 0) {
    const long long Mt = (long long)d->M * nb;
    // measured on the device (tools/gemm16_tune.py, B = 60 shapes): 64x64 tiles (4 waves, 4 workgroups / CU) win every
    // short-K launch -- they are latency-bound --, 128x128 / 8 waves the long-K and very wide ones.  (128x128 / 4 waves, tile 7,
    // wins the isolated 1x1 sweep by 8-18 % but loses in the pipeline's own launches -- GELU / fp32-out epilogues, N = 2048-4096:
    // +0.5 ms per step -- and stays selectable only.)
    if (d->N <= 32) tile = 4;
    else if (d->N % 128 == 0 && Mt >= 32768 && (d->K >= 1024 || d->N >= 2048 || (d->K >= 256 && Mt >= 500000))) tile = 1;
    else tile = 3;
  }
  const bool af = d->a_f32 != 0;
  switch (tile) {
    case 1: return launch16<128, 128, 4, 2>(p, (int)nb, af, vec, st);   // 8 waves, 32x64 per wave
    case 2: return launch16<128, 64, 4, 1>(p, (int)nb, af, vec, st);    // 4 waves, 32x64 per wave
    case 3: return launch16<64, 64, 2, 2>(p, (int)nb, af, vec, st);     // 4 waves, 32x32 per wave
    case 4: return launch16<128, 32, 4, 1>(p, (int)nb, af, vec, st);    // 4 waves, N <= 32
    case 5: return launch16<64, 128, 2, 2>(p, (int)nb, af, vec, st);    // 4 waves, 32x64 per wave
    case 6: return launch16<256, 64, 8, 1>(p, (int)nb, af, vec, st);    // 8 waves, 32x64 per wave
    case 7: return launch16<128, 128, 2, 2>(p, (int)nb, af, vec, st);   // 4 waves, 64x64 per wave (16 MFMAs per wave per 64-deep slice)
    default: return SMX_EINVAL;
  }
}
