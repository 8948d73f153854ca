// Implicit-GEMM convolution / batched NT-GEMM for gfx950 on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).
//
//   C[g][m][n] = epi(alpha * sum_k A[g][m][k] * Bt[g][n][k])
//
// A rows are gathered from an NHWC tensor on the fly (m -> pixel, k -> (ky,kx,c)); Bt is
// [N][K] with k contiguous.  Both operand tiles are staged global -> registers -> LDS as
// [rows][32 k + 4 pad] fp32 (row stride 36 dwords: conflict-free for the 16-lane groups of
// ds_read_b128), double-buffered, one barrier per 32-deep K slice.  A wave64 reads its
// fragments with ONE ds_read_b128 per 32x8 sub-tile: lane l holds row (l&31), k = 4*(l>>5)+j,
// j=0..3, and MFMA j of the 8-deep step pairs k=j (lanes 0-31) with k=4+j (lanes 32-63) --
// the k pairing is free because A and Bt fragments use the same map.
//
// Tile configs (256 threads = 4 waves): 128x128, 128x64, 128x32, 64x128, 64x64.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>
#include "smx.h"
#include "smx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 36;

struct GP {
  const float* a; const float* bt; float* c; const float* bias; const float* res;
  long long a_bs0, a_bs1, bt_bs0, bt_bs1, c_bs0, c_bs1, res_bs0, res_bs1;
  int nb1;
  int M, N, K;
  int lda, ldb, ldc, ldres;
  int Hin, Win, Cin, Ho, Wo, kh, kw, stride, pad_t, pad_l, up2;
  int cin_bk;      // Cin % BK == 0: a k-slice never straddles two filter taps (running tap counters)
  int act; float alpha; int bias_per_row; int d2s_p, d2s_c;
  int tiles_n; int is1x1;
  int xcd_swizzle; int variant;
  int ksplit; float* ws;            // split-K: blockIdx.z owns a K range, raw partials -> ws[z][M][N]
  int fast;                         // the buffer-load loader applies (host-checked: vector layout, Cin % 32 == 0, no upsampling, tile-relative offsets < 2^31)
};

// MODE 2 loader (round 5).  On gfx950 a VALU instruction costs an fp32 MFMA kernel its full issue time (profiles/r05_winograd_valu_vs_mfma.txt), and the
// gather below spent ~114 of them per 16-32 MFMAs on 64-bit addresses, zero-fills and predicates.  Here every operand quad is ONE buffer load: the
// resource (tile base) and the slice's offset (k0, or the filter tap's pixel offset + channel) are wave-uniform SGPR values, the thread's row offset is
// computed once per block, and a row that must read zero (M / N tail, padding) presents an offset past the buffer instead of taking a branch:
// 0 VALU per quad for a 1x1 layer / the weights, 5 for a k x k layer (the tap's bit of the row's validity masks -> v_cndmask of the offset).
constexpr unsigned GC_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gc_resource(const float* base) {
  // (the base IS wave-uniform; saying so keeps hipcc from wrapping every load of the bigger tiles in a waterfall loop)
  const unsigned long long u = (unsigned long long)(uintptr_t)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>((uintptr_t)(((unsigned long long)hi << 32) | lo)), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 gc_fetch(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));   // (bit_cast of the builtin's own vector type: an implicit
  return make_float4(v.x, v.y, v.z, v.w);                                                                // conversion to an ext_vector made hipcc load ONE dword)
}

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

template <int BM, int BN, int WGM, int WGN, int LMODE>     // LMODE 0: any layout (scalar gather), 1: float4 gather, 2: float4 buffer loads (p.fast)
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_conv_kernel(GP p) {
  constexpr bool VEC = LMODE >= 1, FAST = LMODE == 2;
  constexpr int NT = 64 * WGM * WGN, RPP = NT / 8;   // threads; staging rows per pass (8 float4 per 32-k row)
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int RA = (BM + RPP - 1) / RPP, RB = (BN + RPP - 1) / RPP;   // rows per thread in the staging pass
  constexpr bool GA = (BM % RPP) != 0, GB = (BN % RPP) != 0;            // tile has fewer rows than one pass
  static_assert(TM >= 1 && TN >= 1, "tile / wave grid mismatch");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                   // [2][BM][LDS_LD]
  float* Bs = smem + 2 * BM * LDS_LD;                 // [2][BN][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  // XCD-aware tile order: the dispatcher puts block b on XCD b%8 (each XCD has a private L2).
  // Give every XCD a CONTIGUOUS range of logical tiles (tile_n fastest, then tile_m), so the
  // N-tiles of one M-tile and the 3x3 halos of neighbouring M-tiles hit the same L2 instead of
  // being fetched over the fabric by 4-8 different XCDs.
  int logical = blockIdx.x;
  if (p.xcd_swizzle) {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = logical % p.tiles_n, tile_m = logical / p.tiles_n;
  const int g = blockIdx.y, g0 = g / p.nb1, g1 = g - g0 * p.nb1;
  const float* __restrict__ A = p.a + g0 * p.a_bs0 + g1 * p.a_bs1;
  const float* __restrict__ Bt = p.bt + g0 * p.bt_bs0 + g1 * p.bt_bs1;

  // ---- per-thread staging coordinates ------------------------------------------------
  const int c4 = tid & 7, r0 = tid >> 3;
  long long a_base[RA]; int a_iy0[RA], a_ix0[RA]; bool a_ok[RA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = tile_m * BM + r0 + RPP * i;
    a_ok[i] = m < p.M && (!GA || r0 + RPP * i < BM);
    if (p.is1x1) {
      a_base[i] = (long long)m * p.lda; a_iy0[i] = 0; a_ix0[i] = 0;
    } else {
      int img = m / HoWo, rem = m - img * HoWo, oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_base[i] = (long long)img * p.Hin * p.Win * p.lda;
      a_iy0[i] = oy * p.stride - p.pad_t; a_ix0[i] = ox * p.stride - p.pad_l;
    }
  }
  long long b_base[RB]; bool b_ok[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    int n = tile_n * BN + r0 + RPP * i;
    b_ok[i] = n < p.N && (!GB || r0 + RPP * i < BN); b_base[i] = (long long)n * p.ldb;
  }
  const int Hlim = p.up2 ? 2 * p.Hin : p.Hin, Wlim = p.up2 ? 2 * p.Win : p.Win;
  // MODE 2: tile-relative byte offsets, validity masks (bit ky of f_my: input row iy0 + ky is inside the frame; bit kx of f_mx likewise)
  unsigned f_av[RA], f_my[RA], f_mx[RA], f_bv[RB];
  const int f_img0 = FAST && !p.is1x1 ? (tile_m * BM) / HoWo : 0;
  const long long f_abias = FAST && !p.is1x1 ? ((long long)p.pad_t * p.Win + p.pad_l) * p.lda : 0;          // floats: keeps a padded row's offset non-negative
  const __amdgpu_buffer_rsrc_t f_ra = gc_resource(p.is1x1 ? A + (long long)tile_m * BM * p.lda : A + (long long)f_img0 * p.Hin * p.Win * p.lda - f_abias);
  const __amdgpu_buffer_rsrc_t f_rb = gc_resource(Bt + (long long)tile_n * BN * p.ldb);
  if (FAST) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int row = r0 + RPP * i, m = tile_m * BM + row;
      const bool ok = m < p.M && (!GA || row < BM);
      if (p.is1x1) {
        f_av[i] = ok ? (unsigned)(row * p.lda + c4 * 4) * 4u : GC_OOB; f_my[i] = 1u; f_mx[i] = 1u;
      } else {
        const int img = m / HoWo, rem = m - img * HoWo, oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
        f_av[i] = (unsigned)(((long long)(img - f_img0) * p.Hin + iy0) * p.Win + ix0) * (unsigned)p.lda * 4u + (unsigned)(f_abias + c4 * 4) * 4u;
        unsigned my = 0u, mx = 0u;
        for (int k = 0; k < p.kh; ++k) my |= (unsigned)(iy0 + k >= 0 && iy0 + k < p.Hin) << k;
        for (int k = 0; k < p.kw; ++k) mx |= (unsigned)(ix0 + k >= 0 && ix0 + k < p.Win) << k;
        f_my[i] = ok ? my : 0u; f_mx[i] = mx;
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = r0 + RPP * i, n = tile_n * BN + row;
      f_bv[i] = (n < p.N && (!GB || row < BN)) ? (unsigned)(row * p.ldb + c4 * 4) * 4u : GC_OOB;
    }
  }

  float4 areg[RA], breg[RB];
  const int nslices_all = (p.K + BK - 1) / BK;
  int s_begin = 0, s_end = nslices_all;
  if (p.ksplit > 1) {
    const int per = (nslices_all + p.ksplit - 1) / p.ksplit;
    s_begin = blockIdx.z * per; s_end = min(nslices_all, s_begin + per);
  }
  int tap_c0 = 0, tap_ky = 0, tap_kx = 0;             // VEC path: running (ky,kx,c0) of the slice
  if (VEC && s_begin > 0) {
    const int k0 = s_begin * BK, tap = k0 / p.Cin;
    tap_c0 = k0 - tap * p.Cin; tap_ky = tap / p.kw; tap_kx = tap - tap_ky * p.kw;
  }

  auto load_slice = [&](int k0) {
    if (FAST) {
      const unsigned sa = p.is1x1 ? (unsigned)k0 * 4u : (unsigned)((tap_ky * p.Win + tap_kx) * p.lda + tap_c0) * 4u;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        unsigned vo = f_av[i];
        if (!p.is1x1) vo = ((f_my[i] >> tap_ky) & (f_mx[i] >> tap_kx) & 1u) ? vo : GC_OOB;
        areg[i] = gc_fetch(f_ra, vo, sa);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) breg[i] = gc_fetch(f_rb, f_bv[i], (unsigned)k0 * 4u);
      tap_c0 += BK;
      if (tap_c0 >= p.Cin) { tap_c0 = 0; if (++tap_kx == p.kw) { tap_kx = 0; ++tap_ky; } }
    } else if (VEC) {
      // Cin % 32 == 0: the slice lies inside one tap (running counters).  Cin % 4 == 0 only (e.g. the
      // 36-channel keypoint head): this lane's float4 column has its own (tap, channel) -- two integer
      // divisions per slice per lane, shared by all its rows; K % 32 != 0 needs the k < K guard.
      int l_ky = tap_ky, l_kx = tap_kx, l_c = tap_c0 + c4 * 4; bool kin = true;
      if (!p.cin_bk) {
        const int k = k0 + c4 * 4; kin = k < p.K;
        const int tap = k / p.Cin; l_c = k - tap * p.Cin; l_ky = tap / p.kw; l_kx = tap - l_ky * p.kw;
      }
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_ok[i] && kin) {
          if (p.is1x1) {
            v = *reinterpret_cast<const float4*>(A + a_base[i] + k0 + c4 * 4);
          } else {
            int iy = a_iy0[i] + l_ky, ix = a_ix0[i] + l_kx;
            // up2 == 1: nearest x2 (Upsample folded into the gather); up2 == 2: ZERO-INSERT x2 -- the data gradient of a
            // stride-2 convolution is a stride-1 convolution over the zero-stuffed output gradient: odd virtual positions are 0
            if (iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim && !(p.up2 == 2 && ((iy | ix) & 1))) {
              if (p.up2) { iy >>= 1; ix >>= 1; }
              v = *reinterpret_cast<const float4*>(A + a_base[i] + ((long long)iy * p.Win + ix) * p.lda + l_c);
            }
          }
        }
        areg[i] = v;
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b_ok[i] && kin) v = *reinterpret_cast<const float4*>(Bt + b_base[i] + k0 + c4 * 4);
        breg[i] = v;
      }
      tap_c0 += BK;
      if (tap_c0 >= p.Cin) { tap_c0 = 0; if (++tap_kx == p.kw) { tap_kx = 0; ++tap_ky; } }
    } else {
      // generic path: any Cin / K / alignment -- per-element index math, scalar loads
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int k = k0 + c4 * 4 + e; float x = 0.f;
          if (a_ok[i] && k < p.K) {
            int tap = k / p.Cin, cc = k - tap * p.Cin;
            int ky = tap / p.kw, kx = tap - ky * p.kw;
            int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
            if (iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim && !(p.up2 == 2 && ((iy | ix) & 1))) {
              if (p.up2) { iy >>= 1; ix >>= 1; }
              x = A[a_base[i] + ((long long)iy * p.Win + ix) * p.lda + cc];
            }
          }
          v[e] = x;
        }
        areg[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int k = k0 + c4 * 4 + e;
          v[e] = (b_ok[i] && k < p.K) ? Bt[b_base[i] + k] : 0.f;
        }
        breg[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };
  auto store_slice = [&](int buf) {
    float* as = As + buf * BM * LDS_LD; float* bs = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < RA; ++i) if (!GA || r0 + RPP * i < BM) *reinterpret_cast<float4*>(as + (r0 + RPP * i) * LDS_LD + c4 * 4) = areg[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) if (!GB || r0 + RPP * i < BN) *reinterpret_cast<float4*>(bs + (r0 + RPP * i) * LDS_LD + c4 * 4) = breg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nslices = s_end - s_begin;
  const int frag_off = (lane & 31) * LDS_LD + (lane >> 5) * 4;
  const bool early = (p.variant & 1) != 0, prio = (p.variant & 2) != 0;
  auto compute = [&](int buf) {
    const float* as = As + buf * BM * LDS_LD + (wm * WTM) * LDS_LD + frag_off;
    const float* bs = Bs + buf * BN * LDS_LD + (wn * WTN) * LDS_LD + frag_off;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(bs + j * 32 * LDS_LD + kk * 8);
      if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
      if (prio) __builtin_amdgcn_s_setprio(0);
    }
  };
  if (!early) {
    if (nslices > 0) { load_slice(s_begin * BK); store_slice(0); }
    __syncthreads();
    for (int s = 0; s < nslices; ++s) {
      const int buf = s & 1;
      if (s + 1 < nslices) load_slice((s_begin + s + 1) * BK);
      compute(buf);
      if (s + 1 < nslices) store_slice(buf ^ 1);
      __syncthreads();
    }
  } else {
    // store-early: the LDS write of slice s+1 opens iteration s (its loads were issued a full
    // iteration ago), the loads of slice s+2 follow, then the MFMAs, then the only barrier.
    if (nslices > 0) { load_slice(s_begin * BK); store_slice(0); }
    if (nslices > 1) load_slice((s_begin + 1) * BK);
    __syncthreads();
    for (int s = 0; s < nslices; ++s) {
      const int buf = s & 1;
      if (s + 1 < nslices) store_slice(buf ^ 1);
      if (s + 2 < nslices) load_slice((s_begin + s + 2) * BK);
      compute(buf);
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
  float* __restrict__ C = p.c + g0 * p.c_bs0 + g1 * p.c_bs1;
  const float* __restrict__ R = p.res ? p.res + g0 * p.res_bs0 + g1 * p.res_bs1 : nullptr;
  const int mrow0 = tile_m * BM + wm * WTM + 4 * (lane >> 5);
  if (!p.d2s_p) {
    // plain store.  Two phases so that every residual / row-bias load of the tile is in flight
    // before the first use (one s_waitcnt per tile column instead of one per element).  The activation and the
    // tile-inside-the-matrix test are resolved once per block (MODE: 0 identity, 1 relu / leaky relu in slope form,
    // 2 generic; FULL: no per-element bounds tests) -- a per-value `switch (p.act)` compiles to scalar branches around
    // every one of the 16*TM*TN values.
    auto store_tile = [&](auto mode, auto fulltag) __attribute__((always_inline)) {
      constexpr int MODE = decltype(mode)::value;
      constexpr bool FULL = decltype(fulltag)::value;
      const float slope = p.act == SMX_ACT_RELU ? 0.f : 0.2f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = tile_n * BN + wn * WTN + j * 32 + (lane & 31);
        const bool nok = FULL || n < p.N;
        const float bn = (nok && p.bias && !p.bias_per_row) ? p.bias[n] : 0.f;
        float rv[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
            float t = 0.f;
            if (R && nok && (FULL || m < p.M)) t = R[(long long)m * p.ldres + n];
            rv[i][r] = t;
          }
        float rb[TM][16];
        if (p.bias && p.bias_per_row) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
              rb[i][r] = (FULL || m < p.M) ? p.bias[m] : 0.f;
            }
        } else {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) rb[i][r] = bn;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
            float v = p.alpha * acc[i][j][r] + rb[i][r];
            if (MODE == 1) v = fmaxf(v, 0.f) + slope * fminf(v, 0.f);
            if (MODE == 2) v = apply_act(v, p.act);
            if (MODE == 3) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
            v += rv[i][r];
            if (nok && (FULL || m < p.M)) C[(long long)m * p.ldc + n] = v;
          }
      }
    };
    const bool full = (tile_m * BM + BM <= p.M) && (tile_n * BN + BN <= p.N);
    const int amode = p.act == SMX_ACT_NONE ? 0 : ((p.act == SMX_ACT_RELU || p.act == SMX_ACT_LRELU02) ? 1 : (p.act == SMX_ACT_GELU ? 3 : 2));
    if (full) {
      if (amode == 0) store_tile(std::integral_constant<int, 0>{}, std::true_type{});
      else if (amode == 1) store_tile(std::integral_constant<int, 1>{}, std::true_type{});
      else if (amode == 3) store_tile(std::integral_constant<int, 3>{}, std::true_type{});
      else store_tile(std::integral_constant<int, 2>{}, std::true_type{});
    } else {
      if (amode == 0) store_tile(std::integral_constant<int, 0>{}, std::false_type{});
      else if (amode == 1) store_tile(std::integral_constant<int, 1>{}, std::false_type{});
      else store_tile(std::integral_constant<int, 2>{}, std::false_type{});
    }
    return;
  }
  // depth-to-space store (un-patchify): n = (p1*p+p2)*dc + c
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = tile_n * BN + wn * WTN + j * 32 + (lane & 31);
    if (n >= p.N) continue;
    const float bn = p.bias ? p.bias[n] : 0.f;
    const int dq = n / p.d2s_c, dcn = n - dq * p.d2s_c;
    const int p1 = dq / p.d2s_p, p2 = dq - p1 * p.d2s_p;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
        if (m >= p.M) continue;
        float v = apply_act(p.alpha * acc[i][j][r] + bn, p.act);
        const int img = m / HoWo, rem = m - img * HoWo, oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const long long pix = ((long long)img * (p.Ho * p.d2s_p) + oy * p.d2s_p + p1) * (p.Wo * p.d2s_p) + ox * p.d2s_p + p2;
        if (R) v += R[pix * p.ldres + dcn];
        C[pix * p.ldc + dcn] = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GP p) {
  const long long total = (long long)p.M * p.N;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / p.N), n = (int)(i - (long long)m * p.N);
    float acc = 0.f;
    for (int z = 0; z < p.ksplit; ++z) acc += p.ws[(long long)z * total + i];
    float v = p.alpha * acc;
    if (p.bias) v += p.bias[p.bias_per_row ? m : n];
    v = apply_act(v, p.act);
    if (p.res) v += p.res[(long long)m * p.ldres + n];
    p.c[(long long)m * p.ldc + n] = v;
  }
}

template <int BM, int BN, int WGM, int WGN>
int launch_cfg(const GP& p, int nb, bool vec, hipStream_t st) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  GP q = p; q.tiles_n = tiles_n;
  dim3 grid(tiles_m * tiles_n, nb, p.ksplit > 1 ? p.ksplit : 1), block(64 * WGM * WGN);
  size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
  // MODE 2's offsets are 32-bit and relative to the tile: a tile's rows span at most BM / (Ho Wo) + 2 images of the input
  const long long span = ((long long)BM / ((long long)p.Ho * p.Wo) + 2) * p.Hin * p.Win * p.lda * 4 + ((long long)p.pad_t * p.Win + p.pad_l + 8) * p.lda * 4;
  q.fast = (vec && p.fast && (p.is1x1 || span < 0x7fffffffLL) && (long long)BM * p.lda * 4 < 0x7fffffffLL && (long long)BN * p.ldb * 4 + (long long)p.K * 4 < 0x7fffffffLL) ? 1 : 0;
  if (q.fast) {
    auto k = gemm_conv_kernel<BM, BN, WGM, WGN, 2>;
    if (lds > 64 * 1024) SMX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SMX_LAUNCH(k, grid, block, lds, st, q);
  } else if (vec) {
    auto k = gemm_conv_kernel<BM, BN, WGM, WGN, 1>;
    if (lds > 64 * 1024) SMX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SMX_LAUNCH(k, grid, block, lds, st, q);
  } else {
    auto k = gemm_conv_kernel<BM, BN, WGM, WGN, 0>;
    if (lds > 64 * 1024) SMX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SMX_LAUNCH(k, grid, block, lds, st, q);
  }
  if (p.ksplit > 1) {
    int blocks = (int)(((long long)p.M * p.N + 255) / 256); if (blocks > 4096) blocks = 4096;
    SMX_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, q);
  }
  return smx_launch_status();
}

}  // namespace

extern "C" int smx_gemm_conv_f32(const smx_gemm_desc* d, void* stream) {
  if (!d || !d->a || !d->bt || !d->c) return SMX_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->nb0 <= 0 || d->nb1 <= 0) return SMX_EINVAL;
  if (d->kh <= 0 || d->kw <= 0 || d->Cin <= 0 || d->K != d->kh * d->kw * d->Cin) return SMX_EINVAL;
  if (d->Ho <= 0 || d->Wo <= 0 || d->M % (d->Ho * d->Wo) != 0) return SMX_EINVAL;
  if (d->stride <= 0 || d->lda < d->Cin || d->ldb < d->K) return SMX_EINVAL;
  if (d->d2s_p ? (d->d2s_c <= 0 || d->N != d->d2s_p * d->d2s_p * d->d2s_c || d->ldc < d->d2s_c) : d->ldc < d->N) return SMX_EINVAL;
  const long long nb = (long long)d->nb0 * d->nb1;
  if (nb > 65535) return SMX_EINVAL;
  GP p;
  p.a = d->a; p.bt = d->bt; p.c = d->c; p.bias = d->bias; p.res = d->res;
  p.a_bs0 = d->a_bs0; p.a_bs1 = d->a_bs1; p.bt_bs0 = d->bt_bs0; p.bt_bs1 = d->bt_bs1;
  p.c_bs0 = d->c_bs0; p.c_bs1 = d->c_bs1; p.res_bs0 = d->res_bs0; p.res_bs1 = d->res_bs1;
  p.nb1 = d->nb1; p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldres = d->res ? d->ldres : 0;
  p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo;
  p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.up2 = d->up2;
  p.act = d->act; p.alpha = d->alpha; p.bias_per_row = d->bias_per_row; p.d2s_p = d->d2s_p; p.d2s_c = d->d2s_c;
  p.tiles_n = 1;
  p.xcd_swizzle = smx_tune(SMX_TUNE_GEMM_XCD_SWIZZLE) ? 1 : 0;
  p.variant = smx_tune(SMX_TUNE_GEMM_VARIANT);
  p.ksplit = d->ksplit > 1 ? d->ksplit : 1; p.ws = d->ws;
  if (p.ksplit > 1 && (!d->ws || nb != 1 || d->d2s_p || p.ksplit > (d->K + BK - 1) / BK)) return SMX_EINVAL;
  p.is1x1 = (d->kh == 1 && d->kw == 1 && d->stride == 1 && !d->up2 && d->pad_t == 0 && d->pad_l == 0 &&
             d->Hin == d->Ho && d->Win == d->Wo) ? 1 : 0;
  p.cin_bk = d->Cin % BK == 0 ? 1 : 0;
  p.fast = (p.cin_bk && !d->up2 && d->kh <= 32 && d->kw <= 32 && smx_tune(SMX_TUNE_GEMM_LOADER) != 0) ? 1 : 0;
  const bool vec = (d->Cin % 4 == 0) && (d->lda % 4 == 0) && (d->ldb % 4 == 0) &&
                   (((uintptr_t)d->a & 15) == 0) && (((uintptr_t)d->bt & 15) == 0) &&
                   (d->a_bs0 % 4 == 0) && (d->a_bs1 % 4 == 0) && (d->bt_bs0 % 4 == 0) && (d->bt_bs1 % 4 == 0);
  hipStream_t st = (hipStream_t)stream;
  int tile = d->tile;
  if (tile == 0) {
    // measured on MI355X (profiles/r01_b_gemm_tile_tuning.txt): 16 resident waves per CU win --
    // 64x64 tiles (4 waves, 4 blocks/CU) almost everywhere, 128x128 with 8 waves for big-M, N%128==0
    if (d->N <= 32) tile = 3;
    else if (d->N % 128 == 0 && (long long)d->M * nb >= 49152 && d->K <= 256 && d->N <= 512 && !d->d2s_p) tile = 12;   // short K, big M: 256x128, 16 waves
    else if (d->N % 128 == 0 && (long long)d->M * nb >= 24576) tile = 8;
    else if (d->N > 64 && d->N < 128 && d->K >= 1024 && (long long)d->M * nb >= 24576) tile = 8;   // the 7x7 keypoint / jacobian head (N = 76, K = 1764): 4.8 -> 4.0 ms at B = 300
    else tile = 5;
  }
  switch (tile) {
    case 1: return launch_cfg<128, 128, 2, 2>(p, (int)nb, vec, st);
    case 2: return launch_cfg<128, 64, 2, 2>(p, (int)nb, vec, st);
    case 3: return launch_cfg<128, 32, 4, 1>(p, (int)nb, vec, st);
    case 4: return launch_cfg<64, 128, 1, 4>(p, (int)nb, vec, st);
    case 5: return launch_cfg<64, 64, 2, 2>(p, (int)nb, vec, st);
    case 6: return launch_cfg<256, 128, 4, 2>(p, (int)nb, vec, st);   // 8 waves, 64x64 per wave
    case 7: return launch_cfg<256, 64, 4, 2>(p, (int)nb, vec, st);    // 8 waves, 64x32 per wave
    case 8: return launch_cfg<128, 128, 4, 2>(p, (int)nb, vec, st);   // 8 waves, 32x64 per wave
    case 9: return launch_cfg<128, 128, 4, 4>(p, (int)nb, vec, st);   // 16 waves, 32x32 per wave
    case 10: return launch_cfg<128, 64, 4, 2>(p, (int)nb, vec, st);   // 8 waves, 32x32 per wave
    case 11: return launch_cfg<64, 128, 2, 4>(p, (int)nb, vec, st);   // 8 waves, 32x32 per wave
    case 12: return launch_cfg<256, 128, 8, 2>(p, (int)nb, vec, st);  // 16 waves, 32x64 per wave
    case 13: return launch_cfg<256, 64, 8, 2>(p, (int)nb, vec, st);   // 16 waves, 32x32 per wave
    case 14: return launch_cfg<128, 32, 4, 1>(p, (int)nb, vec, st);
    case 15: return launch_cfg<64, 32, 2, 1>(p, (int)nb, vec, st);    // 2 waves
    default: return SMX_EINVAL;
  }
}
