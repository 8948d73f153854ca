// Training-step GEMM side (backward of the convolution / Linear / batched-GEMM family) for gfx950.
//
//   * wgrad_kernel: the WEIGHT gradient of a convolution as a "TN" GEMM on the fp32 MFMA,
//         dW[co][(ky,kx,ci)] = sum_m dY[m][co] * X~[m][(ky,kx,ci)],
//     m over the B*Ho*Wo output pixels (the REDUCTION runs over rows, unlike the forward's NT form), X~ the implicit
//     im2col of the NHWC input (zero pad, stride, nearest-x2) gathered on the fly.  Both operands are staged as
//     [32 pixels][64 columns] LDS tiles (coalesced 16-B global reads along the channel axis) and the MFMA fragments are
//     read "down the columns": lane l supplies column (l & 31) of pixel p (lanes 0-31) and p+8 (lanes 32-63) -- with a
//     row pitch of 68 floats the two half-waves hit disjoint bank halves, so every fragment is ONE conflict-free
//     ds_read_b32.  The pixel axis is split over blockIdx.z (few output tiles, very long reduction): raw partials go
//     to a workspace and wgrad_reduce_kernel adds them IN A FIXED ORDER (deterministic, no atomics) while scattering
//     into the parameter's own layout (OIHW / [out][in] / transposed) and accumulating onto an existing .grad.
//     The same kernel is the batched TN GEMM of the AttnBlock backward (dV = P^T dH, dK = dS^T Q).
//   * colsum: per-column sums over rows (bias gradients, per-channel statistics), two deterministic stages.
//   * pack kernels: parameter layout (OIHW) -> the forward's [Cout][kh][kw][Cin] and the data-gradient's
//     [Cin][kh'][kw'][Cout] (taps flipped): the data gradient is then a FORWARD convolution through gemm_conv /
//     winograd (smx_gemm_conv_f32 with up2 = 2 zero-insertion for stride-2 layers), no separate dgrad kernel.
//   * batched transpose, activation backward.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// operand quads by buffer loads (see gemm_conv.hip, "MODE 2 loader"): SGPR resource + SGPR slice offset + a 32-bit per-thread offset; a quad that must
// read zero (row past the split, padding, column past K) presents an offset past the buffer -- no branch, no 64-bit vector address
constexpr unsigned WG_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_resource(const float* base) {
  const unsigned long long u = (unsigned long long)(uintptr_t)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>((uintptr_t)(((unsigned long long)hi << 32) | lo)), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 wg_fetch(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  return make_float4(v.x, v.y, v.z, v.w);
}

inline int grid_for(long long total) { int g = smx_cdiv(total, 256); return g > 16384 ? 16384 : (g < 1 ? 1 : g); }

struct WG {
  const float* dy; const float* x; float* ws; float* bias_ws;      // bias_ws [nb][msplit][Cout] (nullable): column sums of dy per pixel split
  long long dy_bs, x_bs;
  int M, Cout, K, ldy, ldx;
  int Hin, Win, Cin, Ho, Wo, kh, kw, stride, pad_t, pad_l, up2;
  int is1x1, vec_x, vec_y;
  int msplit, mper, tiles_k;
  int fast;                                                       // host-checked: vector layouts, Cout % 4 == 0, 32-bit byte offsets -> the buffer-load loader
  int adv_y, adv_x;                                              // 32 pixels further along the row-major (oy, ox) walk: 32 / Wo rows and 32 % Wo columns
};

// Block tile TC (output channels) x TKT (k columns), 4 waves as 2 x 2, each wave (TC/2) x (TKT/2) = (TC/64) x (TKT/64) accumulators of 32 x 32.
// Instantiated at 64 x 64 (124 / 128 VGPRs, 34 KB of LDS: four blocks per CU).  128 x 128 tiles (twice the flops per staged byte, half the
// fragment reads per MFMA) were built and measured SLOWER on every layer of the step (two blocks per CU at 256 VGPRs: 38-54 TF against
// 52-69): the kernel is bound by how many independent waves hide the gather latency, not by the L2 -> LDS stream.
// LDS row pitch TC + 4 (TKT + 4): 8 * pitch = 32 mod 64, so rows p and p + 8 (the two half-waves of a fragment read) sit in opposite bank
// halves and every fragment is one conflict-free ds_read_b32.
// BF16: the same tiles, staged in fp32 as they are read, but contracted on v_mfma_f32_32x32x16_bf16 -- each lane rounds its 8 pixels of a
// column to bf16 (RNE) on the way from LDS to the MFMA, fp32 accumulate: what torch.autocast(bfloat16) does to this GEMM (the bias
// gradient keeps summing the unrounded dy).  2 MFMAs per 32-pixel slice and accumulator instead of 16.
template <bool BF16, int TC, int TKT, bool FAST = false>
__global__ __launch_bounds__(256, 4) void wgrad_kernel(WG p) {
  constexpr int PA = TC + 4, PB = TKT + 4;                 // LDS pitches
  constexpr int CA4 = TC / 4, CB4 = TKT / 4;               // float4 columns per staged row
  constexpr int RA = 256 / CA4, RB = 256 / CB4;            // rows covered per staging pass
  constexpr int NA = 32 / RA, NB = 32 / RB;                // staging passes per 32-pixel slice
  constexpr int AI = TC / 64, BJ = TKT / 64;               // accumulators per wave: AI x BJ
  __shared__ __attribute__((aligned(16))) float As[2][32 * PA];
  __shared__ __attribute__((aligned(16))) float Bs[2][32 * PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_k = blockIdx.x % p.tiles_k, tile_c = blockIdx.x / p.tiles_k;
  const int co0 = tile_c * TC, k0 = tile_k * TKT;
  const int g = blockIdx.y, z = blockIdx.z;
  const float* __restrict__ DY = p.dy + (long long)g * p.dy_bs;
  const float* __restrict__ X = p.x + (long long)g * p.x_bs;
  const int m_begin = z * p.mper, m_end = min(p.M, m_begin + p.mper);

  // staging coordinates: this thread's float4 column is fixed per operand; it covers rows ra0 + RA * i (A) and rb0 + RB * i (B)
  const int ca4 = tid % CA4, ra0 = tid / CA4;
  const int cb4 = tid % CB4, rb0 = tid / CB4;
  // the B columns k0 + 4 cb4 + e are fixed for the whole block: resolve (tap, channel) once
  int b_ky[4], b_kx[4], b_cc[4]; bool b_in[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = k0 + cb4 * 4 + e;
    b_in[e] = k < p.K;
    const int tap = b_in[e] ? k / p.Cin : 0;
    b_cc[e] = b_in[e] ? k - tap * p.Cin : 0; b_ky[e] = tap / p.kw; b_kx[e] = tap - b_ky[e] * p.kw;
  }
  const int HoWo = p.Ho * p.Wo;
  const int Hlim = p.up2 ? 2 * p.Hin : p.Hin, Wlim = p.up2 ? 2 * p.Win : p.Win;

  // running output coordinates of this thread's NB staged rows (no integer division per slice: the address arithmetic of the gather was
  // as expensive as the slice's MFMAs): one division when the block starts, then +32 pixels per slice along the (image, oy, ox) walk
  int cy[NB], cx[NB]; long long cbase[NB];
  const long long img_stride = (long long)p.Hin * p.Win * p.ldx;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int m = m_begin + rb0 + RB * i;
    const int img = m / HoWo, rem = m - img * HoWo;
    cy[i] = rem / p.Wo; cx[i] = rem - cy[i] * p.Wo; cbase[i] = (long long)img * img_stride;
  }
  // TWO register sets: the loads of slice s + 2 are issued while slice s is being multiplied (one slice of MFMAs, ~0.4 us, does not
  // cover the latency of a gathered L2 / HBM load; with a single set every slice sat out the rest of it at the store)
  float4 areg0[NA], breg0[NB], areg1[NA], breg1[NB];
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);            // bias gradient: this thread's 4 dy columns summed over its rows (tile_k == 0 blocks)
  const bool do_bias = p.bias_ws != nullptr && tile_k == 0;
  // FAST: resources at the split's first row (dy; x of a 1x1 layer) or at the group's first image (x of a k x k layer); per-thread offsets once
  const __amdgpu_buffer_rsrc_t f_ry = wg_resource(DY + (long long)m_begin * p.ldy);
  const __amdgpu_buffer_rsrc_t f_rx = wg_resource(p.is1x1 ? X + (long long)m_begin * p.ldx : X);
  unsigned f_ya[NA], f_xb[NB]; int f_img[NB];
  if (FAST) {
#pragma unroll
    for (int i = 0; i < NA; ++i) f_ya[i] = (co0 + ca4 * 4 + 3 < p.Cout) ? (unsigned)((ra0 + RA * i) * p.ldy + co0 + ca4 * 4) * 4u : WG_OOB;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      f_xb[i] = (k0 + cb4 * 4 + 3 < p.K) ? (unsigned)((rb0 + RB * i) * p.ldx + k0 + cb4 * 4) * 4u : WG_OOB;
      f_img[i] = (m_begin + rb0 + RB * i) / HoWo;
    }
  }
  auto load_slice = [&](int m0, float4 (&areg)[NA], float4 (&breg)[NB]) {
    if (FAST) {
      const bool tail = m0 + 32 > m_end;                           // wave-uniform: only a split's last slice has rows to blank
      const unsigned sy = (unsigned)(m0 - m_begin) * (unsigned)p.ldy * 4u;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        unsigned vo = f_ya[i];
        if (tail && m0 + ra0 + RA * i >= m_end) vo = WG_OOB;
        const float4 a = wg_fetch(f_ry, vo, sy);
        areg[i] = a;
        if (do_bias) { bsum.x += a.x; bsum.y += a.y; bsum.z += a.z; bsum.w += a.w; }
      }
      if (p.is1x1) {
        const unsigned sx = (unsigned)(m0 - m_begin) * (unsigned)p.ldx * 4u;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          unsigned vo = f_xb[i];
          if (tail && m0 + rb0 + RB * i >= m_end) vo = WG_OOB;
          breg[i] = wg_fetch(f_rx, vo, sx);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          int iy = cy[i] * p.stride - p.pad_t + b_ky[0], ix = cx[i] * p.stride - p.pad_l + b_kx[0];
          const bool ok = b_in[3] && iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim && !(tail && m0 + rb0 + RB * i >= m_end);
          if (p.up2) { iy >>= 1; ix >>= 1; }
          const unsigned off = (unsigned)(((f_img[i] * p.Hin + iy) * p.Win + ix) * p.ldx + b_cc[0]) * 4u;
          breg[i] = wg_fetch(f_rx, ok ? off : WG_OOB, 0u);
          cx[i] += p.adv_x; cy[i] += p.adv_y;
          if (cx[i] >= p.Wo) { cx[i] -= p.Wo; ++cy[i]; }
          while (cy[i] >= p.Ho) { cy[i] -= p.Ho; ++f_img[i]; }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int m = m0 + ra0 + RA * i;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < m_end) {
        const int co = co0 + ca4 * 4;
        if (p.vec_y && co + 3 < p.Cout) {
          a = *reinterpret_cast<const float4*>(DY + (long long)m * p.ldy + co);
        } else {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (co + e < p.Cout) ? DY[(long long)m * p.ldy + co + e] : 0.f;
          a = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      areg[i] = a;
      if (do_bias) { bsum.x += a.x; bsum.y += a.y; bsum.z += a.z; bsum.w += a.w; }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int m = m0 + rb0 + RB * i;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < m_end) {
        if (p.is1x1) {
          const int k = k0 + cb4 * 4;
          if (p.vec_x && k + 3 < p.K) {
            b = *reinterpret_cast<const float4*>(X + (long long)m * p.ldx + k);
          } else {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (k + e < p.K) ? X[(long long)m * p.ldx + k + e] : 0.f;
            b = make_float4(v[0], v[1], v[2], v[3]);
          }
        } else {
          const long long base = cbase[i];
          const int iy0 = cy[i] * p.stride - p.pad_t, ix0 = cx[i] * p.stride - p.pad_l;
          if (p.vec_x && b_in[3] && b_cc[3] == b_cc[0] + 3) {          // the four columns share one tap: one 16-B load
            int iy = iy0 + b_ky[0], ix = ix0 + b_kx[0];
            if (iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim) {
              if (p.up2) { iy >>= 1; ix >>= 1; }
              b = *reinterpret_cast<const float4*>(X + base + ((long long)iy * p.Win + ix) * p.ldx + b_cc[0]);
            }
          } else {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = 0.f;
              if (b_in[e]) {
                int iy = iy0 + b_ky[e], ix = ix0 + b_kx[e];
                if (iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim) {
                  if (p.up2) { iy >>= 1; ix >>= 1; }
                  t = X[base + ((long long)iy * p.Win + ix) * p.ldx + b_cc[e]];
                }
              }
              v[e] = t;
            }
            b = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
      breg[i] = b;
      if (!p.is1x1) {                                            // the walk to the same row of the next slice
        cx[i] += p.adv_x; cy[i] += p.adv_y;
        if (cx[i] >= p.Wo) { cx[i] -= p.Wo; ++cy[i]; }
        while (cy[i] >= p.Ho) { cy[i] -= p.Ho; cbase[i] += img_stride; }
      }
    }
  };
  auto store_slice = [&](int buf, const float4 (&areg)[NA], const float4 (&breg)[NB]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<float4*>(&As[buf][(ra0 + RA * i) * PA + ca4 * 4]) = areg[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(&Bs[buf][(rb0 + RB * i) * PB + cb4 * 4]) = breg[i];
  };

  f32x16 acc[AI][BJ];
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int j = 0; j < BJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int wm = wave >> 1, wn = wave & 1;                 // wave tile: co rows (TC/2) wm.., k columns (TKT/2) wn..
  // fragment base: pixel (l >> 5) * 8 within a 16-pixel group, column (l & 31) of the wave's part
  const int fa = (lane >> 5) * 8 * PA + wm * (TC / 2) + (lane & 31);
  const int fb = (lane >> 5) * 8 * PB + wn * (TKT / 2) + (lane & 31);

  const int nslices = (m_end - m_begin + 31) / 32;
  auto multiply = [&](int buf) {
    const float* as = &As[buf][fa];
    const float* bs = &Bs[buf][fb];
    if constexpr (BF16) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {                          // pixels 16 h + 0..7 (lanes 0-31) and 16 h + 8..15 (lanes 32-63)
        bf16x8 af[AI], bfr[BJ];
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = as[(16 * h + j) * PA + 32 * i];
          af[i] = __builtin_bit_cast(bf16x8, pack8(v));
        }
#pragma unroll
        for (int jj = 0; jj < BJ; ++jj) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = bs[(16 * h + j) * PB + 32 * jj];
          bfr[jj] = __builtin_bit_cast(bf16x8, pack8(v));
        }
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
          for (int jj = 0; jj < BJ; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[jj], acc[i][jj], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int prow = (j & 7) + 16 * (j >> 3);            // pixels prow (lanes 0-31) and prow + 8 (lanes 32-63)
        float af[AI], bfr[BJ];
#pragma unroll
        for (int i = 0; i < AI; ++i) af[i] = as[prow * PA + 32 * i];
#pragma unroll
        for (int jj = 0; jj < BJ; ++jj) bfr[jj] = bs[prow * PB + 32 * jj];
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
          for (int jj = 0; jj < BJ; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[jj], acc[i][jj], 0, 0, 0);
      }
    }
  };
  if (nslices > 0) {
    load_slice(m_begin, areg0, breg0);
    store_slice(0, areg0, breg0);
    if (nslices > 1) load_slice(m_begin + 32, areg0, breg0);        // slice 1 -> set 0, slice 2 -> set 1, slice 3 -> set 0, ...
    __syncthreads();
    for (int s = 0; s < nslices; s += 2) {
      if (s + 2 < nslices) load_slice(m_begin + (s + 2) * 32, areg1, breg1);
      multiply(0);
      if (s + 1 < nslices) store_slice(1, areg0, breg0);
      __syncthreads();
      if (s + 1 >= nslices) break;
      if (s + 3 < nslices) load_slice(m_begin + (s + 3) * 32, areg0, breg0);
      multiply(1);
      if (s + 2 < nslices) store_slice(0, areg1, breg1);
      __syncthreads();
    }
  }
  if (do_bias) {
    // the dy tile streamed through this block anyway: its column sums are the bias gradient of this pixel split.  RA row-lanes
    // (tid / CA4) hold partial sums of the same 4 columns: fixed-order LDS finish (the staging buffers are free now)
    float* red = &As[0][0];
    __syncthreads();
    *reinterpret_cast<float4*>(red + (ra0 * CA4 + ca4) * 4) = bsum;
    __syncthreads();
    if (tid < TC) {
      const int col = tid;                                     // column co0 + col: float4 ca4 = col >> 2, element col & 3
      float t = 0.f;
      for (int r = 0; r < RA; ++r) t += red[(r * CA4 + (col >> 2)) * 4 + (col & 3)];
      if (co0 + col < p.Cout) p.bias_ws[((long long)g * p.msplit + z) * p.Cout + co0 + col] = t;
    }
  }
  // raw partial tile -> ws[(g * msplit + z)][Cout][K]
  float* W = p.ws + ((long long)g * p.msplit + z) * p.Cout * p.K;
#pragma unroll
  for (int jj = 0; jj < BJ; ++jj) {
    const int kcol = k0 + wn * (TKT / 2) + 32 * jj + (lane & 31);
    if (kcol < p.K) {
#pragma unroll
      for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + wm * (TC / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (co < p.Cout) W[(long long)co * p.K + kcol] = acc[i][jj][r];
        }
    }
  }
}

// out (+)= sum_z ws[g][z][co][k], scattered into the destination's layout:
//   layout 0: OIHW parameter  out[co][ci][ky][kx]           (k = (ky*kw + kx)*Cin + ci)
//   layout 1: row-major       out[g][co * ldo + k]          (Linear [out][in]; patch-embedding Linear [out][(p1 p2 c)]; batched GEMM C)
//   layout 2: transposed      out[g][k * ldo + co]
template <int VEC, int NG>
__device__ __forceinline__ void wgrad_reduce_body(float* __restrict__ red_raw, const int bid, const int nblk,
                                                  const float* __restrict__ ws, float* __restrict__ out, long long out_bs,
                                                  int nb, int msplit, int Cout, int K, int Cin, int khw, int layout, int ldo,
                                                  int accumulate, float alpha, const float* __restrict__ bias_ws, float* __restrict__ bias_out) {
  const long long per = (long long)Cout * K, total = per * nb;
  // the bias gradient rides in the same launch (it was a third launch of ~10 us per layer, 450 per step): the LAST block also sums
  // the nb * msplit column-sum partials per channel, in a fixed order
  if (bias_out && bid == nblk - 1) {
    for (int c = threadIdx.x; c < Cout; c += 256) {
      float t = 0.f;
      for (int k = 0; k < nb * msplit; ++k) t += bias_ws[(long long)k * Cout + c];
      t *= alpha;
      bias_out[c] = accumulate ? bias_out[c] + t : t;
    }
  }
  // 128 consecutive output elements per block (32 lanes x float4 = 512 B per partial row), 8 lane groups walking the pixel splits (up to
  // ~100 partial tiles) interleaved, finished through LDS in a FIXED order.  (One thread per element walking all splits alone left a
  // 36,864-element layer on 144 blocks: 21 us per launch; 4-byte loads of 128-B rows: 19 us.)  VEC = 1: the scalar form for layers whose
  // element count per group is not a multiple of 4.
  // NG = 32: 8 lanes x float4 = one 128-B line per partial row and 32 groups over the splits -- the region weight gradient leaves up to 256
  // partials of a 36,864-element layer: four times the blocks, a quarter of the dependent loads per thread.
  constexpr int LG = 256 / NG;
  float (*red)[LG * VEC] = reinterpret_cast<float (*)[LG * VEC]>(red_raw);       // [NG][LG * VEC] <= 1024 floats
  const int l = threadIdx.x % LG, q = threadIdx.x / LG;
  for (long long base = bid * ((long long)LG * VEC); base < total; base += (long long)nblk * (LG * VEC)) {
    const long long i = base + (long long)l * VEC;                // VEC == 4: per % 4 == 0, so the 4 elements share g and are contiguous
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    int g = 0; long long j = 0;
    if (i < total) {
      g = (int)(i / per); j = i - (long long)g * per;
      const float* w = ws + (long long)g * msplit * per + j;
      int zz = q;
      if (VEC == 4) {                        // four partial rows in flight per lane (the region kernel's splits run to 512: one dependent
        for (; zz + 3 * NG < msplit; zz += 4 * NG) { // HBM round trip per split was most of this kernel); the order of the additions is unchanged
          const float4 v0 = *reinterpret_cast<const float4*>(w + (long long)zz * per);
          const float4 v1 = *reinterpret_cast<const float4*>(w + (long long)(zz + NG) * per);
          const float4 v2 = *reinterpret_cast<const float4*>(w + (long long)(zz + 2 * NG) * per);
          const float4 v3 = *reinterpret_cast<const float4*>(w + (long long)(zz + 3 * NG) * per);
          s[0] += v0.x; s[VEC > 1 ? 1 : 0] += v0.y; s[VEC > 2 ? 2 : 0] += v0.z; s[VEC > 3 ? 3 : 0] += v0.w;
          s[0] += v1.x; s[VEC > 1 ? 1 : 0] += v1.y; s[VEC > 2 ? 2 : 0] += v1.z; s[VEC > 3 ? 3 : 0] += v1.w;
          s[0] += v2.x; s[VEC > 1 ? 1 : 0] += v2.y; s[VEC > 2 ? 2 : 0] += v2.z; s[VEC > 3 ? 3 : 0] += v2.w;
          s[0] += v3.x; s[VEC > 1 ? 1 : 0] += v3.y; s[VEC > 2 ? 2 : 0] += v3.z; s[VEC > 3 ? 3 : 0] += v3.w;
        }
      }
      for (; zz < msplit; zz += NG) {
        if (VEC == 4) {
          const float4 v = *reinterpret_cast<const float4*>(w + (long long)zz * per);
          s[0] += v.x; s[VEC > 1 ? 1 : 0] += v.y; s[VEC > 2 ? 2 : 0] += v.z; s[VEC > 3 ? 3 : 0] += v.w;
        } else {
          s[0] += w[(long long)zz * per];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[q][l * VEC + e] = s[e];
    __syncthreads();
    if (q == 0 && i < total) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int c = l * VEC + e;
        float t = 0.f;
#pragma unroll
        for (int o8 = 0; o8 < NG; o8 += 8)
          t += ((red[o8][c] + red[o8 + 1][c]) + (red[o8 + 2][c] + red[o8 + 3][c])) + ((red[o8 + 4][c] + red[o8 + 5][c]) + (red[o8 + 6][c] + red[o8 + 7][c]));
        t *= alpha;
        const long long je = j + e;
        const int co = (int)(je / K), k = (int)(je - (long long)co * K);
        long long o;
        if (layout == 0) { const int tap = k / Cin, ci = k - tap * Cin; o = ((long long)co * Cin + ci) * khw + tap; }
        else if (layout == 1) o = (long long)co * ldo + k;
        else o = (long long)k * ldo + co;
        float* d = out + (long long)g * out_bs + o;
        *d = accumulate ? *d + t : t;
      }
    }
    __syncthreads();
  }
}

template <int VEC, int NG = 8>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long long out_bs,
                                                           int nb, int msplit, int Cout, int K, int Cin, int khw, int layout, int ldo,
                                                           int accumulate, float alpha, const float* __restrict__ bias_ws, float* __restrict__ bias_out) {
  __shared__ float red[1024];
  wgrad_reduce_body<VEC, NG>(red, (int)blockIdx.x, (int)gridDim.x, ws, out, out_bs, nb, msplit, Cout, K, Cin, khw, layout, ldo, accumulate, alpha, bias_ws, bias_out);
}

// Every split reduce of a backward pass in ONE launch (smx_wgrad_reduce_batch): a device table of items sorted by first_block; block b
// finishes the item whose block range holds it exactly as the per-layer launch would have (same block count, same order of additions:
// bit-identical results).  The per-layer reduces were ~460 launches of 5-50 us per training step (7.7 ms), most of it launch latency
// and tails.  Items of one launch must not share an output (the host plan puts a parameter's second call site into the next launch).
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const smx_reduce_item* __restrict__ items, int n) {
  __shared__ float red[1024];
  int lo = 0, hi = n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (items[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
  const smx_reduce_item it = items[lo];
  const int bid = (int)blockIdx.x - it.first_block;
  if (bid >= it.nblocks) return;
  if (it.kind == 2) wgrad_reduce_body<4, 32>(red, bid, it.nblocks, it.ws, it.out, 0, 1, it.msplit, it.Cout, it.K, it.Cin, it.khw, it.layout, it.ldo, it.accumulate, it.alpha, it.bias_ws, it.bias_out);
  else if (it.kind == 1) wgrad_reduce_body<4, 8>(red, bid, it.nblocks, it.ws, it.out, 0, 1, it.msplit, it.Cout, it.K, it.Cin, it.khw, it.layout, it.ldo, it.accumulate, it.alpha, it.bias_ws, it.bias_out);
  else wgrad_reduce_body<1, 8>(red, bid, it.nblocks, it.ws, it.out, 0, 1, it.msplit, it.Cout, it.K, it.Cin, it.khw, it.layout, it.ldo, it.accumulate, it.alpha, it.bias_ws, it.bias_out);
}

// ---- column sums: part[chunk][C] = sum over the chunk's rows; then out[c] (+)= sum_chunk part[chunk][c] (fixed order) ----
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int ld, long long P, int C, int rows_per_chunk,
                                                             float* __restrict__ part) {
  __shared__ float red[256 * 4];
  int cw = 1; while (cw < C && cw < 256) cw <<= 1;
  const int tx = threadIdx.x % cw, ty = threadIdx.x / cw, rpar = 256 / cw;
  const long long r_begin = (long long)blockIdx.x * rows_per_chunk, r_end = min(P, r_begin + rows_per_chunk);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (long long r = r_begin + ty; r < r_end; r += rpar) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int c = tx + q * cw; if (c < C) s[q] += x[r * ld + c]; }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[threadIdx.x * 4 + q] = s[q];
  __syncthreads();
  if (ty == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = tx + q * cw;
      if (c < C) {
        float t = 0.f;
        for (int y = 0; y < rpar; ++y) t += red[(y * cw + tx) * 4 + q];
        part[(long long)blockIdx.x * C + c] = t;
      }
    }
  }
}

__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ part, int nchunk, int C, float* __restrict__ out,
                                                             int accumulate, float alpha) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < nchunk; ++k) s += part[(long long)k * C + c];
  s *= alpha;
  out[c] = accumulate ? out[c] + s : s;
}

// ---- parameter packing ---------------------------------------------------------------------------------------------
// mode 0: OIHW -> [Cout][(ky,kx,ci)]            (the forward operand)
// mode 1: OIHW -> [Cin][(kh-1-ky, kw-1-kx, co)] (the data-gradient operand: transposed channels, flipped taps)
template <typename TO>
__device__ __forceinline__ void pack_weight_elem(const float* __restrict__ w, TO* __restrict__ o, long long i, int Cout, int Cin, int kh, int kw, int mode) {
  // i indexes the OUTPUT so that writes are coalesced
  if (mode == 0) {
    const int K = kh * kw * Cin; const int co = (int)(i / K), k = (int)(i - (long long)co * K);
    const int tap = k / Cin, ci = k - tap * Cin;
    St<TO>::st(o + i, w[((long long)co * Cin + ci) * kh * kw + tap]);
  } else {
    const int K = kh * kw * Cout; const int ci = (int)(i / K), k = (int)(i - (long long)ci * K);
    const int tap = k / Cout, co = k - tap * Cout; const int ky = kh - 1 - tap / kw, kx = kw - 1 - tap % kw;
    St<TO>::st(o + i, w[((long long)co * Cin + ci) * kh * kw + ky * kw + kx]);
  }
}

template <typename TO>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, TO* __restrict__ o, int Cout, int Cin, int kh, int kw, int mode) {
  const long long total = (long long)Cout * Cin * kh * kw;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) pack_weight_elem<TO>(w, o, i, Cout, Cin, kh, kw, mode);
}

// Winograd-domain weights U = G g G^T of F(2x2,3x3) in the fragment order winograd.hip reads ([16 f][N/32][C/8][2][32][4]:
// lane l = 32 hh + r <-> output channel 32 nt + r, input channels 8 s + 4 hh + e), straight from the OIHW parameter:
//   mode 0 (forward):       N = Cout, C = Cin,  g[n][a][b][c] = w[n][c][a][b]
//   mode 1 (data gradient): N = Cin,  C = Cout, g[n][a][b][c] = w[c][n][2-a][2-b]
// so the training step's 3x3 convolutions (forward AND data gradient) run on the fused Winograd kernel with this step's weights
__device__ __forceinline__ void pack_winograd_u_elem(const float* __restrict__ w, float* __restrict__ u, long long i, int Cout, int Cin, int mode) {
  const int N = mode ? Cin : Cout, Cc = mode ? Cout : Cin;
  const int n32 = (N + 31) / 32, c8 = Cc / 8;
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  long long t = i;
  const int e = (int)(t & 3); t >>= 2; const int r = (int)(t & 31); t >>= 5; const int hh = (int)(t & 1); t >>= 1;
  const int s = (int)(t % c8); t /= c8; const int nt = (int)(t % n32); const int f = (int)(t / n32);
  float out = 0.f;
  if (f < 16) {
    const int n = nt * 32 + r, c = 8 * s + 4 * hh + e, fi = f >> 2, fj = f & 3;
    if (n < N) {
      double acc = 0.0;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
          const float g = mode ? w[((long long)c * Cin + n) * 9 + (2 - a) * 3 + (2 - b)] : w[((long long)n * Cin + c) * 9 + a * 3 + b];
          acc += G[fi][a] * (double)g * G[fj][b];
        }
      out = (float)acc;
    }
  }
  u[i] = out;                                             // f >= 16: the prefetch pad behind the last fragment (zeros)
}

__global__ __launch_bounds__(256) void pack_winograd_u_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int mode, long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) pack_winograd_u_elem(w, u, i, Cout, Cin, mode);
}

// 3x3 weights by (row, column) PAIR of the OIHW parameter: one thread reads the pair's nine taps (36 contiguous bytes; a wave reads one
// contiguous run in mode 0) and writes nine outputs, each coalesced across the wave -- the per-element form issued nine strided 4-byte
// loads per pair.  Same values, same destinations as pack_weight_elem.
template <typename TO>
__device__ __forceinline__ void pack_weight_pair9(const float* __restrict__ w, TO* __restrict__ o, long long g, int Cout, int Cin, int mode) {
  // mode 0: g = co * Cin + ci (output [co][tap][ci]);  mode 1: g = ci * Cout + co (output [ci][flipped tap][co])
  int co, ci;
  if (mode == 0) { co = (int)(g / Cin); ci = (int)(g - (long long)co * Cin); }
  else { ci = (int)(g / Cout); co = (int)(g - (long long)ci * Cout); }
  const float* src = w + ((long long)co * Cin + ci) * 9;
  float v[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) v[t] = src[t];
  if (mode == 0) {
    TO* dst = o + (long long)co * 9 * Cin + ci;
#pragma unroll
    for (int t = 0; t < 9; ++t) St<TO>::st(dst + (long long)t * Cin, v[t]);
  } else {
    TO* dst = o + (long long)ci * 9 * Cout + co;
#pragma unroll
    for (int t = 0; t < 9; ++t) St<TO>::st(dst + (long long)t * Cout, v[8 - t]);     // tap t reads (ky, kx) = (2 - t / 3, 2 - t % 3) = source tap 8 - t
  }
}

// The same values by (n, c) GROUP: one thread loads the nine weights once and forms one frequency ROW (4 of the 16 outputs) from them, with the
// per-element function's own expression and order of additions (bit-identical).  The per-element form re-read the nine weights for each
// of the 16 outputs: 250 M elements x 9 strided loads per fp32 training step.  Block b of an item covers groups [64 b, 64 b + 64); the
// block behind the last group writes the 1024 zeros of the prefetch pad.
__device__ __forceinline__ void pack_winograd_u_group(const float* __restrict__ w, float* __restrict__ u, long long g, int fi, int Cout, int Cin, int mode) {
  const int N = mode ? Cin : Cout, Cc = mode ? Cout : Cin;
  const int n32 = (N + 31) / 32, c8 = Cc / 8;
  const long long plane = (long long)n32 * 32 * Cc;
  if (g >= plane) {                                       // pad: 64 groups x 16 zeros
    const long long o = 16 * plane + (g - plane) * 16 + 4 * fi;
    if (g - plane < 64) { u[o] = 0.f; u[o + 1] = 0.f; u[o + 2] = 0.f; u[o + 3] = 0.f; }
    return;
  }
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  long long t = g;
  const int e = (int)(t & 3); t >>= 2; const int r = (int)(t & 31); t >>= 5; const int hh = (int)(t & 1); t >>= 1;
  const int s = (int)(t % c8); const int nt = (int)(t / c8);
  const int n = nt * 32 + r, c = 8 * s + 4 * hh + e;
  float gw[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
      gw[a][b] = (n < N) ? (mode ? w[((long long)c * Cin + n) * 9 + (2 - a) * 3 + (2 - b)] : w[((long long)n * Cin + c) * 9 + a * 3 + b]) : 0.f;
#pragma unroll
  for (int fj = 0; fj < 4; ++fj) {
    float out = 0.f;
    if (n < N) {
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc += G[fi][a] * (double)gw[a][b] * G[fj][b];
      out = (float)acc;
    }
    u[(long long)(4 * fi + fj) * plane + g] = out;
  }
}

// every packing of a step in ONE launch: a table of items (sorted by first_block), block b works on the item whose block range holds it,
// 1024 output elements per block.  The per-layer launches were ~660 per training step at 6-11 us each, most of it launch latency.
__global__ __launch_bounds__(256) void pack_batch_kernel(const smx_pack_item* __restrict__ items, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (items[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
  const smx_pack_item it = items[lo];
  const long long base = (long long)((int)blockIdx.x - it.first_block) * 1024;
  if (it.kind != SMX_PACK_WINOGRAD_U && it.kh == 3 && it.kw == 3) {     // 3x3: one thread per (co, ci) pair; the item's first total / 9 threads work
    const long long g = (base >> 2) + threadIdx.x;        // block b of the item = threads [256 b, 256 b + 256)
    if (g < it.total / 9) {
      if (it.kind == SMX_PACK_F32) pack_weight_pair9<float>(it.w, (float*)it.out, g, it.cout, it.cin, it.mode);
      else pack_weight_pair9<bf16_t>(it.w, (bf16_t*)it.out, g, it.cout, it.cin, it.mode);
    }
    return;
  }
  if (it.kind == SMX_PACK_WINOGRAD_U) {                   // 1024 outputs per block = 64 (n, c) groups x 16 frequencies: 4 threads per group
    pack_winograd_u_group(it.w, (float*)it.out, (base >> 4) + (threadIdx.x >> 2), threadIdx.x & 3, it.cout, it.cin, it.mode);
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long long i = base + e * 256 + threadIdx.x;
    if (i >= it.total) break;
    if (it.kind == SMX_PACK_F32) pack_weight_elem<float>(it.w, (float*)it.out, i, it.cout, it.cin, it.kh, it.kw, it.mode);
    else if (it.kind == SMX_PACK_BF16) pack_weight_elem<bf16_t>(it.w, (bf16_t*)it.out, i, it.cout, it.cin, it.kh, it.kw, it.mode);
    else pack_winograd_u_elem(it.w, (float*)it.out, i, it.cout, it.cin, it.mode);
  }
}

// ---- batched transpose: y[g][c][r] = x[g][r][c] -------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, int ldx, long long x_bs, float* __restrict__ y, int ldy,
                                                        long long y_bs, int R, int C) {
  __shared__ float t[32][33];
  const float* X = x + (long long)blockIdx.z * x_bs; float* Y = y + (long long)blockIdx.z * y_bs;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) if (r0 + j < R && c0 + tx < C) t[j][tx] = X[(long long)(r0 + j) * ldx + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8) if (c0 + j < C && r0 + tx < R) Y[(long long)(c0 + j) * ldy + r0 + tx] = t[tx][j];
}

// ---- activation backward: gx = g * f'(.)  with the reference tensor the derivative is cheapest from -----------------
//   RELU / LRELU02 / SIGMOID: ref = the activation's OUTPUT y;  SWISH / GELU: ref = its INPUT x
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ ref, int ldr,
                                                      float* __restrict__ gx, int ldo, long long P, int C, int act) {
  const long long total = P * C;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C; const int c = (int)(i - r * C);
    const float gv = g[r * ldg + c], v = ref[r * ldr + c];
    float d;
    switch (act) {
      case SMX_ACT_RELU: d = v > 0.f ? 1.f : 0.f; break;
      case SMX_ACT_LRELU02: d = v > 0.f ? 1.f : 0.2f; break;
      case SMX_ACT_SIGMOID: d = v * (1.f - v); break;
      case SMX_ACT_SWISH: { const float s = 1.f / (1.f + expf(-v)); d = s * (1.f + v * (1.f - s)); break; }
      case SMX_ACT_GELU: d = 0.5f * (1.f + erff(v * 0.70710678118654752440f)) + v * 0.39894228040143267794f * expf(-0.5f * v * v); break;
      default: d = 1.f; break;
    }
    gx[r * ldo + c] = gv * d;
  }
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long long P, int C, int act) {
  const long long total = P * C;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C; const int c = (int)(i - r * C);
    const float v = x[r * ldx + c];
    float o;
    switch (act) {
      case SMX_ACT_RELU: o = v > 0.f ? v : 0.f; break;
      case SMX_ACT_LRELU02: o = v > 0.f ? v : 0.2f * v; break;
      case SMX_ACT_SWISH: o = v / (1.f + expf(-v)); break;
      case SMX_ACT_GELU: o = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); break;
      case SMX_ACT_SIGMOID: o = 1.f / (1.f + expf(-v)); break;
      default: o = v; break;
    }
    y[r * ldy + c] = o;
  }
}

// y[.., :C] (ld ldy) += alpha * x[.., :C] (ld ldx): gradient accumulation into channel slices of concat buffers
__global__ __launch_bounds__(256) void axpy_slice_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long long P, int C, float alpha) {
  const long long total = P * C;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C; const int c = (int)(i - r * C);
    y[r * ldy + c] += alpha * x[r * ldx + c];
  }
}

}  // namespace

extern "C" int64_t smx_wgrad_ws_floats(int nb, int M, int Cout, int K, int* msplit_out) {
  if (nb <= 0 || M <= 0 || Cout <= 0 || K <= 0) return 0;
  // 64 x 64 blocks, 4 resident per CU; at least 256 pixels per block, partial workspace <= 64 MB
  const long long tiles = (long long)smx_cdiv(Cout, 64) * smx_cdiv(K, 64) * nb;
  // ONE resident round: floor, not ceil -- 1026 blocks on 1024 slots run as two rounds, the second one with two blocks (measured: the
  // 64 -> 64 @ 256^2 layer took 402 us with 114 splits of 9 tiles)
  const long long slots = 1024;
  long long ms = tiles >= slots ? 1 : slots / tiles;
  const long long max_by_m = (M + 255) / 256;
  if (ms > max_by_m) ms = max_by_m;
  const long long cap = (64LL << 20) / 4 / ((long long)nb * Cout * K);
  if (ms > cap) ms = cap;
  if (ms < 1) ms = 1;
  if (msplit_out) *msplit_out = (int)ms;
  return (int64_t)nb * ms * Cout * K + (int64_t)nb * ms * Cout;      // weight partials + the bias-gradient partials
}

/* the same, told the convolution's geometry: 3x3 / stride 1 / pad 1 layers with 64-multiple channel counts and 32-multiple widths go to
 * the region kernel (train_wgrad_region.hip), which wants more, smaller pixel splits (one block owns all nine taps of a 64 x 64 tile) */
extern "C" int64_t smx_wgrad_conv_ws_floats(int nb, int M, int Cout, int Cin, int Hin, int Win, int Ho, int Wo, int kh, int kw, int stride,
                                            int pad_t, int pad_l, int up2, int* msplit_out) {
  if (nb <= 0 || M <= 0 || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || Ho <= 0 || Wo <= 0) return 0;
  if (M % (Ho * Wo) == 0 && smx_wgrad_region_shape_ok(nb, Cout, Cin, Hin, Win, Ho, Wo, kh, kw, stride, pad_t, pad_l, up2)) {
    const int ms = smx_wgrad_region_split(M, Cout, Cin, Ho, Wo, nullptr);
    if (msplit_out) *msplit_out = ms;
    return (int64_t)ms * Cout * 9 * Cin + (int64_t)ms * Cout;
  }
  return smx_wgrad_ws_floats(nb, M, Cout, kh * kw * Cin, msplit_out);
}

// which reduce form a (workspace, split, size) gets and on how many blocks: shared by the per-layer launch and smx_wgrad_reduce_describe
static int reduce_kind(const float* ws, int nb, int msplit, long long per_g, int* nblk) {
  if (per_g % 4 == 0 && (((uintptr_t)ws) & 15) == 0 && msplit >= 64 && (long long)nb * per_g <= (1LL << 19)) { *nblk = grid_for((long long)nb * per_g * 8); return 2; }
  if (per_g % 4 == 0 && (((uintptr_t)ws) & 15) == 0) { *nblk = grid_for((long long)nb * per_g * 2); return 1; }
  *nblk = grid_for((long long)nb * per_g * 8);
  return 0;
}

static int wgrad_launch(bool bf16, const float* dy, int ldy, int64_t dy_bs, const float* x, int ldx, int64_t x_bs, int nb, int M, int Cout,
                        int Hin, int Win, int Cin, int Ho, int Wo, int kh, int kw, int stride, int pad_t, int pad_l, int up2,
                        float* ws, int msplit, float* out, int64_t out_bs, int layout, int ldo, int accumulate, float alpha,
                        float* bias_out, void* stream) {
  if (!dy || !x || !ws || !out || nb <= 0 || M <= 0 || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return SMX_EINVAL;
  if (msplit < 1 || nb > 65535 || msplit > 65535 || layout < 0 || layout > 2 || ldy < Cout || ldx < Cin) return SMX_EINVAL;
  if (Ho <= 0 || Wo <= 0 || M % (Ho * Wo) != 0 || (up2 != 0 && up2 != 1)) return SMX_EINVAL;
  WG p;
  p.dy = dy; p.x = x; p.ws = ws; p.dy_bs = dy_bs; p.x_bs = x_bs;
  // the bias gradient (column sums of dy) rides along: partials live behind the weight partials in ws
  p.bias_ws = bias_out ? ws + (long long)nb * msplit * Cout * (kh * kw * Cin) : nullptr;
  p.M = M; p.Cout = Cout; p.K = kh * kw * Cin; p.ldy = ldy; p.ldx = ldx;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.kh = kh; p.kw = kw; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l; p.up2 = up2;
  p.is1x1 = (kh == 1 && kw == 1 && stride == 1 && !up2 && pad_t == 0 && pad_l == 0 && Hin == Ho && Win == Wo) ? 1 : 0;
  if (layout != 0 && ldo < (layout == 1 ? p.K : Cout)) return SMX_EINVAL;
  p.vec_x = (Cin % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0 && x_bs % 4 == 0) ? 1 : 0;
  p.vec_y = (ldy % 4 == 0 && (((uintptr_t)dy) & 15) == 0 && dy_bs % 4 == 0) ? 1 : 0;
  p.msplit = msplit;
  p.mper = ((M + msplit - 1) / msplit + 31) / 32 * 32;
  p.adv_y = 32 / Wo; p.adv_x = 32 % Wo;
  hipStream_t st = (hipStream_t)stream;
  p.tiles_k = smx_cdiv(p.K, 64);
  const long long tiles = (long long)smx_cdiv(Cout, 64) * p.tiles_k;
  if (tiles > 2147483647LL) return SMX_EINVAL;
  const dim3 grid((unsigned)tiles, nb, msplit);
  const bool region = smx_wgrad_region_shape_ok(nb, Cout, Cin, Hin, Win, Ho, Wo, kh, kw, stride, pad_t, pad_l, up2) && p.vec_x && p.vec_y &&
                      (((uintptr_t)ws) & 15) == 0 && msplit == smx_wgrad_region_split(M, Cout, Cin, Ho, Wo, nullptr);
  if (region) {
    const int rc = smx_wgrad_region_launch(bf16, dy, ldy, x, ldx, M, Cout, Hin, Win, Cin, Ho, Wo, up2, ws, p.bias_ws, msplit, stream);
    if (rc != SMX_OK) return rc;
  } else {
    // the buffer-load loader: whole quads only (vector layouts, Cout and the taps' channel count multiples of 4) and 32-bit byte offsets
    p.fast = (p.vec_x && p.vec_y && Cout % 4 == 0 && Cin % 4 == 0 && smx_tune(SMX_TUNE_GEMM_LOADER) != 0 &&
              (long long)(p.mper + 64) * ldy * 4 < 0x7fffffffLL && (long long)(p.mper + 64) * ldx * 4 < 0x7fffffffLL &&
              (p.is1x1 || ((long long)(M / (Ho * Wo)) + 1) * Hin * Win * ldx * 4 < 0x7fffffffLL)) ? 1 : 0;
    if (bf16) { if (p.fast) SMX_LAUNCH((wgrad_kernel<true, 64, 64, true>), grid, dim3(256), 0, st, p); else SMX_LAUNCH((wgrad_kernel<true, 64, 64>), grid, dim3(256), 0, st, p); }
    else { if (p.fast) SMX_LAUNCH((wgrad_kernel<false, 64, 64, true>), grid, dim3(256), 0, st, p); else SMX_LAUNCH((wgrad_kernel<false, 64, 64>), grid, dim3(256), 0, st, p); }
  }
  if (accumulate & 2) return nb == 1 ? smx_launch_status() : SMX_EINVAL;      // deferred: the partials stay in ws for smx_wgrad_reduce_batch
  const long long per_g = (long long)Cout * p.K;
  int nblk = 0;
  const int kind = reduce_kind(ws, nb, msplit, per_g, &nblk);
  if (kind == 2)
    SMX_LAUNCH((wgrad_reduce_kernel<4, 32>), dim3(nblk), dim3(256), 0, st, ws, out, (long long)out_bs, nb, msplit,
               Cout, p.K, Cin, kh * kw, layout, ldo, accumulate, alpha, p.bias_ws, bias_out);
  else if (kind == 1)
    SMX_LAUNCH(wgrad_reduce_kernel<4>, dim3(nblk), dim3(256), 0, st, ws, out, (long long)out_bs, nb, msplit,
               Cout, p.K, Cin, kh * kw, layout, ldo, accumulate, alpha, p.bias_ws, bias_out);
  else
    SMX_LAUNCH(wgrad_reduce_kernel<1>, dim3(nblk), dim3(256), 0, st, ws, out, (long long)out_bs, nb, msplit,
               Cout, p.K, Cin, kh * kw, layout, ldo, accumulate, alpha, p.bias_ws, bias_out);
  return smx_launch_status();
}

extern "C" int smx_wgrad_reduce_describe(const float* ws, int msplit, float* out, int Cout, int Cin, int kh, int kw, int layout, int ldo,
                                         int accumulate, float alpha, float* bias_out, smx_reduce_item* item) {
  if (!ws || !out || !item || msplit < 1 || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || layout < 0 || layout > 2) return SMX_EINVAL;
  const int K = kh * kw * Cin;
  item->ws = ws; item->out = out; item->bias_out = bias_out;
  item->bias_ws = bias_out ? ws + (long long)msplit * Cout * K : nullptr;
  item->msplit = msplit; item->Cout = Cout; item->K = K; item->Cin = Cin; item->khw = kh * kw; item->layout = layout; item->ldo = ldo;
  item->accumulate = accumulate & 1; item->alpha = alpha; item->first_block = 0;
  item->kind = reduce_kind(ws, 1, msplit, (long long)Cout * K, &item->nblocks);
  return SMX_OK;
}

extern "C" int smx_wgrad_reduce_batch(const smx_reduce_item* items_dev, int n_items, int n_blocks, void* stream) {
  if (!items_dev || n_items <= 0 || n_blocks <= 0) return SMX_EINVAL;
  SMX_LAUNCH(wgrad_reduce_batch_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n_items);
  return smx_launch_status();
}

extern "C" int smx_wgrad_f32(const float* dy, int ldy, int64_t dy_bs, const float* x, int ldx, int64_t x_bs, int nb, int M, int Cout,
                             int Hin, int Win, int Cin, int Ho, int Wo, int kh, int kw, int stride, int pad_t, int pad_l, int up2,
                             float* ws, int msplit, float* out, int64_t out_bs, int layout, int ldo, int accumulate, float alpha,
                             float* bias_out, void* stream) {
  return wgrad_launch(false, dy, ldy, dy_bs, x, ldx, x_bs, nb, M, Cout, Hin, Win, Cin, Ho, Wo, kh, kw, stride, pad_t, pad_l, up2, ws, msplit, out, out_bs,
                      layout, ldo, accumulate, alpha, bias_out, stream);
}

/* the same contraction on the bf16 MFMA (operands rounded to bf16 on the way to the matrix cores, fp32 accumulate and fp32 result):
 * the weight gradient of the bf16-compute training mode (torch.autocast(bfloat16) semantics for F.conv2d / F.linear backward). */
extern "C" int smx_wgrad_mfma16_f32(const float* dy, int ldy, int64_t dy_bs, const float* x, int ldx, int64_t x_bs, int nb, int M, int Cout,
                                    int Hin, int Win, int Cin, int Ho, int Wo, int kh, int kw, int stride, int pad_t, int pad_l, int up2,
                                    float* ws, int msplit, float* out, int64_t out_bs, int layout, int ldo, int accumulate, float alpha,
                                    float* bias_out, void* stream) {
  return wgrad_launch(true, dy, ldy, dy_bs, x, ldx, x_bs, nb, M, Cout, Hin, Win, Cin, Ho, Wo, kh, kw, stride, pad_t, pad_l, up2, ws, msplit, out, out_bs,
                      layout, ldo, accumulate, alpha, bias_out, stream);
}

static void colsum_chunks(long long P, long long* rows, long long* nchunk) {
  *rows = 256; *nchunk = (P + *rows - 1) / *rows;
  if (*nchunk > 2048) { *rows = (P + 2047) / 2048; *nchunk = (P + *rows - 1) / *rows; }
}

extern "C" int64_t smx_colsum_ws_floats(int64_t P, int C) {
  if (P <= 0 || C <= 0) return 0;
  long long rows, nchunk; colsum_chunks(P, &rows, &nchunk);
  return nchunk * (C < 1024 ? C : 1024);
}

extern "C" int smx_colsum_f32(const float* x, int ld, int64_t P, int C, float* ws, float* out, int accumulate, float alpha, void* stream) {
  if (!x || !ws || !out || P <= 0 || C <= 0 || ld < C) return SMX_EINVAL;
  long long rows, nchunk; colsum_chunks(P, &rows, &nchunk);
  hipStream_t st = (hipStream_t)stream;
  for (int c0 = 0; c0 < C; c0 += 1024) {              // wide rows (the un-patchify bias: p*p*C columns) go in column blocks of 1024
    const int cw = C - c0 < 1024 ? C - c0 : 1024;
    SMX_LAUNCH(colsum_partial_kernel, dim3((unsigned)nchunk), dim3(256), 0, st, x + c0, ld, (long long)P, cw, (int)rows, ws);
    SMX_LAUNCH(partial_reduce_kernel, dim3(smx_cdiv(cw, 256)), dim3(256), 0, st, ws, (int)nchunk, cw, out + c0, accumulate, alpha);
  }
  return smx_launch_status();
}

extern "C" int smx_partial_reduce_f32(const float* part, int nchunk, int C, float* out, int accumulate, float alpha, void* stream) {
  if (!part || !out || nchunk <= 0 || C <= 0) return SMX_EINVAL;
  SMX_LAUNCH(partial_reduce_kernel, dim3(smx_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, part, nchunk, C, out, accumulate, alpha);
  return smx_launch_status();
}

extern "C" int smx_pack_weight_f32(const float* w_oihw, float* packed, int Cout, int Cin, int kh, int kw, int mode, void* stream) {
  if (!w_oihw || !packed || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || (mode != 0 && mode != 1)) return SMX_EINVAL;
  SMX_LAUNCH(pack_weight_kernel<float>, dim3(grid_for((long long)Cout * Cin * kh * kw)), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, Cout, Cin, kh, kw, mode);
  return smx_launch_status();
}

/* the same packing rounded to bfloat16 (RNE) on the way out: the weight operand of the bf16-compute training mode, one launch per layer
 * and step instead of pack + convert */
extern "C" int smx_pack_weight_bf16(const float* w_oihw, void* packed, int Cout, int Cin, int kh, int kw, int mode, void* stream) {
  if (!w_oihw || !packed || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || (mode != 0 && mode != 1)) return SMX_EINVAL;
  SMX_LAUNCH(pack_weight_kernel<bf16_t>, dim3(grid_for((long long)Cout * Cin * kh * kw)), dim3(256), 0, (hipStream_t)stream, w_oihw, (bf16_t*)packed, Cout, Cin,
             kh, kw, mode);
  return smx_launch_status();
}

extern "C" int64_t smx_winograd_u_floats(int N, int C) { return (N > 0 && C > 0) ? 16LL * ((N + 31) / 32) * 32 * C + 1024 : 0; }

extern "C" int smx_pack_winograd_u_f32(const float* w_oihw, float* u, int Cout, int Cin, int mode, void* stream) {
  if (!w_oihw || !u || Cout <= 0 || Cin <= 0 || (mode != 0 && mode != 1)) return SMX_EINVAL;
  const int N = mode ? Cin : Cout, Cc = mode ? Cout : Cin;
  if (Cc % 8 != 0) return SMX_EINVAL;
  const long long total = smx_winograd_u_floats(N, Cc);
  SMX_LAUNCH(pack_winograd_u_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w_oihw, u, Cout, Cin, mode, total);
  return smx_launch_status();
}

extern "C" int smx_pack_batch(const smx_pack_item* items_dev, int n_items, int n_blocks, void* stream) {
  if (!items_dev || n_items <= 0 || n_blocks <= 0) return SMX_EINVAL;
  SMX_LAUNCH(pack_batch_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n_items);
  return smx_launch_status();
}

extern "C" int smx_transpose_f32(const float* x, int ldx, int64_t x_bs, float* y, int ldy, int64_t y_bs, int nb, int R, int C, void* stream) {
  if (!x || !y || nb <= 0 || nb > 65535 || R <= 0 || C <= 0 || ldx < C || ldy < R) return SMX_EINVAL;
  SMX_LAUNCH(transpose_kernel, dim3(smx_cdiv(C, 32), smx_cdiv(R, 32), nb), dim3(256), 0, (hipStream_t)stream, x, ldx, (long long)x_bs, y, ldy, (long long)y_bs, R, C);
  return smx_launch_status();
}

extern "C" int smx_act_bwd_f32(const float* g, int ldg, const float* ref, int ldr, float* gx, int ldo, int64_t P, int C, int act, void* stream) {
  if (!g || !ref || !gx || P <= 0 || C <= 0 || ldg < C || ldr < C || ldo < C) return SMX_EINVAL;
  SMX_LAUNCH(act_bwd_kernel, dim3(grid_for((long long)P * C)), dim3(256), 0, (hipStream_t)stream, g, ldg, ref, ldr, gx, ldo, (long long)P, C, act);
  return smx_launch_status();
}

extern "C" int smx_act_f32(const float* x, int ldx, float* y, int ldy, int64_t P, int C, int act, void* stream) {
  if (!x || !y || P <= 0 || C <= 0 || ldx < C || ldy < C) return SMX_EINVAL;
  SMX_LAUNCH(act_fwd_kernel, dim3(grid_for((long long)P * C)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, (long long)P, C, act);
  return smx_launch_status();
}

extern "C" int smx_axpy_slice_f32(const float* x, int ldx, float* y, int ldy, int64_t P, int C, float alpha, void* stream) {
  if (!x || !y || P <= 0 || C <= 0 || ldx < C || ldy < C) return SMX_EINVAL;
  SMX_LAUNCH(axpy_slice_kernel, dim3(grid_for((long long)P * C)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, (long long)P, C, alpha);
  return smx_launch_status();
}
