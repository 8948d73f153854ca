// 7x7 / stride 1 convolution heads of the motion estimator in the bf16 configuration (configs[2]): the stacked keypoint + jacobian head
// (archs/keypoint_detector_arch.py:60-86: 7x7 valid, 35 -> 10 + 40 channels here stacked to N = 76) and the mask + occlusion head
// (archs/dense_motion_arch.py:118-161: 7x7 pad 3, 128 -> 16 + 1) on fp32 storage, "bf16x3" arithmetic.
//
// Keypoints, jacobians and the deformation masks stay out of bf16, so through round 3 these two launches ran the fp32
// implicit GEMM in both configurations: 10.3 ms of a 161 ms bf16 step, bound by L2 -> LDS traffic (every input pixel is staged 49 times:
// 30.8 GB for the mask head at B = 300), not by the matrix pipe.  Here:
//   * region-direct: a block owns an 8 x 32 output tile and stages the (8+6) x (32+6) input region ONCE per 16-channel slice; all 49
//     taps read their MFMA operands straight out of it;
//   * every fp32 value is split x = hi + lo with hi = bf16(x), lo = bf16(x - hi) while it is staged (weights: at pack time), and a product
//     is three v_mfma_f32_32x32x16_bf16: w_hi x_hi + w_hi x_lo + w_lo x_hi, fp32 accumulate -- relative error ~2^-17 per product (the
//     dropped w_lo x_lo term and the rounding of the lo parts), 3/16 of the fp32 pipe's time for the same contraction;
//   * weights: fragment-ordered pack [slice][ky][kx][n tile][hi | lo][64 lanes][8] (smx_conv7_bf16x3_pack); one ky row of a slice
//     (7 taps) goes global -> LDS by LDS-DMA, double buffered, one barrier per ky row = 42 NT MFMAs per wave;
//   * region pixels keep 32 B per plane (hi / lo), the 16-B half a lane reads XOR-swizzled by bit 3 of the pixel's column
//     (conflict-free ds_read_b128 for every tap offset: the scheme of conv3x3_bf16_t32.hip);
//   * 4 waves x 2 output rows x 32 pixels x NT n-tiles; accumulators are [n][pixel]; fp32 output with bias.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TH = 8, TW = 32, RH = TH + 6, RW = TW + 6, RPX = RH * RW;   // 14 x 38 = 532 region pixels
constexpr int PLANE_B = RPX * 32;                                         // 17,024 B per plane (hi | lo)

struct C7 {
  const float* x; const bf16_t* wp; const float* bias; float* y;
  int lda, ldc, B, H, W, Cin, N, Ho, Wo, pad, act, tiles_y, tiles_x, nslices;
  const float* whdr;               // F16: {max |w| bits, 1 / (the power of two the weights were scaled by)} in front of the pack
};

typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ float c7_act(float v, int act) {
  switch (act) {
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    case SMX_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case SMX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}
// x = hi + lo (both bf16, RNE)
__device__ __forceinline__ void split8(const float (&f)[8], uint4& hi, uint4& lo) {
  hi = pack8(f);
  float h[8]; unpack8(hi, h);
  const float d[8] = {f[0] - h[0], f[1] - h[1], f[2] - h[2], f[3] - h[3], f[4] - h[4], f[5] - h[5], f[6] - h[6], f[7] - h[7]};
  lo = pack8(d);
}

// x = h1 + h2, both IEEE half (round to nearest even): the f16x3 form's two levels
__device__ __forceinline__ void split8h(const float (&f)[8], uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { h[q] = cvt2_f16(f[2 * q], f[2 * q + 1]); l[q] = cvt2_f16(f[2 * q] - f16lo(h[q]), f[2 * q + 1] - f16hi(h[q])); }
  hi = make_uint4(h[0], h[1], h[2], h[3]); lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// F16 = true: the "f16x3" arithmetic (csrc/winograd_bf3.hip's): two IEEE-half levels per operand, three v_mfma_f32_32x32x16_f16 products -- fp32-grade, so the
// fp32 configuration's heads run here too.  The weights arrive scaled by a power of two (pack header); the input is scaled by the block: a first pass over the
// block's region x ALL channel slices finds its largest |value| (the region comes from L2 the second time), the power of two that puts it into [2, 4) rides on
// every value before the split and is divided out in the epilogue, exactly.
template <int NT, bool F16 = false>
__global__ __launch_bounds__(256, 2) void conv7_bf16x3_kernel(C7 p) {
  constexpr int WROW_B = 7 * NT * 2 * 1024;                               // one ky row of a slice: [kx][nt][hi | lo][1 KB]
  constexpr int NCH = (RPX * 2 + 255) / 256;                              // 8-channel chunks of the region per thread: 5
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Rh = smem;                                               // [RPX][32 B]
  unsigned char* Rl = smem + PLANE_B;
  unsigned char* Ws = smem + 2 * PLANE_B;                                 // [2][WROW_B]
  const unsigned lds_w = (unsigned)(uintptr_t)((lds_void*)Ws);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int bx = bid % p.tiles_x; bid /= p.tiles_x;
  const int by = bid % p.tiles_y; const int img = bid / p.tiles_y;
  const float* __restrict__ X = p.x + (long long)img * p.H * p.W * p.lda;
  const int iy0 = by * TH - p.pad, ix0 = bx * TW - p.pad;                 // input coordinates of region pixel (0, 0)
  float sv = 1.f;                                                         // F16: the block's power-of-two input scale

  // ---- region staging: chunk = (pixel, 8-channel half); fp32 -> registers -> hi / lo planes --------------------------------------
  float4 ra[NCH], rb[NCH];
  auto load_region = [&](int c0) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int item = tid + 256 * k;
      const int px = item >> 1, half = item & 1;
      const int ry = px / RW, rx = px - ry * RW;
      const int iy = iy0 + ry, ix = ix0 + rx;
      const int c = c0 + 8 * half;
      const bool ok = item < RPX * 2 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const float* src = X + ((long long)iy * p.W + ix) * p.lda + c;
      ra[k] = (ok && c + 3 < p.Cin) ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[k] = (ok && c + 7 < p.Cin) ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_region = [&]() {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int item = tid + 256 * k;
      if (item < RPX * 2) {
        const int px = item >> 1, half = item & 1;
        const int rx = px % RW;
        const float f[8] = {ra[k].x * sv, ra[k].y * sv, ra[k].z * sv, ra[k].w * sv, rb[k].x * sv, rb[k].y * sv, rb[k].z * sv, rb[k].w * sv};
        uint4 hi, lo;
        if (F16) split8h(f, hi, lo); else split8(f, hi, lo);
        const int off = px * 32 + ((half ^ ((rx >> 3) & 1)) << 4);
        *reinterpret_cast<uint4*>(Rh + off) = hi;
        *reinterpret_cast<uint4*>(Rl + off) = lo;
      }
    }
  };
  // ---- weights: one ky row of a slice = WROW_B contiguous bytes of the pack, 1 KB per DMA instruction, round-robin over the waves ----
  auto issue_w = [&](int step, int buf) {                                 // step = slice * 7 + ky
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wp) + (F16 ? 16 : 0) + (long long)step * WROW_B + lane * 16;
#pragma unroll
    for (int q = 0; q < (7 * NT * 2 + 3) / 4; ++q) {
      const int i = wave + 4 * q;
      if (i < 7 * NT * 2) glds16(src + i * 1024, lds_w + (unsigned)(buf * WROW_B + i * 1024));
    }
  };

  if (F16) {                                                              // the block's input scale (see the kernel's head comment)
    float tm = 0.f;
    for (int s0 = 0; s0 < p.nslices; ++s0) {
      load_region(s0 * 16);
#pragma unroll
      for (int k = 0; k < NCH; ++k)
        tm = fmaxf(tm, fmaxf(fmaxf(fmaxf(fabsf(ra[k].x), fabsf(ra[k].y)), fmaxf(fabsf(ra[k].z), fabsf(ra[k].w))), fmaxf(fmaxf(fabsf(rb[k].x), fabsf(rb[k].y)), fmaxf(fabsf(rb[k].z), fabsf(rb[k].w)))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tm = fmaxf(tm, __shfl_xor(tm, o, 64));
    float* red = reinterpret_cast<float*>(Ws);                            // (the weight stage is not in use yet)
    if (lane == 0) red[wave] = tm;
    __syncthreads();
    const float bm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    sv = (bm > 0.f && bm < 3.0e38f) ? __builtin_amdgcn_ldexpf(1.f, 2 - __builtin_amdgcn_frexp_expf(bm)) : 1.f;
  }
  f32x16 acc[2][NT];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][j][q] = 0.f;

  // pixel operand: lane l <-> pixel column (l & 31) of the wave's output row, 8-channel half l >> 5
  const int pc = lane & 31, hh = lane >> 5;
  const int nsteps = p.nslices * 7;
  load_region(0);
  issue_w(0, 0);
  for (int step = 0; step < nsteps; ++step) {
    const int s = step / 7, ky = step - s * 7, buf = step & 1;
    if (ky == 0) {                                                         // new slice: every wave is past the old region (barrier at the end of the last step)
      store_region();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // this step's weight row has landed (and the region prefetch, if one was issued)
    __syncthreads();
    if (step + 1 < nsteps) issue_w(step + 1, buf ^ 1);
    if (ky == 4 && s + 1 < p.nslices) load_region((s + 1) * 16);          // consumed at the next slice's first step
    const unsigned char* wb = Ws + buf * WROW_B + lane * 16;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      bf16x8 xh[2], xl[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int col = pc + kx;
        const int off = ((2 * wave + r + ky) * RW + col) * 32 + ((hh ^ ((col >> 3) & 1)) << 4);
        xh[r] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Rh + off));
        xl[r] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Rl + off));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const uint4 whu = *reinterpret_cast<const uint4*>(wb + ((kx * NT + j) * 2) * 1024), wlu = *reinterpret_cast<const uint4*>(wb + ((kx * NT + j) * 2 + 1) * 1024);
        if (F16) {                                                         // smallest products first, the two rows' accumulators alternating
#pragma unroll
          for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int r = 0; r < 2; ++r)
              acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, pr == 1 ? wlu : whu), __builtin_bit_cast(f16x8_t, pr == 0 ? xl[r] : xh[r]), acc[r][j], 0, 0, 0);
        } else {
          const bf16x8 wh = __builtin_bit_cast(bf16x8, whu), wl = __builtin_bit_cast(bf16x8, wlu);
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[r], acc[r][j], 0, 0, 0);
            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[r], acc[r][j], 0, 0, 0);
            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[r], acc[r][j], 0, 0, 0);
          }
        }
      }
    }
    if (ky == 6) __syncthreads();                                          // the region is rewritten at the top of the next step
  }
  // ---- epilogue: lane (pixel pc, hh) holds channels 32 j + 8 g + 4 hh + (0..3) of its pixel ----------------------------------------------
  float* __restrict__ Y = p.y + (long long)img * p.Ho * p.Wo * p.ldc;
  const float osc = F16 ? p.whdr[1] / sv : 1.f;                           // undo the two power-of-two scales (exact)
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int oy = by * TH + 2 * wave + r, ox = bx * TW + pc;
    if (oy < p.Ho && ox < p.Wo) {
      float* yp = Y + ((long long)oy * p.Wo + ox) * p.ldc;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n = 32 * j + (q & 3) + 8 * (q >> 2) + 4 * hh;
          if (n < p.N) yp[n] = c7_act((F16 ? acc[r][j][q] * osc : acc[r][j][q]) + (p.bias ? p.bias[n] : 0.f), p.act);
        }
    }
  }
}

// ---- the fp32 configuration's form: the same region-direct scheme on v_mfma_f32_32x32x2_f32 (exact fp32 products) --------------------------
// The implicit GEMM stages every input pixel 49 times (L2 -> LDS bound: 6.0 + 4.1 ms per step for the two heads against 3.1 + 2.2 ms of
// matrix time); here the region of a 16-channel slice is staged once as [pixel][16 floats] (the 16-B chunk XOR-swizzled by (column >> 2) & 3:
// conflict-free ds_read_b128 over 16 consecutive pixels) and one ds_read_b128 feeds four MFMAs through the free k pairing (lanes 0-31
// hold channels 8 q + t, lanes 32-63 channels 8 q + 4 + t of step t); weights: one ky row per step by LDS-DMA, [kx][nt][q][64 lanes][4].
constexpr int PLANE_F_B = RPX * 64;                                       // 34,048 B: [pixel][16 floats]
template <int NT>
__global__ __launch_bounds__(256, 2) void conv7_f32_kernel(C7 p) {
  constexpr int WROW_B = 7 * NT * 2 * 1024;
  constexpr int NCH = (RPX * 4 + 255) / 256;                              // float4 chunks of the region per thread: 9
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Rg = smem;
  unsigned char* Ws = smem + PLANE_F_B;
  const unsigned lds_w = (unsigned)(uintptr_t)((lds_void*)Ws);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int bx = bid % p.tiles_x; bid /= p.tiles_x;
  const int by = bid % p.tiles_y; const int img = bid / p.tiles_y;
  const float* __restrict__ X = p.x + (long long)img * p.H * p.W * p.lda;
  const int iy0 = by * TH - p.pad, ix0 = bx * TW - p.pad;
  float4 rr[NCH];
  auto load_region = [&](int c0) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int item = tid + 256 * k;
      const int px = item >> 2, ch = item & 3;
      const int ry = px / RW, rx = px - ry * RW;
      const int iy = iy0 + ry, ix = ix0 + rx;
      const int c = c0 + 4 * ch;
      const bool ok = item < RPX * 4 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c + 3 < p.Cin;
      rr[k] = ok ? *reinterpret_cast<const float4*>(X + ((long long)iy * p.W + ix) * p.lda + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_region = [&]() {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int item = tid + 256 * k;
      if (item < RPX * 4) {
        const int px = item >> 2, ch = item & 3, rx = px % RW;
        *reinterpret_cast<float4*>(Rg + px * 64 + ((ch ^ ((rx >> 2) & 3)) << 4)) = rr[k];
      }
    }
  };
  auto issue_w = [&](int step, int buf) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wp) + (long long)step * WROW_B + lane * 16;
#pragma unroll
    for (int q = 0; q < (7 * NT * 2 + 3) / 4; ++q) {
      const int i = wave + 4 * q;
      if (i < 7 * NT * 2) glds16(src + i * 1024, lds_w + (unsigned)(buf * WROW_B + i * 1024));
    }
  };
  f32x16 acc[2][NT];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][j][q] = 0.f;
  const int pc = lane & 31, hh = lane >> 5;
  const int nsteps = p.nslices * 7;
  load_region(0);
  issue_w(0, 0);
  for (int step = 0; step < nsteps; ++step) {
    const int s = step / 7, ky = step - s * 7, buf = step & 1;
    if (ky == 0) store_region();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (step + 1 < nsteps) issue_w(step + 1, buf ^ 1);
    if (ky == 4 && s + 1 < p.nslices) load_region((s + 1) * 16);
    const unsigned char* wb = Ws + buf * WROW_B + lane * 16;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float4 xv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int col = pc + kx;
          xv[r] = *reinterpret_cast<const float4*>(Rg + ((2 * wave + r + ky) * RW + col) * 64 + (((2 * q + hh) ^ ((col >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float4 wv = *reinterpret_cast<const float4*>(wb + ((kx * NT + j) * 2 + q) * 1024);
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, xv[r].x, acc[r][j], 0, 0, 0);
            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, xv[r].y, acc[r][j], 0, 0, 0);
            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, xv[r].z, acc[r][j], 0, 0, 0);
            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, xv[r].w, acc[r][j], 0, 0, 0);
          }
        }
      }
    }
    if (ky == 6) __syncthreads();
  }
  float* __restrict__ Y = p.y + (long long)img * p.Ho * p.Wo * p.ldc;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int oy = by * TH + 2 * wave + r, ox = bx * TW + pc;
    if (oy < p.Ho && ox < p.Wo) {
      float* yp = Y + ((long long)oy * p.Wo + ox) * p.ldc;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n = 32 * j + (q & 3) + 8 * (q >> 2) + 4 * hh;
          if (n < p.N) yp[n] = c7_act(acc[r][j][q] + (p.bias ? p.bias[n] : 0.f), p.act);
        }
    }
  }
}

// w [N][7][7][Cin] fp32 -> [slice][ky][kx][nt][q][64 lanes][4] fp32: lane l holds row 32 nt + (l & 31), channels 16 s + 8 q + 4 (l >> 5) + 0..3
__global__ __launch_bounds__(256) void conv7_f32_pack_kernel(const float* __restrict__ w, float4* __restrict__ wp, int N, int Cin, int NT, int nslices) {
  const long long total = (long long)nslices * 49 * NT * 2 * 64;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    long long f = i >> 6;
    const int q = (int)(f & 1); f >>= 1;
    const int nt = (int)(f % NT); f /= NT;
    const int tap = (int)(f % 49); const int s = (int)(f / 49);
    const int n = 32 * nt + (lane & 31), c0 = 16 * s + 8 * q + 4 * (lane >> 5);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (n < N && c0 + e < Cin) ? w[((long long)n * 49 + tap) * Cin + c0 + e] : 0.f;
    wp[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <int NT>
int c7f_launch(const C7& p, hipStream_t st) {
  constexpr int LDS = PLANE_F_B + 2 * 7 * NT * 2 * 1024;
  SMX_HIP(smx_max_dynamic_lds((const void*)conv7_f32_kernel<NT>, LDS));
  SMX_LAUNCH(conv7_f32_kernel<NT>, dim3((unsigned)((long long)p.B * p.tiles_y * p.tiles_x)), dim3(256), LDS, st, p);
  return smx_launch_status();
}

// w [N][7][7][Cin] fp32 (k contiguous: the forward layout of ops.Conv) -> [slice][ky][kx][nt][hi | lo][64 lanes][8] bf16
__global__ __launch_bounds__(256) void conv7_pack_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int N, int Cin, int NT, int nslices) {
  const long long total = (long long)nslices * 49 * NT * 64;             // one thread per (slice, tap, nt, lane): writes hi and lo
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    long long f = i >> 6;
    const int nt = (int)(f % NT); f /= NT;
    const int tap = (int)(f % 49); const int s = (int)(f / 49);
    const int n = 32 * nt + (lane & 31), c0 = 16 * s + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (n < N && c0 + e < Cin) ? w[((long long)n * 49 + tap) * Cin + c0 + e] : 0.f;
    uint4 hi, lo; split8(v, hi, lo);
    const long long o = ((((long long)s * 49 + tap) * NT + nt) * 2) * 64 + lane;
    wp[o] = hi; wp[o + 64] = lo;
  }
}

template <int NT, bool F16 = false>
int c7_launch(const C7& p, hipStream_t st) {
  constexpr int LDS = 2 * PLANE_B + 2 * 7 * NT * 2 * 1024;
  SMX_HIP(smx_max_dynamic_lds((const void*)conv7_bf16x3_kernel<NT, F16>, LDS));
  SMX_LAUNCH((conv7_bf16x3_kernel<NT, F16>), dim3((unsigned)((long long)p.B * p.tiles_y * p.tiles_x)), dim3(256), LDS, st, p);
  return smx_launch_status();
}

// the f16x3 pack: max |w| (bits, atomicMax) -> the power of two that brings it into [2^11, 2^12) -> two IEEE-half levels of w x that scale in the hi / lo slots of
// the same fragment layout; 16 header bytes {max bits, 1 / scale} in front
__global__ void conv7_f16_absmax_kernel(const float* __restrict__ w, long long n, unsigned* __restrict__ hdr) {
  unsigned m = 0u;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = max(m, __float_as_uint(w[i]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(hdr, m);
}
__global__ __launch_bounds__(256) void conv7_f16_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ wpb, int N, int Cin, int NT, int nslices) {
  const float wmax = __uint_as_float(reinterpret_cast<const unsigned*>(wpb)[0]);
  const float su = (wmax > 0.f && wmax < 3.0e38f) ? __builtin_amdgcn_ldexpf(1.f, 12 - __builtin_amdgcn_frexp_expf(wmax)) : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(wpb)[1] = 1.f / su;
  uint4* wp = reinterpret_cast<uint4*>(wpb + 16);
  const long long total = (long long)nslices * 49 * NT * 64;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    long long f = i >> 6;
    const int nt = (int)(f % NT); f /= NT;
    const int tap = (int)(f % 49); const int s = (int)(f / 49);
    const int n = 32 * nt + (lane & 31), c0 = 16 * s + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (n < N && c0 + e < Cin) ? w[((long long)n * 49 + tap) * Cin + c0 + e] * su : 0.f;
    uint4 hi, lo; split8h(v, hi, lo);
    const long long o = ((((long long)s * 49 + tap) * NT + nt) * 2) * 64 + lane;
    wp[o] = hi; wp[o + 64] = lo;
  }
}

}  // namespace

extern "C" long long smx_conv7_bf16x3_pack_elems(int Cin, int N) {
  if (Cin <= 0 || N <= 0 || N > 96) return -1;
  return (long long)((Cin + 15) / 16) * 49 * ((N + 31) / 32) * 2 * 512;
}

extern "C" int smx_conv7_bf16x3_pack(const float* w, void* wp, int Cin, int N, void* stream) {
  if (!w || !wp || smx_conv7_bf16x3_pack_elems(Cin, N) < 0 || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  const int NT = (N + 31) / 32, ns = (Cin + 15) / 16;
  const long long total = (long long)ns * 49 * NT * 64;
  int g = smx_cdiv(total, 256); if (g > 4096) g = 4096;
  SMX_LAUNCH(conv7_pack_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w, (uint4*)wp, N, Cin, NT, ns);
  return smx_launch_status();
}

extern "C" int smx_conv7_bf16x3_f32(const float* x, int lda, const void* wp, const float* bias, float* y, int ldc, int B, int H, int W, int Cin,
                                    int N, int pad, int act, void* stream) {
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 4 || N <= 0 || N > 96 || (pad != 0 && pad != 3)) return SMX_EINVAL;
  if (lda < Cin || lda % 4 || ldc < N || ((uintptr_t)x & 15) || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  if (act != SMX_ACT_NONE && act != SMX_ACT_RELU && act != SMX_ACT_LRELU02 && act != SMX_ACT_SIGMOID) return SMX_EINVAL;
  C7 p;
  p.x = x; p.wp = (const bf16_t*)wp; p.whdr = nullptr; p.bias = bias; p.y = y; p.lda = lda; p.ldc = ldc; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.N = N;
  p.pad = pad; p.act = act; p.Ho = H + 2 * pad - 6; p.Wo = W + 2 * pad - 6;
  if (p.Ho <= 0 || p.Wo <= 0) return SMX_EINVAL;
  p.tiles_y = smx_cdiv(p.Ho, TH); p.tiles_x = smx_cdiv(p.Wo, TW); p.nslices = (Cin + 15) / 16;
  if ((long long)B * p.tiles_y * p.tiles_x > 2147483647LL || (long long)H * W * lda > 2147483647LL) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int NT = (N + 31) / 32;
  return NT == 1 ? c7_launch<1>(p, st) : NT == 2 ? c7_launch<2>(p, st) : c7_launch<3>(p, st);
}

/* The same heads in the "f16x3" arithmetic (two IEEE-half levels per operand, three v_mfma_f32_32x32x16_f16 products: fp32-grade -- the form the fp32 configuration
 * uses for its big launches).  wp = smx_conv7_f16_pack: 16 header bytes + the layout of smx_conv7_bf16x3_pack (2 x smx_conv7_bf16x3_pack_elems + 16 bytes); the weights
 * are scaled by a power of two chosen on the device, the input by the block (a first pass over its region finds the largest |value|). */
extern "C" int smx_conv7_f16_pack(const float* w, void* wp, int Cin, int N, void* stream) {
  if (!w || !wp || smx_conv7_bf16x3_pack_elems(Cin, N) < 0 || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  const int NT = (N + 31) / 32, ns = (Cin + 15) / 16;
  SMX_HIP(hipMemsetAsync(wp, 0, 16, (hipStream_t)stream));
  const long long n = (long long)N * 49 * Cin, total = (long long)ns * 49 * NT * 64;
  int gb = smx_cdiv(n, 256); if (gb > 2048) gb = 2048;
  SMX_LAUNCH(conv7_f16_absmax_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, w, n, (unsigned*)wp);
  int g = smx_cdiv(total, 256); if (g > 4096) g = 4096;
  SMX_LAUNCH(conv7_f16_pack_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)wp, N, Cin, NT, ns);
  return smx_launch_status();
}

extern "C" int smx_conv7_f16_f32(const float* x, int lda, const void* wp, const float* bias, float* y, int ldc, int B, int H, int W, int Cin,
                                 int N, int pad, int act, void* stream) {
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 4 || N <= 0 || N > 96 || (pad != 0 && pad != 3)) return SMX_EINVAL;
  if (lda < Cin || lda % 4 || ldc < N || ((uintptr_t)x & 15) || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  if (act != SMX_ACT_NONE && act != SMX_ACT_RELU && act != SMX_ACT_LRELU02 && act != SMX_ACT_SIGMOID) return SMX_EINVAL;
  C7 p;
  p.x = x; p.wp = (const bf16_t*)wp; p.whdr = (const float*)wp; p.bias = bias; p.y = y; p.lda = lda; p.ldc = ldc; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.N = N;
  p.pad = pad; p.act = act; p.Ho = H + 2 * pad - 6; p.Wo = W + 2 * pad - 6;
  if (p.Ho <= 0 || p.Wo <= 0) return SMX_EINVAL;
  p.tiles_y = smx_cdiv(p.Ho, TH); p.tiles_x = smx_cdiv(p.Wo, TW); p.nslices = (Cin + 15) / 16;
  if ((long long)B * p.tiles_y * p.tiles_x > 2147483647LL || (long long)H * W * lda > 2147483647LL) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int NT = (N + 31) / 32;
  return NT == 1 ? c7_launch<1, true>(p, st) : NT == 2 ? c7_launch<2, true>(p, st) : c7_launch<3, true>(p, st);
}

/* the fp32 configuration's form of the same heads: exact fp32 products on v_mfma_f32_32x32x2_f32, region-direct; wp from smx_conv7_f32_pack
 * (smx_conv7_bf16x3_pack_elems(Cin, N) / 2 floats: the two packs have the same byte size) */
extern "C" int smx_conv7_f32_pack(const float* w, float* wp, int Cin, int N, void* stream) {
  if (!w || !wp || smx_conv7_bf16x3_pack_elems(Cin, N) < 0 || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  const int NT = (N + 31) / 32, ns = (Cin + 15) / 16;
  const long long total = (long long)ns * 49 * NT * 2 * 64;
  int g = smx_cdiv(total, 256); if (g > 4096) g = 4096;
  SMX_LAUNCH(conv7_f32_pack_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w, (float4*)wp, N, Cin, NT, ns);
  return smx_launch_status();
}

extern "C" int smx_conv7_f32(const float* x, int lda, const float* wp, const float* bias, float* y, int ldc, int B, int H, int W, int Cin,
                             int N, int pad, int act, void* stream) {
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 4 || N <= 0 || N > 96 || (pad != 0 && pad != 3)) return SMX_EINVAL;
  if (lda < Cin || lda % 4 || ldc < N || ((uintptr_t)x & 15) || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  if (act != SMX_ACT_NONE && act != SMX_ACT_RELU && act != SMX_ACT_LRELU02 && act != SMX_ACT_SIGMOID) return SMX_EINVAL;
  C7 p;
  p.x = x; p.wp = (const bf16_t*)wp; p.whdr = nullptr; p.bias = bias; p.y = y; p.lda = lda; p.ldc = ldc; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.N = N;
  p.pad = pad; p.act = act; p.Ho = H + 2 * pad - 6; p.Wo = W + 2 * pad - 6;
  if (p.Ho <= 0 || p.Wo <= 0) return SMX_EINVAL;
  p.tiles_y = smx_cdiv(p.Ho, TH); p.tiles_x = smx_cdiv(p.Wo, TW); p.nslices = (Cin + 15) / 16;
  if ((long long)B * p.tiles_y * p.tiles_x > 2147483647LL || (long long)H * W * lda > 2147483647LL) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int NT = (N + 31) / 32;
  return NT == 1 ? c7f_launch<1>(p, st) : NT == 2 ? c7f_launch<2>(p, st) : c7f_launch<3>(p, st);
}
