// Training (BASELINE configs[4] "perceptual loss"): the pieces of the reference's MultiScalePyramidPerceptualLoss
// (losses/losses.py:293-387) that are not convolutions, NHWC fp32 for gfx950:
//   * AntiAliasInterpolation2d (:341-387): zero-pad (ka, ka), depthwise Gaussian K x K (the same kernel for every channel), keep every
//     `step`-th output -- computed only at the kept outputs; its adjoint as a gather over the (<= ceil(K/step)^2) outputs an input reaches;
//   * the 2 x 2 / stride-2 max pooling of the VGG19 feature stack (torchvision cfg "E") and its backward (gradient to the FIRST maximum
//     of the window in scan order, like ATen's saved indices);
//   * per-channel affine (the (x - mean) / std input normalisation of archs/vgg_arch.py:203; its backward is the same kernel with shift 0).
// All HBM-bound streaming kernels: float4 along the channel axis where the channel count allows, one thread per output element otherwise.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "smx.h"
#include "smx_common.h"

namespace {

inline int grid_for(long long total) { int g = smx_cdiv(total, 256); return g > 16384 ? 16384 : (g < 1 ? 1 : g); }

__global__ __launch_bounds__(256) void aa_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, float* __restrict__ y, int ldy,
                                                     long long total, int H, int W, int C, int K, int step) {
  const int Ho = (H + step - 1) / step, Wo = (W + step - 1) / step, ka = K / 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long p = i / C;
    const int ox = (int)(p % Wo); p /= Wo; const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
    const float* xb = x + (long long)b * H * W * ldx + c;
    float acc = 0.f;
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * step + ky - ka; if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < K; ++kx) {
        const int ix = ox * step + kx - ka; if (ix < 0 || ix >= W) continue;
        acc += xb[((long long)iy * W + ix) * ldx] * w[ky * K + kx];
      }
    }
    y[(((long long)b * Ho + oy) * Wo + ox) * ldy + c] = acc;
  }
}

// dx[b][iy][ix][c] = sum over kept outputs (oy, ox) with 0 <= iy - oy*step + ka < K (same in x) of w[ky][kx] * gy[b][oy][ox][c]
__global__ __launch_bounds__(256) void aa_bwd_kernel(const float* __restrict__ gy, int ldg, const float* __restrict__ w, float* __restrict__ dx, int ldx,
                                                     long long total, int H, int W, int C, int K, int step) {
  const int Ho = (H + step - 1) / step, Wo = (W + step - 1) / step, ka = K / 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long p = i / C;
    const int ix = (int)(p % W); p /= W; const int iy = (int)(p % H); const int b = (int)(p / H);
    // oy*step in [iy + ka - K + 1, iy + ka]
    int oy0 = (iy + ka - K + 1 + step - 1) / step; if (iy + ka - K + 1 < 0) oy0 = 0;
    int oy1 = (iy + ka) / step; if (oy1 > Ho - 1) oy1 = Ho - 1;
    int ox0 = (ix + ka - K + 1 + step - 1) / step; if (ix + ka - K + 1 < 0) ox0 = 0;
    int ox1 = (ix + ka) / step; if (ox1 > Wo - 1) ox1 = Wo - 1;
    const float* gb = gy + (long long)b * Ho * Wo * ldg + c;
    float acc = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      const int ky = iy - oy * step + ka;
      for (int ox = ox0; ox <= ox1; ++ox) {
        const int kx = ix - ox * step + ka;
        acc += w[ky * K + kx] * gb[((long long)oy * Wo + ox) * ldg];
      }
    }
    dx[(((long long)b * H + iy) * W + ix) * ldx + c] = acc;
  }
}

__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long long total4,
                                                       int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, cq = C >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % cq); long long p = i / cq;
    const int ox = (int)(p % Wo); p /= Wo; const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
    const float* xb = x + (((long long)b * H + 2 * oy) * W + 2 * ox) * ldx + c4 * 4;
    const float4 a = *reinterpret_cast<const float4*>(xb), bq = *reinterpret_cast<const float4*>(xb + ldx);
    const float4 c = *reinterpret_cast<const float4*>(xb + (long long)W * ldx), d = *reinterpret_cast<const float4*>(xb + (long long)(W + 1) * ldx);
    *reinterpret_cast<float4*>(y + (((long long)b * Ho + oy) * Wo + ox) * ldy + c4 * 4) =
        make_float4(fmaxf(fmaxf(a.x, bq.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, bq.y), fmaxf(c.y, d.y)), fmaxf(fmaxf(a.z, bq.z), fmaxf(c.z, d.z)),
                    fmaxf(fmaxf(a.w, bq.w), fmaxf(c.w, d.w)));
  }
}

// one thread per OUTPUT window and 4 channels: the gradient goes to the first maximum in scan order (0,0), (0,1), (1,0), (1,1); the other three get 0
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gy, int ldg,
                                                           float* __restrict__ dx, int ldo, long long total4, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, cq = C >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % cq); long long p = i / cq;
    const int ox = (int)(p % Wo); p /= Wo; const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
    const long long base = ((long long)b * H + 2 * oy) * W + 2 * ox;
    const float* xb = x + base * ldx + c4 * 4;
    float v[4][4];
    const long long offs[4] = {0, 1, (long long)W, (long long)W + 1};
#pragma unroll
    for (int t = 0; t < 4; ++t) { const float4 q = *reinterpret_cast<const float4*>(xb + offs[t] * ldx); v[t][0] = q.x; v[t][1] = q.y; v[t][2] = q.z; v[t][3] = q.w; }
    const float4 gq = *reinterpret_cast<const float4*>(gy + (((long long)b * Ho + oy) * Wo + ox) * ldg + c4 * 4);
    const float g[4] = {gq.x, gq.y, gq.z, gq.w};
    float o[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int arg = 0; float m = v[0][e];
#pragma unroll
      for (int t = 1; t < 4; ++t) if (v[t][e] > m) { m = v[t][e]; arg = t; }
#pragma unroll
      for (int t = 0; t < 4; ++t) o[t][e] = (t == arg) ? g[e] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(dx + (base + offs[t]) * ldo + c4 * 4) = make_float4(o[t][0], o[t][1], o[t][2], o[t][3]);
  }
}

__global__ __launch_bounds__(256) void chan_affine_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ y, int ldy, long long total, int C) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); const long long p = i / C;
    y[p * ldy + c] = x[p * ldx + c] * scale[c] + (shift ? shift[c] : 0.f);
  }
}

}  // namespace

/* y[b][oy][ox][c] = sum_k w[ky][kx] x[b][oy*step + ky - K/2][ox*step + kx - K/2][c] (zero outside), Ho = ceil(H / step); w [K][K], K odd */
extern "C" int smx_antialias_nhwc_f32(const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W, int C, int K, int step, void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || !(K & 1) || step <= 0 || ldx < C || ldy < C) return SMX_EINVAL;
  const long long total = (long long)B * ((H + step - 1) / step) * ((W + step - 1) / step) * C;
  SMX_LAUNCH(aa_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, w, y, ldy, total, H, W, C, K, step);
  return smx_launch_status();
}

/* adjoint of smx_antialias_nhwc_f32: dx [B][H][W][C] (written) from gy [B][Ho][Wo][C] */
extern "C" int smx_antialias_nhwc_bwd_f32(const float* gy, int ldg, const float* w, float* dx, int ldx, int B, int H, int W, int C, int K, int step,
                                          void* stream) {
  if (!gy || !w || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || !(K & 1) || step <= 0 || ldx < C || ldg < C) return SMX_EINVAL;
  const long long total = (long long)B * H * W * C;
  SMX_LAUNCH(aa_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, gy, ldg, w, dx, ldx, total, H, W, C, K, step);
  return smx_launch_status();
}

extern "C" int smx_maxpool2_f32(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, void* stream) {
  if (!x || !y || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || C % 4 || ldx % 4 || ldy % 4 || ldx < C || ldy < C) return SMX_EINVAL;
  if ((((uintptr_t)x) | ((uintptr_t)y)) & 15) return SMX_EINVAL;
  const long long total4 = (long long)B * (H / 2) * (W / 2) * (C / 4);
  SMX_LAUNCH(maxpool2_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, total4, H, W, C);
  return smx_launch_status();
}

/* x: the pooling INPUT (the window maxima are re-derived from it), gy [B][H/2][W/2][C] -> dx [B][H][W][C] (every element written) */
extern "C" int smx_maxpool2_bwd_f32(const float* x, int ldx, const float* gy, int ldg, float* dx, int ldo, int B, int H, int W, int C, void* stream) {
  if (!x || !gy || !dx || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || C % 4 || ldx % 4 || ldg % 4 || ldo % 4 || ldx < C || ldg < C || ldo < C)
    return SMX_EINVAL;
  if ((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)dx)) & 15) return SMX_EINVAL;
  const long long total4 = (long long)B * (H / 2) * (W / 2) * (C / 4);
  SMX_LAUNCH(maxpool2_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, ldx, gy, ldg, dx, ldo, total4, H, W, C);
  return smx_launch_status();
}

/* y[p][c] = x[p][c] * scale[c] + shift[c] (shift may be null) over P pixels */
extern "C" int smx_chan_affine_f32(const float* x, int ldx, const float* scale, const float* shift, float* y, int ldy, int64_t P, int C, void* stream) {
  if (!x || !scale || !y || P <= 0 || C <= 0 || ldx < C || ldy < C) return SMX_EINVAL;
  const long long total = (long long)P * C;
  SMX_LAUNCH(chan_affine_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, scale, shift, y, ldy, total, C);
  return smx_launch_status();
}
