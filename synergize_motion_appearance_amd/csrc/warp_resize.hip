// A7 fused backward warp (flow resize + bilinear gather + occlusion), bilinear resize,
// 2x2 average pool and the strided anti-alias downsample -- gfx950, NHWC.
//
// Warp layout: one pixel is served by C/4 consecutive lanes (float4 = 16 B per lane), so a
// wave64 covers 64*4/C pixels and every bilinear tap is one fully coalesced C*4-byte run
// (256 B .. 1 KiB) -- the NHWC layout turns the "gather" into four contiguous row reads.
// The flow / occlusion taps are read by the first lane of each pixel group and broadcast
// with a wavefront shuffle.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smx.h"
#include "smx_common.h"

namespace {

// align_corners=True source index + weights (ATen area_pixel_compute_source_index)
__device__ __forceinline__ void ac_src(int o, int in, int out, int& i0, int& i1, float& l0, float& l1) {
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float real = scale * o;
  i0 = (int)real; if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = fminf(fmaxf(real - i0, 0.f), 1.f); l0 = 1.f - l1;
}

template <int LPP>  // lanes per pixel = C/4 (power of two, <= 64)
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ feat, long long feat_bs,
                                                   const float* __restrict__ flow, const float* __restrict__ occ,
                                                   float* __restrict__ out, long long npix, int H, int W, int C,
                                                   int Hf, int Wf) {
  const long long gt = blockIdx.x * 256LL + threadIdx.x;
  const long long pix = gt / LPP; const int sub = (int)(gt % LPP);
  const bool live = pix < npix;
  const long long pp = live ? pix : npix - 1;
  const int HW = H * W; const int b = (int)(pp / HW); const int rem = (int)(pp - (long long)b * HW);
  const int y = rem / W, x = rem - y * W;
  float gx = 0.f, gy = 0.f, oc = 1.f;
  if (sub == 0) {
    const float* fb = flow + (long long)b * Hf * Wf * 2;
    if (Hf == H && Wf == W) {
      gx = fb[(y * Wf + x) * 2]; gy = fb[(y * Wf + x) * 2 + 1];
      if (occ) oc = occ[(long long)b * Hf * Wf + y * Wf + x];
    } else {
      int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
      ac_src(y, Hf, H, y0, y1, ly0, ly1); ac_src(x, Wf, W, x0, x1, lx0, lx1);
      const float2 f00 = *reinterpret_cast<const float2*>(fb + (y0 * Wf + x0) * 2);
      const float2 f01 = *reinterpret_cast<const float2*>(fb + (y0 * Wf + x1) * 2);
      const float2 f10 = *reinterpret_cast<const float2*>(fb + (y1 * Wf + x0) * 2);
      const float2 f11 = *reinterpret_cast<const float2*>(fb + (y1 * Wf + x1) * 2);
      gx = ly0 * (lx0 * f00.x + lx1 * f01.x) + ly1 * (lx0 * f10.x + lx1 * f11.x);
      gy = ly0 * (lx0 * f00.y + lx1 * f01.y) + ly1 * (lx0 * f10.y + lx1 * f11.y);
      if (occ) {
        const float* ob = occ + (long long)b * Hf * Wf;
        oc = ly0 * (lx0 * ob[y0 * Wf + x0] + lx1 * ob[y0 * Wf + x1]) + ly1 * (lx0 * ob[y1 * Wf + x0] + lx1 * ob[y1 * Wf + x1]);
      }
    }
  }
  if (LPP > 1) {
    const int src = (threadIdx.x & 63) & ~(LPP - 1);
    gx = __shfl(gx, src, 64); gy = __shfl(gy, src, 64); oc = __shfl(oc, src, 64);
  }
  if (!live) return;
  // grid_sample, bilinear, zeros padding, align_corners=True
  const float ix = ((gx + 1.f) / 2.f) * (W - 1), iy = ((gy + 1.f) / 2.f) * (H - 1);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = ix - fx, ty = iy - fy;
  const float wnw = (1.f - tx) * (1.f - ty), wne = tx * (1.f - ty), wsw = (1.f - tx) * ty, wse = tx * ty;
  const float* fbase = feat + (long long)b * feat_bs + sub * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto tap = [&](int yy, int xx, float w) {
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const float4 v = *reinterpret_cast<const float4*>(fbase + ((long long)yy * W + xx) * C);
      acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
    }
  };
  // NaN / huge coordinates: the comparisons below fail and every tap is skipped (zeros), as in ATen
  if (ix > -2.f && ix < (float)W + 1.f && iy > -2.f && iy < (float)H + 1.f) {
    tap(y0, x0, wnw); tap(y0, x0 + 1, wne); tap(y0 + 1, x0, wsw); tap(y0 + 1, x0 + 1, wse);
  }
  if (occ) { acc.x *= oc; acc.y *= oc; acc.z *= oc; acc.w *= oc; }
  *reinterpret_cast<float4*>(out + pix * C + sub * 4) = acc;
}

__global__ __launch_bounds__(256) void resize_ac_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                        long long total, int Hin, int Win, int Hout, int Wout, int C) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long p = i / C;
    const int ox = (int)(p % Wout); p /= Wout; const int oy = (int)(p % Hout); const int b = (int)(p / Hout);
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    ac_src(oy, Hin, Hout, y0, y1, ly0, ly1); ac_src(ox, Win, Wout, x0, x1, lx0, lx1);
    const float* xb = x + (long long)b * Hin * Win * ldx + c;
    const float v = ly0 * (lx0 * xb[((long long)y0 * Win + x0) * ldx] + lx1 * xb[((long long)y0 * Win + x1) * ldx]) +
                    ly1 * (lx0 * xb[((long long)y1 * Win + x0) * ldx] + lx1 * xb[((long long)y1 * Win + x1) * ldx]);
    y[(((long long)b * Hout + oy) * Wout + ox) * ldy + c] = v;
  }
}

__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                       long long total, int Hin, int Win, int C) {
  const int Ho = Hin / 2, Wo = Win / 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long p = i / C;
    const int ox = (int)(p % Wo); p /= Wo; const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
    const float* xb = x + (((long long)b * Hin + 2 * oy) * Win + 2 * ox) * ldx + c;
    const float v = (xb[0] + xb[ldx] + xb[(long long)Win * ldx] + xb[(long long)(Win + 1) * ldx]) * 0.25f;
    y[(((long long)b * Ho + oy) * Wo + ox) * ldy + c] = v;
  }
}

__global__ __launch_bounds__(256) void antialias_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        float* __restrict__ out, int ldo, long long total, int C, int H,
                                                        int W, int K, int step) {
  const int Ho = (H + step - 1) / step, Wo = (W + step - 1) / step, ka = K / 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long p = i / C;
    const int ox = (int)(p % Wo); p /= Wo; const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
    const float* ib = img + ((long long)b * C + c) * H * W; const float* wc = w + c * K * K;
    float acc = 0.f;
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * step + ky - ka; if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < K; ++kx) {
        const int ix = ox * step + kx - ka; if (ix < 0 || ix >= W) continue;
        acc += ib[iy * W + ix] * wc[ky * K + kx];
      }
    }
    out[(((long long)b * Ho + oy) * Wo + ox) * ldo + c] = acc;
  }
}

inline int grid_for(long long total) { int g = smx_cdiv(total, 256); return g > 16384 ? 16384 : (g < 1 ? 1 : g); }

}  // namespace

extern "C" int smx_warp_nhwc_f32(const float* feat, int feat_batch, const float* flow, const float* occ,
                                 float* out, int B, int H, int W, int C, int Hf, int Wf, void* stream) {
  if (!feat || !flow || !out || B <= 0 || H <= 1 || W <= 1 || Hf <= 1 || Wf <= 1) return SMX_EINVAL;
  if (feat_batch != 1 && feat_batch != B) return SMX_EINVAL;
  const int lpp = C / 4;
  if (C % 4 != 0 || lpp < 1 || lpp > 64 || (lpp & (lpp - 1)) != 0) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long long npix = (long long)B * H * W; const long long feat_bs = feat_batch == 1 ? 0 : (long long)H * W * C;
  dim3 grid(smx_cdiv(npix * lpp, 256)), block(256);
#define SMX_WARP(L) hipLaunchKernelGGL(warp_kernel<L>, grid, block, 0, st, feat, feat_bs, flow, occ, out, npix, H, W, C, Hf, Wf)
  switch (lpp) {
    case 1: SMX_WARP(1); break; case 2: SMX_WARP(2); break; case 4: SMX_WARP(4); break; case 8: SMX_WARP(8); break;
    case 16: SMX_WARP(16); break; case 32: SMX_WARP(32); break; default: SMX_WARP(64); break;
  }
#undef SMX_WARP
  return smx_launch_status();
}

extern "C" int smx_resize_bilinear_ac_nhwc_f32(const float* x, int ldx, float* y, int ldy, int B, int Hin, int Win,
                                               int Hout, int Wout, int C, void* stream) {
  if (!x || !y || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C <= 0 || ldx < C || ldy < C) return SMX_EINVAL;
  const long long total = (long long)B * Hout * Wout * C;
  hipLaunchKernelGGL(resize_ac_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, total, Hin, Win, Hout, Wout, C);
  return smx_launch_status();
}

extern "C" int smx_avgpool2_nhwc_f32(const float* x, int ldx, float* y, int ldy, int B, int Hin, int Win, int C, void* stream) {
  if (!x || !y || B <= 0 || Hin < 2 || Win < 2 || (Hin & 1) || (Win & 1) || C <= 0 || ldx < C || ldy < C) return SMX_EINVAL;
  const long long total = (long long)B * (Hin / 2) * (Win / 2) * C;
  hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, total, Hin, Win, C);
  return smx_launch_status();
}

extern "C" int smx_antialias_down_f32(const float* img_nchw, const float* w, float* out, int ldo, int B, int C,
                                      int H, int W, int K, int step, void* stream) {
  if (!img_nchw || !w || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || !(K & 1) || step <= 0 || ldo < C) return SMX_EINVAL;
  const long long total = (long long)B * ((H + step - 1) / step) * ((W + step - 1) / step) * C;
  hipLaunchKernelGGL(antialias_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img_nchw, w, out, ldo, total, C, H, W, K, step);
  return smx_launch_status();
}
