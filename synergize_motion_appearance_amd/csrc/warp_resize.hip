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
#include <stdlib.h>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

namespace {

// align_corners=True source index + weights (ATen area_pixel_compute_source_index)
__device__ __forceinline__ void ac_src(int o, int in, int out, int& i0, int& i1, float& l0, float& l1) {
#pragma clang fp contract(off)   // scale*o must not fuse into the subtraction below: every kernel (and ATen) sees the same source index
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float real = scale * o;
  i0 = (int)real; if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = fminf(fmaxf(real - i0, 0.f), 1.f); l0 = 1.f - l1;
}

// LPP = lanes per pixel = C/4 (power of two, <= 64); PPT = pixels per thread.  A wave's life is a
// dependent chain (flow taps -> shuffle -> 4 feature taps -> store), so one pixel per thread is
// latency-bound (measured 3.6 TB/s at s=256); PPT independent pixels per thread put 4*PPT
// feature loads in flight per lane.
// T = storage type of the features (float | bf16_t); flow / occlusion and all coordinate math are fp32 in both.
template <typename T, int LPP, int PPT>
__global__ __launch_bounds__(256) void warp_kernel(const T* __restrict__ feat, long long feat_bs,
                                                   const float* __restrict__ flow, const float* __restrict__ occ,
                                                   T* __restrict__ out, long long npix, int H, int W, int C,
                                                   int Hf, int Wf, int chunks_per_img, int nframes) {
  constexpr int PPB = 256 / LPP;                                     // pixels per block per pass
  const int sub = threadIdx.x % LPP;
  const int HW = H * W;
  // Block order: each XCD (block b -> XCD b%8, private 4 MiB L2) gets a contiguous range of logical
  // chunks, ordered (spatial chunk, frame) with the FRAME fastest: the B frames that warp the same
  // broadcast source region run back to back on one L2, so the source features are fetched over
  // the fabric once per region instead of once per frame (the source map alone is 4x an L2).
  // (all pixel indices are 32-bit: 64-bit integer division is a ~150-instruction software routine)
  int logical = blockIdx.x;
  if (chunks_per_img > 0) {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int region = lin / nframes, fr = lin - region * nframes;
    logical = fr * chunks_per_img + region;
  }
  const int pix0 = logical * PPB * PPT + threadIdx.x / LPP;
  float gx[PPT], gy[PPT], oc[PPT]; int bb[PPT]; bool live[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int pix = pix0 + i * PPB;
    live[i] = pix < (int)npix;
    const int pp = live[i] ? pix : (int)npix - 1;
    const int b = pp / HW; const int rem = pp - b * HW;
    const int y = rem / W, x = rem - y * W;
    bb[i] = b;
    float fxv = 0.f, fyv = 0.f, ov = 1.f;
    const float* fb = flow + (long long)b * Hf * Wf * 2;
    const float* ob = occ ? occ + (long long)b * Hf * Wf : nullptr;
    if (Hf == H && Wf == W) {
      if (sub == 0) {
        fxv = fb[(y * Wf + x) * 2]; fyv = fb[(y * Wf + x) * 2 + 1];
        if (ob) ov = ob[y * Wf + x];
      }
    } else if (LPP >= 16) {
      // lane-parallel flow / occlusion resize: lanes 0-3 of the pixel group fetch the 4 taps of
      // flow.x, 4-7 of flow.y, 8-11 of the occlusion -- ONE load instruction for the 12 values
      // instead of 12 serial scalar loads on one lane -- then a 4-lane shuffle reduction.
      int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
      ac_src(y, Hf, H, y0, y1, ly0, ly1); ac_src(x, Wf, W, x0, x1, lx0, lx1);
      const int tapi = sub & 3, what = sub >> 2;                     // tap (00,01,10,11), quantity (fx, fy, occ)
      const int ty_ = (tapi >> 1) ? y1 : y0, tx_ = (tapi & 1) ? x1 : x0;
      const float wt = ((tapi >> 1) ? ly1 : ly0) * ((tapi & 1) ? lx1 : lx0);
      float val = 0.f;
      if (what < 2) val = fb[(ty_ * Wf + tx_) * 2 + what];
      else if (what == 2 && ob) val = ob[ty_ * Wf + tx_];
      // ATen's association: ly0*(lx0*v00 + lx1*v01) + ly1*(lx0*v10 + lx1*v11)
      float part = ((tapi & 1) ? lx1 : lx0) * val;
      part += __shfl_xor(part, 1, 64);                               // row sums (lx0*v_0 + lx1*v_1)
      part *= (tapi >> 1) ? ly1 : ly0;
      part += __shfl_xor(part, 2, 64);
      (void)wt;
      fxv = part;                                                    // lanes 0-3: gx, 4-7: gy, 8-11: occ
    } else if (sub == 0) {
      int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
      ac_src(y, Hf, H, y0, y1, ly0, ly1); ac_src(x, Wf, W, x0, x1, lx0, lx1);
      const float2 f00 = *reinterpret_cast<const float2*>(fb + (y0 * Wf + x0) * 2);
      const float2 f01 = *reinterpret_cast<const float2*>(fb + (y0 * Wf + x1) * 2);
      const float2 f10 = *reinterpret_cast<const float2*>(fb + (y1 * Wf + x0) * 2);
      const float2 f11 = *reinterpret_cast<const float2*>(fb + (y1 * Wf + x1) * 2);
      fxv = ly0 * (lx0 * f00.x + lx1 * f01.x) + ly1 * (lx0 * f10.x + lx1 * f11.x);
      fyv = ly0 * (lx0 * f00.y + lx1 * f01.y) + ly1 * (lx0 * f10.y + lx1 * f11.y);
      if (ob) ov = ly0 * (lx0 * ob[y0 * Wf + x0] + lx1 * ob[y0 * Wf + x1]) + ly1 * (lx0 * ob[y1 * Wf + x0] + lx1 * ob[y1 * Wf + x1]);
    }
    gx[i] = fxv; gy[i] = fyv; oc[i] = ov;
  }
  {
    const int src = (threadIdx.x & 63) & ~(LPP - 1);
    const bool par = LPP >= 16 && !(Hf == H && Wf == W);
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      if (par) {
        const float g0 = __shfl(gx[i], src, 64), g1 = __shfl(gx[i], src + 4, 64), g2 = __shfl(gx[i], src + 8, 64);
        gx[i] = g0; gy[i] = g1; oc[i] = occ ? g2 : 1.f;
      } else if (LPP > 1) {
        gx[i] = __shfl(gx[i], src, 64); gy[i] = __shfl(gy[i], src, 64); oc[i] = __shfl(oc[i], src, 64);
      }
    }
  }
  // grid_sample, bilinear, zeros padding, align_corners=True: issue all 4*PPT taps, then blend
  float4 v[PPT][4]; float w[PPT][4];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const float ix = ((gx[i] + 1.f) / 2.f) * (W - 1), iy = ((gy[i] + 1.f) / 2.f) * (H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const float tx = ix - fx, ty = iy - fy;
    // NaN / huge coordinates: every comparison fails and all taps are skipped (zeros), as in ATen
    const bool sane = ix > -2.f && ix < (float)W + 1.f && iy > -2.f && iy < (float)H + 1.f;
    const int x0 = sane ? (int)fx : -4, y0 = sane ? (int)fy : -4;
    w[i][0] = (1.f - tx) * (1.f - ty); w[i][1] = tx * (1.f - ty); w[i][2] = (1.f - tx) * ty; w[i][3] = tx * ty;
    const T* fbase = feat + (long long)bb[i] * feat_bs + sub * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
      v[i][k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live[i] && yy >= 0 && yy < H && xx >= 0 && xx < W)
        v[i][k] = St<T>::ld4(fbase + ((long long)yy * W + xx) * C);
    }
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    if (!live[i]) continue;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc.x += v[i][k].x * w[i][k]; acc.y += v[i][k].y * w[i][k]; acc.z += v[i][k].z * w[i][k]; acc.w += v[i][k].w * w[i][k];
    }
    if (occ) { acc.x *= oc[i]; acc.y *= oc[i]; acc.z *= oc[i]; acc.w *= oc[i]; }
    St<T>::st4(out + (long long)(pix0 + i * PPB) * C + sub * 4, acc);
  }
}

// Row-chunk variant for the big launches (LPP >= 16, W a multiple of the block's 4*PPB pixels).
// The per-pixel coordinate work (flow / occlusion bilinear resize, two integer divisions, the
// align_corners index math: ~150 VALU instructions) used to be repeated by all C/4 lanes of a pixel
// and cost more issue slots than the 16 tap loads + 4 stores it feeds (measured at s=256, B=30:
// 236 us -> 166 us; a 4-row tall tile instead of a row chunk measured within 3 % either way).  Here a block owns 4*PPB consecutive pixels of ONE image row, so frame,
// row and x-base are block-uniform (scalar ALU); lane j < NPW of each wave computes the sampling
// position of the wave's j-th pixel ONCE and the pixel's lane group fetches it with three
// wavefront shuffles.  Arithmetic and association order are those of the scalar path of warp_kernel.
// NTS: non-temporal output stores (the output stream is write-once: the verdict's "write-rate fix" experiment, knob `warp_nt`)
typedef float f32x4nt __attribute__((ext_vector_type(4)));
typedef unsigned u32x2nt __attribute__((ext_vector_type(2)));
typedef unsigned u32x4nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_nt(float* p, float4 v) { __builtin_nontemporal_store(f32x4nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4nt*>(p)); }
__device__ __forceinline__ void st4_nt(bf16_t* p, float4 v) { const uint2 q = pack4(v.x, v.y, v.z, v.w); __builtin_nontemporal_store(u32x2nt{q.x, q.y}, reinterpret_cast<u32x2nt*>(p)); }
template <typename T, int LPP, bool NTS = false>
__global__ __launch_bounds__(256) void warp_rows_kernel(const T* __restrict__ feat, long long feat_bs,
                                                        const float* __restrict__ flow, const float* __restrict__ occ,
                                                        T* __restrict__ out, int H, int W, int C, int Hf, int Wf,
                                                        int chunks_per_img, int nframes) {
  constexpr int PPT = 4, PPB = 256 / LPP, GPW = 64 / LPP, NPW = GPW * PPT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPP, g = lane / LPP;
  // (XCD, region, frame) block order -- see warp_kernel
  const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int region = lin / nframes, b = lin - region * nframes;
  const int cpr = W / (PPB * PPT);
  const int y = region / cpr, xb = (region - y * cpr) * (PPB * PPT);
  (void)chunks_per_img;

  float cix = 0.f, ciy = 0.f, coc = 1.f;
  if (lane < NPW) {
    const int pass = lane / GPW, gg = lane % GPW;
    const int x = xb + pass * PPB + wave * GPW + gg;
    const float* fb = flow + (long long)b * Hf * Wf * 2;
    const float* ob = occ ? occ + (long long)b * Hf * Wf : nullptr;
    float fxv, fyv, ov = 1.f;
    if (Hf == H && Wf == W) {
      fxv = fb[(y * Wf + x) * 2]; fyv = fb[(y * Wf + x) * 2 + 1];
      if (ob) ov = ob[y * Wf + x];
    } else {
      int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
      ac_src(y, Hf, H, y0, y1, ly0, ly1); ac_src(x, Wf, W, x0, x1, lx0, lx1);
      const float2 f00 = *reinterpret_cast<const float2*>(fb + (y0 * Wf + x0) * 2);
      const float2 f01 = *reinterpret_cast<const float2*>(fb + (y0 * Wf + x1) * 2);
      const float2 f10 = *reinterpret_cast<const float2*>(fb + (y1 * Wf + x0) * 2);
      const float2 f11 = *reinterpret_cast<const float2*>(fb + (y1 * Wf + x1) * 2);
      // unfused, in ATen's association (and bit-identical to warp_kernel's lane-parallel resize)
      auto bil = [&](float v00, float v01, float v10, float v11) {
#pragma clang fp contract(off)   // (HIP's __fmul_rn / __fadd_rn are plain * and + : only the pragma stops the contraction)
        return __fadd_rn(__fmul_rn(ly0, __fadd_rn(__fmul_rn(lx0, v00), __fmul_rn(lx1, v01))),
                         __fmul_rn(ly1, __fadd_rn(__fmul_rn(lx0, v10), __fmul_rn(lx1, v11))));
      };
      fxv = bil(f00.x, f01.x, f10.x, f11.x);
      fyv = bil(f00.y, f01.y, f10.y, f11.y);
      if (ob) ov = bil(ob[y0 * Wf + x0], ob[y0 * Wf + x1], ob[y1 * Wf + x0], ob[y1 * Wf + x1]);
    }
    cix = ((fxv + 1.f) / 2.f) * (W - 1); ciy = ((fyv + 1.f) / 2.f) * (H - 1); coc = ov;
  }
  float4 v[PPT][4]; float w[PPT][4]; float oc[PPT];
  const T* fbase = feat + (long long)b * feat_bs + sub * 4;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int src = i * GPW + g;
    const float ix = __shfl(cix, src, 64), iy = __shfl(ciy, src, 64);
    oc[i] = __shfl(coc, src, 64);
    const float fx = floorf(ix), fy = floorf(iy);
    const float tx = ix - fx, ty = iy - fy;
    const bool sane = ix > -2.f && ix < (float)W + 1.f && iy > -2.f && iy < (float)H + 1.f;
    const int x0 = sane ? (int)fx : -4, y0 = sane ? (int)fy : -4;
    w[i][0] = (1.f - tx) * (1.f - ty); w[i][1] = tx * (1.f - ty); w[i][2] = (1.f - tx) * ty; w[i][3] = tx * ty;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
      v[i][k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (yy >= 0 && yy < H && xx >= 0 && xx < W)
        v[i][k] = St<T>::ld4(fbase + (yy * W + xx) * C);
    }
  }
  T* ob_ = out + ((long long)b * H * W + (long long)y * W + xb + wave * GPW + g) * C + sub * 4;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc.x += v[i][k].x * w[i][k]; acc.y += v[i][k].y * w[i][k]; acc.z += v[i][k].z * w[i][k]; acc.w += v[i][k].w * w[i][k];
    }
    if (occ) { acc.x *= oc[i]; acc.y *= oc[i]; acc.z *= oc[i]; acc.w *= oc[i]; }
    if (NTS) st4_nt(ob_ + i * PPB * C, acc); else St<T>::st4(ob_ + i * PPB * C, acc);
  }
}

// bf16 storage, EIGHT channels (16 B) per lane: the row-chunk kernel above moves 8 B per lane and tap in bf16 -- half the bytes of the fp32
// launch on the same number of lanes, loads and address computations, so it took the same time (317 us at 256x256x64, B=60: 3.2 TB/s
// algorithmic).  Here a pixel is C/8 lanes, a block owns 8 * PPB' pixels... same structure, half the lanes per pixel, twice the pixels per
// block: the taps stay packed (4 dwords) until they are blended.
template <int LPP>
__global__ __launch_bounds__(256) void warp_rows16_kernel(const bf16_t* __restrict__ feat, long long feat_bs,
                                                          const float* __restrict__ flow, const float* __restrict__ occ,
                                                          bf16_t* __restrict__ out, int H, int W, int C, int Hf, int Wf, int nframes, int nts) {
  constexpr int PPT = 4, PPB = 256 / LPP, GPW = 64 / LPP, NPW = GPW * PPT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPP, g = lane / LPP;
  const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int region = lin / nframes, b = lin - region * nframes;
  const int cpr = W / (PPB * PPT);
  const int y = region / cpr, xb = (region - y * cpr) * (PPB * PPT);

  float cix = 0.f, ciy = 0.f, coc = 1.f;
  if (lane < NPW) {
    const int pass = lane / GPW, gg = lane % GPW;
    const int x = xb + pass * PPB + wave * GPW + gg;
    const float* fb = flow + (long long)b * Hf * Wf * 2;
    const float* ob = occ ? occ + (long long)b * Hf * Wf : nullptr;
    float fxv, fyv, ov = 1.f;
    if (Hf == H && Wf == W) {
      fxv = fb[(y * Wf + x) * 2]; fyv = fb[(y * Wf + x) * 2 + 1];
      if (ob) ov = ob[y * Wf + x];
    } else {
      int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
      ac_src(y, Hf, H, y0, y1, ly0, ly1); ac_src(x, Wf, W, x0, x1, lx0, lx1);
      const float2 f00 = *reinterpret_cast<const float2*>(fb + (y0 * Wf + x0) * 2);
      const float2 f01 = *reinterpret_cast<const float2*>(fb + (y0 * Wf + x1) * 2);
      const float2 f10 = *reinterpret_cast<const float2*>(fb + (y1 * Wf + x0) * 2);
      const float2 f11 = *reinterpret_cast<const float2*>(fb + (y1 * Wf + x1) * 2);
      auto bil = [&](float v00, float v01, float v10, float v11) {
#pragma clang fp contract(off)
        return __fadd_rn(__fmul_rn(ly0, __fadd_rn(__fmul_rn(lx0, v00), __fmul_rn(lx1, v01))),
                         __fmul_rn(ly1, __fadd_rn(__fmul_rn(lx0, v10), __fmul_rn(lx1, v11))));
      };
      fxv = bil(f00.x, f01.x, f10.x, f11.x);
      fyv = bil(f00.y, f01.y, f10.y, f11.y);
      if (ob) ov = bil(ob[y0 * Wf + x0], ob[y0 * Wf + x1], ob[y1 * Wf + x0], ob[y1 * Wf + x1]);
    }
    cix = ((fxv + 1.f) / 2.f) * (W - 1); ciy = ((fyv + 1.f) / 2.f) * (H - 1); coc = ov;
  }
  uint4 v[PPT][4]; float w[PPT][4]; float oc[PPT];
  const bf16_t* fbase = feat + (long long)b * feat_bs + sub * 8;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int src = i * GPW + g;
    const float ix = __shfl(cix, src, 64), iy = __shfl(ciy, src, 64);
    oc[i] = __shfl(coc, src, 64);
    const float fx = floorf(ix), fy = floorf(iy);
    const float tx = ix - fx, ty = iy - fy;
    const bool sane = ix > -2.f && ix < (float)W + 1.f && iy > -2.f && iy < (float)H + 1.f;
    const int x0 = sane ? (int)fx : -4, y0 = sane ? (int)fy : -4;
    w[i][0] = (1.f - tx) * (1.f - ty); w[i][1] = tx * (1.f - ty); w[i][2] = (1.f - tx) * ty; w[i][3] = tx * ty;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
      v[i][k] = make_uint4(0u, 0u, 0u, 0u);
      if (yy >= 0 && yy < H && xx >= 0 && xx < W)
        v[i][k] = *reinterpret_cast<const uint4*>(fbase + (yy * W + xx) * C);
    }
  }
  bf16_t* ob_ = out + ((long long)b * H * W + (long long)y * W + xb + wave * GPW + g) * C + sub * 8;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float f[8];
      unpack8(v[i][k], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e] * w[i][k];
    }
    if (occ) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= oc[i];
    }
    { const uint4 pk_ = pack8(acc); if (nts) __builtin_nontemporal_store(u32x4nt{pk_.x, pk_.y, pk_.z, pk_.w}, reinterpret_cast<u32x4nt*>(ob_ + i * PPB * C)); else *reinterpret_cast<uint4*>(ob_ + i * PPB * C) = pk_; }
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void resize_ac_kernel(const TI* __restrict__ x, int ldx, TO* __restrict__ y, int ldy,
                                                        long long total, int Hin, int Win, int Hout, int Wout, int C) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long p = i / C;
    const int ox = (int)(p % Wout); p /= Wout; const int oy = (int)(p % Hout); const int b = (int)(p / Hout);
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    ac_src(oy, Hin, Hout, y0, y1, ly0, ly1); ac_src(ox, Win, Wout, x0, x1, lx0, lx1);
    const TI* xb = x + (long long)b * Hin * Win * ldx + c;
    const float v = ly0 * (lx0 * St<TI>::ld(xb + ((long long)y0 * Win + x0) * ldx) + lx1 * St<TI>::ld(xb + ((long long)y0 * Win + x1) * ldx)) +
                    ly1 * (lx0 * St<TI>::ld(xb + ((long long)y1 * Win + x0) * ldx) + lx1 * St<TI>::ld(xb + ((long long)y1 * Win + x1) * ldx));
    St<TO>::st(y + (((long long)b * Hout + oy) * Wout + ox) * ldy + c, v);
  }
}

// four channels per lane (16 B fp32 / 8 B bf16 per tap) when the channel count and the row strides allow: the scalar form above spends a
// full address computation (two divisions, the align_corners index math) per ELEMENT
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void resize_ac4_kernel(const TI* __restrict__ x, int ldx, TO* __restrict__ y, int ldy,
                                                         long long total4, int Hin, int Win, int Hout, int Wout, int c4n) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n); long long p = i / c4n;
    const int ox = (int)(p % Wout); p /= Wout; const int oy = (int)(p % Hout); const int b = (int)(p / Hout);
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    ac_src(oy, Hin, Hout, y0, y1, ly0, ly1); ac_src(ox, Win, Wout, x0, x1, lx0, lx1);
    const TI* xb = x + (long long)b * Hin * Win * ldx + c4 * 4;
    const float4 a = St<TI>::ld4(xb + ((long long)y0 * Win + x0) * ldx), bq = St<TI>::ld4(xb + ((long long)y0 * Win + x1) * ldx);
    const float4 c = St<TI>::ld4(xb + ((long long)y1 * Win + x0) * ldx), d = St<TI>::ld4(xb + ((long long)y1 * Win + x1) * ldx);
    // the scalar kernel's association, per channel
    const float4 v = make_float4(ly0 * (lx0 * a.x + lx1 * bq.x) + ly1 * (lx0 * c.x + lx1 * d.x), ly0 * (lx0 * a.y + lx1 * bq.y) + ly1 * (lx0 * c.y + lx1 * d.y),
                                 ly0 * (lx0 * a.z + lx1 * bq.z) + ly1 * (lx0 * c.z + lx1 * d.z), ly0 * (lx0 * a.w + lx1 * bq.w) + ly1 * (lx0 * c.w + lx1 * d.w));
    St<TO>::st4(y + (((long long)b * Hout + oy) * Wout + ox) * ldy + c4 * 4, v);
  }
}

// Bilinear (align_corners=True) down-sampling reads 4 taps per OUTPUT pixel: when a per-pixel op precedes
// it (to_context: relu(conv1x1(x)) at 256x256, kept at 64x64 -- appmotioncodebook_arch.py:416-418), the op
// only has to be evaluated at those taps (1/4 of the 256x256 pixels; exact, the op is per pixel).
// gather:  t[b][oy][ox][tap][c] = x[b][y_tap][x_tap][c],  taps (y0,x0) (y0,x1) (y1,x0) (y1,x1)
// combine: y = ly0*(lx0*t0 + lx1*t1) + ly1*(lx0*t2 + lx1*t3)   (resize_ac_kernel's association)
template <typename T>
__global__ __launch_bounds__(256) void resize_taps_gather_kernel(const T* __restrict__ x, int ldx, T* __restrict__ t,
                                                                 long long total4, int Hin, int Win, int Hout, int Wout, int c4n) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n); long long p = i / c4n;
    const int tap = (int)(p & 3); p >>= 2;
    const int ox = (int)(p % Wout); p /= Wout; const int oy = (int)(p % Hout); const int b = (int)(p / Hout);
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    ac_src(oy, Hin, Hout, y0, y1, ly0, ly1); ac_src(ox, Win, Wout, x0, x1, lx0, lx1);
    const int yy = (tap >> 1) ? y1 : y0, xx = (tap & 1) ? x1 : x0;
    St<T>::st4(t + i * 4, St<T>::ld4(x + (((long long)b * Hin + yy) * Win + xx) * ldx + c4 * 4));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void resize_taps_combine_kernel(const T* __restrict__ t, T* __restrict__ y, int ldy,
                                                                  long long total4, int Hin, int Win, int Hout, int Wout, int c4n) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n); long long p = i / c4n;
    const int ox = (int)(p % Wout); const long long q = p / Wout; const int oy = (int)(q % Hout);
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    ac_src(oy, Hin, Hout, y0, y1, ly0, ly1); ac_src(ox, Win, Wout, x0, x1, lx0, lx1);
    const T* tp = t + (p * 4 * c4n + c4) * 4;
    const float4 a = St<T>::ld4(tp), b_ = St<T>::ld4(tp + 4 * c4n), c = St<T>::ld4(tp + 8 * c4n), d = St<T>::ld4(tp + 12 * c4n);
    float4 o;
    o.x = ly0 * (lx0 * a.x + lx1 * b_.x) + ly1 * (lx0 * c.x + lx1 * d.x);
    o.y = ly0 * (lx0 * a.y + lx1 * b_.y) + ly1 * (lx0 * c.y + lx1 * d.y);
    o.z = ly0 * (lx0 * a.z + lx1 * b_.z) + ly1 * (lx0 * c.z + lx1 * d.z);
    o.w = ly0 * (lx0 * a.w + lx1 * b_.w) + ly1 * (lx0 * c.w + lx1 * d.w);
    St<T>::st4(y + p * ldy + c4 * 4, o);
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

namespace {
template <typename T>
int warp_launch(const T* feat, int feat_batch, const float* flow, const float* occ, T* out, int B, int H, int W, int C, int Hf, int Wf,
                void* stream) {
  if (!feat || !flow || !out || B <= 0 || H <= 1 || W <= 1 || Hf <= 1 || Wf <= 1) return SMX_EINVAL;
  if (feat_batch != 1 && feat_batch != B) return SMX_EINVAL;
  const int lpp = C / 4;
  if (C % 4 != 0 || lpp < 1 || lpp > 64 || (lpp & (lpp - 1)) != 0) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long long npix = (long long)B * H * W;
  if (npix * lpp > 2147483647LL) return SMX_EINVAL;
  const long long feat_bs = feat_batch == 1 ? 0 : (long long)H * W * C;
  // 4 pixels per thread once the launch is large (>= 8M lanes); small launches keep 1 for parallelism
  const int ppt = (npix * lpp >= (8LL << 20)) ? 4 : 1;
  dim3 grid(smx_cdiv(npix * lpp, 256 * ppt)), block(256);
  if constexpr (sizeof(T) == 2) {
    // bf16: 16 B per lane (8 channels) -- as many bytes per lane and tap as the fp32 launch moves
    const int l8 = C / 8, rc8 = l8 > 0 ? 256 / l8 * 4 : 0;
    if (C % 8 == 0 && (l8 == 8 || l8 == 16 || l8 == 32) && W % rc8 == 0 && npix / rc8 >= 256 && (long long)H * W * C < (1LL << 31) &&
        ((((uintptr_t)feat) | ((uintptr_t)out)) & 15) == 0 && smx_tune(SMX_TUNE_WARP_ROWS)) {
      dim3 grid8((unsigned)((long long)(H * W) / rc8 * B));
      if (l8 == 8) SMX_LAUNCH((warp_rows16_kernel<8>), grid8, block, 0, st, feat, feat_bs, flow, occ, out, H, W, C, Hf, Wf, B, smx_tune(SMX_TUNE_WARP_NT));
      else if (l8 == 16) SMX_LAUNCH((warp_rows16_kernel<16>), grid8, block, 0, st, feat, feat_bs, flow, occ, out, H, W, C, Hf, Wf, B, smx_tune(SMX_TUNE_WARP_NT));
      else SMX_LAUNCH((warp_rows16_kernel<32>), grid8, block, 0, st, feat, feat_bs, flow, occ, out, H, W, C, Hf, Wf, B, smx_tune(SMX_TUNE_WARP_NT));
      return smx_launch_status();
    }
  }
  // whole row chunks and enough blocks: coordinates computed once per pixel (warp_rows_kernel)
  const int rchunk = 256 / (lpp > 0 ? lpp : 1) * 4;
  if (lpp >= 16 && W % rchunk == 0 && npix / rchunk >= 256 && (long long)H * W * C < (1LL << 31) && smx_tune(SMX_TUNE_WARP_ROWS)) {
    const int cpi2 = (H * W) / rchunk;
    dim3 grid2(cpi2 * B);
    if (smx_tune(SMX_TUNE_WARP_NT)) {
      if (lpp == 16) SMX_LAUNCH((warp_rows_kernel<T, 16, true>), grid2, block, 0, st, feat, feat_bs, flow, occ, out, H, W, C, Hf, Wf, cpi2, B);
      else if (lpp == 32) SMX_LAUNCH((warp_rows_kernel<T, 32, true>), grid2, block, 0, st, feat, feat_bs, flow, occ, out, H, W, C, Hf, Wf, cpi2, B);
      else SMX_LAUNCH((warp_rows_kernel<T, 64, true>), grid2, block, 0, st, feat, feat_bs, flow, occ, out, H, W, C, Hf, Wf, cpi2, B);
    }
    else if (lpp == 16) SMX_LAUNCH((warp_rows_kernel<T, 16>), grid2, block, 0, st, feat, feat_bs, flow, occ, out, H, W, C, Hf, Wf, cpi2, B);
    else if (lpp == 32) SMX_LAUNCH((warp_rows_kernel<T, 32>), grid2, block, 0, st, feat, feat_bs, flow, occ, out, H, W, C, Hf, Wf, cpi2, B);
    else SMX_LAUNCH((warp_rows_kernel<T, 64>), grid2, block, 0, st, feat, feat_bs, flow, occ, out, H, W, C, Hf, Wf, cpi2, B);
    return smx_launch_status();
  }
  const int chunk = 256 / lpp * ppt;                                // pixels per block
  int cpi = ((H * W) % chunk == 0 && smx_tune(SMX_TUNE_WARP_REORDER)) ? (H * W) / chunk : 0;
#define SMX_WARP(L) do { if (ppt == 4) SMX_LAUNCH((warp_kernel<T, L, 4>), grid, block, 0, st, feat, feat_bs, flow, occ, out, npix, H, W, C, Hf, Wf, cpi, B); \
                         else SMX_LAUNCH((warp_kernel<T, L, 1>), grid, block, 0, st, feat, feat_bs, flow, occ, out, npix, H, W, C, Hf, Wf, cpi, B); } while (0)
  switch (lpp) {
    case 1: SMX_WARP(1); break; case 2: SMX_WARP(2); break; case 4: SMX_WARP(4); break; case 8: SMX_WARP(8); break;
    case 16: SMX_WARP(16); break; case 32: SMX_WARP(32); break; default: SMX_WARP(64); break;
  }
#undef SMX_WARP
  return smx_launch_status();
}

template <typename TI, typename TO>
int resize_launch(const TI* x, int ldx, TO* y, int ldy, int B, int Hin, int Win, int Hout, int Wout, int C, void* stream) {
  if (!x || !y || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C <= 0 || ldx < C || ldy < C) return SMX_EINVAL;
  const long long total = (long long)B * Hout * Wout * C;
  if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (((uintptr_t)x) & (4 * sizeof(TI) - 1)) == 0 && (((uintptr_t)y) & (4 * sizeof(TO) - 1)) == 0) {
    SMX_LAUNCH((resize_ac4_kernel<TI, TO>), dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, total / 4, Hin, Win, Hout, Wout, C / 4);
    return smx_launch_status();
  }
  SMX_LAUNCH((resize_ac_kernel<TI, TO>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, total, Hin, Win, Hout, Wout, C);
  return smx_launch_status();
}

template <typename T>
int taps_gather_launch(const T* x, int ldx, T* taps, int B, int Hin, int Win, int Hout, int Wout, int C, void* stream) {
  if (!x || !taps || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C <= 0 || C % 4 != 0 || ldx < C || ldx % 4 != 0) return SMX_EINVAL;
  if (((uintptr_t)x & (4 * sizeof(T) - 1)) || ((uintptr_t)taps & (4 * sizeof(T) - 1))) return SMX_EINVAL;
  const long long total4 = (long long)B * Hout * Wout * 4 * (C / 4);
  SMX_LAUNCH(resize_taps_gather_kernel<T>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, ldx, taps, total4,
             Hin, Win, Hout, Wout, C / 4);
  return smx_launch_status();
}

template <typename T>
int taps_combine_launch(const T* taps, T* y, int ldy, int B, int Hin, int Win, int Hout, int Wout, int C, void* stream) {
  if (!taps || !y || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C <= 0 || C % 4 != 0 || ldy < C || ldy % 4 != 0) return SMX_EINVAL;
  if (((uintptr_t)y & (4 * sizeof(T) - 1)) || ((uintptr_t)taps & (4 * sizeof(T) - 1))) return SMX_EINVAL;
  const long long total4 = (long long)B * Hout * Wout * (C / 4);
  SMX_LAUNCH(resize_taps_combine_kernel<T>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, taps, y, ldy,
             total4, Hin, Win, Hout, Wout, C / 4);
  return smx_launch_status();
}
}  // namespace

extern "C" int smx_warp_nhwc_f32(const float* feat, int feat_batch, const float* flow, const float* occ,
                                 float* out, int B, int H, int W, int C, int Hf, int Wf, void* stream) {
  return warp_launch<float>(feat, feat_batch, flow, occ, out, B, H, W, C, Hf, Wf, stream);
}
extern "C" int smx_warp_nhwc_bf16(const void* feat, int feat_batch, const float* flow, const float* occ,
                                  void* out, int B, int H, int W, int C, int Hf, int Wf, void* stream) {
  return warp_launch<bf16_t>((const bf16_t*)feat, feat_batch, flow, occ, (bf16_t*)out, B, H, W, C, Hf, Wf, stream);
}

extern "C" int smx_resize_bilinear_ac_nhwc_f32(const float* x, int ldx, float* y, int ldy, int B, int Hin, int Win,
                                               int Hout, int Wout, int C, void* stream) {
  return resize_launch<float, float>(x, ldx, y, ldy, B, Hin, Win, Hout, Wout, C, stream);
}
extern "C" int smx_resize_bilinear_ac_nhwc_bf16(const void* x, int ldx, void* y, int ldy, int B, int Hin, int Win,
                                                int Hout, int Wout, int C, void* stream) {
  return resize_launch<bf16_t, bf16_t>((const bf16_t*)x, ldx, (bf16_t*)y, ldy, B, Hin, Win, Hout, Wout, C, stream);
}

extern "C" int smx_resize_taps_gather_f32(const float* x, int ldx, float* taps, int B, int Hin, int Win, int Hout, int Wout,
                                          int C, void* stream) {
  return taps_gather_launch<float>(x, ldx, taps, B, Hin, Win, Hout, Wout, C, stream);
}
extern "C" int smx_resize_taps_gather_bf16(const void* x, int ldx, void* taps, int B, int Hin, int Win, int Hout, int Wout,
                                           int C, void* stream) {
  return taps_gather_launch<bf16_t>((const bf16_t*)x, ldx, (bf16_t*)taps, B, Hin, Win, Hout, Wout, C, stream);
}

extern "C" int smx_resize_taps_combine_f32(const float* taps, float* y, int ldy, int B, int Hin, int Win, int Hout, int Wout,
                                           int C, void* stream) {
  return taps_combine_launch<float>(taps, y, ldy, B, Hin, Win, Hout, Wout, C, stream);
}
extern "C" int smx_resize_taps_combine_bf16(const void* taps, void* y, int ldy, int B, int Hin, int Win, int Hout, int Wout,
                                            int C, void* stream) {
  return taps_combine_launch<bf16_t>((const bf16_t*)taps, (bf16_t*)y, ldy, B, Hin, Win, Hout, Wout, C, stream);
}

extern "C" int smx_avgpool2_nhwc_f32(const float* x, int ldx, float* y, int ldy, int B, int Hin, int Win, int C, void* stream) {
  if (!x || !y || B <= 0 || Hin < 2 || Win < 2 || (Hin & 1) || (Win & 1) || C <= 0 || ldx < C || ldy < C) return SMX_EINVAL;
  const long long total = (long long)B * (Hin / 2) * (Win / 2) * C;
  SMX_LAUNCH(avgpool2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, total, Hin, Win, C);
  return smx_launch_status();
}

extern "C" int smx_antialias_down_f32(const float* img_nchw, const float* w, float* out, int ldo, int B, int C,
                                      int H, int W, int K, int step, void* stream) {
  if (!img_nchw || !w || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || !(K & 1) || step <= 0 || ldo < C) return SMX_EINVAL;
  const long long total = (long long)B * ((H + step - 1) / step) * ((W + step - 1) / step) * C;
  SMX_LAUNCH(antialias_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img_nchw, w, out, ldo, total, C, H, W, K, step);
  return smx_launch_status();
}
