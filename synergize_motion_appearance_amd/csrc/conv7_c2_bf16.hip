// BasicMotionEncoder.convf1 (archs/appmotioncodebook_arch.py BasicMotionEncoder: 7x7 / pad 3 convolution of the 2-channel fp32 residual
// flow to 128 channels, ReLU) in the bf16 configuration: fp32 input, bf16 output.
//
// K = 7 * 7 * 2 = 98 and C_in = 2: the implicit GEMM gathered its A tile with scalar 4-byte loads (0.73 ms per call at B = 300 against
// 63 us of output traffic).  Here a block owns an 8 x 32 output tile and keeps the (8+6) x (32+6) x 2 input region in 4 KB of LDS; each
// lane BUILDS its MFMA pixel operand (8 consecutive k = 4 taps x 2 channels) from four 8-byte LDS reads; wave w owns output channels
// [32 w, 32 w + 32) with its weight fragments (7 k-steps, K padded to 112 with zeros) in registers; accumulators are [n][pixel], exchanged
// per output row through a wave-private LDS buffer into 32-byte bf16 stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TH = 8, TW = 32, RH = TH + 6, RW = TW + 6, RPX = RH * RW;   // 14 x 38 region pixels of float2
constexpr int EXP = 36, EX_F = 32 * EXP;

struct C2 {
  const float* x; const bf16_t* wp; const float* bias; bf16_t* y;
  int ldc, B, H, W, N, act, tiles_y, tiles_x;
};

__global__ __launch_bounds__(256, 2) void conv7_c2_kernel(C2 p) {
  __shared__ __attribute__((aligned(16))) float2 Rg[RPX];
  __shared__ __attribute__((aligned(16))) float Ex[4 * EX_F];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int bx = bid % p.tiles_x; bid /= p.tiles_x;
  const int by = bid % p.tiles_y; const int img = bid / p.tiles_y;
  const float2* __restrict__ X = reinterpret_cast<const float2*>(p.x) + (long long)img * p.H * p.W;
  for (int i = tid; i < RPX; i += 256) {
    const int ry = i / RW, rx = i - ry * RW;
    const int iy = by * TH - 3 + ry, ix = bx * TW - 3 + rx;
    Rg[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? X[(long long)iy * p.W + ix] : make_float2(0.f, 0.f);
  }
  const int n0 = (blockIdx.y * 4 + wave) * 32;
  uint4 wf[7];
  {
    const uint4* wp = reinterpret_cast<const uint4*>(p.wp) + (long long)(n0 / 32) * 7 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) wf[ks] = wp[ks * 64];
  }
  const int pc = lane & 31, hh = lane >> 5;
  // region offsets of this lane's taps: k-step ks, half hh -> taps 8 ks + 4 hh + 0..3 (taps >= 49 meet zero weights: clamp to a valid pixel)
  int toff[7][4];
#pragma unroll
  for (int ks = 0; ks < 7; ++ks)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = min(8 * ks + 4 * hh + i, 48);
      const int ky = t / 7, kx = t - 7 * ky;
      toff[ks][i] = ky * RW + kx + pc;
    }
  __syncthreads();
  float* ex = Ex + wave * EX_F;
  const int erow = lane >> 1, eh = lane & 1;
  float bv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bv[e] = p.bias ? p.bias[n0 + 16 * eh + e] : 0.f;
  bf16_t* __restrict__ Y = p.y + (long long)img * p.H * p.W * p.ldc;
#pragma unroll 2
  for (int r = 0; r < TH; ++r) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float2 t = Rg[r * RW + toff[ks][i]]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[ks]), __builtin_bit_cast(bf16x8, pack8(v)), acc, 0, 0, 0);   // [n][pixel]
    }
    // exchange: lane (pixel pc, hh) holds channels 8 g + 4 hh + (0..3) of its pixel -> lane (pixel erow, half eh) stores 16 channels
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(ex + pc * EXP + 8 * g + 4 * hh) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    float o[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(ex + erow * EXP + 16 * eh + 4 * q);
      o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      o[e] += bv[e];
      if (p.act == SMX_ACT_RELU) o[e] = fmaxf(o[e], 0.f);
      else if (p.act == SMX_ACT_LRELU02) o[e] = o[e] > 0.f ? o[e] : 0.2f * o[e];
    }
    const float lo[8] = {o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]}, hi[8] = {o[8], o[9], o[10], o[11], o[12], o[13], o[14], o[15]};
    bf16_t* yp = Y + ((long long)(by * TH + r) * p.W + bx * TW + erow) * p.ldc + n0 + 16 * eh;
    *reinterpret_cast<uint4*>(yp) = pack8(lo);
    *reinterpret_cast<uint4*>(yp + 8) = pack8(hi);
  }
}

// The fp32 configuration's form: v_mfma_f32_32x32x2_f32, one MFMA per tap -- the MFMA's k pair IS the channel pair (lanes 0-31 supply
// channel 0 of tap t at their pixel, lanes 32-63 channel 1), so the pixel operand is one 4-byte LDS read and the wave's 49 weight
// values per lane stay in registers.  fp32 output, 4 x 16-B stores per lane.
__global__ __launch_bounds__(256, 2) void conv7_c2_f32_kernel(const float* __restrict__ x, const float* __restrict__ wpk, const float* __restrict__ bias,
                                                              float* __restrict__ y, int ldc, int H, int W, int act, int tiles_y, int tiles_x) {
  __shared__ __attribute__((aligned(16))) float Rf[RPX * 2];
  __shared__ __attribute__((aligned(16))) float Ex[4 * EX_F];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int bx = bid % tiles_x; bid /= tiles_x;
  const int by = bid % tiles_y; const int img = bid / tiles_y;
  const float2* __restrict__ X = reinterpret_cast<const float2*>(x) + (long long)img * H * W;
  for (int i = tid; i < RPX; i += 256) {
    const int ry = i / RW, rx = i - ry * RW;
    const int iy = by * TH - 3 + ry, ix = bx * TW - 3 + rx;
    const float2 v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? X[(long long)iy * W + ix] : make_float2(0.f, 0.f);
    Rf[2 * i] = v.x; Rf[2 * i + 1] = v.y;
  }
  const int n0 = (blockIdx.y * 4 + wave) * 32;
  float wf[49];                                                           // W[n0 + (l & 31)][tap][c = l >> 5]
  {
    const float* wp = wpk + (long long)(n0 / 32) * 49 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 49; ++t) wf[t] = wp[t * 64];
  }
  const int pc = lane & 31, hh = lane >> 5;
  __syncthreads();
  float* ex = Ex + wave * EX_F;
  const int erow = lane >> 1, eh = lane & 1;
  float bv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bv[e] = bias ? bias[n0 + 16 * eh + e] : 0.f;
  float* __restrict__ Y = y + (long long)img * H * W * ldc;
  for (int r = 0; r < TH; ++r) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const float* rp = Rf + 2 * (r * RW + pc) + hh;
#pragma unroll
    for (int t = 0; t < 49; ++t)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[t], rp[2 * ((t / 7) * RW + (t % 7))], acc, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(ex + pc * EXP + 8 * g + 4 * hh) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    float* yp = Y + ((long long)(by * TH + r) * W + bx * TW + erow) * ldc + n0 + 16 * eh;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 t = *reinterpret_cast<const float4*>(ex + erow * EXP + 16 * eh + 4 * q);
      t.x += bv[4 * q]; t.y += bv[4 * q + 1]; t.z += bv[4 * q + 2]; t.w += bv[4 * q + 3];
      if (act == SMX_ACT_RELU) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
      else if (act == SMX_ACT_LRELU02) { t.x = t.x > 0.f ? t.x : 0.2f * t.x; t.y = t.y > 0.f ? t.y : 0.2f * t.y; t.z = t.z > 0.f ? t.z : 0.2f * t.z; t.w = t.w > 0.f ? t.w : 0.2f * t.w; }
      *reinterpret_cast<float4*>(yp + 4 * q) = t;
    }
  }
}

// w [N][98] fp32 -> [N/32][49 taps][64 lanes]: lane l holds W[32 nt + (l & 31)][tap][l >> 5]
__global__ __launch_bounds__(256) void conv7_c2_f32_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int N) {
  const int total = (N / 32) * 49 * 64;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63, f = i >> 6, t = f % 49, nt = f / 49;
    wp[i] = w[(32 * nt + (lane & 31)) * 98 + 2 * t + (lane >> 5)];
  }
}

// w [N][7][7][2] fp32 (= [N][98], k = tap * 2 + c) -> [N/32][7 k-steps][64 lanes][8] bf16, K padded to 112 with zeros
__global__ __launch_bounds__(256) void conv7_c2_pack_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int N) {
  const int total = (N / 32) * 7 * 64;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63, f = i >> 6, ks = f % 7, nt = f / 7;
    const int n = 32 * nt + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (k0 + e < 98) ? w[n * 98 + k0 + e] : 0.f;
    wp[i] = pack8(v);
  }
}

}  // namespace

extern "C" int smx_conv7_c2_bf16_pack(const float* w, void* wp, int N, void* stream) {
  if (!w || !wp || N <= 0 || N % 128 || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  SMX_LAUNCH(conv7_c2_pack_kernel, dim3(smx_cdiv((N / 32) * 7 * 64, 256)), dim3(256), 0, (hipStream_t)stream, w, (uint4*)wp, N);
  return smx_launch_status();
}

extern "C" int smx_conv7_c2_bf16(const float* x, const void* wp, const float* bias, void* y, int ldc, int B, int H, int W, int N, int act, void* stream) {
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0 || H % TH || W % TW || N <= 0 || N % 128 || ldc < N || ldc % 8) return SMX_EINVAL;
  if (((uintptr_t)x & 7) || ((uintptr_t)wp & 15) || ((uintptr_t)y & 15)) return SMX_EINVAL;
  if (act != SMX_ACT_NONE && act != SMX_ACT_RELU && act != SMX_ACT_LRELU02) return SMX_EINVAL;
  C2 p;
  p.x = x; p.wp = (const bf16_t*)wp; p.bias = bias; p.y = (bf16_t*)y; p.ldc = ldc; p.B = B; p.H = H; p.W = W; p.N = N; p.act = act;
  p.tiles_y = H / TH; p.tiles_x = W / TW;
  const long long blocks = (long long)B * p.tiles_y * p.tiles_x;
  if (blocks > 2147483647LL) return SMX_EINVAL;
  SMX_LAUNCH(conv7_c2_kernel, dim3((unsigned)blocks, N / 128), dim3(256), 0, (hipStream_t)stream, p);
  return smx_launch_status();
}

extern "C" int smx_conv7_c2_f32_pack(const float* w, float* wp, int N, void* stream) {
  if (!w || !wp || N <= 0 || N % 128) return SMX_EINVAL;
  SMX_LAUNCH(conv7_c2_f32_pack_kernel, dim3(smx_cdiv((N / 32) * 49 * 64, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, N);
  return smx_launch_status();
}

extern "C" int smx_conv7_c2_f32(const float* x, const float* wp, const float* bias, float* y, int ldc, int B, int H, int W, int N, int act, void* stream) {
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0 || H % TH || W % TW || N <= 0 || N % 128 || ldc < N || ldc % 4) return SMX_EINVAL;
  if (((uintptr_t)x & 7) || ((uintptr_t)y & 15)) return SMX_EINVAL;
  if (act != SMX_ACT_NONE && act != SMX_ACT_RELU && act != SMX_ACT_LRELU02) return SMX_EINVAL;
  const long long blocks = (long long)B * (H / TH) * (W / TW);
  if (blocks > 2147483647LL) return SMX_EINVAL;
  SMX_LAUNCH(conv7_c2_f32_kernel, dim3((unsigned)blocks, N / 128), dim3(256), 0, (hipStream_t)stream, x, wp, bias, y, ldc, H, W, act, H / TH, W / TW);
  return smx_launch_status();
}
