// 3x3 / stride 1 / pad 1 convolutions with C_out <= 4 (the decoder's image head conv_out 64->3 at
// 256x256, `Generator` archs/vqgan_arch.py:339-342; RefineFlow's stacked [conv2 | convo2] 256->3 at
// 64x64, archs/appmotioncodebook_arch.py:150-167) for gfx950, NHWC fp32.
//
// On the matrix cores these layers pad N = 3 up to a 32-wide MFMA tile (10x wasted passes, 13-15
// algorithmic TFLOP/s measured); they are HBM-shaped (read B*H*W*C_in once, write 3 floats per pixel).
// Here: one pixel = C_in/4 consecutive lanes x float4 (coalesced C_in*4-byte runs, as in the warp);
// every lane keeps the 9 x C_out float4 weights of its 4 channels in registers for its whole life and
// slides a 3x3 register window along an image row, so each input element is fetched 3 times (once per
// window row) instead of 9; the per-pixel C_out dot products are reduced over the pixel's lanes with
// DPP row operations (4 v_add_f32 per value for 16 lanes) + bpermute steps for 32 / 64 lanes.
// Optional fused GroupNorm (+swish) of the producer on the loaded taps (`in_ss`), as in the
// Winograd region loader; bias / activation in the store.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

namespace {

// TX = storage type of the input activation (float | bf16_t); weights, bias and the <= 4-channel output are fp32
template <typename TX>
struct SP {
  const TX* x; const float* w; const float* bias; float* y; const float* in_ss;
  int in_swish, lda, ldc, B, H, W, Cin, Cout, act, run;
};

__device__ __forceinline__ float s_act(float v, int act) {
  switch (act) {
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    case SMX_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case SMX_ACT_SWISH: return v / (1.f + expf(-v));
    case SMX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case SMX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

__device__ __forceinline__ float4 f4fma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(o);
}

// sum over the LPP consecutive lanes of a pixel group (result valid in every lane of the group)
template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
  v = dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);       // row_half_mirror
  v = dpp_add<0x140>(v);       // row_mirror      -> sum over the 16-lane row
  if (LPP >= 32) v += __shfl_xor(v, 16, 64);
  if (LPP >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <typename TX, int LPP, int NO, int RUN, bool SW>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(SP<TX> p) {
  constexpr int GPB = 256 / LPP;                           // pixel groups per block
  const int sub = threadIdx.x % LPP, grp = threadIdx.x / LPP;
  const int runs_per_row = p.W / RUN;
  const long long gid = (long long)blockIdx.x * GPB + grp; // one group = one run of p.run pixels of one row
  const long long total = (long long)p.B * p.H * runs_per_row;
  const bool live = gid < total;
  const long long g2 = live ? gid : total - 1;
  const int b = (int)(g2 / ((long long)p.H * runs_per_row));
  const int rem = (int)(g2 - (long long)b * p.H * runs_per_row);
  const int y = rem / runs_per_row, x0 = (rem - y * runs_per_row) * RUN;

  // this lane's weights: [n][tap] float4 over channels 4*sub .. 4*sub+3 (+64-channel chunks handled by LPP)
  float4 wr[NO][9];
#pragma unroll
  for (int n = 0; n < NO; ++n)
#pragma unroll
    for (int t = 0; t < 9; ++t)
      wr[n][t] = n < p.Cout ? *reinterpret_cast<const float4*>(p.w + ((long long)n * 9 + t) * p.Cin + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = make_float4(0.f, 0.f, 0.f, 0.f);     // scale, shift
  if (p.in_ss) {
    const float* sp = p.in_ss + ((long long)b * p.Cin + sub * 4) * 2;
    const float4 a = *reinterpret_cast<const float4*>(sp), c = *reinterpret_cast<const float4*>(sp + 4);
    s0 = make_float4(a.x, a.z, c.x, c.z); s1 = make_float4(a.y, a.w, c.y, c.w);
  }
  // row validity / base offsets are per group, the column test only bites on the first tap of the first
  // run and the last tap of the last run of a row (x0 is RUN-aligned): no per-tap bounds logic in the loop;
  // 32-bit element offsets (checked on the host)
  const TX* xb = p.x + (long long)b * p.H * p.W * p.lda + sub * 4;
  bool rv[3]; int ro[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) { const int yy = y - 1 + r; rv[r] = yy >= 0 && yy < p.H; ro[r] = (rv[r] ? yy : y) * p.W * p.lda; }
  // Branch-free taps: the address is clamped into the image, the value is zeroed by a select afterwards --
  // with control flow around the loads the compiler cannot hoist the next columns' loads above the FMAs of
  // the current pixel and every step waits a full memory latency (measured 2 us per pixel step).
  // GroupNorm is always applied (identity scale/shift without in_ss); swish is a template switch.
  auto tap = [&](int r, int xx, bool colok) -> float4 {
    float4 v = St<TX>::ld4(xb + ro[r] + (colok ? xx : x0) * p.lda);
    v = make_float4(fmaf(v.x, s0.x, s1.x), fmaf(v.y, s0.y, s1.y), fmaf(v.z, s0.z, s1.z), fmaf(v.w, s0.w, s1.w));
    if (SW) {
      constexpr float L2E = 1.44269504088896340736f;
      v.x *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * v.x));
      v.y *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * v.y));
      v.z *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * v.z));
      v.w *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * v.w));
    }
    const bool ok = rv[r] && colok;
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  };
  float bn[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) bn[n] = (p.bias && n < p.Cout) ? p.bias[n] : 0.f;

  // window columns c0 (x-1), c1 (x), c2 (x+1), rows y-1, y, y+1; the run is fully unrolled (no register
  // rotation).  Measured on the device (B=30, 64->3 at 256x256): 496 us with predicated taps, 330 us
  // branch-free (this form); moving the stores out of the loop (390 us) or an explicit 4-column prefetch
  // (290 VGPRs, 1 wave/SIMD: 653 us) were slower -- the 108 weight registers cap the occupancy at 2.
  float4 c0[3], c1[3], c2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) { c0[r] = tap(r, x0 - 1, x0 > 0); c1[r] = tap(r, x0, true); }
  float* yb = p.y + ((long long)b * p.H + y) * p.W * p.ldc;
#pragma unroll
  for (int i = 0; i < RUN; ++i) {
    const int x = x0 + i;
#pragma unroll
    for (int r = 0; r < 3; ++r) c2[r] = tap(r, x + 1, i + 1 < RUN || x + 1 < p.W);
    float acc[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      // four independent per-component FMA chains, one horizontal add at the end
      float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        a4 = f4fma(c0[r], wr[n][r * 3], a4);
        a4 = f4fma(c1[r], wr[n][r * 3 + 1], a4);
        a4 = f4fma(c2[r], wr[n][r * 3 + 2], a4);
      }
      acc[n] = group_sum<LPP>((a4.x + a4.y) + (a4.z + a4.w));
    }
    if (live && sub == 0) {
#pragma unroll
      for (int n = 0; n < NO; ++n) if (n < p.Cout) yb[(long long)x * p.ldc + n] = s_act(acc[n] + bn[n], p.act);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { c0[r] = c1[r]; c1[r] = c2[r]; }
  }
}

}  // namespace

namespace {
template <typename TX>
int smalln_launch(const TX* x, int lda, const float* w, const float* bias, float* y, int ldc, int B, int H, int W, int Cin, int Cout,
                  int act, const float* in_ss, int in_swish, void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout > 4 || ldc < Cout) return SMX_EINVAL;
  if ((Cin != 64 && Cin != 128 && Cin != 256) || lda % 4 != 0 || lda < Cin) return SMX_EINVAL;
  if (((uintptr_t)x & (4 * sizeof(TX) - 1)) || ((uintptr_t)w & 15) || (in_ss && ((uintptr_t)in_ss & 15))) return SMX_EINVAL;
  SP<TX> p;
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.in_ss = in_ss; p.in_swish = in_swish; p.lda = lda; p.ldc = ldc;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.act = act;
  if (W % 4 != 0 || (long long)H * W * lda > 2147483647LL) return SMX_EINVAL;
  const int run = (W % 16 == 0) ? 16 : 4;
  p.run = run;
  const int lpp = Cin / 4;
  const long long groups = (long long)B * H * (W / run);
  const long long blocks = (groups + 256 / lpp - 1) / (256 / lpp);
  if (blocks > 2147483647LL) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)blocks), block(256);
  const bool sw = in_ss && in_swish;
#define SMX_SMALL_S(L, R, S) do { if (Cout <= 3) SMX_LAUNCH((conv3x3_small_kernel<TX, L, 3, R, S>), grid, block, 0, st, p); \
                                  else SMX_LAUNCH((conv3x3_small_kernel<TX, L, 4, R, S>), grid, block, 0, st, p); } while (0)
#define SMX_SMALL_R(L, R) do { if (sw) SMX_SMALL_S(L, R, true); else SMX_SMALL_S(L, R, false); } while (0)
#define SMX_SMALL(L) do { if (run == 16) SMX_SMALL_R(L, 16); else SMX_SMALL_R(L, 4); } while (0)
  if (lpp == 16) SMX_SMALL(16); else if (lpp == 32) SMX_SMALL(32); else SMX_SMALL(64);
#undef SMX_SMALL
#undef SMX_SMALL_R
#undef SMX_SMALL_S
  return smx_launch_status();
}
}  // namespace

extern "C" int smx_conv3x3_smalln_f32(const float* x, int lda, const float* w, const float* bias, float* y, int ldc,
                                      int B, int H, int W, int Cin, int Cout, int act, const float* in_ss, int in_swish,
                                      void* stream) {
  return smalln_launch<float>(x, lda, w, bias, y, ldc, B, H, W, Cin, Cout, act, in_ss, in_swish, stream);
}
/* bf16 input activation, fp32 weights / bias / output (the image head and the RefineFlow outputs stay fp32) */
extern "C" int smx_conv3x3_smalln_bf16(const void* x, int lda, const float* w, const float* bias, float* y, int ldc,
                                       int B, int H, int W, int Cin, int Cout, int act, const float* in_ss, int in_swish,
                                       void* stream) {
  return smalln_launch<bf16_t>((const bf16_t*)x, lda, w, bias, y, ldc, B, H, W, Cin, Cout, act, in_ss, in_swish, stream);
}
