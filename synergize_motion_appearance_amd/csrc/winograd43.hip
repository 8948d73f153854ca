// Fused Winograd F(4x4,3x3) convolution for gfx950 (fp32, v_mfma_f32_32x32x2_f32) -- the big 3x3 / stride 1 / pad 1 launches.
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A   on 4x4 output tiles (6x6 input patches): 36 multiplies per 16 outputs and channel =
//   2.25 per output, against 4 for F(2x2,3x3) (winograd.hip) and 9 for the direct convolution: 1.78x fewer MFMA passes than the
//   kernel that is 66 % of the fp32 step.  Transform coefficients reach 8 (A) and 5 (B): measured error vs F.conv2d ~5e-5 in fp32.
//
// The F(2x2,3x3) kernel keeps the input transform in registers, one frequency ROW per wave; with 36 frequencies that costs either
// 96+ accumulator registers per wave beside the transform temporaries (no third wave per SIMD) or every wave re-reading the 6x6
// patches of its tiles from LDS (LDS reads become the limiter).  So this kernel is organised around LDS as a V exchange:
//
//   * block = 32 tiles (4 x 8 tiles = 16 x 32 output pixels of one image) x 32 output channels, 12 waves (768 threads), one block
//     per CU = 3 waves per SIMD at <= 168 VGPRs; wave w owns frequencies 3w .. 3w+2 (48 accumulator registers);
//   * per 8-channel step every thread transforms a THIRD of one (tile, channel) patch -- frequency rows {1,2}, {3,4} or {0,5}, which
//     share their sub-expressions -- from the staged input region (planar [channel][18 rows][40 px] in LDS: tile stride 4 px = one
//     16-B slot, row pitch 40 = 8 slots mod 16, so the 16-lane groups of a ds_read_b128 hit 16 distinct slots) and writes its 12
//     values of V = B^T d B to the V buffer [36 f][2 channel quads][32 tiles][4]; the transform is done ONCE per block and step
//     (not once per wave), 48 FMAs per thread;
//   * consumers read their three B fragments from that buffer with one conflict-free ds_read_b128 each (lane = tile x channel quad,
//     exactly the MFMA operand) and the A fragments U = G g G^T straight from L2 (packed at weight-load time in fragment order, as
//     in winograd.hip), prefetched one step ahead: 12 MFMAs per wave and step;
//   * V is double buffered, so the transform of step k+1 overlaps the MFMAs of step k: ONE barrier per step;
//   * the input region is staged 16 channels at a time (64 contiguous bytes per pixel from HBM/L2), in two 8-channel halves: a
//     half is overwritten as soon as the step that read it has produced its V, the global loads for it were issued a step earlier;
//     the GroupNorm(+swish) of the producing layer is folded into that staging pass (in_ss), as in the F(2x2,3x3) kernel;
//   * epilogue: the 36 accumulator tiles go through LDS in two batches of 18 frequencies (pitch 36: conflict-free ds_write_b128);
//     256 threads = (tile, channel quad) apply Y = A^T M A, bias / activation / residual, store 16 pixels x 16 B each (128 B
//     contiguous per pixel across the 8 quads) and reduce the Welford partials of the stored tile for the next GroupNorm.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <mutex>
#include <stdlib.h>
#include <type_traits>
#include "smx.h"
#include "smx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TB_H = 16, TB_W = 32;                 // output pixels per block
constexpr int RGH = 18, RGW = 34, RPX = 40;         // staged region (pixels), LDS row pitch (floats)
constexpr int CHS = RGH * RPX + 4;                  // floats per channel plane: 724 = 4 mod 8, so the two channel quads a lane pair stages
                                                    // (planes 4 apart: +2896 floats = +16 banks) never meet in a bank
constexpr int HALF = 8 * CHS;                       // one 8-channel half of the region buffer (floats)
constexpr int VBUF = 36 * 2 * 32 * 4;               // one V buffer (floats): [36 f][2 quads][32 tiles][4]
constexpr int NTHR = 768;
constexpr int LDS_FLOATS = 2 * HALF + 2 * VBUF;     // 11520 + 18432 = 29952 floats = 119,808 B
constexpr int MP = 36;                              // epilogue exchange pitch (floats per tile row)
static_assert(18 * 32 * MP <= LDS_FLOATS, "epilogue batch must fit the main loop's LDS");

struct W4 {
  const float* x; const float* u; const float* bias; const float* res; float* y; float* stats;
  const float* in_ss; int in_swish;
  int lda, ldc, ldres;
  int B, H, W, Cin, Cout, act;
  int tiles_y, tiles_x, n32;
  int abl;                          // timing-only ablation bits (SMX_TOOLS builds): 1 no transform, 2 no MFMA, 4 no staging, 8 no U loads, 16 no barrier, 32 no epilogue
};

__device__ __forceinline__ float act43(float v, int act) {
  switch (act) {
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    case SMX_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case SMX_ACT_SWISH: return v / (1.f + expf(-v));
    case SMX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case SMX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// 1-D input transform B^T x of six values (Lavin & Gray, F(4,3)):
//   t0 = 4 x0 - 5 x2 + x4          t1 = (x4 - 4 x2) + (x3 - 4 x1)     t2 = (x4 - 4 x2) - (x3 - 4 x1)
//   t3 = (x4 - x2) + 2 (x3 - x1)   t4 = (x4 - x2) - 2 (x3 - x1)       t5 = 4 x1 - 5 x3 + x5
__device__ __forceinline__ void bt6(const float (&x)[6], float (&t)[6]) {
  const float a = fmaf(-4.f, x[2], x[4]), b = fmaf(-4.f, x[1], x[3]);
  const float c = x[4] - x[2], d = x[3] - x[1];
  t[0] = fmaf(4.f, x[0], fmaf(-5.f, x[2], x[4]));
  t[1] = a + b; t[2] = a - b;
  t[3] = fmaf(2.f, d, c); t[4] = fmaf(-2.f, d, c);
  t[5] = fmaf(4.f, x[1], fmaf(-5.f, x[3], x[5]));
}

__global__ __launch_bounds__(NTHR, 1) void winograd43_kernel(W4 p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const region = smem;                         // [2 halves][8 ch][18][40]
  float* const vbuf = smem + 2 * HALF;                // [2][36][2][32][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int bx = bid % p.tiles_x; bid /= p.tiles_x;
  const int by = bid % p.tiles_y; const int img = bid / p.tiles_y;
  const int nblk = blockIdx.y;
  const float* __restrict__ X = p.x + (long long)img * p.H * p.W * p.lda;

  // ---- region staging: 612 pixels x 4 channel quads (16 channels) per slice, 4 items per thread ----------------------------------
  // quads 0,1 -> half A (the slice's first 8-channel step), quads 2,3 -> half B.  The HALF is wave-uniform (wave & 1): a store pass is
  // executed by the six waves that own that half only; a lane pair covers the two quads of its half for one pixel (32 contiguous bytes
  // of the NHWC row), consecutive lane pairs consecutive pixels -> conflict-free ds_write_b32 into the planar layout.
  const int myq = 2 * (wave & 1) + (lane & 1);
  int goff[4], loff[4]; bool gok[4], lok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = (lane >> 1) + 32 * (wave >> 1) + 192 * k;
    lok[k] = px < RGH * RGW;
    const int ry = px / RGW, rx = px - ry * RGW;
    const int iy = by * TB_H - 1 + ry, ix = bx * TB_W - 1 + rx;
    gok[k] = lok[k] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    goff[k] = (iy * p.W + ix) * p.lda + myq * 4;
    loff[k] = (myq >> 1) * HALF + ((myq & 1) * 4) * CHS + ry * RPX + rx;    // channel plane (myq&1)*4 + e of half myq>>1
  }
  float4 stage[4];
  float4 ssa = make_float4(1.f, 0.f, 1.f, 0.f), ssb = ssa;
  const int loader = p.in_ss ? (p.in_swish ? 2 : 1) : 0;
  auto load_region = [&](int c0) __attribute__((always_inline)) {
    if (loader) {
      const float* sp = p.in_ss + ((long long)img * p.Cin + c0 + myq * 4) * 2;
      ssa = *reinterpret_cast<const float4*>(sp); ssb = *reinterpret_cast<const float4*>(sp + 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gok[k]) v = *reinterpret_cast<const float4*>(X + goff[k] + c0);
      stage[k] = v;
    }
  };
  auto store_region = [&]() __attribute__((always_inline)) {      // this thread's items (all in ONE half: myq >> 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 v = stage[k];
      if (loader) {
        v = make_float4(fmaf(v.x, ssa.x, ssa.y), fmaf(v.y, ssa.z, ssa.w), fmaf(v.z, ssb.x, ssb.y), fmaf(v.w, ssb.z, ssb.w));
        if (loader == 2) {
          constexpr float L2E = 1.44269504088896340736f;
          v.x *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * v.x));
          v.y *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * v.y));
          v.z *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * v.z));
          v.w *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * v.w));
        }
        v.x = gok[k] ? v.x : 0.f; v.y = gok[k] ? v.y : 0.f; v.z = gok[k] ? v.z : 0.f; v.w = gok[k] ? v.w : 0.f;   // zero padding stays zero
      }
      if (lok[k]) {
        float* d = region + loff[k];
        d[0] = v.x; d[CHS] = v.y; d[2 * CHS] = v.z; d[3 * CHS] = v.w;
      }
    }
  };

  // ---- producer role: (tile, channel, frequency-row pair) ----------------------------------------------------------------------
  const int part = wave >> 2;                         // 0: rows {1,2}, 1: rows {3,4}, 2: rows {0,5}   (wave-uniform)
  // lanes run over (channel-in-quad e, tile): the V slot of item i is ((quad * 32 + tile) * 4 + e) = i itself, so the 12 ds_write_b32
  // of a wave hit 64 consecutive floats (tile-major lanes wrote at a stride of 4 floats: 4-way bank conflicts on every store)
  const int item = tid & 255;
  const int ptile = (item >> 2) & 31, pch = (item >> 7) * 4 + (item & 3);     // tile, channel of the 8-channel step
  const int ptr_ = ptile >> 3, ptc = ptile & 7;
  const int prd = pch * CHS + (4 * ptr_) * RPX + 4 * ptc;                     // patch origin inside a half
  const int pwr = item;                                                       // V slot of (tile, channel) inside one frequency
  auto produce = [&](const float* half, float* vb) __attribute__((always_inline)) {
    // rows pass: for each of the 6 patch columns the two frequency rows of this part; then the columns pass per row
    float ra[6], rb[6];
    if (part == 2) {
      float col[6][6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const float4 a = *reinterpret_cast<const float4*>(half + prd + r * RPX);
        const float2 b = *reinterpret_cast<const float2*>(half + prd + r * RPX + 4);
        col[r][0] = a.x; col[r][1] = a.y; col[r][2] = a.z; col[r][3] = a.w; col[r][4] = b.x; col[r][5] = b.y;
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        ra[c] = fmaf(4.f, col[0][c], fmaf(-5.f, col[2][c], col[4][c]));      // row 0
        rb[c] = fmaf(4.f, col[1][c], fmaf(-5.f, col[3][c], col[5][c]));      // row 5
      }
    } else {
      float col[4][6];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 a = *reinterpret_cast<const float4*>(half + prd + (r + 1) * RPX);
        const float2 b = *reinterpret_cast<const float2*>(half + prd + (r + 1) * RPX + 4);
        col[r][0] = a.x; col[r][1] = a.y; col[r][2] = a.z; col[r][3] = a.w; col[r][4] = b.x; col[r][5] = b.y;
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) {                   // col[0..3] = patch rows 1..4
        if (part == 0) {
          const float a = fmaf(-4.f, col[1][c], col[3][c]), b = fmaf(-4.f, col[0][c], col[2][c]);
          ra[c] = a + b; rb[c] = a - b;                                       // rows 1, 2
        } else {
          const float a = col[3][c] - col[1][c], b = col[2][c] - col[0][c];
          ra[c] = fmaf(2.f, b, a); rb[c] = fmaf(-2.f, b, a);                  // rows 3, 4
        }
      }
    }
    float va[6], vb_[6];
    bt6(ra, va); bt6(rb, vb_);
    const int ia = part == 0 ? 1 : (part == 1 ? 3 : 0), ib = part == 0 ? 2 : (part == 1 ? 4 : 5);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      vb[(ia * 6 + j) * 256 + pwr] = va[j];
      vb[(ib * 6 + j) * 256 + pwr] = vb_[j];
    }
  };

  // ---- consumer role: frequencies 3 wave .. 3 wave + 2, all 32 output channels of the block -------------------------------------------
  const int cs8 = p.Cin >> 3;
  const float* __restrict__ U0 = p.u + (((long long)(wave * 3) * p.n32 + nblk) * cs8) * 256 + lane * 4;
  const long long ufs = (long long)p.n32 * cs8 * 256;                        // floats between consecutive frequencies
  f32x16 acc[3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float4 ucur[3], unext[3];
  auto uload = [&](int step, float4 (&d)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 3; ++q) d[q] = *reinterpret_cast<const float4*>(U0 + q * ufs + (long long)step * 256);
  };
  const int vrd = (wave * 3) * 256 + lane * 4;        // lane = 32 hh + tile: [f][hh][tile][4]
  auto consume = [&](const float* vb) __attribute__((always_inline)) {
    float4 v[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) v[q] = *reinterpret_cast<const float4*>(vb + vrd + q * 256);
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ucur[q].x, v[q].x, acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ucur[q].y, v[q].y, acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ucur[q].z, v[q].z, acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ucur[q].w, v[q].w, acc[q], 0, 0, 0);
  };

  // ---- main loop: steps of 8 channels; slice s = steps 2s (half A), 2s+1 (half B) ----------------------------------------------------
  const int nsteps = p.Cin >> 3, nslices = p.Cin >> 4;
  load_region(0);
  uload(0, ucur);
  store_region();                                     // both halves of slice 0 (each thread stores its own half)
  if (nslices > 1) load_region(16);
  __syncthreads();
  produce(region, vbuf);                              // V(0) from half A
  __syncthreads();
  for (int k = 0; k < nsteps; ++k) {
    const bool more = k + 1 < nsteps;
    if (more && !(p.abl & 8)) uload(k + 1, unext);
    // the half step k read (k even: A, odd: B) is dead: V(k) is complete.  The waves that own that half refill it with the next
    // slice (requested two steps ago) and request the slice after that.
    const int s_next = (k >> 1) + 1;
    if (s_next < nslices && (k & 1) == (wave & 1) && !(p.abl & 4)) {   // this wave owns the half that just died
      store_region();
      if (s_next + 1 < nslices) load_region((s_next + 1) * 16);     // two full steps until these registers are stored
    }
    // (measured, tools/wino43_ablate.sh: the phases of a step ADD -- all twelve waves leave the barrier together and run
    // [staging | transform | MFMA] in lock-step.  Staggering the order per wave (MFMAs first on two of a SIMD's three waves) duplicates
    // both bodies under a wave-uniform branch and spilled 292 B per lane at the 168-register budget: 0.26 -> 0.19.)
    if (more && !(p.abl & 1)) produce(region + ((k + 1) & 1) * HALF, vbuf + ((k + 1) & 1) * VBUF);   // needs the half of step k+1: stored >= 1 barrier ago
    __builtin_amdgcn_s_setprio(1);
    if (!(p.abl & 2)) consume(vbuf + (k & 1) * VBUF);
    __builtin_amdgcn_s_setprio(0);
    if (more) {
#pragma unroll
      for (int q = 0; q < 3; ++q) ucur[q] = unext[q];
    }
    if (!(p.abl & 16)) __syncthreads();
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------------------------------
  if (p.abl & 32) { if (acc[0][0] + acc[1][1] + acc[2][2] == 123.456f) p.y[0] = 1.f; return; }
  // exchange [18 f][32 tiles][36] per batch; lane (t, hh) holds tile t, channels 8g + 4hh + e in registers 4g + e
  float* const mb = smem;
  const int et = lane & 31, ehh = lane >> 5;
  const int etile = (tid & 255) >> 3, en4 = (tid & 7) * 4;                  // item threads (tid < 256): (tile, channel quad)
  float4 yv[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) yv[a][b] = make_float4(0.f, 0.f, 0.f, 0.f);
  // (requesting the residual tile here, into the output accumulators, so that its HBM round trip overlaps the two exchanges, was
  // measured: with the 48 main-loop accumulators still live it spills 280 B per lane at the 168-register budget, 0.26 -> 0.22)
  auto f4fma = [](float s, const float4& a, float4& d) { d.x = fmaf(s, a.x, d.x); d.y = fmaf(s, a.y, d.y); d.z = fmaf(s, a.z, d.z); d.w = fmaf(s, a.w, d.w); };
  // A^T rows: [1 1 1 1 1 0], [0 1 -1 2 -2 0], [0 1 1 4 4 0], [0 1 -1 8 -8 1]
  const float AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -2.f, 0.f}, {0.f, 1.f, 1.f, 4.f, 4.f, 0.f}, {0.f, 1.f, -1.f, 8.f, -8.f, 1.f}};
#pragma unroll
  for (int batch = 0; batch < 2; ++batch) {
    // waves 0..5 own frequencies 0..17 (rows 0..2), waves 6..11 frequencies 18..35 (rows 3..5)
    if ((wave >= 6) == (batch == 1)) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int fl = (wave - 6 * batch) * 3 + q;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(mb + (fl * 32 + et) * MP + 8 * g + 4 * ehh) =
              make_float4(acc[q][4 * g], acc[q][4 * g + 1], acc[q][4 * g + 2], acc[q][4 * g + 3]);
      }
    }
    __syncthreads();
    if (tid < 256) {
#pragma unroll
      for (int il = 0; il < 3; ++il) {
        const int i = 3 * batch + il;
        float4 m[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) m[j] = *reinterpret_cast<const float4*>(mb + ((il * 6 + j) * 32 + etile) * MP + en4);
        // Z_i[b] = sum_j M[i][j] A[j][b]
        float4 z[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          z[b] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int j = 0; j < 6; ++j) if (AT[b][j] != 0.f) f4fma(AT[b][j], m[j], z[b]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
          if (AT[a][i] != 0.f) {
#pragma unroll
            for (int b = 0; b < 4; ++b) f4fma(AT[a][i], z[b], yv[a][b]);
          }
      }
    }
    __syncthreads();
  }
  float* const red = smem;                            // stats exchange [32 tiles][32 n][2]
  if (tid < 256) {
    const int n = nblk * 32 + en4;
    const int oy = by * TB_H + 4 * (etile >> 3), ox = bx * TB_W + 4 * (etile & 7);
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bq = *reinterpret_cast<const float4*>(p.bias + n);
    float* __restrict__ Yi = p.y + (long long)img * p.H * p.W * p.ldc;
    const float* __restrict__ Ri = p.res ? p.res + (long long)img * p.H * p.W * p.ldres : nullptr;
    // residual rows are fetched one output row ahead of their use (2 x 4 float4 live instead of 16: all 16 at once spill)
    float4 rn[4];
    auto load_res = [&](int a) __attribute__((always_inline)) {
#pragma unroll
      for (int b = 0; b < 4; ++b) rn[b] = *reinterpret_cast<const float4*>(Ri + ((oy + a) * p.W + ox + b) * p.ldres + n);
    };
    if (Ri) load_res(0);
    // the activation is resolved ONCE per block (identity / slope form / generic): a per-value switch is ~190 scalar branches around
    // 64 results.  The statistics of the stored tile are accumulated on the fly as shifted sums {sum (v - pivot), sum (v - pivot)^2}
    // (pivot = the tile's first stored value per channel: no cancellation for 16 values), so no result outlives its store.
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, piv[4] = {0.f, 0.f, 0.f, 0.f};
    auto finish = [&](auto mode) __attribute__((always_inline)) {
      constexpr int MODE = decltype(mode)::value;
      const float slope = p.act == SMX_ACT_RELU ? 0.f : 0.2f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float4 rc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) rc[b] = rn[b];
        if (Ri && a < 3) load_res(a + 1);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          float o[4] = {yv[a][b].x + bq.x, yv[a][b].y + bq.y, yv[a][b].z + bq.z, yv[a][b].w + bq.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (MODE == 1) o[e] = fmaxf(o[e], 0.f) + slope * fminf(o[e], 0.f);
            if (MODE == 2) o[e] = act43(o[e], p.act);
          }
          if (Ri) { o[0] += rc[b].x; o[1] += rc[b].y; o[2] += rc[b].z; o[3] += rc[b].w; }
          *reinterpret_cast<float4*>(Yi + ((oy + a) * p.W + ox + b) * p.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (a == 0 && b == 0) piv[e] = o[e];
            const float d = o[e] - piv[e];
            s1[e] += d; s2[e] = fmaf(d, d, s2[e]);
          }
        }
      }
    };
    if (p.act == SMX_ACT_NONE) finish(std::integral_constant<int, 0>{});
    else if (p.act == SMX_ACT_RELU || p.act == SMX_ACT_LRELU02) finish(std::integral_constant<int, 1>{});
    else finish(std::integral_constant<int, 2>{});
    if (p.stats) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ms = s1[e] * 0.0625f;                                 // mean of the shifted values
        red[((etile * 32) + en4 + e) * 2] = piv[e] + ms;
        red[((etile * 32) + en4 + e) * 2 + 1] = fmaxf(s2[e] - s1[e] * ms, 0.f);
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    if (tid < 32) {                                   // Chan merge of the 32 tiles (16 values each) of channel tid
      float a = 0.f, b = 0.f;
      for (int tl = 0; tl < 32; ++tl) { a += red[(tl * 32 + tid) * 2]; b += red[(tl * 32 + tid) * 2 + 1]; }
      a *= (1.f / 32.f);
      float c2 = 0.f;
      for (int tl = 0; tl < 32; ++tl) { const float d = red[(tl * 32 + tid) * 2] - a; c2 = fmaf(d, d, c2); }
      const long long chunk = ((long long)img * p.tiles_y + by) * p.tiles_x + bx;
      float* o2 = p.stats + (chunk * p.Cout + nblk * 32 + tid) * 2;
      o2[0] = a; o2[1] = b + 16.f * c2;
    }
  }
}

}  // namespace

/* F(4x4,3x3) form of smx_winograd_conv3x3_f32 for the big launches: H % 16 == 0, W % 32 == 0, Cin % 16 == 0, Cout % 32 == 0; 16-B aligned
 * rows (lda, ldc, ldres multiples of 4).  u43: U = G g G^T (6x6 frequencies) packed [36 f][Cout/32][Cin/8][64 lanes][4] like the
 * F(2x2,3x3) weights.  stats_part (optional): [B][(H/16)*(W/32)][Cout][2] = {mean, M2} of each stored 16x32-pixel block. */
extern "C" int smx_winograd43_conv3x3_f32(const float* x, int lda, const float* u43, const float* bias, const float* res, int ldres,
                                          float* y, int ldc, int B, int H, int W, int Cin, int Cout, int act, const float* in_ss,
                                          int in_swish, float* stats_part, void* stream) {
  if (!x || !u43 || !y || B <= 0 || H <= 0 || W <= 0) return SMX_EINVAL;
  if (H % TB_H || W % TB_W || Cin % 16 || Cout % 32 || lda % 4 || ldc % 4 || lda < Cin || ldc < Cout) return SMX_EINVAL;
  if ((((uintptr_t)x) | ((uintptr_t)u43) | ((uintptr_t)y)) & 15) return SMX_EINVAL;
  if (res && (ldres % 4 || ldres < Cout || (((uintptr_t)res) & 15))) return SMX_EINVAL;
  if ((bias && (((uintptr_t)bias) & 15)) || (in_ss && (((uintptr_t)in_ss) & 15))) return SMX_EINVAL;
  if ((long long)H * W * lda > 2147483647LL || (long long)H * W * ldc > 2147483647LL || (long long)H * W * (res ? ldres : 0) > 2147483647LL) return SMX_EINVAL;
  W4 p;
  p.x = x; p.u = u43; p.bias = bias; p.res = res; p.y = y; p.stats = stats_part; p.in_ss = in_ss; p.in_swish = in_swish;
  p.lda = lda; p.ldc = ldc; p.ldres = res ? ldres : 0;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.act = act;
  p.tiles_y = H / TB_H; p.tiles_x = W / TB_W; p.n32 = Cout / 32;
#ifdef SMX_TOOLS
  { const char* e = getenv("SMX_W43_ABL"); p.abl = e ? atoi(e) : 0; }
#else
  p.abl = 0;
#endif
  const long long blocks = (long long)B * p.tiles_y * p.tiles_x;
  if (blocks > 2147483647LL) return SMX_EINVAL;
  SMX_HIP(smx_max_dynamic_lds((const void*)winograd43_kernel, LDS_FLOATS * 4));
  SMX_LAUNCH(winograd43_kernel, dim3((unsigned)blocks, Cout / 32), dim3(NTHR), (size_t)LDS_FLOATS * 4, (hipStream_t)stream, p);
  return smx_launch_status();
}
