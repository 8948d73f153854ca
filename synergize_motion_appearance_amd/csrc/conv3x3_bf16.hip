// 3x3 / stride 1 / pad 1 convolution on the bf16 MFMA, "region-direct": the configs[2] counterpart of the fused
// Winograd kernel (same call sites: ResBlock / Upsample / SFT / conv-FFN / RefineFlow 3x3 convolutions, ~85 % of the flops).
//
// At 16x the fp32 matrix rate these layers are bound by bytes, not by the matrix pipe -- HBM first, then L2 -> LDS and LDS
// reads -- so the kernel is organised around moving each input byte ONCE:
//   * a block owns a 16x16-pixel output tile x 64 output channels; per 64-channel slice the raw (16+2)x(16+2)-pixel input
//     region is staged once into LDS (GroupNorm(+swish) of the producer and the nearest-x2 upsampling folded into the
//     staging pass) and ALL NINE taps read their MFMA A operands straight out of it: im2col never exists, not even in LDS
//     (the implicit-GEMM kernel stages 9 shifted copies: 9x the global->LDS traffic for the same bytes);
//   * weights: the big launches (16x16 tiles, SLAB) keep ALL NINE taps' [64 n][32 k] tiles of a 32-channel slice in LDS -- one staging
//     phase and two barriers per 72 MFMAs per wave; the small ones (8x16 tiles, 3 blocks / CU) stream a [64 n][64 k] tile per tap,
//     double buffered, one barrier per tap;
//   * 4 waves, each 64 pixels x 64 channels (2 x 2 MFMA tiles of 32x32x16): 1 KB of LDS reads per MFMA;
//   * the epilogue transposes the accumulators through LDS and stores 16-B chunks of 8 channels (bias / activation / residual
//     fused) and can emit the Welford partials {mean, M2} of the stored tile for the NEXT GroupNorm (stats_part), so a
//     ResBlock's activations are read by the convolutions only.
// LDS: SLAB 18*18 px * 80 B + 9 * 64 * 80 B = 72 KB (the epilogue's 256 x 68 fp32 tile reuses it) -> 2 blocks / CU; per-tap form:
// region 46.7 KB (25.9 KB at 8x16) + 2 x 9.2 KB weights.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TW = 16, RW = TW + 2;                                        // output tile width, staged region width
constexpr int PIXB = 144;                                                  // bytes per region pixel / weight row in LDS (128 + 16 pad)
constexpr int BN = 64, CS = 64;                                            // output channels per block, channels per slice
constexpr int NT = 256;
constexpr int WT_B = BN * PIXB;                                            // 9,216
constexpr int CLD = BN + 4;
// TH = output tile height: 16 (wave = 64 px x 64 n, 2 blocks / CU) or 8 (wave = 32 px x 64 n, ~100 VGPRs, 3 blocks / CU = 12 waves:
// more loads in flight per CU for the latency-bound small-C_in layers, a two-step-deep weight ring without spills)
template <int TH> struct Geo {
  static constexpr int RH = TH + 2, RPX = RH * RW, REGION_B = RPX * PIXB;
  static constexpr int LDS_B = (REGION_B + 2 * WT_B) > (TH * TW * CLD * 4) ? (REGION_B + 2 * WT_B) : (TH * TW * CLD * 4);
};

struct CP {
  const bf16_t* x; const bf16_t* w; const float* bias; const void* res; bf16_t* y; float* stats; const float* in_ss;
  int in_swish, res_f32;
  const bf16_t* mul; int ldmul; float sft_w;   // SFT epilogue: y = res + sft_w * (res * mul + conv)  (bf16 res / mul, 16 B-aligned rows)
  int lda, ldc, ldres, ldw;
  int B, H, W, Cin, Cout, up2, act;
  int tiles_y, tiles_x, ntiles, tpb;
};

__device__ __forceinline__ float c_act(float v, int act) {
  switch (act) {
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    case SMX_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case SMX_ACT_SWISH: return v / (1.f + expf(-v));
    case SMX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case SMX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// SLAB (the big launches, 16x16 tiles): 32-channel slices with ALL NINE taps' weights of the slice in LDS -- one staging phase
// (region + 9 x [64 n][32 k] weights, 72 KB single-buffered) and two barriers per 72 MFMAs per wave.  The per-step trace of the
// tap-streaming form (one [64 n][64 k] weight tile, one barrier, one store/load phase per tap = per 16 MFMAs) read: compute
// 1.25k cycles, weight store / next request / barrier 1.05k, loop overhead 0.35k per step -- the matrix pipe idled through more
// than half of every step, and neither deeper weight prefetch, nor skipping the barriers, nor pipelining the LDS reads moved it.
// F32: fp32 STORAGE on both sides (the bf16-compute training mode, `smx_conv3x3_mfma16_f32`): the region is read as fp32 and rounded to
// bf16 (RNE) on its way into LDS, the output (and the residual) are fp32 -- torch.autocast(bfloat16)'s arithmetic for F.conv2d on fp32
// tensors; no fused GroupNorm loader / statistics in this form.
template <int TH, bool SLAB = false, bool F32 = false>
__global__ __launch_bounds__(NT, TH == 16 ? 2 : 3) void conv3x3_bf16_kernel(CP p) {
  constexpr int RPX = Geo<TH>::RPX, TI = TH / 8;                                 // TI: 32-pixel A fragments per wave
  constexpr int CSL = SLAB ? 32 : CS;                                            // channels per staged slice
  constexpr int PB = SLAB ? 80 : PIXB;                                           // bytes per region pixel / weight row in LDS (data + 16 pad)
  constexpr int CPP = CSL / 8, CPPL = SLAB ? 2 : 3;                              // 16-B chunks per pixel (and its log2)
  constexpr int REGION_B = RPX * PB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Rg = smem;                       // region [RPX][PB]
  unsigned char* Ws = smem + REGION_B;            // weights [2][BN][PIXB]  (SLAB: [9 taps][BN][PB])
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order: block b runs on XCD b%8; give each XCD a contiguous range of (image, tile) so neighbouring tiles'
  // halos hit the same L2
  int logical = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bx = logical % p.tiles_x; logical /= p.tiles_x;
  const int by = logical % p.tiles_y; const int img = logical / p.tiles_y;
  const int n0 = blockIdx.y * BN;
  const int Hs = p.up2 ? p.H >> 1 : p.H, Ws_ = p.up2 ? p.W >> 1 : p.W;
  const bf16_t* __restrict__ X = p.x + (F32 ? 0LL : (long long)img * Hs * Ws_ * p.lda);
  const float* __restrict__ Xf = reinterpret_cast<const float*>(p.x) + (F32 ? (long long)img * Hs * Ws_ * p.lda : 0LL);

  // ---- region staging: RPX * 8 chunks of 16 B per slice, 256 threads -> 11 chunks per thread (the last partially) -----
  constexpr int NCH = (RPX * CPP + NT - 1) / NT;
  uint4 rreg[NCH];
  // load_region only ISSUES the global loads -- the region chunks and, once per slice, the GroupNorm {scale, shift} pairs of
  // this thread's 8 channels (item & 7 == tid & 7 for all its chunks); store_region, three taps later, applies the fused
  // GroupNorm(+swish) branch-free and writes LDS.  (Normalising right after the load made every wave sit out the region's
  // HBM round trip in the middle of the tap loop.)
  float4 ssv[4];
  const int loader = p.in_ss ? (p.in_swish ? 2 : 1) : 0;
  auto chunk_ok = [&](int k, int& g) __attribute__((always_inline)) -> bool {   // in-image offsets fit 32 bits (launcher check)
    // `g` is ALWAYS a legal offset (coordinates clamped into the frame, a thread past the region's last chunk re-reads that chunk): the loads
    // are issued unconditionally, back to back (hipcc wraps every `if (ok) load` in its own pair of branches -- csrc/winograd.hip, round 5:
    // +2.5 % from this alone); the padding is zeroed when the chunk goes to LDS
    const int item = tid + NT * k;
    const int px = min(item >> CPPL, RPX - 1), c8 = item & (CPP - 1);
    const int ry = px / RW, rx = px - ry * RW;
    int iy = by * TH - 1 + ry, ix = bx * TW - 1 + rx;
    const bool ok = item < RPX * CPP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    iy = min(max(iy, 0), p.H - 1); ix = min(max(ix, 0), p.W - 1);
    if (p.up2) { iy >>= 1; ix >>= 1; }
    g = (iy * Ws_ + ix) * p.lda + c8 * 8;
    return ok;
  };
  auto load_region = [&](int c0) __attribute__((always_inline)) {
    if (loader) {
      const float* sp = p.in_ss + ((long long)img * p.Cin + c0 + (tid & (CPP - 1)) * 8) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) ssv[e] = *reinterpret_cast<const float4*>(sp + 4 * e);
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      int g; (void)chunk_ok(k, g);
      uint4 v;
      if (F32) {
        const float4 a = *reinterpret_cast<const float4*>(Xf + g + c0), b = *reinterpret_cast<const float4*>(Xf + g + c0 + 4);
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        v = pack8(f);
      } else v = *reinterpret_cast<const uint4*>(X + g + c0);
      rreg[k] = v;
    }
  };
  auto store_items = [&](auto mode) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode)::value;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int item = tid + NT * k;
      uint4 v = rreg[k];
      int g; const bool ok = chunk_ok(k, g);
      if (MODE >= 1) {
        float f[8]; unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const float4 s4 = ssv[e >> 1];
          f[e] = fmaf(f[e], s4.x, s4.y); f[e + 1] = fmaf(f[e + 1], s4.z, s4.w);
        }
        if (MODE == 2) {
          constexpr float L2E = 1.44269504088896340736f;
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * f[e]));
        }
        v = pack8(f);
      }
      if (!ok) v = make_uint4(0u, 0u, 0u, 0u);                            // the conv's zero padding stays exactly 0
      if (item < RPX * CPP) *reinterpret_cast<uint4*>(Rg + (item >> CPPL) * PB + (item & (CPP - 1)) * 16) = v;
    }
  };
  auto store_region = [&]() __attribute__((always_inline)) {
    if (loader == 2) store_items(std::integral_constant<int, 2>{});
    else if (loader == 1) store_items(std::integral_constant<int, 1>{});
    else store_items(std::integral_constant<int, 0>{});
  };
  // ---- weight tile: [BN][64 k] bf16 of (tap, slice): 64 rows x 8 chunks = 512 chunks -> 2 per thread -----------------
  uint4 wreg[2];
  const int wr0 = tid >> 3, wc8 = tid & 7;
  auto load_w = [&](int tap, int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = n0 + wr0 + 32 * i;
      wreg[i] = n < p.Cout ? *reinterpret_cast<const uint4*>(p.w + (n * p.ldw + tap * p.Cin + c0 + wc8 * 8)) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_w = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(Ws + buf * WT_B + (wr0 + 32 * i) * PIXB + wc8 * 16) = wreg[i];
  };

  f32x16 acc[TI][2];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // A operand: lane l <-> MFMA row l&31 = pixel (tile row 2*TI*wave + 2*i + ((l&31)>>4), col l&15), k chunk (l>>5)
  const int arow = lane & 31, hh = lane >> 5;
  const int abase = ((2 * TI * wave + (arow >> 4)) * RW + (arow & 15)) * PB + hh * 16;   // + i*2*RW*PB + tap offset + kk*32
  const int bbase = arow * PB + hh * 16;                                                 // + j*32*PB + kk*32

  // one step = one (slice, tap): this step's weight tile is in Ws[step & 1]; the next one is requested at the top of the step
  // and written to the other buffer at its end.  (Measured dead ends at the 256-VGPR / 2-waves-per-SIMD budget: a two-step-deep
  // weight ring -- two statically indexed register sets, loop unrolled by two -- and a multi-tile loop that holds the next
  // tile's region in registers across the epilogue both spill 32-89 VGPRs.)
  const int nsl = p.Cin / CS, nsteps = nsl * 9;
  auto compute = [&](int step) __attribute__((always_inline)) {
    const int s = step / 9, tap = step - s * 9, buf = step & 1;
    const int ky = tap / 3, kx = tap - ky * 3;
    const unsigned char* ap = Rg + abase + (ky * RW + kx) * PIXB;
    const unsigned char* bp = Ws + buf * WT_B + bbase;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[TI], bfr[2];
#pragma unroll
      for (int i = 0; i < TI; ++i) af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ap + i * 2 * RW * PIXB + kk * 32));
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bp + j * 32 * PIXB + kk * 32));
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // A = weights: the tile is [n][pixel]
    }
  };
  if (SLAB) {
    constexpr int NWC = 9 * BN * CPP / NT;                               // 9 weight chunks per thread per slice
    uint4 wq[NWC];
    auto load_wslab = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < NWC; ++k) {
        const int item = tid + NT * k, row = item >> CPPL, c4 = item & (CPP - 1);     // row = tap * 64 + n
        const int n = n0 + (row & 63);
        wq[k] = n < p.Cout ? *reinterpret_cast<const uint4*>(p.w + (n * p.ldw + (row >> 6) * p.Cin + c0 + c4 * 8)) : make_uint4(0u, 0u, 0u, 0u);
      }
    };
    auto store_wslab = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < NWC; ++k) {
        const int item = tid + NT * k;
        *reinterpret_cast<uint4*>(Ws + (item >> CPPL) * PB + (item & (CPP - 1)) * 16) = wq[k];
      }
    };
    const int nsl32 = p.Cin / CSL;
    load_wslab(0); load_region(0);
    store_wslab(); store_region();
    __syncthreads();
    for (int s = 0; s < nsl32; ++s) {
      if (s + 1 < nsl32) { load_wslab((s + 1) * CSL); load_region((s + 1) * CSL); }   // in flight over the slice's 72 MFMAs
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const unsigned char* ap = Rg + abase + ((tap / 3) * RW + (tap % 3)) * PB;
        const unsigned char* bp = Ws + tap * BN * PB + bbase;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8 af[TI], bfr[2];
#pragma unroll
          for (int i = 0; i < TI; ++i) af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ap + i * 2 * RW * PB + kk * 32));
#pragma unroll
          for (int j = 0; j < 2; ++j) bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bp + j * 32 * PB + kk * 32));
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
      }
      __syncthreads();                                                   // every wave is past its last read of this slice
      if (s + 1 < nsl32) { store_wslab(); store_region(); __syncthreads(); }
    }
  } else {
  load_region(0); store_region();
  if (TH == 16) {
    // one step = one (slice, tap): this step's weight tile is in Ws[step & 1]; the next one is requested at the top of the step
    // and written to the other buffer at its end (a deeper ring spills at this variant's 2-waves-per-SIMD budget)
    load_w(0, 0); store_w(0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
      const int s = step / 9, tap = step - s * 9;
      if (step + 1 < nsteps) { const int s1 = (step + 1) / 9; load_w(step + 1 - s1 * 9, s1 * CS); }
      if (tap == 5 && s + 1 < nsl) load_region((s + 1) * CS);          // in flight over the last taps of this slice
      compute(step);
      if (step + 1 < nsteps) store_w((step & 1) ^ 1);
      __syncthreads();
      if (tap == 8 && s + 1 < nsl) { store_region(); __syncthreads(); } // every wave is past its last read of the old slice
    }
  } else {
    // 8x16 tiles (122 VGPRs): two statically indexed weight register sets, loop unrolled by two -- the tile of step g is requested
    // at the top of step g-2 and written to LDS at the end of step g-1, so an L2 round trip has two steps to complete
    uint4 wa[2], wb[2];
    auto ldw = [&](uint4 (&wr)[2], int step) __attribute__((always_inline)) {
      const int s1 = step / 9, tap = step - s1 * 9;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = n0 + wr0 + 32 * i;
        wr[i] = n < p.Cout ? *reinterpret_cast<const uint4*>(p.w + (n * p.ldw + tap * p.Cin + s1 * CS + wc8 * 8)) : make_uint4(0u, 0u, 0u, 0u);
      }
    };
    auto stw = [&](const uint4 (&wr)[2], int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(Ws + buf * WT_B + (wr0 + 32 * i) * PIXB + wc8 * 16) = wr[i];
    };
    auto body = [&](int step, uint4 (&mine)[2], const uint4 (&other)[2]) __attribute__((always_inline)) {   // `mine` held this step's tile (already in LDS)
      const int s = step / 9, tap = step - s * 9;
      if (step + 2 < nsteps) ldw(mine, step + 2);
      if (tap == 5 && s + 1 < nsl) load_region((s + 1) * CS);
      compute(step);
      if (step + 1 < nsteps) stw(other, (step & 1) ^ 1);
      __syncthreads();
      if (tap == 8 && s + 1 < nsl) { store_region(); __syncthreads(); }
    };
    ldw(wa, 0); ldw(wb, 1);
    stw(wa, 0);
    __syncthreads();
    for (int step = 0; step < nsteps; step += 2) {
      body(step, wa, wb);
      if (step + 1 < nsteps) body(step + 1, wb, wa);
    }
  }

  }   // !SLAB

  // ---- epilogue: block exchange through LDS -> 16-B chunks of 8 channels ---------------------------------------------------
  // The accumulator tile is [n][pixel] (A = weights): lane (pixel = l&31, hh) holds channels 8g + 4hh + (0..3) of its pixel in
  // registers 4g..4g+3, so the exchange is 4 ds_write_b128 per tile (pitch 68 floats: conflict-free) instead of 16 ds_write_b32.
  float* Cs = reinterpret_cast<float*>(smem);                            // [TH*16 px][CLD]
  const int cq = tid & 7;                                                // this thread's 8-channel chunk (same for all its pixels)
  const int nc = n0 + cq * 8;
  const bool full = nc + 7 < p.Cout;
  const bool al = (p.ldc % (F32 ? 4 : 8) == 0) && ((((uintptr_t)p.y) & 15) == 0) &&
                  (!p.res || (p.res_f32 ? ((p.ldres % 4 == 0) && ((((uintptr_t)p.res) & 15) == 0)) : ((p.ldres % 8 == 0) && ((((uintptr_t)p.res) & 15) == 0))));
  // per-image bases (64-bit once, wave-uniform); per-value offsets are 32-bit (H*W*ld < 2^31, launcher check)
  const long long ipix = (long long)img * p.H * p.W;
  const bf16_t* __restrict__ R16 = reinterpret_cast<const bf16_t*>(p.res) + ipix * p.ldres;
  const float* __restrict__ R32 = reinterpret_cast<const float*>(p.res) + ipix * p.ldres;
  const bf16_t* __restrict__ M16 = p.mul + ipix * p.ldmul;
  bf16_t* __restrict__ Y16 = p.y + (F32 ? 0LL : ipix * p.ldc);
  float* __restrict__ Y32 = reinterpret_cast<float*>(p.y) + (F32 ? ipix * p.ldc : 0LL);
  constexpr int NPASS = TH * TW / 32;                                    // pixels of the tile, 32 per pass
  // a bf16 residual is requested for ALL passes before the accumulators go through LDS: one HBM round trip overlapped with the
  // exchange instead of NPASS of them in sequence behind it
  const bool pre = p.res && !p.res_f32 && full && al;
  const bool sft = pre && p.mul;                                         // the launcher only passes mul with the fast-path layout
  uint4 rq[NPASS], mq[NPASS];
  if (pre) {
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
      const int px = (tid >> 3) + 32 * it;
      const int opix = (by * TH + (px >> 4)) * p.W + bx * TW + (px & 15);
      rq[it] = *reinterpret_cast<const uint4*>(R16 + opix * p.ldres + nc);
      if (sft) mq[it] = *reinterpret_cast<const uint4*>(M16 + opix * p.ldmul + nc);
    }
  }
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = (p.bias && nc + e < p.Cout) ? p.bias[nc + e] : 0.f;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int px = (2 * TI * wave + 2 * i + (arow >> 4)) * TW + (arow & 15);
        *reinterpret_cast<float4*>(Cs + px * CLD + j * 32 + 8 * g + 4 * hh) =
            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
  __syncthreads();
  float piv[8], sm[8], sq[8];                                            // Welford partials of what is stored (shifted by the first value)
#pragma unroll
  for (int e = 0; e < 8; ++e) { piv[e] = 0.f; sm[e] = 0.f; sq[e] = 0.f; }
  // the activation is resolved once per block (0 identity, 1 relu / leaky relu in slope form, 2 generic), not per value
  auto passes = [&](auto mode) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode)::value;
    const float slope = p.act == SMX_ACT_RELU ? 0.f : 0.2f;
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
      const int px = (tid >> 3) + 32 * it;
      const int oy = by * TH + (px >> 4), ox = bx * TW + (px & 15);
      const int opix = oy * p.W + ox;
      const float4 v0 = *reinterpret_cast<const float4*>(Cs + px * CLD + cq * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(Cs + px * CLD + cq * 8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] += bv[e];
        if (MODE == 1) v[e] = fmaxf(v[e], 0.f) + slope * fminf(v[e], 0.f);
        if (MODE == 2) v[e] = c_act(v[e], p.act);
      }
      if (nc < p.Cout) {
        if (full && al) {
          if (sft) {
            float q[8], m[8]; unpack8(rq[it], q); unpack8(mq[it], m);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = q[e] + p.sft_w * (q[e] * m[e] + v[e]);
          } else if (pre) {
            float q[8]; unpack8(rq[it], q);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += q[e];
          } else if (p.res) {
            if (p.res_f32) {
              const float4 q0 = *reinterpret_cast<const float4*>(R32 + opix * p.ldres + nc), q1 = *reinterpret_cast<const float4*>(R32 + opix * p.ldres + nc + 4);
              v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w; v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
            } else {
              float q[8]; unpack8(*reinterpret_cast<const uint4*>(R16 + opix * p.ldres + nc), q);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += q[e];
            }
          }
          if (F32) {
            *reinterpret_cast<float4*>(Y32 + opix * p.ldc + nc) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(Y32 + opix * p.ldc + nc + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            const uint4 pk = pack8(v);
            *reinterpret_cast<uint4*>(Y16 + opix * p.ldc + nc) = pk;
            if (p.stats) unpack8(pk, v);                                 // statistics of the values as STORED (bf16-rounded)
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (nc + e >= p.Cout) { v[e] = 0.f; continue; }
            if (p.res) v[e] += p.res_f32 ? R32[opix * p.ldres + nc + e] : bf2f(R16[opix * p.ldres + nc + e]);
            if (F32) { Y32[opix * p.ldc + nc + e] = v[e]; continue; }
            const bf16_t h = f2bf(v[e]);
            Y16[opix * p.ldc + nc + e] = h;
            v[e] = bf2f(h);
          }
        }
      }
      if (p.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (it == 0) piv[e] = v[e];
          const float d = v[e] - piv[e];
          sm[e] += d; sq[e] += d * d;
        }
      }
    }
  };
  if (p.act == SMX_ACT_NONE) passes(std::integral_constant<int, 0>{});
  else if (p.act == SMX_ACT_RELU || p.act == SMX_ACT_LRELU02) passes(std::integral_constant<int, 1>{});
  else passes(std::integral_constant<int, 2>{});
  if (p.stats) {
    // per channel: Chan-merge the 32 threads (8 values each) that share this channel chunk
    __syncthreads();                                                     // Cs is dead: reuse as [32 threads][64 ch][2]
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float ms = sm[e] * (1.f / NPASS);
      red[((tid >> 3) * BN + cq * 8 + e) * 2] = piv[e] + ms;
      red[((tid >> 3) * BN + cq * 8 + e) * 2 + 1] = fmaxf(sq[e] - sm[e] * ms, 0.f);
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      float mean = 0.f;
#pragma unroll 8
      for (int t = 0; t < 32; ++t) mean += red[(t * BN + tid) * 2];
      mean *= (1.f / 32.f);
      float m2 = 0.f;
#pragma unroll 8
      for (int t = 0; t < 32; ++t) { const float d = red[(t * BN + tid) * 2] - mean; m2 += red[(t * BN + tid) * 2 + 1] + (float)NPASS * d * d; }
      const long long chunk = ((long long)img * p.tiles_y + by) * p.tiles_x + bx;
      float* o = p.stats + (chunk * p.Cout + n0 + tid) * 2;
      o[0] = mean; o[1] = m2;
    }
  }
}

}  // namespace

static int conv3x3_bf16_launch(const void* x, int lda, const void* w, int ldw, const float* bias, const void* res, int res_f32,
                               int ldres, const void* mul, int ldmul, float sft_w, void* y, int ldc, int B, int H, int W, int Cin, int Cout,
                               int up2, int act, const float* in_ss, int in_swish, float* stats_part, int tile_h, void* stream, bool f32io = false) {
  if (f32io && (in_ss || stats_part || mul || (res && !res_f32) || lda % 4 != 0)) return SMX_EINVAL;
  if (!x || !w || !y || B <= 0 || Cin <= 0 || Cout <= 0 || (tile_h != 8 && tile_h != 16)) return SMX_EINVAL;
  if (mul && (!res || res_f32 || Cout % 8 || ldc % 8 || ldres % 8 || ldmul % 8 || ldmul < Cout || act != SMX_ACT_NONE ||
              ((((uintptr_t)y) | ((uintptr_t)res) | ((uintptr_t)mul)) & 15))) return SMX_EINVAL;
  const int TH = tile_h;
  if (H % TH != 0 || W % TW != 0 || Cin % CS != 0 || (!f32io && lda % 8 != 0) || lda < Cin || ldc < Cout || ldw < 9 * Cin || ldw % 8 != 0) return SMX_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || (in_ss && ((uintptr_t)in_ss & 15)) || (res && ldres < Cout)) return SMX_EINVAL;
  if (up2 && ((H & 1) || (W & 1))) return SMX_EINVAL;
  CP p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.bias = bias; p.res = res; p.res_f32 = res_f32; p.y = (bf16_t*)y; p.stats = stats_part;
  p.mul = (const bf16_t*)mul; p.ldmul = mul ? ldmul : 0; p.sft_w = sft_w;
  p.in_ss = in_ss; p.in_swish = in_swish; p.lda = lda; p.ldc = ldc; p.ldres = res ? ldres : 0; p.ldw = ldw;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.up2 = up2 ? 1 : 0; p.act = act;
  p.tiles_y = H / TH; p.tiles_x = W / TW;
  const long long blocks = (long long)B * p.tiles_y * p.tiles_x;
  if (blocks > 2147483647LL || (long long)(up2 ? H / 2 : H) * (up2 ? W / 2 : W) * lda > 2147483647LL) return SMX_EINVAL;
  if ((long long)H * W * ldc > 2147483647LL || (long long)H * W * (res ? ldres : 0) > 2147483647LL || (long long)H * W * (mul ? ldmul : 0) > 2147483647LL ||
      (long long)Cout * ldw > 2147483647LL) return SMX_EINVAL;
  SMX_HIP(smx_max_dynamic_lds((const void*)conv3x3_bf16_kernel<16>, Geo<16>::LDS_B));
  p.ntiles = (int)blocks; p.tpb = 1;
  dim3 grid((unsigned)blocks, (Cout + BN - 1) / BN);
  constexpr int SLAB_LDS = Geo<16>::RPX * 80 + 9 * BN * 80;            // 72,000 B (>= the 69,632 B epilogue exchange)
  if (f32io) {
    SMX_HIP(smx_max_dynamic_lds((const void*)(conv3x3_bf16_kernel<16, true, true>), SLAB_LDS));
    if (TH == 16) SMX_LAUNCH((conv3x3_bf16_kernel<16, true, true>), grid, dim3(NT), SLAB_LDS, (hipStream_t)stream, p);
    else SMX_LAUNCH((conv3x3_bf16_kernel<8, false, true>), grid, dim3(NT), Geo<8>::LDS_B, (hipStream_t)stream, p);
    return smx_launch_status();
  }
  if (TH == 16 && Cin % 32 == 0 && smx_tune(SMX_TUNE_CONV16_SLAB)) {
    SMX_HIP(smx_max_dynamic_lds((const void*)(conv3x3_bf16_kernel<16, true>), SLAB_LDS));
    SMX_LAUNCH((conv3x3_bf16_kernel<16, true>), grid, dim3(NT), SLAB_LDS, (hipStream_t)stream, p);
  }
  else if (TH == 16) SMX_LAUNCH(conv3x3_bf16_kernel<16>, grid, dim3(NT), Geo<16>::LDS_B, (hipStream_t)stream, p);
  else SMX_LAUNCH(conv3x3_bf16_kernel<8>, grid, dim3(NT), Geo<8>::LDS_B, (hipStream_t)stream, p);
  return smx_launch_status();
}

extern "C" int smx_conv3x3_bf16(const void* x, int lda, const void* w, int ldw, const float* bias, const void* res, int res_f32,
                                int ldres, void* y, int ldc, int B, int H, int W, int Cin, int Cout, int up2, int act,
                                const float* in_ss, int in_swish, float* stats_part, int tile_h, void* stream) {
  return conv3x3_bf16_launch(x, lda, w, ldw, bias, res, res_f32, ldres, nullptr, 0, 0.f, y, ldc, B, H, W, Cin, Cout, up2, act, in_ss, in_swish,
                             stats_part, tile_h, stream);
}

extern "C" int smx_conv3x3_sft_bf16(const void* x, int lda, const void* w, int ldw, const float* bias, const void* dec, int lddec,
                                    const void* scale, int ldscale, float sft_w, void* y, int ldc, int B, int H, int W, int Cin, int Cout,
                                    int tile_h, void* stream) {
  if (!dec || !scale) return SMX_EINVAL;
  return conv3x3_bf16_launch(x, lda, w, ldw, bias, dec, 0, lddec, scale, ldscale, sft_w, y, ldc, B, H, W, Cin, Cout, 0, SMX_ACT_NONE, nullptr, 0,
                             nullptr, tile_h, stream);
}

/* fp32 storage, bf16 MFMA: the region-direct kernel for the bf16-COMPUTE training mode (forward and data gradient of the 3x3 / s1 / p1
 * convolutions on fp32 activations: inputs rounded to bf16 while staged, fp32 accumulate, fp32 output / residual). */
extern "C" int smx_conv3x3_mfma16_f32(const float* x, int lda, const void* w, int ldw, const float* bias, const float* res, int ldres,
                                      float* y, int ldc, int B, int H, int W, int Cin, int Cout, int up2, int act, int tile_h, void* stream) {
  return conv3x3_bf16_launch(x, lda, w, ldw, bias, res, 1, ldres, nullptr, 0, 0.f, y, ldc, B, H, W, Cin, Cout, up2, act, nullptr, 0, nullptr, tile_h,
                             stream, true);
}
