// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions, region form (reference step
// models/appmotioncomp_model.py:294-434 -> autograd of archs/vqgan_arch.py:168-191 ResBlock / :135-153 Up/Downsample convolutions).
//
// The generic TN GEMM (train_gemm.hip wgrad_kernel) treats dW[co][(ky,kx,ci)] as Cout x 9 Cin and gives every 64 x 64 tile its own
// block: nine blocks per (co, ci) tile pair stage the SAME dY pixels and nine shifted copies of the same X pixels -- 16 KB of
// L2 -> LDS traffic per 64 MFMAs, ~10 TB/s at the fp32 matrix rate; it sits at 0.3-0.4 of the fp32 pipe and gains nothing from the
// bf16 MFMA (the staging, not the arithmetic, is the time).  Here a block owns 64 output channels x 64 input channels x ALL NINE taps
// (nine 32 x 32 accumulators per wave, 144 VGPRs) and walks DOWN a 32-pixel-wide strip of the image one output row per step:
//   * per step one dY row segment (32 px x 64 co) and ONE new X row (34 px x 64 ci) are staged; the other two X rows of the 3 x 34
//     halo are already in a 4-row LDS ring from the previous steps -- 16.5 KB staged per 9 x 64 MFMAs instead of 9 x 16 KB;
//   * staging is LDS-DMA (global_load_lds_dwordx4, issued from inline asm: cdna_hip_programming 5.7): no staging registers (the
//     accumulators take the register file), no ds_write, the next step's rows land while this step multiplies, one barrier per step;
//     padding lanes fetch a 16-byte block of zeros; nearest-x2 upsampling (Upsample's conv) is folded into the source addresses;
//   * LDS keeps 256 B per pixel UNPADDED (DMA writes 1 KB runs); the 16-B chunk a lane fetches is XOR-swizzled by bit 3 of the pixel's
//     column, so the "down the columns" fragment reads (lanes 0-31: pixel p, lanes 32-63: pixel p + 8) hit opposite bank halves:
//     every fragment is one conflict-free ds_read_b32, as in the generic kernel;
//   * the pixel axis is split over blockIdx.y into runs of strip rows; raw partials go to the generic kernel's workspace layout
//     ws[split][co][(ky,kx,ci)] and wgrad_reduce_kernel finishes them in a fixed order (deterministic, no atomics); the bias gradient
//     (column sums of dY) is taken from the staged dY rows by the ci-tile-0 blocks.
// BF16 = true: the same staging, contracted on v_mfma_f32_32x32x16_bf16 with both operands rounded RNE on the way from LDS
// (torch.autocast(bfloat16)'s arithmetic for this GEMM, as smx_wgrad_mfma16_f32 defines it).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int XPX = 36;                        // pixel slots per ring row (34 used: x0 - 1 .. x0 + 32), 9 DMA instructions of 4 pixels
constexpr int XROW_F = XPX * 64;               // floats per ring row (9216 B)
constexpr int DY_F = 32 * 64;                  // floats per dY row segment (8192 B)
constexpr int LDS_F = 4 * XROW_F + 2 * DY_F;   // 53,248 B

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

struct RG {
  const float* dy; const float* x; float* ws; float* bias_ws;
  int ldy, ldx, Cout, Cin, K;
  int B, H, W, Hin, Win, up2;                  // H x W: the convolution's grid (= output grid); Hin x Win: the stored input (H / 2 with up2)
  int nseg, units, uper, msplit, tiles_ci;
};

typedef __attribute__((address_space(3))) void lds_void;

// one LDS-DMA: 64 lanes x 16 B, lane i: *gsrc(i) -> LDS [lds_dst + 16 i, +16).  M0 is compiler-reserved: saved and restored here.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <bool BF16>
__global__ __launch_bounds__(256, 2) void wgrad_region_kernel(RG p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Xs = smem;                      // [4 ring rows][36 px][64 ci]
  float* const Ds = smem + 4 * XROW_F;         // [2][32 px][64 co]
  const unsigned lds0 = (unsigned)(uintptr_t)((lds_void*)smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_ci = blockIdx.x % p.tiles_ci, tile_co = blockIdx.x / p.tiles_ci;
  const int co0 = tile_co * 64, ci0 = tile_ci * 64;
  const int z = blockIdx.y;
  const int u_begin = z * p.uper, u_end = min(p.units, u_begin + p.uper);

  // ---- this lane's part of the DMA instructions: pixel (lane >> 4) of the instruction's four, stored chunk (lane & 15) ----
  // X row instruction i (0..8): pixels 4 i .. 4 i + 3 of the ring row; wave w issues i = w, w + 4 (and 8 for wave 0).  dY: j = w, w + 4.
  const int cpos = lane & 15, pin = lane >> 4;
  const float* zsrc = g_zero16;
  auto issue_x_row = [&](int img, int x0, int iy, int slot) {           // input row iy of the strip -> ring slot
    const bool rowok = iy >= 0 && iy < p.H;
    const int sy = p.up2 ? (iy >> 1) : iy;
    const float* rowp = p.x + ((long long)img * p.Hin + (rowok ? sy : 0)) * p.Win * p.ldx + ci0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int i = wave + 4 * t;
      if (i < 9) {
        const int px = 4 * i + pin, ix = x0 - 1 + px;
        const int c = cpos ^ (((px >> 3) & 1) << 3);
        const bool ok = rowok && px < 34 && ix >= 0 && ix < p.W;
        const int sx = p.up2 ? (ix >> 1) : ix;
        const float* src = ok ? rowp + (long long)sx * p.ldx + c * 4 : zsrc;
        glds16(src, lds0 + (unsigned)(slot * XROW_F + i * 256) * 4u);
      }
    }
  };
  auto issue_dy_row = [&](int img, int x0, int y, int buf) {
    const float* rowp = p.dy + (((long long)img * p.H + y) * p.W + x0) * p.ldy + co0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = wave + 4 * t;
      const int px = 4 * j + pin;
      const int c = cpos ^ (((px >> 3) & 1) << 3);
      glds16(rowp + (long long)px * p.ldy + c * 4, lds0 + (unsigned)(4 * XROW_F + buf * DY_F + j * 256) * 4u);
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- fragment addressing: lane l supplies column (l & 31) of the wave's 32, pixel p (lanes 0-31) or p + 8 (lanes 32-63) ----
  const int wm = wave >> 1, wn = wave & 1, half = lane >> 5;
  const int chA = wm * 8 + ((lane & 31) >> 2), chB = wn * 8 + ((lane & 31) >> 2), sub = lane & 3;
  // base for a pixel whose (compile-time) column has bit 3 == b; the lane's own pixel is 8 * half further: bit 3 flips with `half`
  const int fa0 = half * 8 * 64 + ((chA ^ ((0 ^ half) << 3)) << 2) + sub, fa1 = half * 8 * 64 + ((chA ^ ((1 ^ half) << 3)) << 2) + sub;
  const int fb0 = half * 8 * 64 + ((chB ^ ((0 ^ half) << 3)) << 2) + sub, fb1 = half * 8 * 64 + ((chB ^ ((1 ^ half) << 3)) << 2) + sub;

  const bool do_bias = p.bias_ws != nullptr && tile_ci == 0;
  float bsum = 0.f;                                                      // column (tid & 63), pixels 8 (tid >> 6) .. + 7 of every dY row
  const int bcol = tid & 63, bgrp = tid >> 6;
  const int boff = bgrp * 8 * 64 + ((((bcol >> 2) ^ ((bgrp & 1) << 3))) << 2) + (bcol & 3);

  for (int u = u_begin; u < u_end; ++u) {
    const int strip = u / p.H, y = u - strip * p.H;
    const int img = strip / p.nseg, x0 = (strip - img * p.nseg) * 32;
    if (u == u_begin || y == 0) {                                        // (re)start of a strip run: the whole 3-row halo, not overlapped
      __syncthreads();
      issue_x_row(img, x0, y - 1, (y + 3) & 3);
      issue_x_row(img, x0, y, y & 3);
      issue_x_row(img, x0, y + 1, (y + 1) & 3);
      issue_dy_row(img, x0, y, u & 1);
      dma_drain();
      __syncthreads();
    }
    if (u + 1 < u_end && y + 1 < p.H) {                                  // next step of the same strip: one new X row + its dY row
      issue_x_row(img, x0, y + 2, (y + 2) & 3);
      issue_dy_row(img, x0, y + 1, (u + 1) & 1);
    }
    const float* dyb = Ds + (u & 1) * DY_F;
    if (do_bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum += dyb[boff + e * 64];
    }
    const float* xr0 = Xs + ((y + 3) & 3) * XROW_F;
    const float* xr1 = Xs + (y & 3) * XROW_F;
    const float* xr2 = Xs + ((y + 1) & 3) * XROW_F;
    if constexpr (BF16) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = dyb[fa0 + (16 * h + j) * 64];  // bit 3 of 16 h + j is 0
        const bf16x8 af = __builtin_bit_cast(bf16x8, pack8(v));
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* xr = ky == 0 ? xr0 : (ky == 1 ? xr1 : xr2);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int px = 16 * h + j + kx;
              v[j] = xr[(((px >> 3) & 1) ? fb1 : fb0) + px * 64];
            }
            const bf16x8 bfr = __builtin_bit_cast(bf16x8, pack8(v));
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[ky * 3 + kx], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int prow = (j & 7) + 16 * (j >> 3);                        // pixels prow (lanes 0-31) and prow + 8 (lanes 32-63); bit 3 of prow is 0
        const float a = dyb[fa0 + prow * 64];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* xr = ky == 0 ? xr0 : (ky == 1 ? xr1 : xr2);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int px = prow + kx;
            const float b = xr[(((px >> 3) & 1) ? fb1 : fb0) + px * 64];
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ky * 3 + kx], 0, 0, 0);
          }
        }
      }
    }
    dma_drain();
    __syncthreads();
  }

  if (do_bias) {                                                         // four pixel groups per column, fixed order
    float* red = Ds;
    red[bgrp * 64 + bcol] = bsum;
    __syncthreads();
    if (tid < 64) p.bias_ws[(long long)z * p.Cout + co0 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
  }
  // raw partial tile -> ws[z][Cout][K], k = tap * Cin + ci
  float* W = p.ws + (long long)z * p.Cout * p.K;
  const int ci = ci0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      W[(long long)co * p.K + t * p.Cin + ci] = acc[t][r];
    }
}

}  // namespace

// Shapes the region kernel takes (the decision smx_wgrad_conv_ws_floats and the launcher share).
bool smx_wgrad_region_shape_ok(int nb, int Cout, int Cin, int Hin, int Win, int Ho, int Wo, int kh, int kw, int stride, int pad_t, int pad_l, int up2) {
  if (smx_tune(SMX_TUNE_WGRAD_REGION) == 0) return false;
  if (nb != 1 || kh != 3 || kw != 3 || stride != 1 || pad_t != 1 || pad_l != 1) return false;
  if (Cout % 64 || Cin % 64 || Wo % 32 || Ho < 1) return false;
  if (up2 ? (Ho != 2 * Hin || Wo != 2 * Win) : (Ho != Hin || Wo != Win)) return false;
  return true;
}

// Pixel split of the region form: about one block per CU (tuning knob wgrad_slots), at least 4 strip rows per block, workspace <= 128 MB.
int smx_wgrad_region_split(int M, int Cout, int Cin, int Ho, int Wo, int* uper_out) {
  const long long units = (long long)(M / (Ho * Wo)) * (Wo / 32) * Ho;
  const long long tiles = (long long)(Cout / 64) * (Cin / 64);
  const long long slots = smx_tune(SMX_TUNE_WGRAD_SLOTS) > 0 ? smx_tune(SMX_TUNE_WGRAD_SLOTS) : 256;
  long long ms = tiles >= slots ? 1 : slots / tiles;
  const long long by_rows = (units + 3) / 4;
  if (ms > by_rows) ms = by_rows;
  const long long cap = (128LL << 20) / 4 / ((long long)Cout * 9 * Cin);
  if (ms > cap) ms = cap;
  if (ms < 1) ms = 1;
  const long long uper = (units + ms - 1) / ms;
  ms = (units + uper - 1) / uper;                                        // no empty split
  if (uper_out) *uper_out = (int)uper;
  return (int)ms;
}

int smx_wgrad_region_launch(bool bf16, const float* dy, int ldy, const float* x, int ldx, int M, int Cout, int Hin, int Win, int Cin, int Ho, int Wo,
                            int up2, float* ws, float* bias_ws, int msplit, void* stream) {
  int uper = 0;
  if (smx_wgrad_region_split(M, Cout, Cin, Ho, Wo, &uper) != msplit) return SMX_EINVAL;
  RG p;
  p.dy = dy; p.x = x; p.ws = ws; p.bias_ws = bias_ws; p.ldy = ldy; p.ldx = ldx; p.Cout = Cout; p.Cin = Cin; p.K = 9 * Cin;
  p.B = M / (Ho * Wo); p.H = Ho; p.W = Wo; p.Hin = Hin; p.Win = Win; p.up2 = up2;
  p.nseg = Wo / 32; p.units = p.B * p.nseg * Ho; p.uper = uper; p.msplit = msplit; p.tiles_ci = Cin / 64;
  SMX_HIP(smx_max_dynamic_lds((const void*)wgrad_region_kernel<false>, LDS_F * 4));
  SMX_HIP(smx_max_dynamic_lds((const void*)wgrad_region_kernel<true>, LDS_F * 4));
  const dim3 grid((unsigned)((Cout / 64) * (Cin / 64)), (unsigned)msplit);
  if (bf16) SMX_LAUNCH(wgrad_region_kernel<true>, grid, dim3(256), LDS_F * 4, (hipStream_t)stream, p);
  else SMX_LAUNCH(wgrad_region_kernel<false>, grid, dim3(256), LDS_F * 4, (hipStream_t)stream, p);
  return SMX_OK;
}
