// 3x3 / stride 1 / pad 1 convolutions with C_out <= 4 on bf16 storage (configs[2]): the image head 64 -> 3 @ 256^2 (archs/vqgan_arch.py
// Generator's last conv) and the RefineFlow heads (archs/appmotioncodebook_arch.py RefineFlow: C -> 2 / 3), fp32 output.
//
// conv_small.hip runs these on the vector ALUs (the matrix cores would pad N to 32): right for fp32, but at B = 300 the bf16 step spent
// 5.7 ms there -- 2.85 ms for the image head alone against 0.5 ms of HBM time.  On the bf16 MFMA the padding is affordable (725 GFLOP
// executed for the image head = 0.3 ms at the pipe's peak), so this is the region-direct scheme of the other bf16 convolutions with one
// 32-wide N tile:
//   * a block owns an 8 x 32 output tile; per 64-channel group the (8+2) x (32+2) input region is staged ONCE (GroupNorm + swish of the
//     producer folded into the staging pass, as in conv_small.hip) as four 16-channel planes of 32 B per pixel, the 16-B half a lane reads
//     XOR-swizzled by bit 3 of the pixel's column (conflict-free ds_read_b128 for every tap);
//   * the group's weights -- 9 taps x 4 k-steps x 1 KB fragments, bf16 like every other layer of this configuration -- go global -> LDS
//     by LDS-DMA from a fragment-ordered pack (smx_conv3x3_smalln_mfma_pack);
//   * 4 waves x 2 output rows: 72 MFMAs per wave and group between two barriers; accumulators are [n][pixel], so lanes 0-31 hold the
//     <= 4 real channels of their pixel in registers 0-3: fp32 stores with bias / activation, no exchange.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TH = 8, TW = 32, RH = TH + 2, RW = TW + 2, RPX = RH * RW;   // 10 x 34 = 340 region pixels
constexpr int PLANE_B = RPX * 32;                                         // 10,880 B per 16-channel plane
constexpr int W_B = 9 * 4 * 1024;                                         // 36,864 B: [tap][k-step][64 lanes][16 B]
constexpr int LDS_B = 4 * PLANE_B + W_B;                                  // 80,384 B -> 2 blocks per CU

struct SN {
  const bf16_t* x; const bf16_t* wp; const float* bias; float* y; const float* in_ss;
  int lda, ldc, B, H, W, Cin, N, act, in_swish, tiles_y, tiles_x;
};

typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ float sn_act(float v, int act) {
  switch (act) {
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    case SMX_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case SMX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

__global__ __launch_bounds__(256, 2) void conv3x3_smalln_mfma16_kernel(SN p) {
  constexpr int NCH = (RPX * 8 + 255) / 256;                              // 16-B chunks (8 channels) of a 64-channel group per thread: 11
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Rg = smem;                                               // [4 k-steps][RPX][32 B]
  unsigned char* Ws = smem + 4 * PLANE_B;
  const unsigned lds_w = (unsigned)(uintptr_t)((lds_void*)Ws);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int bx = bid % p.tiles_x; bid /= p.tiles_x;
  const int by = bid % p.tiles_y; const int img = bid / p.tiles_y;
  const bf16_t* __restrict__ X = p.x + (long long)img * p.H * p.W * p.lda;
  const int c8 = tid & 7;                                                 // this thread's 8-channel chunk of the group (item & 7 == tid & 7 for all its items)

  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[r][q] = 0.f;
  const int pc = lane & 31, hh = lane >> 5;
  const int ngroups = p.Cin / 64;
  for (int g = 0; g < ngroups; ++g) {
    if (g) __syncthreads();                                               // every wave is past the previous group's region and weights
    // weights of the group: 36 DMA instructions of 1 KB, 9 per wave
    {
      const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wp) + (long long)g * W_B + lane * 16;
#pragma unroll
      for (int q = 0; q < 9; ++q) { const int i = wave * 9 + q; glds16(src + i * 1024, lds_w + (unsigned)(i * 1024)); }
    }
    // region: global -> registers -> (GroupNorm, swish) -> LDS
    float4 ssv[4];
    if (p.in_ss) {
      const float* sp = p.in_ss + ((long long)img * p.Cin + g * 64 + c8 * 8) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) ssv[e] = *reinterpret_cast<const float4*>(sp + 4 * e);
    }
    uint4 rreg[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int item = tid + 256 * k, px = item >> 3;
      const int ry = px / RW, rx = px - ry * RW;
      const int iy = by * TH - 1 + ry, ix = bx * TW - 1 + rx;
      const bool ok = item < RPX * 8 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      rreg[k] = ok ? *reinterpret_cast<const uint4*>(X + ((long long)iy * p.W + ix) * p.lda + g * 64 + c8 * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int item = tid + 256 * k, px = item >> 3;
      if (item < RPX * 8) {
        const int ry = px / RW, rx = px - ry * RW;
        const int iy = by * TH - 1 + ry, ix = bx * TW - 1 + rx;
        uint4 v = rreg[k];
        if (p.in_ss && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {      // the conv's zero padding stays exactly 0
          float f[8]; unpack8(v, f);
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const float4 s4 = ssv[e >> 1];
            f[e] = fmaf(f[e], s4.x, s4.y); f[e + 1] = fmaf(f[e + 1], s4.z, s4.w);
          }
          if (p.in_swish) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] / (1.f + expf(-f[e]));
          }
          v = pack8(f);
        }
        // chunk c8: k-step c8 >> 1, half c8 & 1 of its 16-channel plane
        *reinterpret_cast<uint4*>(Rg + (c8 >> 1) * PLANE_B + px * 32 + (((c8 & 1) ^ ((rx >> 3) & 1)) << 4)) = v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned char* wb = Ws + lane * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const bf16x8 wf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wb + (tap * 4 + ks) * 1024));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int col = pc + kx;
          const bf16x8 xf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(
              Rg + ks * PLANE_B + ((2 * wave + r + ky) * RW + col) * 32 + ((hh ^ ((col >> 3) & 1)) << 4)));
          acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[r], 0, 0, 0);     // [n][pixel]
        }
      }
  }
  // lanes 0-31 hold channels 0-3 of pixel (row, pc) in registers 0-3 (lanes 32-63: channels 4-7, never real here)
  if (hh == 0) {
    float* __restrict__ Y = p.y + (long long)img * p.H * p.W * p.ldc;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int oy = by * TH + 2 * wave + r, ox = bx * TW + pc;
      float* yp = Y + ((long long)oy * p.W + ox) * p.ldc;
#pragma unroll
      for (int n = 0; n < 4; ++n)
        if (n < p.N) yp[n] = sn_act(acc[r][n] + (p.bias ? p.bias[n] : 0.f), p.act);
    }
  }
}

// w [N][3][3][Cin] fp32 -> [Cin/64][tap][k-step][64 lanes][8] bf16: lane l holds row n = l & 31 (zeros for n >= N), channels 64 g + 16 ks + 8 (l >> 5) + 0..7
__global__ __launch_bounds__(256) void smalln_mfma_pack_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int N, int Cin) {
  const long long total = (long long)(Cin / 64) * 9 * 4 * 64;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    long long f = i >> 6;
    const int ks = (int)(f & 3); f >>= 2;
    const int tap = (int)(f % 9); const int g = (int)(f / 9);
    const int n = lane & 31, c0 = 64 * g + 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = n < N ? w[((long long)n * 9 + tap) * Cin + c0 + e] : 0.f;
    wp[i] = pack8(v);
  }
}

}  // namespace

extern "C" long long smx_conv3x3_smalln_mfma_pack_elems(int Cin, int N) {
  if (Cin <= 0 || Cin % 64 || N <= 0 || N > 4) return -1;
  return (long long)(Cin / 64) * 9 * 4 * 512;
}

extern "C" int smx_conv3x3_smalln_mfma_pack(const float* w, void* wp, int Cin, int N, void* stream) {
  if (!w || !wp || smx_conv3x3_smalln_mfma_pack_elems(Cin, N) < 0 || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  const long long total = (long long)(Cin / 64) * 9 * 4 * 64;
  int g = smx_cdiv(total, 256); if (g > 1024) g = 1024;
  SMX_LAUNCH(smalln_mfma_pack_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w, (uint4*)wp, N, Cin);
  return smx_launch_status();
}

extern "C" int smx_conv3x3_smalln_mfma_bf16(const void* x, int lda, const void* wp, const float* bias, float* y, int ldc, int B, int H, int W, int Cin,
                                            int Cout, int act, const float* in_ss, int in_swish, void* stream) {
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0 || H % TH || W % TW || smx_conv3x3_smalln_mfma_pack_elems(Cin, Cout) < 0) return SMX_EINVAL;
  if (lda < Cin || lda % 8 || ldc < Cout || ((uintptr_t)x & 15) || ((uintptr_t)wp & 15) || (in_ss && ((uintptr_t)in_ss & 15))) return SMX_EINVAL;
  if (act != SMX_ACT_NONE && act != SMX_ACT_RELU && act != SMX_ACT_LRELU02 && act != SMX_ACT_SIGMOID) return SMX_EINVAL;
  if ((long long)H * W * lda > 2147483647LL) return SMX_EINVAL;
  SN p;
  p.x = (const bf16_t*)x; p.wp = (const bf16_t*)wp; p.bias = bias; p.y = y; p.in_ss = in_ss; p.lda = lda; p.ldc = ldc; p.B = B; p.H = H; p.W = W;
  p.Cin = Cin; p.N = Cout; p.act = act; p.in_swish = in_swish; p.tiles_y = H / TH; p.tiles_x = W / TW;
  const long long blocks = (long long)B * p.tiles_y * p.tiles_x;
  if (blocks > 2147483647LL) return SMX_EINVAL;
  SMX_HIP(smx_max_dynamic_lds((const void*)conv3x3_smalln_mfma16_kernel, LDS_B));
  SMX_LAUNCH(conv3x3_smalln_mfma16_kernel, dim3((unsigned)blocks), dim3(256), LDS_B, (hipStream_t)stream, p);
  return smx_launch_status();
}
