// Short-K Linear / 1x1 convolution on bf16 storage, "row-panel" form (configs[2]): the token Linears of the transformer layers
// (nn.MultiheadAttention in/out projections, archs/appmotioncodebook_arch.py:69-70,101-115: M = B * 1024 tokens, K = 256, N = 256 / 512)
// and the other K = 128 / 256 1x1 convolutions of the path (SFT / fuse / to_context 1x1s).
//
// The implicit-GEMM kernel (gemm_bf16.hip) gives these launches 64x64 tiles: a block lives for 16 MFMAs per wave (K = 256 is four
// 64-deep slices), 19,200 blocks per launch at B = 300 -- per-block latency (request, LDS, barrier, transposed epilogue) times 19 rounds
// of blocks, 2.5x the launch's HBM time.  Here the weights never move and the rows stream:
//   * a block (4 waves) is PERSISTENT over 32-row tiles and owns 128 * NT output columns; wave w owns columns [32 NT w, 32 NT (w+1))
//     for every tile and keeps its B fragments -- NT x K/16 x 16 B per lane, 128 VGPRs at NT = 2, K = 256 -- in REGISTERS for its whole
//     life, loaded once from a fragment-ordered pack (smx_gemm_rp_bf16_pack: one coalesced 1 KB run per fragment).  Weights cost no
//     LDS space, no LDS reads and no per-tile traffic;
//   * the A tile (32 rows x K bf16 = 16 KB) goes global -> LDS by LDS-DMA (global_load_lds_dwordx4 from inline asm), double buffered:
//     tile t+1 lands while tile t is multiplied and stored; LDS keeps rows unpadded (the DMA writes 1 KB runs) and the 16-B chunk a
//     lane fetches is XOR-swizzled by (row & 15), so the 16 rows of a ds_read_b128 lane group hit 16 distinct bank quads;
//   * per 16-deep step a wave reads ONE A fragment (16 B per lane) for NT MFMAs; with N <= 256 every A row is read from HBM once;
//   * epilogue per wave and 32 x 32 tile, wave-private (no block barrier): accumulators are [n][row] (A operand = weights), so the
//     exchange is ds_write_b128; bias / activation / bf16 residual in fp32, 2 x 16-B stores of 16 channels per lane;
//   * one barrier per tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TM = 32;                         // rows per tile
constexpr int EXP = 36;                        // epilogue exchange pitch (floats): conflict-free ds_write_b128 over 16 rows
constexpr int EX_F = 32 * EXP;                 // floats per wave

struct RP {
  const bf16_t* a; const bf16_t* wp; const float* bias; const bf16_t* res; bf16_t* c;
  int lda, ldres, ldc, M, N, K, act, tiles;
  int d2s_p, d2s_c, Ho, Wo;                    // un-patchify (depth-to-space) store: n = (p1 * p + p2) * d2s_c + c -> pixel (oy p + p1, ox p + p2), channel c
};

typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {     // 64 lanes x 16 B -> LDS [lds_dst, +1 KB); M0 saved / restored
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void mem_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float rp_act(float v, int act) {
  switch (act) {
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    case SMX_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case SMX_ACT_SWISH: return v / (1.f + expf(-v));
    case SMX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case SMX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// KS = K / 16 (8 | 16), NT = 32-column tiles per wave (1 | 2)
// D2S: the un-patchify store (its own instantiation: the plain form sits at the 256-VGPR edge)
template <int KS, int NT, bool D2S = false>
__global__ __launch_bounds__(256, 2) void gemm_rp_bf16_kernel(RP p) {
  constexpr int CPR = KS * 2;                  // 16-B chunks per A row
  constexpr int ROWB = CPR * 16;               // bytes per A row in LDS
  constexpr int ATILE = TM * ROWB;             // 16 KB at K = 256
  constexpr int NDMA = ATILE / 1024 / 4;       // DMA instructions per wave and tile (4 | 2)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                    // [2][TM][ROWB]
  float* Ex = reinterpret_cast<float*>(smem + 2 * ATILE);   // [4 waves][32][EXP]
  const unsigned lds0 = (unsigned)(uintptr_t)((lds_void*)smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = (blockIdx.y * 4 + wave) * 32 * NT;          // this wave's first column

  // ---- B fragments: registers for the block's whole life --------------------------------------------------------------------
  uint4 bf[NT][KS];
  {
    const uint4* wp = reinterpret_cast<const uint4*>(p.wp) + ((long long)(n0 / 32) * KS) * 64 + lane;     // [n-tile][k-step][lane]
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bf[j][ks] = wp[(j * KS + ks) * 64];
  }
  // ---- A tile DMA: instruction q of this wave covers LDS bytes [(wave * NDMA + q) * 1024, +1024) of the tile -----------------------
  int dsrc[NDMA];                              // element offset of this lane's source chunk inside a tile
#pragma unroll
  for (int q = 0; q < NDMA; ++q) {
    const int slot = (wave * NDMA + q) * 64 + lane;            // 16-B slot in the tile
    const int row = slot / CPR, pos = slot % CPR;
    dsrc[q] = row * p.lda + ((pos ^ (row & 15)) << 3);
  }
  auto issue = [&](int t, int buf) {
    const bf16_t* base = p.a + (long long)t * TM * p.lda;
#pragma unroll
    for (int q = 0; q < NDMA; ++q) glds16(base + dsrc[q], lds0 + (unsigned)(buf * ATILE + (wave * NDMA + q) * 1024));
  };
  // A fragment: lane (row = l & 31, k half = l >> 5) reads logical chunk 2 ks + half of its row
  const int arow = lane & 31, ahalf = lane >> 5;
  const int aoff = arow * ROWB;
  float* ex = Ex + wave * EX_F;
  const int erow = lane >> 1, eh = lane & 1;   // epilogue read-back: row, 16-column half
  float bv[NT][16];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[j][e] = p.bias ? p.bias[n0 + 32 * j + 16 * eh + e] : 0.f;

  // un-patchify store: this lane's 16-column runs never straddle a sub-pixel (d2s_c % 16 == 0): resolve (p1, p2, c) once
  int d_p1[NT], d_p2[NT], d_oc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    d_p1[j] = d_p2[j] = d_oc[j] = 0;
    if (D2S) {
      const int nc = n0 + 32 * j + 16 * eh, dq = nc / p.d2s_c;
      d_oc[j] = nc - dq * p.d2s_c; d_p1[j] = dq / p.d2s_p; d_p2[j] = dq - d_p1[j] * p.d2s_p;
    }
  }
  int t = blockIdx.x, it = 0;
  if (t < p.tiles) issue(t, 0);
  for (; t < p.tiles; t += gridDim.x, ++it) {
    const int buf = it & 1;
    mem_drain();                               // this wave's part of tile t has landed (and its stores of the previous tile are out)
    __syncthreads();                           // everyone's has; everyone is past its reads of the other buffer
    if (t + (int)gridDim.x < p.tiles) issue(t + gridDim.x, buf ^ 1);
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned char* ab = As + buf * ATILE + aoff;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int pos = (2 * ks + ahalf) ^ (arow & 15);
      const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ab + pos * 16));
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[j][ks]), af, acc[j], 0, 0, 0);   // [n][row]
    }
    // ---- epilogue: per 32 x 32 tile through the wave's own exchange buffer ---------------------------------------------------------
    const long long grow = (long long)t * TM + erow;
    int d_img = 0, d_oy = 0, d_ox = 0;
    if (D2S) {
      const int hw = p.Ho * p.Wo;
      d_img = (int)(grow / hw); const int rem = (int)(grow - (long long)d_img * hw);
      d_oy = rem / p.Wo; d_ox = rem - d_oy * p.Wo;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nc = n0 + 32 * j + 16 * eh;
      uint4 r0 = make_uint4(0u, 0u, 0u, 0u), r1 = r0;
      if (p.res) {
        const bf16_t* rp = p.res + grow * p.ldres + nc;
        r0 = *reinterpret_cast<const uint4*>(rp); r1 = *reinterpret_cast<const uint4*>(rp + 8);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(ex + arow * EXP + 8 * g + 4 * ahalf) = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
      float v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(ex + erow * EXP + 16 * eh + 4 * q);
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += bv[j][e];
      if (p.act != SMX_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = rp_act(v[e], p.act);
      }
      if (p.res) {
        float q0[8], q1[8]; unpack8(r0, q0); unpack8(r1, q1);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] += q0[e]; v[8 + e] += q1[e]; }
      }
      const float lo[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}, hi[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
      bf16_t* cp = p.c + grow * p.ldc + nc;
      if (D2S) {
        const long long opix = ((long long)d_img * (p.Ho * p.d2s_p) + d_oy * p.d2s_p + d_p1[j]) * (p.Wo * p.d2s_p) + d_ox * p.d2s_p + d_p2[j];
        cp = p.c + opix * p.ldc + d_oc[j];
      }
      *reinterpret_cast<uint4*>(cp) = pack8(lo);
      *reinterpret_cast<uint4*>(cp + 8) = pack8(hi);
    }
  }
}

// W [N][ldw] (row n: K contiguous bf16) -> [N/32][K/16][64 lanes][8]: lane l of fragment (n-tile, k-step) holds row 32 nt + (l & 31),
// k = 16 ks + 8 (l >> 5) + 0..7 (the A-operand fragment of v_mfma_f32_32x32x16_bf16)
__global__ __launch_bounds__(256) void gemm_rp_pack_kernel(const bf16_t* __restrict__ w, int ldw, uint4* __restrict__ wp, int N, int K) {
  const int KS = K / 16;
  const long long total = (long long)(N / 32) * KS * 64;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    const long long f = i >> 6;
    const int ks = (int)(f % KS), nt = (int)(f / KS);
    wp[i] = *reinterpret_cast<const uint4*>(w + (long long)(nt * 32 + (lane & 31)) * ldw + ks * 16 + (lane >> 5) * 8);
  }
}

template <int KS, int NT, bool D2S = false>
int rp_launch(const RP& p, hipStream_t st) {
  constexpr int LDS = 2 * TM * KS * 32 + 4 * EX_F * 4;
  SMX_HIP(smx_max_dynamic_lds((const void*)gemm_rp_bf16_kernel<KS, NT, D2S>, LDS));
  const int ny = p.N / (128 * NT);
  int gx = 512 / ny; if (gx < 1) gx = 1; if (gx > p.tiles) gx = p.tiles;
  SMX_LAUNCH((gemm_rp_bf16_kernel<KS, NT, D2S>), dim3(gx, ny), dim3(256), LDS, st, p);
  return smx_launch_status();
}

}  // namespace

extern "C" int smx_gemm_rp_bf16_ok(long long M, int N, int K) {
  return (M > 0 && M % TM == 0 && M <= 2147483647LL && (K == 128 || K == 256) && N > 0 && N % 128 == 0) ? 1 : 0;
}

extern "C" int smx_gemm_rp_bf16_pack(const void* w, int ldw, void* wp, int N, int K, void* stream) {
  if (!w || !wp || N <= 0 || N % 32 || K <= 0 || K % 16 || ldw < K || ldw % 8 || ((uintptr_t)w & 15) || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  const long long total = (long long)(N / 32) * (K / 16) * 64;
  int g = smx_cdiv(total, 256); if (g > 4096) g = 4096;
  SMX_LAUNCH(gemm_rp_pack_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, ldw, (uint4*)wp, N, K);
  return smx_launch_status();
}

static int rp_bf16_launch(const void* a, int lda, const void* wp, const float* bias, const void* res, int ldres, void* c, int ldc,
                          long long M, int N, int K, int act, int d2s_p, int d2s_c, int Ho, int Wo, void* stream) {
  if (!a || !wp || !c || !smx_gemm_rp_bf16_ok(M, N, K)) return SMX_EINVAL;
  if (d2s_p) {
    if (d2s_p < 1 || d2s_c <= 0 || d2s_c % 16 || N != d2s_p * d2s_p * d2s_c || ldc < d2s_c || ldc % 8 || res || Ho <= 0 || Wo <= 0 || M % ((long long)Ho * Wo)) return SMX_EINVAL;
  } else if (ldc < N) return SMX_EINVAL;
  if (lda < K || lda % 8 || ldc % 8 || (res && (ldres < N || ldres % 8))) return SMX_EINVAL;
  if (((uintptr_t)a | (uintptr_t)wp | (uintptr_t)c | (uintptr_t)res) & 15) return SMX_EINVAL;
  if ((long long)TM * lda > 2147483647LL) return SMX_EINVAL;
  RP p;
  p.a = (const bf16_t*)a; p.wp = (const bf16_t*)wp; p.bias = bias; p.res = (const bf16_t*)res; p.c = (bf16_t*)c;
  p.lda = lda; p.ldres = res ? ldres : 0; p.ldc = ldc; p.M = (int)M; p.N = N; p.K = K; p.act = act; p.tiles = (int)(M / TM);
  p.d2s_p = d2s_p; p.d2s_c = d2s_c; p.Ho = Ho; p.Wo = Wo;
  hipStream_t st = (hipStream_t)stream;
  const bool two = N % 256 == 0;
  if (d2s_p) {                                 // one 32-column tile per wave: room for the store's coordinates
    if (K == 256) return rp_launch<16, 1, true>(p, st);
    return rp_launch<8, 1, true>(p, st);
  }
  if (K == 256) return two ? rp_launch<16, 2>(p, st) : rp_launch<16, 1>(p, st);
  return two ? rp_launch<8, 2>(p, st) : rp_launch<8, 1>(p, st);
}

extern "C" int smx_gemm_rp_bf16(const void* a, int lda, const void* wp, const float* bias, const void* res, int ldres, void* c, int ldc,
                                long long M, int N, int K, int act, void* stream) {
  return rp_bf16_launch(a, lda, wp, bias, res, ldres, c, ldc, M, N, K, act, 0, 0, 0, 0, stream);
}

/* the same with the un-patchify (depth-to-space) store of the patch Linears: c is [B][Ho p][Wo p][ldc >= d2s_c], M = B * Ho * Wo tokens */
extern "C" int smx_gemm_rp_d2s_bf16(const void* a, int lda, const void* wp, const float* bias, void* c, int ldc, long long M, int N, int K, int act,
                                    int d2s_p, int d2s_c, int Ho, int Wo, void* stream) {
  if (d2s_p < 1) return SMX_EINVAL;
  return rp_bf16_launch(a, lda, wp, bias, nullptr, 0, c, ldc, M, N, K, act, d2s_p, d2s_c, Ho, Wo, stream);
}
