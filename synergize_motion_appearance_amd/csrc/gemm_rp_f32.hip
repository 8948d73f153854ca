// Short-K Linear / 1x1 convolution in fp32, "row-panel" form: the fp32 (BASELINE configs[1]) counterpart of gemm_rp_bf16.hip for the token
// Linears of the transformer layers (nn.MultiheadAttention in/out projections, archs/appmotioncodebook_arch.py:69-70,101-115: M = B * 1024
// tokens, K = 256, N = 256 / 512) and the other K = 128 / 256 1x1 convolutions.  The implicit-GEMM kernel runs these launches at 0.56-0.63
// of the fp32 matrix pipe: K = 256 is eight 32-deep slices, so a block's prologue / epilogue weigh as much as its MFMAs.  Here:
//   * a block is PERSISTENT over 32-row tiles; wave w owns output columns [32 w, 32 w + 32) of the block's 32 NW for every tile and keeps
//     its weight fragments -- K / 2 floats per lane, 128 VGPRs at K = 256 -- in REGISTERS for its whole life (fragment-ordered pack,
//     smx_gemm_rp_f32_pack: one coalesced 1 KB run per 8-deep group).  NW = 8 waves cover N = 256 in one block: every A row is read once;
//   * the A tile (32 rows x K floats = 32 KB) goes global -> LDS by LDS-DMA (inline asm), double buffered; rows stay unpadded and the 16-B
//     chunk a lane fetches is XOR-swizzled by (row & 15): conflict-free ds_read_b128 over 16 rows;
//   * one ds_read_b128 feeds FOUR v_mfma_f32_32x32x2_f32 (the MFMA's k pairing is free: lanes 0-31 hold k = 8 i + t, lanes 32-63
//     k = 8 i + 4 + t of group i; the weight pack follows the same pairing): 128 MFMAs per wave per tile with 32 LDS reads;
//   * wave-private epilogue (accumulators are [n][row]: ds_write_b128 exchange), bias / activation / residual, 4 x 16-B stores per lane
//     = whole 128-B lines; one barrier per tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smx.h"
#include "smx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TM = 32;                         // rows per tile
constexpr int EXP = 36;                        // epilogue exchange pitch (floats)
constexpr int EX_F = 32 * EXP;

struct RPF {
  const float* a; const float* wp; const float* bias; const float* res; float* c;
  int lda, ldres, ldc, M, N, K, act, tiles;
  int d2s_p, d2s_c, Ho, Wo;                    // un-patchify (depth-to-space) store: n = (p1 * p + p2) * d2s_c + c -> pixel (oy p + p1, ox p + p2), channel c
};

typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void mem_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float rpf_act(float v, int act) {
  switch (act) {
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    case SMX_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case SMX_ACT_SWISH: return v / (1.f + expf(-v));
    case SMX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case SMX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// KG = K / 8 (16 | 32): 8-deep groups; NW = waves per block (4 | 8) = 32-column tiles per block
template <int KG, int NW, bool D2S = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void gemm_rp_f32_kernel(RPF p) {
  constexpr int CPR = KG * 2;                  // 16-B chunks per A row
  constexpr int ROWB = CPR * 16;
  constexpr int ATILE = TM * ROWB;             // 32 KB at K = 256
  constexpr int NDMA = ATILE / 1024 / NW;      // DMA instructions per wave and tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  float* Ex = reinterpret_cast<float*>(smem + 2 * ATILE);
  const unsigned lds0 = (unsigned)(uintptr_t)((lds_void*)smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = (blockIdx.y * NW + wave) * 32;

  float4 bf[KG];                               // weights of column n0 + (l & 31): k = 8 i + 4 (l >> 5) + 0..3
  {
    const float4* wp = reinterpret_cast<const float4*>(p.wp) + ((long long)(n0 / 32) * KG) * 64 + lane;
#pragma unroll
    for (int i = 0; i < KG; ++i) bf[i] = wp[i * 64];
  }
  int dsrc[NDMA];
#pragma unroll
  for (int q = 0; q < NDMA; ++q) {
    const int slot = (wave * NDMA + q) * 64 + lane;
    const int row = slot / CPR, pos = slot % CPR;
    dsrc[q] = row * p.lda + ((pos ^ (row & 15)) << 2);
  }
  auto issue = [&](int t, int buf) {
    const float* base = p.a + (long long)t * TM * p.lda;
#pragma unroll
    for (int q = 0; q < NDMA; ++q) glds16(base + dsrc[q], lds0 + (unsigned)(buf * ATILE + (wave * NDMA + q) * 1024));
  };
  const int arow = lane & 31, ahalf = lane >> 5;
  const int aoff = arow * ROWB;
  float* ex = Ex + wave * EX_F;
  const int erow = lane >> 1, eh = lane & 1;
  float bv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bv[e] = p.bias ? p.bias[n0 + 16 * eh + e] : 0.f;

  int d_p1 = 0, d_p2 = 0, d_oc = 0;            // un-patchify: this lane's 16-column run never straddles a sub-pixel (d2s_c % 16 == 0)
  if (D2S) {
    const int nc = n0 + 16 * eh, dq = nc / p.d2s_c;
    d_oc = nc - dq * p.d2s_c; d_p1 = dq / p.d2s_p; d_p2 = dq - d_p1 * p.d2s_p;
  }
  int t = blockIdx.x, it = 0;
  if (t < p.tiles) issue(t, 0);
  for (; t < p.tiles; t += gridDim.x, ++it) {
    const int buf = it & 1;
    mem_drain();
    __syncthreads();
    if (t + (int)gridDim.x < p.tiles) issue(t + gridDim.x, buf ^ 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned char* ab = As + buf * ATILE + aoff;
#pragma unroll
    for (int i = 0; i < KG; ++i) {
      const int pos = (2 * i + ahalf) ^ (arow & 15);
      const float4 af = *reinterpret_cast<const float4*>(ab + pos * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[i].x, af.x, acc, 0, 0, 0);     // A operand = weights: the tile is [n][row]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[i].y, af.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[i].z, af.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[i].w, af.w, acc, 0, 0, 0);
    }
    const long long grow = (long long)t * TM + erow;
    const int nc = n0 + 16 * eh;
    float4 rq[4];
    if (p.res) {
      const float* rp = p.res + grow * p.ldres + nc;
#pragma unroll
      for (int q = 0; q < 4; ++q) rq[q] = *reinterpret_cast<const float4*>(rp + 4 * q);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(ex + arow * EXP + 8 * g + 4 * ahalf) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    float* cp = p.c + grow * p.ldc + nc;
    if (D2S) {
      const int hw = p.Ho * p.Wo;
      const int img = (int)(grow / hw), rem = (int)(grow - (long long)img * hw);
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const long long opix = ((long long)img * (p.Ho * p.d2s_p) + oy * p.d2s_p + d_p1) * (p.Wo * p.d2s_p) + ox * p.d2s_p + d_p2;
      cp = p.c + opix * p.ldc + d_oc;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 x = *reinterpret_cast<const float4*>(ex + erow * EXP + 16 * eh + 4 * q);
      x.x += bv[4 * q]; x.y += bv[4 * q + 1]; x.z += bv[4 * q + 2]; x.w += bv[4 * q + 3];
      if (p.act != SMX_ACT_NONE) { x.x = rpf_act(x.x, p.act); x.y = rpf_act(x.y, p.act); x.z = rpf_act(x.z, p.act); x.w = rpf_act(x.w, p.act); }
      if (p.res) { x.x += rq[q].x; x.y += rq[q].y; x.z += rq[q].z; x.w += rq[q].w; }
      *reinterpret_cast<float4*>(cp + 4 * q) = x;
    }
  }
}

// W [N][ldw] (row n: K contiguous floats) -> [N/32][K/8][64 lanes][4]: lane l of group (n-tile, i) holds row 32 nt + (l & 31),
// k = 8 i + 4 (l >> 5) + 0..3
__global__ __launch_bounds__(256) void gemm_rp_f32_pack_kernel(const float* __restrict__ w, int ldw, float4* __restrict__ wp, int N, int K) {
  const int KG = K / 8;
  const long long total = (long long)(N / 32) * KG * 64;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    const long long f = i >> 6;
    const int g = (int)(f % KG), nt = (int)(f / KG);
    wp[i] = *reinterpret_cast<const float4*>(w + (long long)(nt * 32 + (lane & 31)) * ldw + g * 8 + (lane >> 5) * 4);
  }
}

template <int KG, int NW, bool D2S = false>
int rpf_launch(const RPF& p, hipStream_t st) {
  constexpr int LDS = 2 * TM * KG * 32 + NW * EX_F * 4;
  SMX_HIP(smx_max_dynamic_lds((const void*)gemm_rp_f32_kernel<KG, NW, D2S>, LDS));
  const int ny = p.N / (32 * NW);
  int gx = (NW == 8 ? 256 : 512) / ny; if (gx < 1) gx = 1; if (gx > p.tiles) gx = p.tiles;
  SMX_LAUNCH((gemm_rp_f32_kernel<KG, NW, D2S>), dim3(gx, ny), dim3(64 * NW), LDS, st, p);
  return smx_launch_status();
}

}  // namespace

extern "C" int smx_gemm_rp_f32_ok(long long M, int N, int K) {
  return (M > 0 && M % TM == 0 && M <= 2147483647LL && (K == 128 || K == 256) && N > 0 && N % 128 == 0) ? 1 : 0;
}

extern "C" int smx_gemm_rp_f32_pack(const float* w, int ldw, float* wp, int N, int K, void* stream) {
  if (!w || !wp || N <= 0 || N % 32 || K <= 0 || K % 8 || ldw < K || ldw % 4 || ((uintptr_t)w & 15) || ((uintptr_t)wp & 15)) return SMX_EINVAL;
  const long long total = (long long)(N / 32) * (K / 8) * 64;
  int g = smx_cdiv(total, 256); if (g > 4096) g = 4096;
  SMX_LAUNCH(gemm_rp_f32_pack_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w, ldw, (float4*)wp, N, K);
  return smx_launch_status();
}

static int rp_f32_launch(const float* a, int lda, const float* wp, const float* bias, const float* res, int ldres, float* c, int ldc,
                         long long M, int N, int K, int act, int d2s_p, int d2s_c, int Ho, int Wo, void* stream) {
  if (!a || !wp || !c || !smx_gemm_rp_f32_ok(M, N, K)) return SMX_EINVAL;
  if (d2s_p) {
    if (d2s_p < 1 || d2s_c <= 0 || d2s_c % 16 || N != d2s_p * d2s_p * d2s_c || N % 256 || ldc < d2s_c || res || Ho <= 0 || Wo <= 0 || M % ((long long)Ho * Wo)) return SMX_EINVAL;
  } else if (ldc < N) return SMX_EINVAL;
  if (lda < K || lda % 4 || ldc % 4 || (res && (ldres < N || ldres % 4))) return SMX_EINVAL;
  if (((uintptr_t)a | (uintptr_t)wp | (uintptr_t)c | (uintptr_t)res) & 15) return SMX_EINVAL;
  if ((long long)TM * lda > 2147483647LL) return SMX_EINVAL;
  RPF p;
  p.a = a; p.wp = wp; p.bias = bias; p.res = res; p.c = c;
  p.lda = lda; p.ldres = res ? ldres : 0; p.ldc = ldc; p.M = (int)M; p.N = N; p.K = K; p.act = act; p.tiles = (int)(M / TM);
  p.d2s_p = d2s_p; p.d2s_c = d2s_c; p.Ho = Ho; p.Wo = Wo;
  hipStream_t st = (hipStream_t)stream;
  if (d2s_p) return K == 256 ? rpf_launch<32, 8, true>(p, st) : rpf_launch<16, 8, true>(p, st);
  const bool wide = N % 256 == 0;
  if (K == 256) return wide ? rpf_launch<32, 8>(p, st) : rpf_launch<32, 4>(p, st);
  return wide ? rpf_launch<16, 8>(p, st) : rpf_launch<16, 4>(p, st);
}

extern "C" int smx_gemm_rp_f32(const float* a, int lda, const float* wp, const float* bias, const float* res, int ldres, float* c, int ldc,
                               long long M, int N, int K, int act, void* stream) {
  return rp_f32_launch(a, lda, wp, bias, res, ldres, c, ldc, M, N, K, act, 0, 0, 0, 0, stream);
}

/* with the un-patchify (depth-to-space) store of the patch Linears: c is [B][Ho p][Wo p][ldc >= d2s_c], M = B * Ho * Wo tokens, N % 256 == 0 */
extern "C" int smx_gemm_rp_d2s_f32(const float* a, int lda, const float* wp, const float* bias, float* c, int ldc, long long M, int N, int K, int act,
                                   int d2s_p, int d2s_c, int Ho, int Wo, void* stream) {
  if (d2s_p < 1) return SMX_EINVAL;
  return rp_f32_launch(a, lda, wp, bias, nullptr, 0, c, ldc, M, N, K, act, d2s_p, d2s_c, Ho, Wo, stream);
}
