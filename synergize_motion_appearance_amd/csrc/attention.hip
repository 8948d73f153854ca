// Fused multi-head attention for the codebook transformers (nn.MultiheadAttention as used by
// TransformerLayer, archs/appmotioncodebook_arch.py:69-70,101-115) -- gfx950, fp32 exact.
//
// Round-1a materialised the [B,H,1024,S] score tensors (QK^T GEMM -> softmax pass -> PV GEMM):
// 33.5 MB per (frame, layer) written and read twice, and K = d_head = 4 / 32 GEMMs that are one
// short K-slice.  Here softmax(q k^T) v is one kernel per layer call, online-softmax, nothing
// but q, k, v, o and the key-padding mask touches HBM.
//
// d_head = 32 (appearance, E=256): fp32 MFMA, "swapped" products so that every softmax
// statistic is lane-local:
//     S^T = K Q^T   (A = K tile from LDS,  B = Q^T fragments held in registers)
//     O^T = V^T P^T (A = V tile from LDS,  B = P^T = exp(S^T - m), straight from the S^T registers)
//   In the 32x32x2 C layout lane l owns query (l&31) and 16 of the 32 keys of a tile (the other 16
//   sit in lane l^32), so the row max / sum are 15 in-register ops + ONE wavefront shuffle, the
//   O rescale is a lane-local scalar, and P feeds the second MFMA with no transpose or LDS trip
//   (the k-pairing of the MFMA is free: A and B fragments just have to agree).
// d_head = 4 (motion, E=32): the 32x32 MFMAs would be 8x padded; the 16-block 4x4x1 MFMA fits exactly
//   (attn_mfma4_kernel below); the VALU kernel (one query per lane, keys/values broadcast from LDS) stays selectable.
// Fully masked rows give NaN exactly like the reference (0/0).
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include <math.h>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// T = storage type of q / k / v / o (float | bf16_t); products and softmax statistics are fp32 in both
template <typename T>
struct AP {
  const T* q; const T* k; const T* v; T* o; const uint8_t* mask;
  long long q_bs, k_bs, v_bs, o_bs;
  int ldq, ldk, ldv, ldo;
  int H, L, S; float scale;
};

// MASK: a key-padding mask exists (the appearance cross-attention); without one (every self-attention call) the per-score compare / select
// against the staged mask bytes -- 49 of the ~180 non-MFMA instructions of a 32-key tile -- and the mask staging are compiled out
// lanes' bit set: `if_set`, otherwise `otherwise` (one v_cndmask with a wave-uniform lane mask in an SGPR pair)
__device__ __forceinline__ float lane_select(unsigned long long lanes, float if_set, float otherwise) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(otherwise), "v"(if_set), "s"(lanes));
  return r;
}

template <typename T, int DH, bool MASK = true>
__global__ __launch_bounds__(256, 2) void attn_mfma_kernel(AP<T> p) {
  constexpr int TK = 32, KLD = DH + 4, KS = DH / 8, DT = DH / 32;
  __shared__ __attribute__((aligned(16))) float Ks[2][TK * KLD];
  __shared__ __attribute__((aligned(16))) float Vs[2][TK * DH];
  __shared__ uint8_t Ms[2][TK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int qrow = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int hh = lane >> 5;
  const T* Q = p.q + b * p.q_bs + (long long)qrow * p.ldq + h * DH;
  const T* K = p.k + b * p.k_bs + h * DH;
  const T* V = p.v + b * p.v_bs + h * DH;
  const uint8_t* M = (MASK && p.mask) ? p.mask + (long long)b * p.S : nullptr;

  float4 qf[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    float4 t = St<T>::ld4(Q + kk * 8 + hh * 4);
    const float sc = p.scale * 1.44269504088896340736f;      // scores in log2 units: softmax via v_exp_f32 (exp2)
    qf[kk] = make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc);
  }
  f32x16 oacc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m = -INFINITY, l = 0.f;

  // staging: TK x DH floats per operand = TK*DH/4 float4, 256 threads
  constexpr int NF4 = TK * DH / 4, PER = (NF4 + 255) / 256;
  float4 kreg[PER], vreg[PER]; uint8_t mreg = 0;
  auto load_tile = [&](int key0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = threadIdx.x + 256 * i;
      if (f < NF4) {
        const int r = f / (DH / 4), c4 = f % (DH / 4);
        kreg[i] = St<T>::ld4(K + (long long)(key0 + r) * p.ldk + c4 * 4);
        vreg[i] = St<T>::ld4(V + (long long)(key0 + r) * p.ldv + c4 * 4);
      }
    }
    if (MASK && M && threadIdx.x < TK) mreg = M[key0 + threadIdx.x];
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = threadIdx.x + 256 * i;
      if (f < NF4) {
        const int r = f / (DH / 4), c4 = f % (DH / 4);
        *reinterpret_cast<float4*>(&Ks[buf][r * KLD + c4 * 4]) = kreg[i];
        *reinterpret_cast<float4*>(&Vs[buf][r * DH + c4 * 4]) = vreg[i];
      }
    }
    if (MASK && threadIdx.x < TK) Ms[buf][threadIdx.x] = M ? mreg : 0;
  };

  const int ntiles = p.S / TK;
  bool seen_live = !MASK;                                       // wave-uniform: some key of the tiles so far is not masked
  load_tile(0); store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile((t + 1) * TK);
    // S^T = K Q^T
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* kp = &Ks[buf][(lane & 31) * KLD + hh * 4];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const float4 kf = *reinterpret_cast<const float4*>(kp + kk * 8);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[kk].x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[kk].y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[kk].z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[kk].w, s, 0, 0, 0);
    }
    // A masked key is masked for EVERY query, so the tile's mask is one wave-uniform 32-bit word: the per-score select takes its lane mask from two
    // scalar bit tests (keys k and k + 4 of the two half-waves) instead of an LDS byte, a compare and a select per score, a tile without masked keys
    // pays nothing, and "every key so far masked" is a scalar flag, not a per-lane select on alpha and on each probability.
    unsigned mb = 0u;
    if (MASK) {
      mb = (unsigned)__ballot(Ms[buf][lane & 31] != 0);
      seen_live = seen_live || mb != 0xFFFFFFFFu;
    }
    if (!MASK || seen_live) {
      if (MASK && mb != 0u) {
        const float ninf = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k0 = (r & 3) + 8 * (r >> 2);
          const unsigned long long lanes = (unsigned long long)(0u - ((mb >> k0) & 1u)) | ((unsigned long long)(0u - ((mb >> (k0 + 4)) & 1u)) << 32);
          s[r] = lane_select(lanes, ninf, s[r]);
        }
      }
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m, tmax);                      // finite: this tile or an earlier one holds a live key
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_new); psum += s[r]; }
      psum += __shfl_xor(psum, 32, 64);
      l = l * alpha + psum; m = m_new;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
      // O^T += V^T P^T
      const float* vp = &Vs[buf][(lane & 31)];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
        for (int d = 0; d < DT; ++d)
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[key * DH + d * 32], s[r], oacc[d], 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }
  T* O = p.o + b * p.o_bs + (long long)qrow * p.ldo + h * DH;
  const float inv = 1.f / l;                                  // l == 0 (fully masked row) -> inf * 0 = NaN like the reference
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 w = make_float4(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
      if (l == 0.f) w = make_float4(NAN, NAN, NAN, NAN);
      St<T>::st4(O + d * 32 + 8 * g + 4 * hh, w);
    }
}

// d_head = 32 on the bf16 MFMA (configs[2]): the same swapped-product scheme on v_mfma_f32_32x32x16_bf16.
//   S^T = K Q^T   A = K tile [key][d] from LDS (lane: key l&31, d chunk 8*(l>>5)), B = Q^T fragments in registers
//   O^T = V^T P^T A = V^T tile [d][key] from LDS,  B = P^T = exp2(S^T - m) packed to bf16 straight from the S^T registers
// The MFMA only needs A and B to agree on which k sits in which of a lane's 8 slots, so P keeps the accumulator's own
// key order (a lane holds keys {0-3, 8-11} + 4*(l>>5) of each 16) and V^T is WRITTEN to LDS with its key columns in that
// order -- no cross-lane exchange (v_permlane32_swap) on the softmax path.  At 16x the fp32 matrix rate the tile is bound by
// the softmax VALU work, not the matrix pipe: 64 keys per step (one barrier per 64 keys), scale folded into the exp2 argument.
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_mfma16_kernel(AP<bf16_t> p) {
  static_assert(DH == 32, "one 32-wide d tile");
  constexpr int TK = 64, KROW = DH * 2 + 16, VROW = TK * 2 + 16;          // LDS row bytes (padded)
  __shared__ __attribute__((aligned(16))) unsigned char Ks[2][TK * KROW];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[2][DH * VROW];
  __shared__ uint8_t Ms[2][TK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int qrow = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int hh = lane >> 5;
  const bf16_t* Q = p.q + b * p.q_bs + (long long)qrow * p.ldq + h * DH;
  const bf16_t* K = p.k + b * p.k_bs + h * DH;
  const bf16_t* V = p.v + b * p.v_bs + h * DH;
  const uint8_t* M = p.mask ? p.mask + (long long)b * p.S : nullptr;
  const float c = p.scale * 1.44269504088896340736f;                     // scores in log2 units

  bf16x8 qf[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) qf[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Q + kk * 16 + hh * 8));
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float m = -INFINITY, l = 0.f;

  // staging: K tile 64 keys x 4 chunks(16 B) = 256 chunks, V tile likewise: one chunk of each per thread
  const int skey = threadIdx.x >> 2, sc = threadIdx.x & 3;
  // LDS column of a key inside V^T: per 16 keys the order {0-3, 8-11, 4-7, 12-15}
  const int kq = skey & 15;
  const int vcol = (skey & ~15) + (kq < 4 ? kq : kq < 8 ? kq + 4 : kq < 12 ? kq - 4 : kq);
  uint4 kreg, vreg; uint8_t mreg = 0;
  auto load_tile = [&](int key0) {
    kreg = *reinterpret_cast<const uint4*>(K + (long long)(key0 + skey) * p.ldk + sc * 8);
    vreg = *reinterpret_cast<const uint4*>(V + (long long)(key0 + skey) * p.ldv + sc * 8);
    if (M && threadIdx.x < TK) mreg = M[key0 + threadIdx.x];
  };
  auto store_tile = [&](int buf) {
    *reinterpret_cast<uint4*>(&Ks[buf][skey * KROW + sc * 16]) = kreg;
    const uint32_t w4[4] = {vreg.x, vreg.y, vreg.z, vreg.w};
#pragma unroll
    for (int e = 0; e < 8; ++e)                                          // transpose: V[key][8sc+e] -> Vt[8sc+e][col(key)]
      *reinterpret_cast<uint16_t*>(&Vt[buf][(sc * 8 + e) * VROW + vcol * 2]) = (uint16_t)(e & 1 ? w4[e >> 1] >> 16 : w4[e >> 1] & 0xffffu);
    if (threadIdx.x < TK) Ms[buf][threadIdx.x] = M ? mreg : 0;
  };

  const int ntiles = p.S / TK;
  load_tile(0); store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile((t + 1) * TK);
    // S^T = K Q^T for the two 32-key halves of the tile
    f32x16 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
      const unsigned char* kp = &Ks[buf][(u * 32 + (lane & 31)) * KROW + hh * 16];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kp + kk * 32)), qf[kk], s[u], 0, 0, 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (Ms[buf][key]) s[u][r] = -INFINITY;
        tmax = fmaxf(tmax, s[u][r]);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax * c);
    const bool dead = (m_new == -INFINITY);                              // every key so far masked
    const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m - m_new);
    float psum = 0.f;
    bf16x8 pf[4];                                                        // P^T fragments: 4 MFMAs x 16 keys
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float e[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { e[r] = dead ? 0.f : __builtin_amdgcn_exp2f(fmaf(s[u][r], c, -m_new)); psum += e[r]; }
      // accumulator regs 0-7 = keys {0-3, 8-11} + 4hh of the first 16, regs 8-15 = the same of the second 16
      pf[2 * u] = __builtin_bit_cast(bf16x8, make_uint4(pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])));
      pf[2 * u + 1] = __builtin_bit_cast(bf16x8, make_uint4(pack2(e[8], e[9]), pack2(e[10], e[11]), pack2(e[12], e[13]), pack2(e[14], e[15])));
    }
    psum += __shfl_xor(psum, 32, 64);
    l = l * alpha + psum; m = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
    // O^T += V^T P^T: A = V^T rows d = lane&31, this lane's 8 key slots of each 16-key group
    const unsigned char* vp = &Vt[buf][(lane & 31) * VROW + hh * 16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(vp + g * 32)), pf[g], oacc, 0, 0, 0);
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }
  bf16_t* O = p.o + b * p.o_bs + (long long)qrow * p.ldo + h * DH;
  const float inv = 1.f / l;                                             // l == 0 (fully masked row) -> NaN like the reference
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 w = make_float4(oacc[4 * g] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv);
    if (l == 0.f) w = make_float4(NAN, NAN, NAN, NAN);
    St<bf16_t>::st4(O + 8 * g + 4 * hh, w);
  }
}

// d_head = 32 on fp32 storage with fp32-grade products on the BF16 matrix pipe ("bf16x6", the arithmetic of csrc/winograd_bf3.hip / gemm_rp_bf3.hip): every
// operand is split exactly three ways into bf16 (x = x1 + x2 + x3) and a product is the six bf16 MFMA products down to 2^-24.  The fp32 kernel above is
// matrix-bound (64 fp32 MFMAs = 4096 pipe cycles per wave and 64 keys); here the same keys cost 48 bf16 MFMAs (1536 cycles) and the work moves to the VALU:
//   * Q^T (pre-scaled by scale * log2 e in fp32, like the fp32 kernel) is split once per block into registers;
//   * a K / V tile of 64 keys is split while it is staged (each thread 8 floats of one key of each: global -> registers one tile ahead -> three bf16 level
//     planes in LDS; V is written transposed, in the S^T accumulator's key order, like attn_mfma16_kernel);
//   * P^T = exp2(S^T - m) is split in registers straight from the S^T accumulators (NP = 3 levels: six products; NP = 2: P carries 16 significand bits and the
//     PV product is five);
//   * the S^T tiles of the two key halves alternate on the pipe; the O^T accumulation runs on two chains.
// Softmax statistics fp32 and lane-local exactly as in the kernels above; fully masked rows give NaN like the reference.
__device__ __forceinline__ unsigned ab3_cvt2(float a, float b) { return cvt2_bf16(a, b); }
// eight fp32 -> NL bf16 level fragments (round to nearest even at every level)
template <int NL>
__device__ __forceinline__ void ab3_split8(const float (&v)[8], uint4 (&lv)[3]) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h[q] = ab3_cvt2(v[2 * q], v[2 * q + 1]);
    const float r0 = v[2 * q] - __uint_as_float(h[q] << 16), r1 = v[2 * q + 1] - __uint_as_float(h[q] & 0xffff0000u);
    m[q] = ab3_cvt2(r0, r1);
    l[q] = NL > 2 ? ab3_cvt2(r0 - __uint_as_float(m[q] << 16), r1 - __uint_as_float(m[q] & 0xffff0000u)) : 0u;
  }
  lv[0] = make_uint4(h[0], h[1], h[2], h[3]); lv[1] = make_uint4(m[0], m[1], m[2], m[3]); lv[2] = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ bf16x8 ab3_f(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// NW = waves per block (4: 128 queries, two blocks per CU | 8: 256 queries, one block per CU -- the K / V split is shared by twice the queries)
template <bool MASK, int NP, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void attn_bf3_kernel(AP<float> p) {
  constexpr int DH = 32, TK = 64, KROW = DH * 2 + 16, VROW = TK * 2 + 16;  // LDS row bytes (padded)
  constexpr int KPL = TK * KROW, VPL = DH * VROW;                         // one level plane
  __shared__ __attribute__((aligned(16))) unsigned char Ks[2][3 * KPL];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[2][3 * VPL];
  __shared__ uint8_t Ms[2][TK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int qrow = blockIdx.x * (32 * NW) + wave * 32 + (lane & 31);
  const int hh = lane >> 5;
  const float* Q = p.q + b * p.q_bs + (long long)qrow * p.ldq + h * DH;
  const float* K = p.k + b * p.k_bs + h * DH;
  const float* V = p.v + b * p.v_bs + h * DH;
  const uint8_t* M = (MASK && p.mask) ? p.mask + (long long)b * p.S : nullptr;

  uint4 qf[2][3];
  {
    const float sc = p.scale * 1.44269504088896340736f;                   // scores in log2 units
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const float4 a0 = *reinterpret_cast<const float4*>(Q + kk * 16 + hh * 8), a1 = *reinterpret_cast<const float4*>(Q + kk * 16 + hh * 8 + 4);
      const float v[8] = {a0.x * sc, a0.y * sc, a0.z * sc, a0.w * sc, a1.x * sc, a1.y * sc, a1.z * sc, a1.w * sc};
      ab3_split8<3>(v, qf[kk]);
    }
  }
  f32x16 oacc, oacb;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = oacb[r] = 0.f;
  float m = -INFINITY, l = 0.f;

  // staging: a tile is 64 keys x 32 floats of K and of V.  NW = 4: 256 groups of 8 floats, one of each operand per thread; NW = 8: 512 groups of 4
  constexpr int GF = NW == 4 ? 8 : 4, GPK = DH / GF;                      // floats per group, groups per key
  const int skey = threadIdx.x / GPK, sg = threadIdx.x % GPK;
  const int kq = skey & 15;
  const int vcol = (skey & ~15) + (kq < 4 ? kq : kq < 8 ? kq + 4 : kq < 12 ? kq - 4 : kq);       // per 16 keys the order {0-3, 8-11, 4-7, 12-15}
  float4 kreg[GF / 4], vreg[GF / 4]; uint8_t mreg = 0;
  auto load_tile = [&](int key0) {
    const float* kp = K + (long long)(key0 + skey) * p.ldk + sg * GF;
    const float* vp = V + (long long)(key0 + skey) * p.ldv + sg * GF;
#pragma unroll
    for (int i = 0; i < GF / 4; ++i) { kreg[i] = *reinterpret_cast<const float4*>(kp + 4 * i); vreg[i] = *reinterpret_cast<const float4*>(vp + 4 * i); }
    if (MASK && M && threadIdx.x < TK) mreg = M[key0 + threadIdx.x];
  };
  auto store_tile = [&](int buf) {
    float kv[8], vv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { kv[i] = 0.f; vv[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < GF / 4; ++i) {
      kv[4 * i] = kreg[i].x; kv[4 * i + 1] = kreg[i].y; kv[4 * i + 2] = kreg[i].z; kv[4 * i + 3] = kreg[i].w;
      vv[4 * i] = vreg[i].x; vv[4 * i + 1] = vreg[i].y; vv[4 * i + 2] = vreg[i].z; vv[4 * i + 3] = vreg[i].w;
    }
    uint4 kl[3], vl[3];
    ab3_split8<3>(kv, kl);                                                // (GF == 4: the upper half is zeros the compiler drops)
    ab3_split8<3>(vv, vl);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      unsigned char* kd = &Ks[buf][s3 * KPL + skey * KROW + sg * (2 * GF)];
      if (GF == 8) *reinterpret_cast<uint4*>(kd) = kl[s3]; else *reinterpret_cast<uint2*>(kd) = make_uint2(kl[s3].x, kl[s3].y);
      const uint32_t w4[4] = {vl[s3].x, vl[s3].y, vl[s3].z, vl[s3].w};
#pragma unroll
      for (int e = 0; e < GF; ++e)                                        // transpose: V[key][GF sg + e] -> Vt[GF sg + e][col(key)]
        *reinterpret_cast<uint16_t*>(&Vt[buf][s3 * VPL + (sg * GF + e) * VROW + vcol * 2]) = (uint16_t)(e & 1 ? w4[e >> 1] >> 16 : w4[e >> 1] & 0xffffu);
    }
    if (MASK && threadIdx.x < TK) Ms[buf][threadIdx.x] = M ? mreg : 0;
  };

  const int ntiles = p.S / TK;
  load_tile(0); store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile((t + 1) * TK);
    // S^T = K Q^T for the two 32-key halves of the tile, the halves alternating on the pipe; smallest products first
    f32x16 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 kf[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) kf[u][s3] = *reinterpret_cast<const uint4*>(&Ks[buf][s3 * KPL + (u * 32 + (lane & 31)) * KROW + hh * 16 + kk * 32]);
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};   // (K level, Q level): k1q3 k3q1 k2q2 k1q2 k2q1 k1q1
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab3_f(kf[u][PA[pr]]), ab3_f(qf[kk][PB[pr]]), s[u], 0, 0, 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (MASK) {
          const int key = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (Ms[buf][key]) s[u][r] = -INFINITY;
        }
        tmax = fmaxf(tmax, s[u][r]);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax);
    const bool dead = MASK && (m_new == -INFINITY);                       // every key so far masked
    const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m - m_new);
    float psum = 0.f;
    uint4 pf[4][3];                                                       // P^T fragments: 4 groups of 16 keys x levels
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float e[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { e[r] = dead ? 0.f : __builtin_amdgcn_exp2f(s[u][r] - m_new); psum += e[r]; }
      // accumulator regs 0-7 = keys {0-3, 8-11} + 4 hh of the first 16, regs 8-15 = the same of the second 16
      const float e0[8] = {e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]}, e1[8] = {e[8], e[9], e[10], e[11], e[12], e[13], e[14], e[15]};
      ab3_split8<NP>(e0, pf[2 * u]);
      ab3_split8<NP>(e1, pf[2 * u + 1]);
    }
    psum += __shfl_xor(psum, 32, 64);
    l = l * alpha + psum; m = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {             // no lane's running maximum moved (the usual tile after the first few): nothing to rescale
#pragma unroll
      for (int r = 0; r < 16; ++r) { oacc[r] *= alpha; oacb[r] *= alpha; }
    }
    // O^T += V^T P^T: A = V^T rows d = lane & 31, this lane's 8 key slots of each 16-key group; two chains (the leading products / the 2^-16-class ones)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint4 vf[3];
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) vf[s3] = *reinterpret_cast<const uint4*>(&Vt[buf][s3 * VPL + (lane & 31) * VROW + hh * 16 + g * 32]);
      if (NP > 2) oacb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab3_f(vf[0]), ab3_f(pf[g][2]), oacb, 0, 0, 0);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab3_f(vf[1]), ab3_f(pf[g][0]), oacc, 0, 0, 0);
      oacb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab3_f(vf[2]), ab3_f(pf[g][0]), oacb, 0, 0, 0);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab3_f(vf[0]), ab3_f(pf[g][1]), oacc, 0, 0, 0);
      oacb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab3_f(vf[1]), ab3_f(pf[g][1]), oacb, 0, 0, 0);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab3_f(vf[0]), ab3_f(pf[g][0]), oacc, 0, 0, 0);
    }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }
  float* O = p.o + b * p.o_bs + (long long)qrow * p.ldo + h * DH;
  const float inv = 1.f / l;                                              // l == 0 (fully masked row) -> NaN like the reference
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 w = make_float4((oacc[4 * g] + oacb[4 * g]) * inv, (oacc[4 * g + 1] + oacb[4 * g + 1]) * inv, (oacc[4 * g + 2] + oacb[4 * g + 2]) * inv,
                           (oacc[4 * g + 3] + oacb[4 * g + 3]) * inv);
    if (l == 0.f) w = make_float4(NAN, NAN, NAN, NAN);
    *reinterpret_cast<float4*>(O + 8 * g + 4 * hh) = w;
  }
}

// The same kernel in the "f16x3" arithmetic: every operand as TWO IEEE-half levels (11 + 11 significand bits), three v_mfma_f32_32x32x16_f16 products per multiply
// (a1 b1 + a1 b2 + a2 b1; the dropped terms are 2^-22 relative) -- half the MFMAs and a 3-instruction split per value instead of 5.5.  A half's range needs care:
//   * scores: what matters is the ABSOLUTE error of s = q . k (log2 units); q (pre-scaled) and k are taken as they are -- below 2^-3 a value's second level is a
//     subnormal half with an absolute error <= 2^-25, far below the 2^-22 relative error of the values that dominate a score;
//   * P' = 2^10 exp2(s - m): the 2^10 rides in the exponent argument and cancels in O = sum(P' v) / sum(P');
//   * V is scaled by a power of two sv chosen by the block from its first 64 keys (largest |v| into [2, 4)); a later tile that outgrows the 2^12 headroom raises a
//     flag; the block picks a new sv, stages that tile again from the fp32 values still in registers, and the O accumulators follow the scale of the tile they take in
//     (exact power-of-two rescale).
__device__ __forceinline__ void ah_split8(const float (&v)[8], uint4& h, uint4& l) {
  unsigned hh_[4], ll_[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    hh_[q] = cvt2_f16(v[2 * q], v[2 * q + 1]);
    ll_[q] = cvt2_f16(v[2 * q] - f16lo(hh_[q]), v[2 * q + 1] - f16hi(hh_[q]));
  }
  h = make_uint4(hh_[0], hh_[1], hh_[2], hh_[3]); l = make_uint4(ll_[0], ll_[1], ll_[2], ll_[3]);
}
__device__ __forceinline__ f16x8_t ah_f(uint4 v) { return __builtin_bit_cast(f16x8_t, v); }

template <bool MASK, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void attn_f16_kernel(AP<float> p) {
  constexpr int DH = 32, TK = 64, KROW = DH * 2 + 16, VROW = TK * 2 + 16;
  constexpr int KPL = TK * KROW, VPL = DH * VROW;
  __shared__ __attribute__((aligned(16))) unsigned char Ks[2][2 * KPL];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[2][2 * VPL];
  __shared__ uint8_t Ms[2][TK];
  __shared__ float Red[NW + 1];                                           // wave maxima of |V| (scale decisions), [NW]: the growth flag
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int qrow = blockIdx.x * (32 * NW) + wave * 32 + (lane & 31);
  const int hh = lane >> 5;
  const float* Q = p.q + b * p.q_bs + (long long)qrow * p.ldq + h * DH;
  const float* K = p.k + b * p.k_bs + h * DH;
  const float* V = p.v + b * p.v_bs + h * DH;
  const uint8_t* M = (MASK && p.mask) ? p.mask + (long long)b * p.S : nullptr;

  uint4 qf[2][2];
  {
    const float sc = p.scale * 1.44269504088896340736f;                   // scores in log2 units
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const float4 a0 = *reinterpret_cast<const float4*>(Q + kk * 16 + hh * 8), a1 = *reinterpret_cast<const float4*>(Q + kk * 16 + hh * 8 + 4);
      const float v[8] = {a0.x * sc, a0.y * sc, a0.z * sc, a0.w * sc, a1.x * sc, a1.y * sc, a1.z * sc, a1.w * sc};
      ah_split8(v, qf[kk][0], qf[kk][1]);
    }
  }
  f32x16 oacc, oacb;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = oacb[r] = 0.f;
  float m = -INFINITY, l = 0.f;

  constexpr int GF = NW == 4 ? 8 : 4, GPK = DH / GF;
  const int skey = threadIdx.x / GPK, sg = threadIdx.x % GPK;
  const int kq = skey & 15;
  const int vcol = (skey & ~15) + (kq < 4 ? kq : kq < 8 ? kq + 4 : kq < 12 ? kq - 4 : kq);
  float4 kreg[GF / 4], vreg[GF / 4]; uint8_t mreg = 0;
  float sv = 1.f, osv = 1.f, svbuf[2] = {1.f, 1.f}, omax = 0.f;           // V scale of the coming stores | of the O accumulators | of each staged tile; largest scaled |v| stored
  bool unscaled = false;
  auto load_tile = [&](int key0) {
    const float* kp = K + (long long)(key0 + skey) * p.ldk + sg * GF;
    const float* vp = V + (long long)(key0 + skey) * p.ldv + sg * GF;
#pragma unroll
    for (int i = 0; i < GF / 4; ++i) { kreg[i] = *reinterpret_cast<const float4*>(kp + 4 * i); vreg[i] = *reinterpret_cast<const float4*>(vp + 4 * i); }
    if (MASK && M && threadIdx.x < TK) mreg = M[key0 + threadIdx.x];
  };
  auto store_tile = [&](int buf) {
    float kv[8], vv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { kv[i] = 0.f; vv[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < GF / 4; ++i) {
      kv[4 * i] = kreg[i].x; kv[4 * i + 1] = kreg[i].y; kv[4 * i + 2] = kreg[i].z; kv[4 * i + 3] = kreg[i].w;
      vv[4 * i] = vreg[i].x * sv; vv[4 * i + 1] = vreg[i].y * sv; vv[4 * i + 2] = vreg[i].z * sv; vv[4 * i + 3] = vreg[i].w * sv;
      omax = fmaxf(omax, fmaxf(fmaxf(fabsf(vv[4 * i]), fabsf(vv[4 * i + 1])), fmaxf(fabsf(vv[4 * i + 2]), fabsf(vv[4 * i + 3]))));
    }
    uint4 kl[2], vl[2];
    ah_split8(kv, kl[0], kl[1]);
    ah_split8(vv, vl[0], vl[1]);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      unsigned char* kd = &Ks[buf][s2 * KPL + skey * KROW + sg * (2 * GF)];
      if (GF == 8) *reinterpret_cast<uint4*>(kd) = kl[s2]; else *reinterpret_cast<uint2*>(kd) = make_uint2(kl[s2].x, kl[s2].y);
      const uint32_t w4[4] = {vl[s2].x, vl[s2].y, vl[s2].z, vl[s2].w};
#pragma unroll
      for (int e = 0; e < GF; ++e)
        *reinterpret_cast<uint16_t*>(&Vt[buf][s2 * VPL + (sg * GF + e) * VROW + vcol * 2]) = (uint16_t)(e & 1 ? w4[e >> 1] >> 16 : w4[e >> 1] & 0xffffu);
    }
    if (MASK && threadIdx.x < TK) Ms[buf][threadIdx.x] = M ? mreg : 0;
    svbuf[buf] = sv;
  };
  // the block's largest |value|: `mine` reduced over the block (two barriers inside)
  auto block_max = [&](float mine) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine = fmaxf(mine, __shfl_xor(mine, o, 64));
    __syncthreads();
    if (lane == 0) Red[wave] = mine;
    __syncthreads();
    float bm = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) bm = fmaxf(bm, Red[w]);
    return bm;
  };

  const int ntiles = p.S / TK;
  load_tile(0);
  {
    float tm = 0.f;
#pragma unroll
    for (int i = 0; i < GF / 4; ++i) tm = fmaxf(tm, fmaxf(fmaxf(fabsf(vreg[i].x), fabsf(vreg[i].y)), fmaxf(fabsf(vreg[i].z), fabsf(vreg[i].w))));
    const float bm = block_max(tm);
    if (threadIdx.x == 0) Red[NW] = 0.f;
    unscaled = !(bm > 0.f && bm < 3.0e38f);
    sv = unscaled ? 1.f : __builtin_amdgcn_ldexpf(1.f, 2 - __builtin_amdgcn_frexp_expf(bm));
    osv = sv;
  }
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile((t + 1) * TK);
    f32x16 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 kf[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) kf[u][s2] = *reinterpret_cast<const uint4*>(&Ks[buf][s2 * KPL + (u * 32 + (lane & 31)) * KROW + hh * 16 + kk * 32]);
      constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};               // (K level, Q level): k1q2 k2q1 k1q1
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_f(kf[u][PA[pr]]), ah_f(qf[kk][PB[pr]]), s[u], 0, 0, 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (MASK) {
          const int key = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (Ms[buf][key]) s[u][r] = -INFINITY;
        }
        tmax = fmaxf(tmax, s[u][r]);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax);
    const bool dead = MASK && (m_new == -INFINITY);
    const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m - m_new);
    const float mo = 10.f - m_new;                                        // P' = 2^10 P
    float psum = 0.f;
    uint4 pf[4][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float e[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { e[r] = dead ? 0.f : __builtin_amdgcn_exp2f(s[u][r] + mo); psum += e[r]; }
      const float e0[8] = {e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]}, e1[8] = {e[8], e[9], e[10], e[11], e[12], e[13], e[14], e[15]};
      ah_split8(e0, pf[2 * u][0], pf[2 * u][1]);
      ah_split8(e1, pf[2 * u + 1][0], pf[2 * u + 1][1]);
    }
    psum += __shfl_xor(psum, 32, 64);
    l = l * alpha + psum; m = m_new;
    float resc = alpha;                                                   // softmax rescale x (rarely) the step to this tile's V scale
    if (svbuf[buf] != osv) { resc *= svbuf[buf] / osv; osv = svbuf[buf]; }
    if (__builtin_amdgcn_ballot_w64(resc != 1.f) != 0ull) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { oacc[r] *= resc; oacb[r] *= resc; }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint4 vf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) vf[s2] = *reinterpret_cast<const uint4*>(&Vt[buf][s2 * VPL + (lane & 31) * VROW + hh * 16 + g * 32]);
      f32x16& c0 = (g & 1) ? oacc : oacb; f32x16& c1 = (g & 1) ? oacb : oacc;
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_f(vf[0]), ah_f(pf[g][1]), c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_f(vf[1]), ah_f(pf[g][0]), c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_f(vf[0]), ah_f(pf[g][0]), c0, 0, 0, 0);
    }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    if (omax > 4096.f || (unscaled && omax > 0.f)) Red[NW] = 1.f;
    __syncthreads();
    if (Red[NW] != 0.f) {                                                 // rare: a tile outgrew the headroom (or the first tiles were all zero): new scale for the tiles after it
      const float bm = block_max(omax) / sv;                             // (true magnitude; finite: omax is taken before the conversion to half)
      if (bm > 0.f && bm < 3.0e38f) { sv = __builtin_amdgcn_ldexpf(1.f, 2 - __builtin_amdgcn_frexp_expf(bm)); unscaled = false; }
      omax = 0.f;
      if (t + 1 < ntiles) store_tile(buf ^ 1);                           // the tile just staged may hold overflowed halves: stage it again (its fp32 values are still in registers)
      if (threadIdx.x == 0) Red[NW] = 0.f;
      __syncthreads();
    }
  }
  float* O = p.o + b * p.o_bs + (long long)qrow * p.ldo + h * DH;
  const float inv = 1.f / (l * osv);                                      // l == 0 (fully masked row) -> NaN like the reference
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 w = make_float4((oacc[4 * g] + oacb[4 * g]) * inv, (oacc[4 * g + 1] + oacb[4 * g + 1]) * inv, (oacc[4 * g + 2] + oacb[4 * g + 2]) * inv,
                           (oacc[4 * g + 3] + oacb[4 * g + 3]) * inv);
    if (l == 0.f) w = make_float4(NAN, NAN, NAN, NAN);
    *reinterpret_cast<float4*>(O + 8 * g + 4 * hh) = w;
  }
}

// The fused AttnBlock core of the fp32 configuration (ONE head of d = 256, V given transposed: what attnblock32_kernel computes on the fp32 MFMA at 0.98 of that
// pipe) in the f16x3 arithmetic of attn_f16_kernel: q (pre-scaled), k as they are, P' = 2^10 P, V^T under the block's power-of-two scale (re-staged / followed by the
// accumulators when a tile outgrows it).  One wave per SIMD: a wave owns 32 queries, the two-level Q^T fragments (128 registers) and all eight O^T tiles (128); per
// 32-key tile 96 half MFMAs (48 for S^T on two chains, 48 for O^T, two d tiles alternating) against the fp32 kernel's 256 MFMAs of twice the length.
__global__ __launch_bounds__(256, 1) void attnblock_f16_kernel(AP<float> p) {
  constexpr int DH = 256, TK = 32, NKK = DH / 16, NDT = DH / 32;
  constexpr int KROW = DH * 2 + 16, VROW = TK * 2 + 16;                   // LDS row bytes (padded): 528 | 80
  constexpr int KPL = TK * KROW, VPL = DH * VROW;                         // one level plane: 16,896 | 20,480 B
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;                                               // [2 buffers][2 levels][KPL]
  unsigned char* Vt = smem + 4 * KPL;                                     // [2 buffers][2 levels][VPL]
  float* Red = reinterpret_cast<float*>(smem + 4 * KPL + 4 * VPL);        // [4 wave maxima][flag]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int qrow = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int hh = lane >> 5;
  const float* Q = p.q + b * p.q_bs + (long long)qrow * p.ldq;
  const float* K = p.k + b * p.k_bs;
  const float* V = p.v + b * p.v_bs;                                      // V^T: [d][key]

  uint4 qf[NKK][2];
  {
    const float sc = p.scale * 1.44269504088896340736f;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const float4 a0 = *reinterpret_cast<const float4*>(Q + kk * 16 + hh * 8), a1 = *reinterpret_cast<const float4*>(Q + kk * 16 + hh * 8 + 4);
      const float v[8] = {a0.x * sc, a0.y * sc, a0.z * sc, a0.w * sc, a1.x * sc, a1.y * sc, a1.z * sc, a1.w * sc};
      ah_split8(v, qf[kk][0], qf[kk][1]);
    }
  }
  f32x16 oacc[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  float sv = 1.f, osv = 1.f, svbuf[2] = {1.f, 1.f}, omax = 0.f;
  bool unscaled = false;

  // staging: K tile 32 keys x 32 chunks of 8 floats, V^T tile 256 rows x 4 chunks of 8 keys: four chunks of each per thread
  float4 kreg[4][2], vreg[4][2];
  auto load_tile = [&](int key0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = tid + 256 * j;
      const float* kp = K + (long long)(key0 + (g >> 5)) * p.ldk + (g & 31) * 8;
      const float* vp = V + (long long)(g >> 2) * p.ldv + key0 + (g & 3) * 8;
      kreg[j][0] = *reinterpret_cast<const float4*>(kp); kreg[j][1] = *reinterpret_cast<const float4*>(kp + 4);
      vreg[j][0] = *reinterpret_cast<const float4*>(vp); vreg[j][1] = *reinterpret_cast<const float4*>(vp + 4);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = tid + 256 * j;
      const float kv[8] = {kreg[j][0].x, kreg[j][0].y, kreg[j][0].z, kreg[j][0].w, kreg[j][1].x, kreg[j][1].y, kreg[j][1].z, kreg[j][1].w};
      float vv[8] = {vreg[j][0].x * sv, vreg[j][0].y * sv, vreg[j][0].z * sv, vreg[j][0].w * sv, vreg[j][1].x * sv, vreg[j][1].y * sv, vreg[j][1].z * sv, vreg[j][1].w * sv};
#pragma unroll
      for (int e = 0; e < 8; e += 2) omax = fmaxf(omax, fmaxf(fabsf(vv[e]), fabsf(vv[e + 1])));
      uint4 kl[2], vl[2];
      ah_split8(kv, kl[0], kl[1]);
      ah_split8(vv, vl[0], vl[1]);
      // V^T row d, keys 8 kc .. 8 kc + 7 of the tile: per 16 keys the columns run {0-3, 8-11, 4-7, 12-15} (the S^T accumulator's key order)
      const int d = g >> 2, kc = g & 3;
      unsigned char* vd = Vt + (buf * 2) * VPL + d * VROW + (kc >> 1) * 32 + (kc & 1) * 8;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        *reinterpret_cast<uint4*>(Ks + (buf * 2 + s2) * KPL + (g >> 5) * KROW + (g & 31) * 16) = kl[s2];
        *reinterpret_cast<uint2*>(vd + s2 * VPL) = make_uint2(vl[s2].x, vl[s2].y);
        *reinterpret_cast<uint2*>(vd + s2 * VPL + 16) = make_uint2(vl[s2].z, vl[s2].w);
      }
    }
    svbuf[buf] = sv;
  };
  auto block_max = [&](float mine) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine = fmaxf(mine, __shfl_xor(mine, o, 64));
    __syncthreads();
    if (lane == 0) Red[wave] = mine;
    __syncthreads();
    return fmaxf(fmaxf(Red[0], Red[1]), fmaxf(Red[2], Red[3]));
  };

  const int ntiles = p.S / TK;
  load_tile(0);
  {
    float tm = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) tm = fmaxf(tm, fmaxf(fmaxf(fabsf(vreg[j][i].x), fabsf(vreg[j][i].y)), fmaxf(fabsf(vreg[j][i].z), fabsf(vreg[j][i].w))));
    const float bm = block_max(tm);
    if (tid == 0) Red[4] = 0.f;
    unscaled = !(bm > 0.f && bm < 3.0e38f);
    sv = unscaled ? 1.f : __builtin_amdgcn_ldexpf(1.f, 2 - __builtin_amdgcn_frexp_expf(bm));
    osv = sv;
  }
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile((t + 1) * TK);
    // S^T = K Q^T on two chains (consecutive MFMAs never share an accumulator)
    f32x16 sa, sb;
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = sb[r] = 0.f;
    const unsigned char* kb = Ks + (buf * 2) * KPL + (lane & 31) * KROW + hh * 16;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const uint4 k1 = *reinterpret_cast<const uint4*>(kb + kk * 32), k2 = *reinterpret_cast<const uint4*>(kb + KPL + kk * 32);
      f32x16& c0 = (kk & 1) ? sa : sb; f32x16& c1 = (kk & 1) ? sb : sa;
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_f(k1), ah_f(qf[kk][1]), c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_f(k2), ah_f(qf[kk][0]), c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_f(k1), ah_f(qf[kk][0]), c0, 0, 0, 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] += sb[r]; tmax = fmaxf(tmax, sa[r]); }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    const float mo = 10.f - m_new;                                        // P' = 2^10 P
    float psum = 0.f, e[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { e[r] = __builtin_amdgcn_exp2f(sa[r] + mo); psum += e[r]; }
    uint4 pf[2][2];
    {
      const float e0[8] = {e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]}, e1[8] = {e[8], e[9], e[10], e[11], e[12], e[13], e[14], e[15]};
      ah_split8(e0, pf[0][0], pf[0][1]);
      ah_split8(e1, pf[1][0], pf[1][1]);
    }
    psum += __shfl_xor(psum, 32, 64);
    l = l * alpha + psum; m = m_new;
    float resc = alpha;
    if (svbuf[buf] != osv) { resc *= svbuf[buf] / osv; osv = svbuf[buf]; }
    if (__builtin_amdgcn_ballot_w64(resc != 1.f) != 0ull) {
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= resc;
    }
    // O^T += V^T P^T: two d tiles at a time, their MFMAs alternating
    const unsigned char* vb = Vt + (buf * 2) * VPL + (lane & 31) * VROW + hh * 16;
#pragma unroll
    for (int dp = 0; dp < NDT / 2; ++dp)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        uint4 vf[2][2];
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) vf[q2][s2] = *reinterpret_cast<const uint4*>(vb + s2 * VPL + (2 * dp + q2) * 32 * VROW + g * 32);
        constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};               // (V level, P level): v1p2 v2p1 v1p1
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2)
            oacc[2 * dp + q2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_f(vf[q2][PA[pr]]), ah_f(pf[g][PB[pr]]), oacc[2 * dp + q2], 0, 0, 0);
      }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    if (omax > 4096.f || (unscaled && omax > 0.f)) Red[4] = 1.f;
    __syncthreads();
    if (Red[4] != 0.f) {
      const float bm = block_max(omax) / sv;
      if (bm > 0.f && bm < 3.0e38f) { sv = __builtin_amdgcn_ldexpf(1.f, 2 - __builtin_amdgcn_frexp_expf(bm)); unscaled = false; }
      omax = 0.f;
      if (t + 1 < ntiles) store_tile(buf ^ 1);
      if (tid == 0) Red[4] = 0.f;
      __syncthreads();
    }
  }
  float* O = p.o + b * p.o_bs + (long long)qrow * p.ldo;
  const float inv = 1.f / (l * osv);
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(O + d * 32 + 8 * g + 4 * hh) = make_float4(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
}

// The vqgan AttnBlock core (archs/vqgan_arch.py:229-253: ONE head of d = C = 256 over the 32 x 32 tokens) on bf16 storage, fused:
// softmax(q k^T / sqrt(C)) v as one kernel, the [B, N, N] score tensor never exists (the three-launch form wrote and re-read 1.26 GB of
// fp32 scores per call at B = 300).  Same swapped-product scheme as attn_mfma16_kernel with the d axis 8 tiles wide: Q^T fragments
// (16 k-steps) and the eight O^T accumulator tiles live in registers (64 + 128 VGPRs), K tiles [32 keys][256] and V^T tiles
// [256][32 keys] stream global -> LDS by LDS-DMA (no staging registers: Q^T + O^T take the register file; rows unpadded, 16-B chunks
// XOR-swizzled at the source address for conflict-free fragment reads) through a double-buffered stage.  V arrives TRANSPOSED ([C][N],
// what the value projection GEMM writes with bias_per_row); the S^T accumulator's key order {0-3, 8-11} + 4 (l >> 5) is met by reading
// the V^T fragment as two 8-byte pieces of the row.  The O
// rescale (128 multiplies per lane) is skipped while no lane's running maximum moved.  fp32 statistics, P rounded to bf16 for the PV
// product, fp32 accumulate -- the arithmetic of the three-launch form (its PV GEMM rounded the fp32 probabilities while staging).
typedef __attribute__((address_space(3))) void ab_lds_void;
__device__ __forceinline__ void ab_glds16(const void* gsrc, unsigned lds_dst) {     // LDS-DMA: 64 lanes x 16 B -> LDS [lds_dst, +1 KB); M0 saved / restored
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int DH>
__global__ __launch_bounds__(512, 1) void attnblock16_kernel(AP<bf16_t> p) {
  static_assert(DH == 256, "the AttnBlock width of this path");
  // 8 waves: wave pair (2 g, 2 g + 1) shares the 32 queries of group g and splits the d axis (4 of the 8 O^T tiles each); both compute
  // S^T (16 of a wave's 24 MFMAs per tile: the matrix pipe is far from the bound here) -- with all eight tiles in one wave the kernel
  // spilled, and every scratch reload's vmcnt(0) also waited for the tile prefetch in flight
  constexpr int TK = 32, KROW = DH * 2, VROW = TK * 2, DT = DH / 64, KK = DH / 16;
  constexpr int KTILE = TK * KROW, VTILE = DH * VROW;                     // 16 KB + 16 KB per stage, rows unpadded (the DMA writes 1 KB runs)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NST = 4;                                                  // stages: tiles t+1 .. t+3 in flight while tile t is multiplied (one block per CU:
                                                                          // nothing else covers the L2 / HBM round trip of a tile, ~3 tile times)
  unsigned char* Ks = smem;                                               // [NST][TK][KROW]: 16-B chunk c of key row r at position c ^ (r & 15)
  unsigned char* Vt = smem + NST * KTILE;                                 // [NST][DH][VROW]: chunk c of row d at position c ^ ((d >> 2) & 3)
  const unsigned lds0 = (unsigned)(uintptr_t)((ab_lds_void*)smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const int qrow = blockIdx.x * 128 + (wave >> 1) * 32 + (lane & 31);
  const int hh = lane >> 5, dbase = (wave & 1) * (DH / 2);
  const bf16_t* Q = p.q + b * p.q_bs + (long long)qrow * p.ldq;
  const bf16_t* K = p.k + b * p.k_bs;
  const bf16_t* VT = p.v + b * p.v_bs;                                    // [DH][ldv >= S]
  const float c = p.scale * 1.44269504088896340736f;

  bf16x8 qf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) qf[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Q + kk * 16 + hh * 8));
  f32x16 oacc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m = -INFINITY, l = 0.f;

  // staging by LDS-DMA (no registers in flight): per tile 16 + 16 instructions of 1 KB, four of each per wave
  // two K and two V^T instructions per wave: slot (wave * 2 + i) * 64 + lane: K row = 4 wave + 2 i + (lane >> 5), position lane & 31;
  // V^T row = 32 wave + 16 i + (lane >> 2), position lane & 3
  auto issue = [&](int key0, int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kr = 4 * wave + 2 * i + hh;
      ab_glds16(K + (long long)(key0 + kr) * p.ldk + (((lane & 31) ^ (kr & 15)) << 3), lds0 + (unsigned)(buf * KTILE + (wave * 2 + i) * 1024));
      const int vr = 32 * wave + 16 * i + (lane >> 2);
      ab_glds16(VT + (long long)vr * p.ldv + key0 + (((lane & 3) ^ ((vr >> 2) & 3)) << 3), lds0 + (unsigned)(NST * KTILE + buf * VTILE + (wave * 2 + i) * 1024));
    }
  };
  const int krow = lane & 31, ksw = krow & 15;
  const int vsw = (krow >> 2) & 3;                                         // rows 32 d + (l & 31): (row >> 2) & 3 does not depend on the d tile

  const int ntiles = p.S / TK;
  for (int t = 0; t < NST - 1 && t < ntiles; ++t) issue(t * TK, t);
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & (NST - 1);
    // four DMA instructions per wave and tile, nothing else in flight: tile t has landed once at most the younger tiles' remain
    if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                       // everyone's part of tile t is there; everyone is past tile t - 1
    if (t + NST - 1 < ntiles) issue((t + NST - 1) * TK, (t + NST - 1) & (NST - 1));
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const unsigned char* kp = Ks + buf * KTILE + krow * KROW;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kp + (((2 * kk + hh) ^ ksw) << 4))), qf[kk], s, 0, 0, 0);
    float tmax = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax * c);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float psum = 0.f, e[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { e[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c, -m_new)); psum += e[r]; }
    // accumulator regs 0-7 = keys {0-3, 8-11} + 4 hh of the first 16, regs 8-15 = the same of the second 16
    bf16x8 pf[2];
    pf[0] = __builtin_bit_cast(bf16x8, make_uint4(pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])));
    pf[1] = __builtin_bit_cast(bf16x8, make_uint4(pack2(e[8], e[9]), pack2(e[10], e[11]), pack2(e[12], e[13]), pack2(e[14], e[15])));
    psum += __shfl_xor(psum, 32, 64);
    l = l * alpha + psum;
    if (__any(m_new != m)) {                                               // wave-uniform: some lane's maximum moved
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
    }
    m = m_new;
    // O^T += V^T P^T: A = V^T row d, the SAME 8 keys this lane's P fragment holds: two 8-byte pieces of the row (keys 16 g + 4 hh + 0..3 and + 8)
    const unsigned char* vp = Vt + buf * VTILE + (dbase + krow) * VROW + hh * 8;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint2 lo = *reinterpret_cast<const uint2*>(vp + d * 32 * VROW + (((2 * g) ^ vsw) << 4));
        const uint2 hi = *reinterpret_cast<const uint2*>(vp + d * 32 * VROW + (((2 * g + 1) ^ vsw) << 4));
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y)), pf[g], oacc[d], 0, 0, 0);
      }
  }
  bf16_t* O = p.o + b * p.o_bs + (long long)qrow * p.ldo;
  const float inv = 1.f / l;
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      St<bf16_t>::st4(O + dbase + 32 * d + 8 * g + 4 * hh, make_float4(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv));
}

// The same AttnBlock core in the fp32 configuration (configs[1], the reference's own arithmetic): ONE kernel on v_mfma_f32_32x32x2_f32,
// no [B, N, N] score tensor, no softmax_rows pass.  The contraction is matrix-bound here (322 GFLOP per call at B = 300 against ~1 GB of
// traffic), so the shape is the opposite of the bf16 kernel's: ONE wave per SIMD (4 waves, 128 queries per block, one block per CU) that owns 32
// queries AND all eight O^T tiles -- Q^T fragments (128 VGPRs, pre-scaled by scale * log2 e) + O^T accumulators (128 VGPRs) at the 512-register
// budget -- and issues 256 MFMAs per 32-key tile (128 for S^T = K Q^T, 128 for O^T += V^T P^T) against 64 ds_read_b128: nothing but the matrix
// pipe is near a limit.  K tiles [32 keys][256] and V^T tiles [256][32 keys] stream global -> LDS by LDS-DMA into a double-buffered stage
// (64 KB each), 16-B chunks XOR-swizzled at the source address (K: chunk c of key row r at position c ^ (r & 15); V^T: chunk c of row d at
// position c ^ (d & 7)) so that the fragment reads of 8 consecutive lanes hit 8 distinct 16-B bank slots.  One lane's ds_read_b128 feeds
// FOUR MFMAs through the free k pairing (A and B only have to agree): S^T step kk multiplies d = 8 kk + 4 hh + t, PV group g multiplies keys
// 8 g + 4 hh + t -- exactly the keys the S^T accumulator's registers 4 g + t hold, so P feeds the second product straight from registers.
// Softmax statistics fp32, lane-local + one wavefront shuffle; the O rescale is skipped while no lane's running maximum moved.
template <int DH>
__global__ __launch_bounds__(256, 1) void attnblock32_kernel(AP<float> p) {
  static_assert(DH == 256, "the AttnBlock width of this path");
  constexpr int TK = 32, KROW = DH * 4, VROW = TK * 4, DT = DH / 32, KK = DH / 8;
  constexpr int KTILE = TK * KROW, VTILE = DH * VROW;                     // 32 KB + 32 KB per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;                                               // [2][TK][KROW]
  unsigned char* Vt = smem + 2 * KTILE;                                   // [2][DH][VROW]
  const unsigned lds0 = (unsigned)(uintptr_t)((ab_lds_void*)smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const int qrow = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int hh = lane >> 5;
  const float* Q = p.q + b * p.q_bs + (long long)qrow * p.ldq;
  const float* K = p.k + b * p.k_bs;
  const float* VT = p.v + b * p.v_bs;                                     // [DH][ldv >= S]
  const float sc = p.scale * 1.44269504088896340736f;                     // scores in log2 units: softmax via v_exp_f32

  float4 qf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const float4 t = *reinterpret_cast<const float4*>(Q + kk * 8 + hh * 4);
    qf[kk] = make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc);
  }
  f32x16 oacc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m = -INFINITY, l = 0.f;

  // per tile 32 + 32 DMA instructions of 1 KB (8 + 8 per wave): a K instruction is one key row (64 chunks), a V^T instruction is 8 d rows x 8 chunks
  auto issue = [&](int key0, int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kr = wave * 8 + i;
      ab_glds16(K + (long long)(key0 + kr) * p.ldk + ((lane ^ (kr & 15)) << 2), lds0 + (unsigned)(buf * KTILE + kr * KROW));
      const int vr = (wave * 8 + i) * 8 + (lane >> 3);
      ab_glds16(VT + (long long)vr * p.ldv + key0 + (((lane & 7) ^ (vr & 7)) << 2), lds0 + (unsigned)(2 * KTILE + buf * VTILE + (wave * 8 + i) * 1024));
    }
  };
  const int krow = lane & 31, ksw = krow & 15, vsw = krow & 7;            // V^T rows 32 dt + krow: (row & 7) does not depend on the d tile
  const int ntiles = p.S / TK;
  issue(0, 0);
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's 16 pieces of tile t (requested a whole tile ago)
    __syncthreads();                                                       // everyone's pieces are there; everyone is past tile t - 1
    if (t + 1 < ntiles) issue((t + 1) * TK, buf ^ 1);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const unsigned char* kp = Ks + buf * KTILE + krow * KROW;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const float4 kf = *reinterpret_cast<const float4*>(kp + (((2 * kk + hh) ^ ksw) << 4));
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[kk].x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[kk].y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[kk].z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[kk].w, s, 0, 0, 0);
    }
    float tmax = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_new); psum += s[r]; }
    psum += __shfl_xor(psum, 32, 64);
    l = l * alpha + psum;
    if (__any(m_new != m)) {                                               // wave-uniform: some lane's maximum moved
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
    }
    m = m_new;
    // O^T += V^T P^T: A = V^T row d, keys 8 g + 4 hh + (0..3) = one 16-B chunk; B = the S^T registers 4 g + (0..3) of the same keys
    const unsigned char* vp = Vt + buf * VTILE + krow * VROW;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 vf = *reinterpret_cast<const float4*>(vp + d * 32 * VROW + (((2 * g + hh) ^ vsw) << 4));
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, s[4 * g + 0], oacc[d], 0, 0, 0);
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, s[4 * g + 1], oacc[d], 0, 0, 0);
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, s[4 * g + 2], oacc[d], 0, 0, 0);
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, s[4 * g + 3], oacc[d], 0, 0, 0);
      }
  }
  float* O = p.o + b * p.o_bs + (long long)qrow * p.ldo;
  const float inv = 1.f / l;
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(O + 32 * d + 8 * g + 4 * hh) = make_float4(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
}

// d_head = 4: one query per lane, 64 queries per block; the block's 4 waves split the S keys
// (wave w owns keys [w*S/4, (w+1)*S/4)), K/V broadcast from LDS, 8 keys per online-softmax step,
// and the four partial (m, l, acc) states are merged through LDS at the end.
template <typename T>
__global__ __launch_bounds__(256) void attn_valu4_kernel(AP<T> p) {
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  float4* Ks = smem4;                       // [S]
  float4* Vs = smem4 + p.S;                 // [S]
  uint8_t* Ms = reinterpret_cast<uint8_t*>(smem4 + 2 * p.S);   // [S]
  float* part = reinterpret_cast<float*>(Ms + p.S);             // [4][64][6]
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qrow = blockIdx.x * 64 + lane;
  float4 q = St<T>::ld4(p.q + b * p.q_bs + (long long)qrow * p.ldq + h * 4);
  { const float sc = p.scale * 1.44269504088896340736f; q = make_float4(q.x * sc, q.y * sc, q.z * sc, q.w * sc); }
  const T* K = p.k + b * p.k_bs + h * 4;
  const T* V = p.v + b * p.v_bs + h * 4;
  const uint8_t* M = p.mask ? p.mask + (long long)b * p.S : nullptr;
  for (int i = threadIdx.x; i < p.S; i += 256) {
    Ks[i] = St<T>::ld4(K + (long long)i * p.ldk);
    Vs[i] = St<T>::ld4(V + (long long)i * p.ldv);
    Ms[i] = M ? M[i] : 0;
  }
  __syncthreads();
  float m = -INFINITY, l = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int per = p.S >> 2, g0 = wave * per;
  for (int g = g0; g < g0 + per; g += 8) {
    float s[8]; float tmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 kk = Ks[g + i];
      float t = q.x * kk.x + q.y * kk.y + q.z * kk.z + q.w * kk.w;
      if (Ms[g + i]) t = -INFINITY;
      s[i] = t; tmax = fmaxf(tmax, t);
    }
    const float m_new = fmaxf(m, tmax);
    if (m_new == -INFINITY) continue;
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float4 a2 = make_float4(acc.x * alpha, acc.y * alpha, acc.z * alpha, acc.w * alpha);
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float pe = __builtin_amdgcn_exp2f(s[i] - m_new);
      const float4 vv = Vs[g + i];
      ps += pe; a2.x += pe * vv.x; a2.y += pe * vv.y; a2.z += pe * vv.z; a2.w += pe * vv.w;
    }
    acc = a2; l = l * alpha + ps; m = m_new;
  }
  float* pp = part + (wave * 64 + lane) * 6;
  pp[0] = m; pp[1] = l; pp[2] = acc.x; pp[3] = acc.y; pp[4] = acc.z; pp[5] = acc.w;
  __syncthreads();
  if (wave == 0) {
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, part[(w * 64 + lane) * 6]);
    float ll = 0.f; float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* q6 = part + (w * 64 + lane) * 6;
      const float f = (q6[0] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(q6[0] - mm);
      ll += q6[1] * f; a.x += q6[2] * f; a.y += q6[3] * f; a.z += q6[4] * f; a.w += q6[5] * f;
    }
    // ll == 0 (every key masked) -> 0/0 = NaN like the reference
    St<T>::st4(p.o + b * p.o_bs + (long long)qrow * p.ldo + h * 4, make_float4(a.x / ll, a.y / ll, a.z / ll, a.w / ll));
  }
}


// d_head = 4 on the matrix pipe: v_mfma_f32_4x4x1_16B_f32 computes 16 independent 4x4 (K = 1) outer products per
// instruction -- block = lane / 4.  With the products swapped the way the d_head = 32 kernel does it,
//     S^T block [4 keys][4 queries]  = sum_d K[key][d] * Q[query][d]    (A = K column d, B = this lane's q.d)
//     O^T block [4 d][4 queries]    += V^T[d][key] * P^T[key][query]    (A = V row of the key, B = this lane's p)
// lane l owns query l in BOTH results (block b covers queries 4b..4b+3 = its own 4 lanes; every block is fed the same
// 4 keys), so scores, softmax statistics and the output accumulators are lane-local: no shuffles.  The 8 FMAs per
// (query, key) leave the VALU, which is left with the softmax itself (max, sub, v_exp_f32, sum, rescale) -- the
// VALU kernel spent ~70 issue cycles per key and wave, this one ~36 with the matrix pipe (16) running beside it.
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <typename T, bool MASK>
__global__ __launch_bounds__(256, 2) void attn_mfma4_kernel(AP<T> p) {
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  float4* Ks = smem4;                                            // [S] key-major: K[key][0..3]
  float* Vt = reinterpret_cast<float*>(smem4 + p.S);             // [4][S] d-major
  float* Mf = reinterpret_cast<float*>(smem4 + 2 * p.S);         // [S] (MASK) 0 / -inf per key: the score accumulators START from it
  float* part = Mf + (MASK ? p.S : 0);                           // [4][64][6]
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l4 = lane & 3;
  const int qrow = blockIdx.x * 64 + lane;
  float4 q = St<T>::ld4(p.q + b * p.q_bs + (long long)qrow * p.ldq + h * 4);
  { const float sc = p.scale * 1.44269504088896340736f; q = make_float4(q.x * sc, q.y * sc, q.z * sc, q.w * sc); }
  const T* K = p.k + b * p.k_bs + h * 4;
  const T* V = p.v + b * p.v_bs + h * 4;
  const uint8_t* M = p.mask ? p.mask + (long long)b * p.S : nullptr;
  for (int i = threadIdx.x; i < p.S; i += 256) {
    Ks[i] = St<T>::ld4(K + (long long)i * p.ldk);
    const float4 v = St<T>::ld4(V + (long long)i * p.ldv);
    Vt[i] = v.x; Vt[p.S + i] = v.y; Vt[2 * p.S + i] = v.z; Vt[3 * p.S + i] = v.w;
    if (MASK) Mf[i] = M[i] ? -INFINITY : 0.f;
  }
  __syncthreads();
  float m = -INFINITY, l = 0.f;
  f32x4m acc[4];                                                 // 4 independent chains (key e of every 4-key block)
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = f32x4m{0.f, 0.f, 0.f, 0.f};
  const int per = p.S >> 2, g0 = wave * per;
  const float* vrow = Vt + l4 * p.S;
  for (int g = g0; g < g0 + per; g += 16) {
    f32x4m s[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 kf = Ks[g + 4 * c + l4];                      // A: K[key g+4c+(lane&3)][0..3]
      f32x4m c0 = f32x4m{0.f, 0.f, 0.f, 0.f};
      if (MASK) { const float4 mf = *reinterpret_cast<const float4*>(Mf + g + 4 * c); c0 = f32x4m{mf.x, mf.y, mf.z, mf.w}; }
      s[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(kf.x, q.x, c0, 0, 0, 0);
      s[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(kf.y, q.y, s[c], 0, 0, 0);
      s[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(kf.z, q.z, s[c], 0, 0, 0);
      s[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(kf.w, q.w, s[c], 0, 0, 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int i = 0; i < 4; ++i) tmax = fmaxf(tmax, s[c][i]);
    }
    const float m_new = fmaxf(m, tmax);
    if (MASK && m_new == -INFINITY) continue;                    // every key so far masked (per lane: keep the state)
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float ps = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) { s[c][i] = __builtin_amdgcn_exp2f(s[c][i] - m_new); ps += s[c][i]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] *= alpha;
    l = l * alpha + ps; m = m_new;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 vf = *reinterpret_cast<const float4*>(vrow + g + 4 * c);       // A: V[key g+4c+e][d = lane&3], e = 0..3
      acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(vf.x, s[c][0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(vf.y, s[c][1], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(vf.z, s[c][2], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(vf.w, s[c][3], acc[3], 0, 0, 0);
    }
  }
  const f32x4m a4 = acc[0] + acc[1] + acc[2] + acc[3];
  float* pp = part + (wave * 64 + lane) * 6;
  pp[0] = m; pp[1] = l; pp[2] = a4[0]; pp[3] = a4[1]; pp[4] = a4[2]; pp[5] = a4[3];
  __syncthreads();
  if (wave == 0) {
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, part[(w * 64 + lane) * 6]);
    float ll = 0.f; float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* q6 = part + (w * 64 + lane) * 6;
      const float f = (q6[0] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(q6[0] - mm);
      ll += q6[1] * f; a.x += q6[2] * f; a.y += q6[3] * f; a.z += q6[4] * f; a.w += q6[5] * f;
    }
    // ll == 0 (every key masked) -> 0/0 = NaN like the reference
    St<T>::st4(p.o + b * p.o_bs + (long long)qrow * p.ldo + h * 4, make_float4(a.x / ll, a.y / ll, a.z / ll, a.w / ll));
  }
}

// d_head = 4 on bf16 storage (configs[2]): the same swapped-product scheme on v_mfma_f32_4x4x4_16B_bf16 -- K = 4 is the whole head
// dimension for S^T = K Q^T and four keys for O^T += V^T P^T, so a 4-key group costs TWO matrix instructions instead of eight
// (4x4x1 fp32: one per d and one per key).  The operands are the stored bf16 values themselves (a K row of this head is the 8-byte A
// fragment, the lane's q row the B fragment: no conversion, no pre-scaling -- the softmax scale enters as the fma in front of v_exp_f32);
// P is rounded to bf16 for the second product like in the d_head = 32 bf16 kernel, the row sum keeps the unrounded values.  32 keys per
// step (one max / rescale per 32 keys).  Scores, statistics and accumulators stay lane-local fp32.
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <bool MASK>
__global__ __launch_bounds__(256, 2) void attn_mfma4_bf16_kernel(AP<bf16_t> p) {
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  uint2* Ks = reinterpret_cast<uint2*>(smem4);                   // [S] key-major: K[key][0..3] (4 x bf16)
  bf16_t* Vt = reinterpret_cast<bf16_t*>(Ks + p.S);              // [4][S] d-major
  float* Mf = reinterpret_cast<float*>(Vt + 4 * p.S);            // [S] (MASK) 0 / -inf per key
  float* part = Mf + (MASK ? p.S : 0);                           // [4][64][6]
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l4 = lane & 3;
  const int qrow = blockIdx.x * 64 + lane;
  const s16x4 qf = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(p.q + b * p.q_bs + (long long)qrow * p.ldq + h * 4));
  const float sc = p.scale * 1.44269504088896340736f;
  const bf16_t* K = p.k + b * p.k_bs + h * 4;
  const bf16_t* V = p.v + b * p.v_bs + h * 4;
  const uint8_t* M = p.mask ? p.mask + (long long)b * p.S : nullptr;
  for (int i = threadIdx.x; i < p.S; i += 256) {
    Ks[i] = *reinterpret_cast<const uint2*>(K + (long long)i * p.ldk);
    const uint2 v = *reinterpret_cast<const uint2*>(V + (long long)i * p.ldv);
    Vt[i] = (bf16_t)(v.x & 0xffffu); Vt[p.S + i] = (bf16_t)(v.x >> 16); Vt[2 * p.S + i] = (bf16_t)(v.y & 0xffffu); Vt[3 * p.S + i] = (bf16_t)(v.y >> 16);
    if (MASK) Mf[i] = M[i] ? -INFINITY : 0.f;
  }
  __syncthreads();
  float m = -INFINITY, l = 0.f;
  f32x4m acc[2] = {f32x4m{0.f, 0.f, 0.f, 0.f}, f32x4m{0.f, 0.f, 0.f, 0.f}};
  const int per = p.S >> 2, g0 = wave * per;
  const bf16_t* vrow = Vt + l4 * p.S;
  for (int g = g0; g < g0 + per; g += 32) {
    f32x4m s[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const s16x4 kf = __builtin_bit_cast(s16x4, Ks[g + 4 * c + l4]);
      f32x4m c0 = f32x4m{0.f, 0.f, 0.f, 0.f};
      if (MASK) { const float4 mf = *reinterpret_cast<const float4*>(Mf + g + 4 * c); c0 = f32x4m{mf.x, mf.y, mf.z, mf.w}; }
      s[c] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(kf, qf, c0, 0, 0, 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) tmax = fmaxf(tmax, s[c][i]);
    const float m_new = fmaxf(m, tmax * sc);
    if (MASK && m_new == -INFINITY) continue;                    // every key so far masked (per lane: keep the state)
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float ps = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) { s[c][i] = __builtin_amdgcn_exp2f(fmaf(s[c][i], sc, -m_new)); ps += s[c][i]; }
    acc[0] *= alpha; acc[1] *= alpha;
    l = l * alpha + ps; m = m_new;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const s16x4 vf = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(vrow + g + 4 * c));   // A: V^T[d = lane & 3][keys g+4c .. +3]
      const s16x4 pf = __builtin_bit_cast(s16x4, pack4(s[c][0], s[c][1], s[c][2], s[c][3]));           // B: this lane's p for those keys
      acc[c & 1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(vf, pf, acc[c & 1], 0, 0, 0);
    }
  }
  const f32x4m a4 = acc[0] + acc[1];
  float* pp = part + (wave * 64 + lane) * 6;
  pp[0] = m; pp[1] = l; pp[2] = a4[0]; pp[3] = a4[1]; pp[4] = a4[2]; pp[5] = a4[3];
  __syncthreads();
  if (wave == 0) {
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, part[(w * 64 + lane) * 6]);
    float ll = 0.f; float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* q6 = part + (w * 64 + lane) * 6;
      const float f = (q6[0] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(q6[0] - mm);
      ll += q6[1] * f; a.x += q6[2] * f; a.y += q6[3] * f; a.z += q6[4] * f; a.w += q6[5] * f;
    }
    // ll == 0 (every key masked) -> 0/0 = NaN like the reference
    St<bf16_t>::st4(p.o + b * p.o_bs + (long long)qrow * p.ldo + h * 4, make_float4(a.x / ll, a.y / ll, a.z / ll, a.w / ll));
  }
}

}  // namespace

/* which arithmetic smx_attention_f32 takes for this launch: 0 = exact fp32 products on the fp32 MFMA, 3 / 2 = split-bf16 (six products everywhere / P on two
 * levels) -- knob attn_bf3 (3 | 2, + 16 = at any launch size, + 32 = 128-query blocks; 0 = off), d_head 32, S % 64 == 0, at least 512 blocks' worth of 128 queries. */
extern "C" int smx_attention_f32_uses_bf3(int B, int H, int L, int S, int dh) {
  const int knob = smx_tune(SMX_TUNE_ATTN_BF3), np = knob & 15;
  if (dh != 32 || (np != 2 && np != 3 && np != 4) || B <= 0 || H <= 0 || L <= 0 || L % 128 || S <= 0 || S % 64) return 0;
  if (!(knob & 16) && (long long)(L / 128) * B * H < 512) return 0;
  return np;
}

namespace {
template <typename T>
int attention_launch(const T* q, int ldq, int64_t q_bs, const T* k, int ldk, int64_t k_bs, const T* v, int ldv, int64_t v_bs, T* o, int ldo,
                     int64_t o_bs, const uint8_t* key_mask, int B, int H, int L, int S, int dh, float scale, void* stream) {
  if (!q || !k || !v || !o || B <= 0 || H <= 0 || L <= 0 || S <= 0 || (long long)B * H > 65535) return SMX_EINVAL;
  if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4 || q_bs % 4 || k_bs % 4 || v_bs % 4 || o_bs % 4) return SMX_EINVAL;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & (4 * sizeof(T) - 1)) return SMX_EINVAL;
  AP<T> p{q, k, v, o, key_mask, q_bs, k_bs, v_bs, o_bs, ldq, ldk, ldv, ldo, H, L, S, scale};
  hipStream_t st = (hipStream_t)stream;
  if (dh == 4) {
    if (L % 64 || S % 32 || S > 4096) return SMX_EINVAL;       // all keys of a head live in LDS: 4096 (the 512 variant's 64x64 tokens) = 141 KB
    const size_t lds = (size_t)S * 33 + 4 * 64 * 6 * sizeof(float);
    {
      const void* fns[] = {(const void*)(attn_mfma4_kernel<T, true>), (const void*)(attn_mfma4_kernel<T, false>), (const void*)(attn_valu4_kernel<T>)};
      for (const void* f : fns) SMX_HIP(smx_max_dynamic_lds(f, 4096 * 36 + 4 * 64 * 6 * (int)sizeof(float)));
    }
    if constexpr (sizeof(T) == 2) {
      if (smx_tune(SMX_TUNE_ATTN4_MFMA) == 1 && S % 128 == 0) {          // bf16 storage: the 4x4x4 bf16 MFMA form (knob 2: the fp32 4x4x1 form on bf16 storage)
        {
          const void* fns[] = {(const void*)(attn_mfma4_bf16_kernel<true>), (const void*)(attn_mfma4_bf16_kernel<false>)};
          for (const void* f : fns) SMX_HIP(smx_max_dynamic_lds(f, 4096 * 20 + 4 * 64 * 6 * (int)sizeof(float)));
        }
        if (key_mask) SMX_LAUNCH(attn_mfma4_bf16_kernel<true>, dim3(L / 64, B * H), dim3(256), (size_t)S * 20 + 4 * 64 * 6 * sizeof(float), st, p);
        else SMX_LAUNCH(attn_mfma4_bf16_kernel<false>, dim3(L / 64, B * H), dim3(256), (size_t)S * 16 + 4 * 64 * 6 * sizeof(float), st, p);
        return smx_launch_status();
      }
    }
    if (smx_tune(SMX_TUNE_ATTN4_MFMA) && S % 64 == 0) {
      if (key_mask) SMX_LAUNCH((attn_mfma4_kernel<T, true>), dim3(L / 64, B * H), dim3(256), (size_t)S * 36 + 4 * 64 * 6 * sizeof(float), st, p);
      else SMX_LAUNCH((attn_mfma4_kernel<T, false>), dim3(L / 64, B * H), dim3(256), (size_t)S * 32 + 4 * 64 * 6 * sizeof(float), st, p);
    }
    else SMX_LAUNCH(attn_valu4_kernel<T>, dim3(L / 64, B * H), dim3(256), lds, st, p);
  } else if (dh == 32 || dh == 64) {
    if (L % 128 || S % 32) return SMX_EINVAL;
    if (dh == 32 && sizeof(T) == 2 && S % 64 == 0 && smx_tune(SMX_TUNE_ATTN16)) {
      if constexpr (sizeof(T) == 2) {
        if ((ldq | ldk | ldv) % 8 || ((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return SMX_EINVAL;
        SMX_LAUNCH(attn_mfma16_kernel<32>, dim3(L / 128, B * H), dim3(256), 0, st, p);
      }
    }
    else if (sizeof(T) == 4 && smx_attention_f32_uses_bf3(B, H, L, S, dh) && !(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15)) {
      if constexpr (sizeof(T) == 4) {                         // fp32 storage, big launches: fp32-grade products on the bf16 matrix pipe
        const int np = smx_attention_f32_uses_bf3(B, H, L, S, dh);
        const int nw = (smx_tune(SMX_TUNE_ATTN_BF3) & 32) || L % 256 ? 4 : 8;        // knob + 32: the 128-query blocks
#define AB3_GO(MK, NPV) do { if (nw == 8) SMX_LAUNCH((attn_bf3_kernel<MK, NPV, 8>), dim3(L / 256, B * H), dim3(512), 0, st, p); \
                             else SMX_LAUNCH((attn_bf3_kernel<MK, NPV, 4>), dim3(L / 128, B * H), dim3(256), 0, st, p); } while (0)
        if (np == 4) {
          if (key_mask) { if (nw == 8) SMX_LAUNCH((attn_f16_kernel<true, 8>), dim3(L / 256, B * H), dim3(512), 0, st, p); else SMX_LAUNCH((attn_f16_kernel<true, 4>), dim3(L / 128, B * H), dim3(256), 0, st, p); }
          else { if (nw == 8) SMX_LAUNCH((attn_f16_kernel<false, 8>), dim3(L / 256, B * H), dim3(512), 0, st, p); else SMX_LAUNCH((attn_f16_kernel<false, 4>), dim3(L / 128, B * H), dim3(256), 0, st, p); }
        }
        else if (key_mask) { if (np == 2) AB3_GO(true, 2); else AB3_GO(true, 3); }
        else { if (np == 2) AB3_GO(false, 2); else AB3_GO(false, 3); }
#undef AB3_GO
      }
    }
    else if (dh == 32 && !key_mask) SMX_LAUNCH((attn_mfma_kernel<T, 32, false>), dim3(L / 128, B * H), dim3(256), 0, st, p);
    else if (dh == 32) SMX_LAUNCH((attn_mfma_kernel<T, 32>), dim3(L / 128, B * H), dim3(256), 0, st, p);
    else SMX_LAUNCH((attn_mfma_kernel<T, 64>), dim3(L / 128, B * H), dim3(256), 0, st, p);
  } else {
    return SMX_EINVAL;
  }
  return smx_launch_status();
}
}  // namespace

extern "C" int smx_attention_f32(const float* q, int ldq, int64_t q_bs, const float* k, int ldk, int64_t k_bs,
                                 const float* v, int ldv, int64_t v_bs, float* o, int ldo, int64_t o_bs,
                                 const uint8_t* key_mask, int B, int H, int L, int S, int dh, float scale, void* stream) {
  return attention_launch<float>(q, ldq, q_bs, k, ldk, k_bs, v, ldv, v_bs, o, ldo, o_bs, key_mask, B, H, L, S, dh, scale, stream);
}
extern "C" int smx_attention_bf16(const void* q, int ldq, int64_t q_bs, const void* k, int ldk, int64_t k_bs,
                                  const void* v, int ldv, int64_t v_bs, void* o, int ldo, int64_t o_bs,
                                  const uint8_t* key_mask, int B, int H, int L, int S, int dh, float scale, void* stream) {
  return attention_launch<bf16_t>((const bf16_t*)q, ldq, q_bs, (const bf16_t*)k, ldk, k_bs, (const bf16_t*)v, ldv, v_bs, (bf16_t*)o, ldo, o_bs,
                                  key_mask, B, H, L, S, dh, scale, stream);
}

/* The AttnBlock core fused (attnblock16_kernel): o[b][q][:] = softmax_k(q.k * scale) v, ONE head of d = 256, V given TRANSPOSED. */
extern "C" int smx_attnblock_bf16(const void* q, int ldq, int64_t q_bs, const void* k, int ldk, int64_t k_bs, const void* vt, int ldvt, int64_t vt_bs,
                                  void* o, int ldo, int64_t o_bs, int B, int L, int S, int d, float scale, void* stream) {
  if (!q || !k || !vt || !o || B <= 0 || B > 65535 || d != 256 || L <= 0 || L % 128 || S <= 0 || S % 32) return SMX_EINVAL;
  if (ldq % 8 || ldk % 8 || ldvt % 8 || ldo % 4 || ldq < d || ldk < d || ldvt < S || ldo < d || q_bs % 8 || k_bs % 8 || vt_bs % 8 || o_bs % 4) return SMX_EINVAL;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15 || ((uintptr_t)o & 7)) return SMX_EINVAL;
  AP<bf16_t> p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)o, nullptr, q_bs, k_bs, vt_bs, o_bs, ldq, ldk, ldvt, ldo, 1, L, S, scale};
  constexpr int LDS = 4 * (32 * 256 * 2 + 256 * 32 * 2);     // 4 stages x 32 KB
  SMX_HIP(smx_max_dynamic_lds((const void*)attnblock16_kernel<256>, LDS));
  SMX_LAUNCH(attnblock16_kernel<256>, dim3(L / 128, B), dim3(512), LDS, (hipStream_t)stream, p);
  return smx_launch_status();
}

/* The same core in the fp32 configuration (attnblock32_kernel): fp32 storage, exact fp32 products on v_mfma_f32_32x32x2_f32. */
extern "C" int smx_attnblock_f32(const float* q, int ldq, int64_t q_bs, const float* k, int ldk, int64_t k_bs, const float* vt, int ldvt, int64_t vt_bs,
                                 float* o, int ldo, int64_t o_bs, int B, int L, int S, int d, float scale, void* stream) {
  if (!q || !k || !vt || !o || B <= 0 || B > 65535 || d != 256 || L <= 0 || L % 128 || S <= 0 || S % 32) return SMX_EINVAL;
  if (ldq % 4 || ldk % 4 || ldvt % 4 || ldo % 4 || ldq < d || ldk < d || ldvt < S || ldo < d || q_bs % 4 || k_bs % 4 || vt_bs % 4 || o_bs % 4) return SMX_EINVAL;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)o) & 15) return SMX_EINVAL;
  AP<float> p{q, k, vt, o, nullptr, q_bs, k_bs, vt_bs, o_bs, ldq, ldk, ldvt, ldo, 1, L, S, scale};
  if ((smx_tune(SMX_TUNE_ATTN_BF3) & 15) == 4 && ((smx_tune(SMX_TUNE_ATTN_BF3) & 16) || (long long)(L / 128) * B >= 256)) {
    // big launches: the f16x3 kernel (knob attn_bf3 = 4, the default; + 16: at any launch size)
    constexpr int LDS16 = 4 * 32 * (256 * 2 + 16) + 4 * 256 * (32 * 2 + 16) + 32;
    SMX_HIP(smx_max_dynamic_lds((const void*)attnblock_f16_kernel, LDS16));
    SMX_LAUNCH(attnblock_f16_kernel, dim3(L / 128, B), dim3(256), LDS16, (hipStream_t)stream, p);
    return smx_launch_status();
  }
  constexpr int LDS = 2 * (32 * 256 * 4 + 256 * 32 * 4);     // 2 stages x 64 KB
  SMX_HIP(smx_max_dynamic_lds((const void*)attnblock32_kernel<256>, LDS));
  SMX_LAUNCH(attnblock32_kernel<256>, dim3(L / 128, B), dim3(256), LDS, (hipStream_t)stream, p);
  return smx_launch_status();
}
