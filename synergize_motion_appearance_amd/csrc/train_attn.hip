// Training: backward of the multi-head attention core (nn.MultiheadAttention as the reference uses it,
// archs/appmotioncodebook_arch.py:88-126; forward: attention.hip) for gfx950, flash style -- the [L][S] probability matrix is
// never materialised: it is recomputed tile by tile from q, k and the saved row statistics.
//
//   P = softmax(scale q k^T + mask),  O = P V
//   D_i  = sum_d dO_id O_id
//   dV_j = sum_i P_ij dO_i          dP_ij = dO_i . V_j          dS_ij = P_ij (dP_ij - D_i)
//   dQ_i = scale sum_j dS_ij K_j    dK_j  = scale sum_i dS_ij Q_i
//
// Two kernels: (1) one thread per QUERY: pass 1 over the keys -> row max / sum (and D_i), pass 2 -> dQ; the statistics
// {m, 1/l, D} are written for kernel (2), one thread per KEY: loops the queries (Q / dO / statistics tiles broadcast from LDS)
// -> dK, dV per sample.  With a codebook context shared by the batch (cross attention: K/V batch stride 0) the caller sums the
// per-sample dK / dV over the batch in a fixed order (smx_batch_sum_f32) -- no atomics, bit-reproducible.
// d_head is 4 (motion transformer) or 32 (appearance transformer): q / k / v / dO rows live in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "smx.h"
#include "smx_common.h"

namespace {

struct ABP {
  const float* q; const float* k; const float* v; const float* o; const float* d_o; const uint8_t* mask;
  float* dq; float* dk; float* dv; float* stats;
  long long q_bs, k_bs, v_bs;        // element strides between samples (k_bs = v_bs = 0: shared context)
  int ldq, ldk, ldv;
  int B, H, L, S; float scale;
};

// Both kernels: 256 threads = 64 rows (queries / keys) x 4 slices of the other axis.  Thread (row r, slice g = wave index) walks the tile
// entries j = g, g + 4, ... ; the four partial results of a row are combined through LDS in a fixed order.  (One thread per row alone left
// B * H * L / 128 = 256 two-wave blocks for 1024 SIMDs at the training batch: latency-bound at a tenth of the VALU rate.)
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(ABP p) {
  constexpr int TK = 64;
  __shared__ float Ks[TK * DH], Vs[TK * DH];
  __shared__ uint8_t Ms[TK];
  __shared__ float red[4][64][DH > 4 ? DH : 4];        // statistics / dQ partials of the 4 slices
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int rl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + rl;
  const bool ok = i < p.L;
  const int E = p.H * DH;
  float q[DH], gd[DH], acc[DH];
  float Di = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = ok ? p.q[(long long)b * p.q_bs + (long long)i * p.ldq + h * DH + d] * p.scale : 0.f;
    gd[d] = ok ? p.d_o[((long long)b * p.L + i) * E + h * DH + d] : 0.f;
    const float ov = ok ? p.o[((long long)b * p.L + i) * E + h * DH + d] : 0.f;
    Di += gd[d] * ov; acc[d] = 0.f;
  }
  const float* Kb = p.k + (long long)b * p.k_bs + h * DH;
  const float* Vb = p.v + (long long)b * p.v_bs + h * DH;
  float m = -INFINITY, l = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const float il = pass ? 1.f / l : 0.f;
    for (int j0 = 0; j0 < p.S; j0 += TK) {
      __syncthreads();
      for (int t = threadIdx.x; t < TK * DH; t += 256) {
        const int r = t / DH, d = t - r * DH;
        const bool in = j0 + r < p.S;
        Ks[t] = in ? Kb[(long long)(j0 + r) * p.ldk + d] : 0.f;
        if (pass) Vs[t] = in ? Vb[(long long)(j0 + r) * p.ldv + d] : 0.f;
      }
      if (threadIdx.x < TK) Ms[threadIdx.x] = (j0 + threadIdx.x < p.S) ? (p.mask ? p.mask[(long long)b * p.S + j0 + threadIdx.x] : 0) : 1;
      __syncthreads();
      const int n = min(TK, p.S - j0);
      for (int r = g; r < n; r += 4) {
        if (Ms[r]) continue;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) s += q[d] * Ks[r * DH + d];
        if (!pass) {
          if (s > m) { l = l * expf(m - s) + 1.f; m = s; } else l += expf(s - m);
        } else {
          const float pij = expf(s - m) * il;
          float dp = 0.f;
#pragma unroll
          for (int d = 0; d < DH; ++d) dp += gd[d] * Vs[r * DH + d];
          const float ds = pij * (dp - Di);
#pragma unroll
          for (int d = 0; d < DH; ++d) acc[d] += ds * Ks[r * DH + d];
        }
      }
    }
    if (!pass) {                                          // the row's (m, l) over ALL keys from the four slices' (m_g, l_g)
      red[g][rl][0] = m; red[g][rl][1] = l;
      __syncthreads();
      float mm = fmaxf(fmaxf(red[0][rl][0], red[1][rl][0]), fmaxf(red[2][rl][0], red[3][rl][0]));
      float ll = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) ll += (red[k][rl][0] == -INFINITY) ? 0.f : red[k][rl][1] * expf(red[k][rl][0] - mm);
      m = mm; l = ll;
      __syncthreads();
    }
  }
#pragma unroll
  for (int d = 0; d < DH; ++d) red[g][rl][d] = acc[d];
  __syncthreads();
  if (g == 0 && ok) {
#pragma unroll
    for (int d = 0; d < DH; ++d)
      p.dq[((long long)b * p.L + i) * E + h * DH + d] = ((red[0][rl][d] + red[1][rl][d]) + (red[2][rl][d] + red[3][rl][d])) * p.scale;
    float* st = p.stats + (((long long)b * p.H + h) * p.L + i) * 3;
    st[0] = m; st[1] = 1.f / l; st[2] = Di;
  }
}

template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(ABP p) {
  constexpr int TQ = 64;
  __shared__ float Qs[TQ * DH], Gs[TQ * DH], St[TQ * 3];
  __shared__ float red[4][64][2 * DH];                 // dK | dV partials of the 4 slices
  // grid.y = B * H: one (sample, head) per block row also when the context is shared by the batch (k_bs = v_bs = 0) -- the per-sample
  // dK / dV are then summed over the batch by the caller's fixed-order batch sum
  const int bb = blockIdx.y / p.H, h = blockIdx.y - bb * p.H;
  const int rl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + rl;
  const bool ok = j < p.S;
  const int E = p.H * DH;
  float kk[DH], vv[DH], dk[DH], dv[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    kk[d] = ok ? p.k[(long long)bb * p.k_bs + (long long)j * p.ldk + h * DH + d] : 0.f;
    vv[d] = ok ? p.v[(long long)bb * p.v_bs + (long long)j * p.ldv + h * DH + d] : 0.f;
    dk[d] = 0.f; dv[d] = 0.f;
  }
  {
    const int b = bb;
    const bool masked = p.mask && ok && p.mask[(long long)b * p.S + j];
    for (int i0 = 0; i0 < p.L; i0 += TQ) {
      __syncthreads();
      for (int t = threadIdx.x; t < TQ * DH; t += 256) {
        const int r = t / DH, d = t - r * DH;
        const bool in = i0 + r < p.L;
        Qs[t] = in ? p.q[(long long)b * p.q_bs + (long long)(i0 + r) * p.ldq + h * DH + d] * p.scale : 0.f;
        Gs[t] = in ? p.d_o[((long long)b * p.L + i0 + r) * E + h * DH + d] : 0.f;
      }
      for (int t = threadIdx.x; t < TQ * 3; t += 256) {
        const int r = t / 3;
        St[t] = (i0 + r < p.L) ? p.stats[(((long long)b * p.H + h) * p.L + i0 + r) * 3 + (t - r * 3)] : 0.f;
      }
      __syncthreads();
      if (!ok || masked) continue;
      const int n = min(TQ, p.L - i0);
      for (int r = g; r < n; r += 4) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) { s += Qs[r * DH + d] * kk[d]; dp += Gs[r * DH + d] * vv[d]; }
        const float pij = expf(s - St[r * 3]) * St[r * 3 + 1];
        const float ds = pij * (dp - St[r * 3 + 2]);
#pragma unroll
        for (int d = 0; d < DH; ++d) { dv[d] += pij * Gs[r * DH + d]; dk[d] += ds * Qs[r * DH + d]; }   // Qs already carries `scale`
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < DH; ++d) { red[g][rl][d] = dk[d]; red[g][rl][DH + d] = dv[d]; }
  __syncthreads();
  if (g == 0 && ok) {
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      p.dk[((long long)bb * p.S + j) * E + h * DH + d] = (red[0][rl][d] + red[1][rl][d]) + (red[2][rl][d] + red[3][rl][d]);
      p.dv[((long long)bb * p.S + j) * E + h * DH + d] = (red[0][rl][DH + d] + red[1][rl][DH + d]) + (red[2][rl][DH + d] + red[3][rl][DH + d]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// d_head = 32 on the fp32 MFMA (v_mfma_f32_32x32x2_f32), same "swapped product" layout as the forward kernel (attention.hip): one
// wave owns 32 queries (kernel Q) or 32 keys (kernel KV), whose q / dO (k / v) rows live in registers as MFMA B operands; the other
// side streams through LDS in 32-row tiles (pitch 36: conflict-free ds_read_b128 A fragments, and the [row][d] tile read "down the
// d axis" one dword per lane is the TRANSPOSED A operand of the second product -- no transpose pass).
//   kernel Q :  S^T = K Q^T and dP^T = V dO^T are [keys x queries] accumulators (lane = one query, 16 keys): softmax statistics, P and
//               dS are lane-local; dQ^T[d x queries] += K^T dS^T takes dS straight from those accumulator registers (the contraction
//               runs over the keys in the order the C layout holds them).  Two sweeps over the keys: statistics, then gradients.
//   kernel KV:  S = Q K^T and dP = dO V^T are [queries x keys] (lane = one key, 16 queries; the per-query {m, 1/l, D} come from the
//               statistics kernel Q wrote);  dV^T[d x keys] += dO^T P,  dK^T[d x keys] += Q^T dS.
// Scores are kept in log2 units (q pre-scaled by scale * log2 e, v_exp_f32): dS is the gradient w.r.t. the NATURAL score, so
// dQ = scale * sum dS K and dK = ln2 * sum dS (q scale log2 e).
// ---------------------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;

__global__ __launch_bounds__(256, 2) void attn_bwd_q_mfma_kernel(ABP p) {
  constexpr int DH = 32, TK = 32, KLD = DH + 4, KS = DH / 8;
  __shared__ __attribute__((aligned(16))) float Ks[2][TK * KLD];
  __shared__ __attribute__((aligned(16))) float Vs[2][TK * KLD];
  __shared__ uint8_t Ms[2][TK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5;
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int qrow = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int E = p.H * DH;
  const float* Q = p.q + (long long)b * p.q_bs + (long long)qrow * p.ldq + h * DH;
  const float* G = p.d_o + ((long long)b * p.L + qrow) * E + h * DH;
  const float* O = p.o + ((long long)b * p.L + qrow) * E + h * DH;
  const float* K = p.k + (long long)b * p.k_bs + h * DH;
  const float* V = p.v + (long long)b * p.v_bs + h * DH;
  const uint8_t* M = p.mask ? p.mask + (long long)b * p.S : nullptr;
  float4 qf[KS], gf[KS];
  float Di = 0.f;
  const float sc = p.scale * LOG2E;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const float4 t = *reinterpret_cast<const float4*>(Q + kk * 8 + hh * 4);
    qf[kk] = make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc);
    gf[kk] = *reinterpret_cast<const float4*>(G + kk * 8 + hh * 4);
    const float4 ov = *reinterpret_cast<const float4*>(O + kk * 8 + hh * 4);
    Di += gf[kk].x * ov.x + gf[kk].y * ov.y + gf[kk].z * ov.z + gf[kk].w * ov.w;
  }
  Di += __shfl_xor(Di, 32, 64);

  float4 kreg, vreg; uint8_t mreg = 0;
  const int sr = threadIdx.x >> 3, sc4 = threadIdx.x & 7;           // staging: row, float4 column (32 rows x 8 float4 = 256 threads)
  auto load_tile = [&](int key0, bool with_v) {
    kreg = *reinterpret_cast<const float4*>(K + (long long)(key0 + sr) * p.ldk + sc4 * 4);
    if (with_v) vreg = *reinterpret_cast<const float4*>(V + (long long)(key0 + sr) * p.ldv + sc4 * 4);
    if (M && threadIdx.x < TK) mreg = M[key0 + threadIdx.x];
  };
  auto store_tile = [&](int buf, bool with_v) {
    *reinterpret_cast<float4*>(&Ks[buf][sr * KLD + sc4 * 4]) = kreg;
    if (with_v) *reinterpret_cast<float4*>(&Vs[buf][sr * KLD + sc4 * 4]) = vreg;
    if (threadIdx.x < TK) Ms[buf][threadIdx.x] = M ? mreg : 0;
  };
  auto scores = [&](const float* tile, const float4 (&bf)[KS]) {
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
    const float* kp = tile + (lane & 31) * KLD + hh * 4;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const float4 kf = *reinterpret_cast<const float4*>(kp + kk * 8);
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, bf[kk].x, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, bf[kk].y, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, bf[kk].z, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, bf[kk].w, a, 0, 0, 0);
    }
    return a;
  };
  const int ntiles = p.S / TK;
  float m = -INFINITY, l = 0.f;
  f32x16 dq;
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const float il = pass ? 1.f / l : 0.f;
    __syncthreads();                                      // the previous sweep's last tile is still being read
    load_tile(0, pass); store_tile(0, pass);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
      const int buf = t & 1;
      if (t + 1 < ntiles) load_tile((t + 1) * TK, pass);
      f32x16 s = scores(Ks[buf], qf);
      if (!pass) {
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (Ms[buf][(r & 3) + 8 * (r >> 2) + 4 * hh]) s[r] = -INFINITY;
          tmax = fmaxf(tmax, s[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);
        const bool dead = (m_new == -INFINITY);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) psum += dead ? 0.f : __builtin_amdgcn_exp2f(s[r] - m_new);
        psum += __shfl_xor(psum, 32, 64);
        l = l * (dead ? 1.f : __builtin_amdgcn_exp2f(m - m_new)) + psum; m = m_new;
      } else {
        const f32x16 dp = scores(Vs[buf], gf);
        const float* kt = &Ks[buf][lane & 31];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = (r & 3) + 8 * (r >> 2) + 4 * hh;
          const float pij = Ms[buf][key] ? 0.f : __builtin_amdgcn_exp2f(s[r] - m) * il;
          const float ds = pij * (dp[r] - Di);
          dq = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[key * KLD], ds, dq, 0, 0, 0);
        }
      }
      if (t + 1 < ntiles) store_tile(buf ^ 1, pass);
      __syncthreads();
    }
  }
  float* DQ = p.dq + ((long long)b * p.L + qrow) * E + h * DH;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(DQ + 8 * g + 4 * hh) = make_float4(dq[4 * g] * p.scale, dq[4 * g + 1] * p.scale, dq[4 * g + 2] * p.scale, dq[4 * g + 3] * p.scale);
  if (hh == 0) {
    float* st = p.stats + (((long long)b * p.H + h) * p.L + qrow) * 3;
    st[0] = m; st[1] = 1.f / l; st[2] = Di;               // m in log2 units (kernel KV uses the same scaling)
  }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_kv_mfma_kernel(ABP p) {
  constexpr int DH = 32, TQ = 32, KLD = DH + 4, KS = DH / 8;
  __shared__ __attribute__((aligned(16))) float Qs[2][TQ * KLD];
  __shared__ __attribute__((aligned(16))) float Gs[2][TQ * KLD];
  __shared__ float St[2][TQ * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5;
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int krow = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const int E = p.H * DH;
  const float* K = p.k + (long long)b * p.k_bs + (long long)krow * p.ldk + h * DH;
  const float* V = p.v + (long long)b * p.v_bs + (long long)krow * p.ldv + h * DH;
  const bool masked = p.mask && p.mask[(long long)b * p.S + krow];
  float4 kf[KS], vf[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    kf[kk] = *reinterpret_cast<const float4*>(K + kk * 8 + hh * 4);
    vf[kk] = *reinterpret_cast<const float4*>(V + kk * 8 + hh * 4);
  }
  const float* Qb = p.q + (long long)b * p.q_bs + h * DH;
  const float* Gb = p.d_o + (long long)b * p.L * E + h * DH;
  const float* Sb = p.stats + ((long long)b * p.H + h) * p.L * 3;
  const float sc = p.scale * LOG2E;
  float4 qreg, greg; float sreg = 0.f;
  const int sr = threadIdx.x >> 3, sc4 = threadIdx.x & 7;
  auto load_tile = [&](int q0) {
    const float4 t = *reinterpret_cast<const float4*>(Qb + (long long)(q0 + sr) * p.ldq + sc4 * 4);
    qreg = make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc);
    greg = *reinterpret_cast<const float4*>(Gb + (long long)(q0 + sr) * E + sc4 * 4);
    if (threadIdx.x < TQ * 3) sreg = Sb[(long long)q0 * 3 + threadIdx.x];
  };
  auto store_tile = [&](int buf) {
    *reinterpret_cast<float4*>(&Qs[buf][sr * KLD + sc4 * 4]) = qreg;
    *reinterpret_cast<float4*>(&Gs[buf][sr * KLD + sc4 * 4]) = greg;
    if (threadIdx.x < TQ * 3) St[buf][threadIdx.x] = sreg;
  };
  auto scores = [&](const float* tile, const float4 (&bf)[KS]) {
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
    const float* ap = tile + (lane & 31) * KLD + hh * 4;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const float4 af = *reinterpret_cast<const float4*>(ap + kk * 8);
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf[kk].x, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf[kk].y, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf[kk].z, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf[kk].w, a, 0, 0, 0);
    }
    return a;
  };
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
  const int ntiles = p.L / TQ;
  load_tile(0); store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile((t + 1) * TQ);
    const f32x16 s = scores(Qs[buf], kf);                  // [queries x keys]: lane = key (lane & 31), row r = query (r&3) + 8 (r>>2) + 4 hh
    const f32x16 dp = scores(Gs[buf], vf);
    const float* qt = &Qs[buf][lane & 31];
    const float* gt = &Gs[buf][lane & 31];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float pij = masked ? 0.f : __builtin_amdgcn_exp2f(s[r] - St[buf][qi * 3]) * St[buf][qi * 3 + 1];
      const float ds = pij * (dp[r] - St[buf][qi * 3 + 2]);
      dv = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[qi * KLD], pij, dv, 0, 0, 0);
      dk = __builtin_amdgcn_mfma_f32_32x32x2f32(qt[qi * KLD], ds, dk, 0, 0, 0);
    }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }
  float* DK = p.dk + ((long long)b * p.S + krow) * E + h * DH;
  float* DV = p.dv + ((long long)b * p.S + krow) * E + h * DH;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *reinterpret_cast<float4*>(DK + 8 * g + 4 * hh) = make_float4(dk[4 * g] * LN2, dk[4 * g + 1] * LN2, dk[4 * g + 2] * LN2, dk[4 * g + 3] * LN2);
    *reinterpret_cast<float4*>(DV + 8 * g + 4 * hh) = make_float4(dv[4 * g], dv[4 * g + 1], dv[4 * g + 2], dv[4 * g + 3]);
  }
}

template <int DH>
int launch(ABP& p, int shared, hipStream_t st) {
  if (DH == 32 && p.L % 128 == 0 && p.S % 128 == 0 && p.ldq % 4 == 0 && p.ldk % 4 == 0 && p.ldv % 4 == 0 && p.q_bs % 4 == 0 && p.k_bs % 4 == 0 &&
      p.v_bs % 4 == 0 && ((((uintptr_t)p.q) | ((uintptr_t)p.k) | ((uintptr_t)p.v) | ((uintptr_t)p.o) | ((uintptr_t)p.d_o) | ((uintptr_t)p.dq) |
                            ((uintptr_t)p.dk) | ((uintptr_t)p.dv)) & 15) == 0 && smx_tune(SMX_TUNE_ATTN_BWD_MFMA)) {
    SMX_LAUNCH(attn_bwd_q_mfma_kernel, dim3(p.L / 128, p.B * p.H), dim3(256), 0, st, p);
    SMX_LAUNCH(attn_bwd_kv_mfma_kernel, dim3(p.S / 128, p.B * p.H), dim3(256), 0, st, p);
    return smx_launch_status();
  }
  SMX_LAUNCH(attn_bwd_q_kernel<DH>, dim3(smx_cdiv(p.L, 64), p.B * p.H), dim3(256), 0, st, p);
  (void)shared;
  SMX_LAUNCH(attn_bwd_kv_kernel<DH>, dim3(smx_cdiv(p.S, 64), p.B * p.H), dim3(256), 0, st, p);
  return smx_launch_status();
}

}  // namespace

/* q [B][L][ldq] (head h at column h*dh; sample stride q_bs), k / v rows [S] (ld, sample stride k_bs / v_bs; 0 = a context
 * shared by the batch), o / d_o [B][L][H*dh] dense (the forward output and its gradient), key_mask [B][S] bytes or null.
 * -> dq [B][L][H*dh]; dk, dv [B][S][H*dh] PER SAMPLE (a shared context's gradient is their batch sum: smx_batch_sum_f32);
 * stats: B*H*L*3 floats scratch. */
extern "C" int smx_attention_bwd_f32(const float* q, int ldq, int64_t q_bs, const float* k, int ldk, int64_t k_bs,
                                     const float* v, int ldv, int64_t v_bs, const float* o, const float* d_o,
                                     const uint8_t* key_mask, float* dq, float* dk, float* dv, float* stats,
                                     int B, int H, int L, int S, int dh, float scale, void* stream) {
  if (!q || !k || !v || !o || !d_o || !dq || !dk || !dv || !stats || B <= 0 || H <= 0 || L <= 0 || S <= 0) return SMX_EINVAL;
  if ((k_bs == 0) != (v_bs == 0) || (long long)B * H > 65535) return SMX_EINVAL;
  if (key_mask && k_bs == 0) return SMX_EINVAL;                 // a per-sample key mask with a shared context is not a reference case
  ABP p;
  p.q = q; p.k = k; p.v = v; p.o = o; p.d_o = d_o; p.mask = key_mask; p.dq = dq; p.dk = dk; p.dv = dv; p.stats = stats;
  p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
  p.B = B; p.H = H; p.L = L; p.S = S; p.scale = scale;
  const int shared = k_bs == 0 ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  switch (dh) {
    case 4: return launch<4>(p, shared, st);
    case 32: return launch<32>(p, shared, st);
    default: return SMX_EINVAL;
  }
}
