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

template <int DH>
__global__ __launch_bounds__(128) void attn_bwd_q_kernel(ABP p) {
  constexpr int TK = 64;
  __shared__ float Ks[TK * DH], Vs[TK * DH];
  __shared__ uint8_t Ms[TK];
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  const int i = blockIdx.x * 128 + threadIdx.x;
  const bool ok = i < p.L;
  const int E = p.H * DH;
  float q[DH], g[DH], acc[DH];
  float Di = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = ok ? p.q[(long long)b * p.q_bs + (long long)i * p.ldq + h * DH + d] * p.scale : 0.f;
    g[d] = ok ? p.d_o[((long long)b * p.L + i) * E + h * DH + d] : 0.f;
    const float ov = ok ? p.o[((long long)b * p.L + i) * E + h * DH + d] : 0.f;
    Di += g[d] * ov; acc[d] = 0.f;
  }
  const float* Kb = p.k + (long long)b * p.k_bs + h * DH;
  const float* Vb = p.v + (long long)b * p.v_bs + h * DH;
  float m = -INFINITY, l = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const float il = pass ? 1.f / l : 0.f;
    for (int j0 = 0; j0 < p.S; j0 += TK) {
      __syncthreads();
      for (int t = threadIdx.x; t < TK * DH; t += 128) {
        const int r = t / DH, d = t - r * DH;
        const bool in = j0 + r < p.S;
        Ks[t] = in ? Kb[(long long)(j0 + r) * p.ldk + d] : 0.f;
        if (pass) Vs[t] = in ? Vb[(long long)(j0 + r) * p.ldv + d] : 0.f;
      }
      if (threadIdx.x < TK) Ms[threadIdx.x] = (j0 + threadIdx.x < p.S) ? (p.mask ? p.mask[(long long)b * p.S + j0 + threadIdx.x] : 0) : 1;
      __syncthreads();
      const int n = min(TK, p.S - j0);
      for (int r = 0; r < n; ++r) {
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
          for (int d = 0; d < DH; ++d) dp += g[d] * Vs[r * DH + d];
          const float ds = pij * (dp - Di);
#pragma unroll
          for (int d = 0; d < DH; ++d) acc[d] += ds * Ks[r * DH + d];
        }
      }
    }
  }
  if (ok) {
#pragma unroll
    for (int d = 0; d < DH; ++d) p.dq[((long long)b * p.L + i) * E + h * DH + d] = acc[d] * p.scale;
    float* st = p.stats + (((long long)b * p.H + h) * p.L + i) * 3;
    st[0] = m; st[1] = 1.f / l; st[2] = Di;
  }
}

template <int DH>
__global__ __launch_bounds__(128) void attn_bwd_kv_kernel(ABP p) {
  constexpr int TQ = 64;
  __shared__ float Qs[TQ * DH], Gs[TQ * DH], St[TQ * 3];
  // grid.y = B * H: one (sample, head) per block row also when the context is shared by the batch (k_bs = v_bs = 0) -- the per-sample
  // dK / dV are then summed over the batch by the caller's fixed-order batch sum; looping the batch inside a (head, key-tile) block
  // left 8 x H blocks for 256 CUs
  const int bb = blockIdx.y / p.H, h = blockIdx.y - bb * p.H;
  const int j = blockIdx.x * 128 + threadIdx.x;
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
      for (int t = threadIdx.x; t < TQ * DH; t += 128) {
        const int r = t / DH, d = t - r * DH;
        const bool in = i0 + r < p.L;
        Qs[t] = in ? p.q[(long long)b * p.q_bs + (long long)(i0 + r) * p.ldq + h * DH + d] * p.scale : 0.f;
        Gs[t] = in ? p.d_o[((long long)b * p.L + i0 + r) * E + h * DH + d] : 0.f;
      }
      for (int t = threadIdx.x; t < TQ * 3; t += 128) {
        const int r = t / 3;
        St[t] = (i0 + r < p.L) ? p.stats[(((long long)b * p.H + h) * p.L + i0 + r) * 3 + (t - r * 3)] : 0.f;
      }
      __syncthreads();
      if (!ok || masked) continue;
      const int n = min(TQ, p.L - i0);
      for (int r = 0; r < n; ++r) {
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
  if (ok) {
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      p.dk[((long long)bb * p.S + j) * E + h * DH + d] = dk[d];
      p.dv[((long long)bb * p.S + j) * E + h * DH + d] = dv[d];
    }
  }
}

template <int DH>
int launch(ABP& p, int shared, hipStream_t st) {
  SMX_LAUNCH(attn_bwd_q_kernel<DH>, dim3(smx_cdiv(p.L, 128), p.B * p.H), dim3(128), 0, st, p);
  (void)shared;
  SMX_LAUNCH(attn_bwd_kv_kernel<DH>, dim3(smx_cdiv(p.S, 128), p.B * p.H), dim3(128), 0, st, p);
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
