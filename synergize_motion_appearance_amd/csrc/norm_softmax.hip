// GroupNorm(32)+swish on NHWC, LayerNorm(+pos) on tokens, masked row softmax -- gfx950.
// All three are HBM-bound streaming kernels: float4 (16 B/lane) accesses, wave64 shuffles
// for the row reductions, LDS only for the cross-wave step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

namespace {

// ---------------------------------------------------------------------------------------
// GroupNorm: pass 1 -- per (b, chunk, c) partial sum / sumsq
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int gn_ppc(int HW) {
  int p = HW / 64; if (p < 32) p = 32; if (p > 256) p = 256; return p;
}

// Partials are {mean, M2 = sum (x - mean)^2} per (b, chunk, channel) -- Welford/Chan form, NOT {sum, sum^2}:
// var = E[x^2] - mean^2 cancels catastrophically in fp32 once |mean| >> std (real checkpoints after a few
// stacked ResBlocks), whereas per-thread sums shifted by the thread's first value and pairwise Chan merges only
// ever subtract numbers of similar size.
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int ldx, float* __restrict__ part,
                                                         int HW, int C, int ppc, int nch) {
  extern __shared__ float red[];                       // [rows][C][3] = {n, mean, M2}
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cq = C >> 2;                               // float4 columns
  const int tx = threadIdx.x % cq, ty = threadIdx.x / cq, rows = 256 / cq;
  const int p0 = chunk * ppc, p1 = min(p0 + ppc, HW);
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  const T* xb = x + (long long)b * HW * ldx + tx * 4;
  int p = p0 + ty;
  float piv[4] = {0.f, 0.f, 0.f, 0.f};                 // shift: this thread's first value per channel
  if (p < p1) { const float4 v0 = St<T>::ld4(xb + (long long)p * ldx); piv[0] = v0.x; piv[1] = v0.y; piv[2] = v0.z; piv[3] = v0.w; }
  int cnt = 0;
  for (; p + 3 * rows < p1; p += 4 * rows) {             // 4 independent 16-byte loads in flight per lane
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = St<T>::ld4(xb + (long long)(p + u * rows) * ldx);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float d0 = v[u].x - piv[0], d1 = v[u].y - piv[1], d2 = v[u].z - piv[2], d3 = v[u].w - piv[3];
      s[0] += d0; s[1] += d1; s[2] += d2; s[3] += d3;
      q[0] += d0 * d0; q[1] += d1 * d1; q[2] += d2 * d2; q[3] += d3 * d3;
    }
    cnt += 4;
  }
  for (; p < p1; p += rows) {
    float4 v = St<T>::ld4(xb + (long long)p * ldx);
    const float d0 = v.x - piv[0], d1 = v.y - piv[1], d2 = v.z - piv[2], d3 = v.w - piv[3];
    s[0] += d0; s[1] += d1; s[2] += d2; s[3] += d3;
    q[0] += d0 * d0; q[1] += d1 * d1; q[2] += d2 * d2; q[3] += d3 * d3;
    ++cnt;
  }
  float* r = red + ((ty * C) + tx * 4) * 3;
  const float fn = (float)cnt, inv = cnt ? 1.f / fn : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ms = s[e] * inv;                       // mean of the shifted values
    r[3 * e] = fn; r[3 * e + 1] = piv[e] + ms; r[3 * e + 2] = fmaxf(q[e] - s[e] * ms, 0.f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int t = 0; t < rows; ++t) {                   // Chan merge
      const float nb = red[(t * C + c) * 3], mb = red[(t * C + c) * 3 + 1], qb = red[(t * C + c) * 3 + 2];
      if (nb > 0.f) {
        const float nn = n + nb, d = mb - mean, f = nb / nn;
        mean += d * f; m2 += qb + d * d * n * f; n = nn;
      }
    }
    float* o = part + (((long long)b * nch + chunk) * C + c) * 2;
    o[0] = mean; o[1] = m2;
  }
}

// pass 2 -- one wave per (b, group): merge the (chunk, channel) partials {mean, M2} (chunk c holds
// min(ppc, HW - c*ppc) pixels) in double -- group mean first, then M2 + n (mean_i - mean)^2 -- -> per-channel
// scale/shift
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ ss, int HW, int C,
                                                         int groups, int nch, int ppc, float eps, float* __restrict__ mr = nullptr) {
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
  const int cpg = C / groups, items = nch * cpg;
  double s = 0.0;
  for (int i = threadIdx.x; i < items; i += 64) {
    const int chunk = i / cpg, c = g * cpg + (i - chunk * cpg);
    const int n_i = min(ppc, HW - chunk * ppc);
    s += (double)part[(((long long)b * nch + chunk) * C + c) * 2] * n_i;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const double n = (double)HW * cpg, mean = s / n;
  double q = 0.0;
  for (int i = threadIdx.x; i < items; i += 64) {
    const int chunk = i / cpg, c = g * cpg + (i - chunk * cpg);
    const int n_i = min(ppc, HW - chunk * ppc);
    const float* pp = part + (((long long)b * nch + chunk) * C + c) * 2;
    const double d = (double)pp[0] - mean;
    q += (double)pp[1] + d * d * n_i;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const double var = q / n;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (mr && threadIdx.x == 0) { mr[(long long)blockIdx.x * 2] = (float)mean; mr[(long long)blockIdx.x * 2 + 1] = rstd; }   // training: saved for the backward
  for (int j = threadIdx.x; j < cpg; j += 64) {
    const int c = g * cpg + j;
    const float sc = rstd * gamma[c];
    ss[((long long)b * C + c) * 2] = sc;
    ss[((long long)b * C + c) * 2 + 1] = beta[c] - (float)mean * sc;
  }
}

// pass 3 -- y = swish(x*scale + shift)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void gn_apply_kernel(const TI* __restrict__ x, int ldx, TO* __restrict__ y, int ldy,
                                                       const float* __restrict__ ss, long long total4, int HW, int C, int swish) {
  const int cq = C >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % cq); const long long pix = i / cq; const int b = (int)(pix / HW);
    float4 v = St<TI>::ld4(x + pix * ldx + c4 * 4);
    const float4 s0 = *reinterpret_cast<const float4*>(ss + ((long long)b * C + c4 * 4) * 2);
    const float4 s1 = *reinterpret_cast<const float4*>(ss + ((long long)b * C + c4 * 4) * 2 + 4);
    float r[4] = {v.x * s0.x + s0.y, v.y * s0.z + s0.w, v.z * s1.x + s1.y, v.w * s1.z + s1.w};
    if (swish) {
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = r[e] / (1.f + expf(-r[e]));
    }
    St<TO>::st4(y + pix * ldy + c4 * 4, make_float4(r[0], r[1], r[2], r[3]));
  }
}

// ---------------------------------------------------------------------------------------
// LayerNorm (+pos): one wave per token
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

template <typename TS, int EPL>   // TS: token storage type; elements per lane = ceil(E/64)
__global__ __launch_bounds__(256) void layernorm_kernel(const TS* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ pos,
                                                        TS* __restrict__ y, TS* __restrict__ ypos, int T, int E,
                                                        int npos, float eps) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const TS* xr = x + (long long)t * E;
  float v[EPL]; float s = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) { int c = lane + 64 * e; v[e] = c < E ? St<TS>::ld(xr + c) : 0.f; s += v[e]; }
  const float mean = wave_sum(s) / E;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) { int c = lane + 64 * e; float d = c < E ? v[e] - mean : 0.f; q += d * d; }
  const float rstd = 1.f / sqrtf(wave_sum(q) / E + eps);
  const float* pr = pos ? pos + (long long)(t % npos) * E : nullptr;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    int c = lane + 64 * e;
    if (c < E) {
      float o = (v[e] - mean) * rstd * gamma[c] + beta[c];
      St<TS>::st(y + (long long)t * E + c, o);
      if (ypos) St<TS>::st(ypos + (long long)t * E + c, o + pr[c]);
    }
  }
}

// four elements per lane (16 B fp32 / 8 B bf16), LPT = E / 4 lanes per token, 64 / LPT tokens per wave: E = 256 is one token per wave with
// ONE load per lane instead of four, E = 32 packs eight tokens into a wave (the scalar form left half of it idle).  Group reductions by
// xor shuffles below LPT.
template <typename TS, int LPT>
__global__ __launch_bounds__(256) void layernorm4_kernel(const TS* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ pos,
                                                         TS* __restrict__ y, TS* __restrict__ ypos, int T, int E, int npos, float eps) {
  constexpr int TPW = 64 / LPT;
  const int lane = threadIdx.x & 63, sub = lane % LPT;
  const int t = (blockIdx.x * 4 + (threadIdx.x >> 6)) * TPW + lane / LPT;
  const bool live = t < T;
  const int tt = live ? t : T - 1;
  const float4 v = St<TS>::ld4(x + (long long)tt * E + sub * 4);
  float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
  for (int o = 1; o < LPT; o <<= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / E;
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
#pragma unroll
  for (int o = 1; o < LPT; o <<= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.f / sqrtf(q / E + eps);
  if (!live) return;
  const float4 g = *reinterpret_cast<const float4*>(gamma + sub * 4), bq = *reinterpret_cast<const float4*>(beta + sub * 4);
  const float4 o4 = make_float4(dx * rstd * g.x + bq.x, dy * rstd * g.y + bq.y, dz * rstd * g.z + bq.z, dw * rstd * g.w + bq.w);
  St<TS>::st4(y + (long long)t * E + sub * 4, o4);
  if (ypos) {
    const float4 p = *reinterpret_cast<const float4*>(pos + (long long)(t % npos) * E + sub * 4);
    St<TS>::st4(ypos + (long long)t * E + sub * 4, make_float4(o4.x + p.x, o4.y + p.y, o4.z + p.z, o4.w + p.w));
  }
}

// ---------------------------------------------------------------------------------------
// masked row softmax in place: one wave per row, S <= 64*SPL
// ---------------------------------------------------------------------------------------
template <typename TS, int SPL>
__global__ __launch_bounds__(256) void softmax_rows_kernel(TS* __restrict__ s, int ld, long long R, int S, float scale,
                                                           const uint8_t* __restrict__ mask, int rows_per_mask) {
  const int lane = threadIdx.x & 63;
  const long long r = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (r >= R) return;
  TS* row = s + r * ld;
  const uint8_t* mrow = mask ? mask + (r / rows_per_mask) * S : nullptr;
  float v[SPL]; float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < SPL; ++e) {
    int c = lane + 64 * e; float t = -INFINITY;
    if (c < S) { t = St<TS>::ld(row + c) * scale; if (mrow && mrow[c]) t = -INFINITY; }
    v[e] = t; mx = fmaxf(mx, t);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < SPL; ++e) { int c = lane + 64 * e; v[e] = c < S ? expf(v[e] - mx) : 0.f; sum += v[e]; }
  sum = wave_sum(sum);
#pragma unroll
  for (int e = 0; e < SPL; ++e) { int c = lane + 64 * e; if (c < S) St<TS>::st(row + c, v[e] / sum); }
}


// ---------------------------------------------------------------------------------------
// Training: GroupNorm(+swish) backward.  z = act(y), y = xh * gamma + beta, xh = (x - mean_g) * rstd_g.
//   dy = dz * act'(y);  dgamma_c = sum dy * xh;  dbeta_c = sum dy;
//   dx = rstd_g * (gamma_c dy - mean_g(gamma dy) - xh * mean_g(gamma dy xh))       (means over the group's HW * C/G elements)
// pass 1: per (b, chunk, c) {sum dy, sum dy*xh};  pass 2: per (b, group) the two means + per (b, c) totals;  pass 3: dx.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float swish_grad(float y) { const float s = 1.f / (1.f + expf(-y)); return s * (1.f + y * (1.f - s)); }

__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dz, int ldz,
                                                             const float* __restrict__ ss, const float* __restrict__ mr,
                                                             float* __restrict__ part, int HW, int C, int groups, int ppc, int nch, int swish) {
  extern __shared__ float red[];                       // [rows][C][2]
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cq = C >> 2, cpg = C / groups;
  const int tx = threadIdx.x % cq, ty = threadIdx.x / cq, rows = 256 / cq;
  const int p0 = chunk * ppc, p1 = min(p0 + ppc, HW);
  float sc[4], sh[4], mu[4], rs[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = tx * 4 + e, g = c / cpg;
    sc[e] = ss[((long long)b * C + c) * 2]; sh[e] = ss[((long long)b * C + c) * 2 + 1];
    mu[e] = mr[((long long)b * groups + g) * 2]; rs[e] = mr[((long long)b * groups + g) * 2 + 1];
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const float* xb = x + (long long)b * HW * ldx + tx * 4;
  const float* zb = dz + (long long)b * HW * ldz + tx * 4;
  for (int p = p0 + ty; p < p1; p += rows) {
    const float4 xv = *reinterpret_cast<const float4*>(xb + (long long)p * ldx);
    const float4 gv = *reinterpret_cast<const float4*>(zb + (long long)p * ldz);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dy = swish ? gs[e] * swish_grad(xs[e] * sc[e] + sh[e]) : gs[e];
      s1[e] += dy; s2[e] += dy * ((xs[e] - mu[e]) * rs[e]);
    }
  }
  float* r = red + ((ty * C) + tx * 4) * 2;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[2 * e] = s1[e]; r[2 * e + 1] = s2[e]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, q = 0.f;
    for (int t = 0; t < rows; ++t) { a += red[(t * C + c) * 2]; q += red[(t * C + c) * 2 + 1]; }
    float* o = part + (((long long)b * nch + chunk) * C + c) * 2;
    o[0] = a; o[1] = q;
  }
}

// one wave per (b, group): coef[b][g] = {mean_g(gamma dy), mean_g(gamma dy xh)}; bc[b][c] = {sum dy, sum dy xh}
__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                             float* __restrict__ coef, float* __restrict__ bc, int HW, int C, int groups, int nch) {
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
  const int cpg = C / groups;                        // a power of two <= 32 (C a power of two <= 1024, 32 groups)
  // lane = (channel j of the group, chunk slice kq): the 64 lanes walk the chunk partials interleaved and finish with a fixed butterfly --
  // with one lane per channel a C = 64 layer kept 2 lanes busy for 256 sequential chunks (23 us per launch)
  const int lane = threadIdx.x, j = lane & (cpg - 1), kq = lane / cpg, KQ = 64 / cpg;
  const int c = g * cpg + j;
  double a = 0.0, q = 0.0;
  for (int k = kq; k < nch; k += KQ) { const float* pp = part + (((long long)b * nch + k) * C + c) * 2; a += pp[0]; q += pp[1]; }
  for (int o = cpg; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }
  double A = 0.0, Q = 0.0;
  if (kq == 0) {
    bc[((long long)b * C + c) * 2] = (float)a; bc[((long long)b * C + c) * 2 + 1] = (float)q;
    A = a * gamma[c]; Q = q * gamma[c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { A += __shfl_xor(A, o, 64); Q += __shfl_xor(Q, o, 64); }
  if (threadIdx.x == 0) {
    const double n = (double)HW * cpg;
    coef[(long long)blockIdx.x * 2] = (float)(A / n); coef[(long long)blockIdx.x * 2 + 1] = (float)(Q / n);
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dz, int ldz,
                                                           float* __restrict__ dx, int ldo, const float* __restrict__ ss,
                                                           const float* __restrict__ mr, const float* __restrict__ coef,
                                                           const float* __restrict__ gamma, long long total4, int HW, int C, int groups, int swish) {
  const int cq = C >> 2, cpg = C / groups;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % cq); const long long pix = i / cq; const int b = (int)(pix / HW);
    const float4 xv = *reinterpret_cast<const float4*>(x + pix * ldx + c4 * 4);
    const float4 gv = *reinterpret_cast<const float4*>(dz + pix * ldz + c4 * 4);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c4 * 4 + e, g = c / cpg;
      const float sc = ss[((long long)b * C + c) * 2], sh = ss[((long long)b * C + c) * 2 + 1];
      const float mu = mr[((long long)b * groups + g) * 2], rs = mr[((long long)b * groups + g) * 2 + 1];
      const float dy = swish ? gs[e] * swish_grad(xs[e] * sc + sh) : gs[e];
      const float xh = (xs[e] - mu) * rs;
      o[e] = rs * (gamma[c] * dy - coef[((long long)b * groups + g) * 2] - xh * coef[((long long)b * groups + g) * 2 + 1]);
    }
    *reinterpret_cast<float4*>(dx + pix * ldo + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// dgamma[c] += sum_b bc[b][c][1]; dbeta[c] += sum_b bc[b][c][0]
__global__ __launch_bounds__(256) void gn_bwd_params_kernel(const float* __restrict__ bc, float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, q = 0.f;
  for (int b = 0; b < B; ++b) { a += bc[((long long)b * C + c) * 2]; q += bc[((long long)b * C + c) * 2 + 1]; }
  dgamma[c] += q; dbeta[c] += a;
}

// ---------------------------------------------------------------------------------------
// LayerNorm(+pos) backward: one wave per token; dy = g_y + g_ypos; dx as above with the token as the group;
// per-block partial {dbeta, dgamma} rows for a deterministic second stage (smx_partial_reduce_f32)
// ---------------------------------------------------------------------------------------
template <int EPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ gy, const float* __restrict__ gyp,
                                                            float* __restrict__ dx, float* __restrict__ part, int T, int E, float eps, int tpw) {
  __shared__ float red[4][2][64 * EPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dgm[EPL], dbt[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) { dgm[e] = 0.f; dbt[e] = 0.f; }
  for (int k = 0; k < tpw; ++k) {
    const int t = (blockIdx.x * 4 + wave) * tpw + k;
    if (t >= T) break;
    const float* xr = x + (long long)t * E;
    float v[EPL], d[EPL]; float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { const int c = lane + 64 * e; v[e] = c < E ? xr[c] : 0.f; s += v[e]; }
    const float mean = wave_sum(s) / E;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { const int c = lane + 64 * e; const float dd = c < E ? v[e] - mean : 0.f; q += dd * dd; }
    const float rstd = 1.f / sqrtf(wave_sum(q) / E + eps);
    float a = 0.f, bq = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int c = lane + 64 * e;
      float dy = 0.f;
      if (c < E) { if (gy) dy += gy[(long long)t * E + c]; if (gyp) dy += gyp[(long long)t * E + c]; }
      const float xh = c < E ? (v[e] - mean) * rstd : 0.f;
      d[e] = dy; v[e] = xh;
      const float gd = c < E ? gamma[c] * dy : 0.f;
      a += gd; bq += gd * xh;
      dgm[e] += dy * xh; dbt[e] += dy;
    }
    a = wave_sum(a) / E; bq = wave_sum(bq) / E;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int c = lane + 64 * e;
      if (c < E) dx[(long long)t * E + c] = rstd * (gamma[c] * d[e] - a - v[e] * bq);
    }
  }
#pragma unroll
  for (int e = 0; e < EPL; ++e) { red[wave][0][lane + 64 * e] = dbt[e]; red[wave][1][lane + 64 * e] = dgm[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * E; i += 256) {
    const int which = i / E, c = i - which * E;
    part[(long long)blockIdx.x * 2 * E + i] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

// dpos[n][c] += sum_b g[b][n][c]   (the position embedding is added to every sample's tokens)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ g, float* __restrict__ out, int B, long long per) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < per; i += (long long)gridDim.x * 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += g[(long long)b * per + i];
    out[i] += s;
  }
}

// softmax backward in place on dP: dS = scale * P * (dP - sum_j dP P); one wave per row (AttnBlock, scores materialised)
template <int SPL>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, long long R, int S, float scale) {
  const int lane = threadIdx.x & 63;
  const long long r = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* pr = P + r * S; float* dr = dP + r * S;
  float pv[SPL], dv[SPL]; float s = 0.f;
#pragma unroll
  for (int e = 0; e < SPL; ++e) { const int c = lane + 64 * e; pv[e] = c < S ? pr[c] : 0.f; dv[e] = c < S ? dr[c] : 0.f; s += pv[e] * dv[e]; }
  s = wave_sum(s);
#pragma unroll
  for (int e = 0; e < SPL; ++e) { const int c = lane + 64 * e; if (c < S) dr[c] = scale * pv[e] * (dv[e] - s); }
}


// ws[blk][0:E] = dbeta partials, ws[blk][E:2E] = dgamma partials -> accumulate into dbeta / dgamma (fixed order over blocks)
// 32 columns per block, 8 lanes per column walking the row blocks interleaved, fixed-order finish through LDS (one thread per column
// walked all T / 16 row blocks alone: 60 us for 512 columns)
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ ws, int nblk, int E, float* __restrict__ dbeta, float* __restrict__ dgamma) {
  __shared__ float red[8][32];
  const int l = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + l;
  float s = 0.f;
  if (i < 2 * E)
    for (int k = q; k < nblk; k += 8) s += ws[(long long)k * 2 * E + i];
  red[q][l] = s;
  __syncthreads();
  if (q == 0 && i < 2 * E) {
    s = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) + ((red[4][l] + red[5][l]) + (red[6][l] + red[7][l]));
    if (i < E) dbeta[i] += s; else dgamma[i - E] += s;
  }
}

}  // namespace

extern "C" int64_t smx_groupnorm_ws_floats(int B, int HW, int C) {
  int ppc = HW / 64; if (ppc < 32) ppc = 32; if (ppc > 256) ppc = 256;
  const int nch = (HW + ppc - 1) / ppc;
  return (int64_t)B * nch * C * 2 + (int64_t)B * C * 2;
}

namespace {
template <typename T>
int gn_swish_launch(const T* x, int ldx, const float* gamma, const float* beta, T* y, int ldy, int B, int HW, int C, int groups,
                    float eps, int swish, float* ws, void* stream) {
  if (!x || !y || !gamma || !beta || !ws || B <= 0 || HW <= 0) return SMX_EINVAL;
  if (C < 4 || C > 1024 || (C & (C - 1)) != 0 || C % groups != 0 || ldx % 4 != 0 || ldy % 4 != 0 || ldx < C || ldy < C) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int ppc = HW / 64; if (ppc < 32) ppc = 32; if (ppc > 256) ppc = 256;
  const int nch = (HW + ppc - 1) / ppc;
  float* part = ws; float* ss = ws + (int64_t)B * nch * C * 2;
  const int rows = 256 / (C / 4);
  SMX_LAUNCH(gn_partial_kernel<T>, dim3(nch, B), dim3(256), (size_t)rows * C * 3 * sizeof(float), st, x, ldx, part, HW, C, ppc, nch);
  SMX_LAUNCH(gn_finalize_kernel, dim3(B * groups), dim3(64), 0, st, part, gamma, beta, ss, HW, C, groups, nch, ppc, eps);
  const long long total4 = (long long)B * HW * (C / 4);
  int blocks = smx_cdiv(total4, 256); if (blocks > 8192) blocks = 8192;
  SMX_LAUNCH((gn_apply_kernel<T, T>), dim3(blocks), dim3(256), 0, st, x, ldx, y, ldy, ss, total4, HW, C, swish);
  return smx_launch_status();
}

template <typename T>
int gn_stats_launch(const T* x, int ldx, const float* gamma, const float* beta, float* ss, int B, int HW, int C, int groups, float eps,
                    float* ws, void* stream) {
  if (!x || !ss || !gamma || !beta || !ws || B <= 0 || HW <= 0) return SMX_EINVAL;
  if (C < 4 || C > 1024 || (C & (C - 1)) != 0 || C % groups != 0 || ldx % 4 != 0 || ldx < C) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int ppc = HW / 64; if (ppc < 32) ppc = 32; if (ppc > 256) ppc = 256;
  const int nch = (HW + ppc - 1) / ppc;
  const int rows = 256 / (C / 4);
  SMX_LAUNCH(gn_partial_kernel<T>, dim3(nch, B), dim3(256), (size_t)rows * C * 3 * sizeof(float), st, x, ldx, ws, HW, C, ppc, nch);
  SMX_LAUNCH(gn_finalize_kernel, dim3(B * groups), dim3(64), 0, st, ws, gamma, beta, ss, HW, C, groups, nch, ppc, eps);
  return smx_launch_status();
}

template <typename T>
int gn_apply_launch(const T* x, int ldx, const float* ss, T* y, int ldy, int B, int HW, int C, int swish, void* stream) {
  if (!x || !ss || !y || B <= 0 || HW <= 0 || C < 4 || C % 4 != 0 || ldx % 4 != 0 || ldy % 4 != 0 || ldx < C || ldy < C) return SMX_EINVAL;
  const long long total4 = (long long)B * HW * (C / 4);
  int blocks = smx_cdiv(total4, 256); if (blocks > 8192) blocks = 8192;
  SMX_LAUNCH((gn_apply_kernel<T, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, ss, total4, HW, C, swish);
  return smx_launch_status();
}

template <typename T>
int layernorm_launch(const T* x, const float* gamma, const float* beta, const float* pos, T* y, T* y_pos, int Tn, int E, int npos,
                     float eps, void* stream) {
  if (!x || !y || !gamma || !beta || Tn <= 0 || E <= 0 || E > 512 || (y_pos && (!pos || npos <= 0))) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(smx_cdiv(Tn, 4)), block(256);
  const bool al = ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)y_pos)) & (4 * sizeof(T) - 1)) == 0 && ((((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)pos)) & 15) == 0;
  if (al && E == 256) { SMX_LAUNCH((layernorm4_kernel<T, 64>), grid, block, 0, st, x, gamma, beta, pos, y, y_pos, Tn, E, npos, eps); return smx_launch_status(); }
  if (al && E == 32) { SMX_LAUNCH((layernorm4_kernel<T, 8>), dim3(smx_cdiv(Tn, 32)), block, 0, st, x, gamma, beta, pos, y, y_pos, Tn, E, npos, eps); return smx_launch_status(); }
  if (E <= 64) SMX_LAUNCH((layernorm_kernel<T, 1>), grid, block, 0, st, x, gamma, beta, pos, y, y_pos, Tn, E, npos, eps);
  else if (E <= 256) SMX_LAUNCH((layernorm_kernel<T, 4>), grid, block, 0, st, x, gamma, beta, pos, y, y_pos, Tn, E, npos, eps);
  else SMX_LAUNCH((layernorm_kernel<T, 8>), grid, block, 0, st, x, gamma, beta, pos, y, y_pos, Tn, E, npos, eps);
  return smx_launch_status();
}

template <typename T>
int softmax_launch(T* s, int ld, int R, int S, float scale, const uint8_t* mask, int rows_per_mask, void* stream) {
  if (!s || R <= 0 || S <= 0 || S > 4096 || ld < S || (mask && rows_per_mask <= 0)) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(smx_cdiv(R, 4)), block(256);
  if (S <= 256) SMX_LAUNCH((softmax_rows_kernel<T, 4>), grid, block, 0, st, s, ld, (long long)R, S, scale, mask, rows_per_mask);
  else if (S <= 512) SMX_LAUNCH((softmax_rows_kernel<T, 8>), grid, block, 0, st, s, ld, (long long)R, S, scale, mask, rows_per_mask);
  else if (S <= 1024) SMX_LAUNCH((softmax_rows_kernel<T, 16>), grid, block, 0, st, s, ld, (long long)R, S, scale, mask, rows_per_mask);
  else if (S <= 2048) SMX_LAUNCH((softmax_rows_kernel<T, 32>), grid, block, 0, st, s, ld, (long long)R, S, scale, mask, rows_per_mask);
  else SMX_LAUNCH((softmax_rows_kernel<T, 64>), grid, block, 0, st, s, ld, (long long)R, S, scale, mask, rows_per_mask);   // 64x64 tokens: the 512 variant's AttnBlock
  return smx_launch_status();
}
}  // namespace

extern "C" int smx_groupnorm_swish_nhwc_f32(const float* x, int ldx, const float* gamma, const float* beta,
                                            float* y, int ldy, int B, int HW, int C, int groups, float eps,
                                            int swish, float* ws, void* stream) {
  return gn_swish_launch<float>(x, ldx, gamma, beta, y, ldy, B, HW, C, groups, eps, swish, ws, stream);
}
extern "C" int smx_groupnorm_swish_nhwc_bf16(const void* x, int ldx, const float* gamma, const float* beta,
                                             void* y, int ldy, int B, int HW, int C, int groups, float eps,
                                             int swish, float* ws, void* stream) {
  return gn_swish_launch<bf16_t>((const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, B, HW, C, groups, eps, swish, ws, stream);
}

extern "C" int smx_groupnorm_stats_f32(const float* x, int ldx, const float* gamma, const float* beta, float* ss,
                                       int B, int HW, int C, int groups, float eps, float* ws, void* stream) {
  return gn_stats_launch<float>(x, ldx, gamma, beta, ss, B, HW, C, groups, eps, ws, stream);
}
extern "C" int smx_groupnorm_stats_bf16(const void* x, int ldx, const float* gamma, const float* beta, float* ss,
                                        int B, int HW, int C, int groups, float eps, float* ws, void* stream) {
  return gn_stats_launch<bf16_t>((const bf16_t*)x, ldx, gamma, beta, ss, B, HW, C, groups, eps, ws, stream);
}

extern "C" int smx_groupnorm_finalize_f32(const float* part, const float* gamma, const float* beta, float* ss,
                                          int B, int HW, int C, int groups, int nch, float eps, void* stream) {
  if (!part || !gamma || !beta || !ss || B <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups != 0 || nch <= 0 || HW % nch != 0) return SMX_EINVAL;
  const int ppc = HW / nch;                            // equal chunks (the Winograd epilogue: 8x16-pixel blocks)
  SMX_LAUNCH(gn_finalize_kernel, dim3(B * groups), dim3(64), 0, (hipStream_t)stream, part, gamma, beta, ss, HW, C, groups, nch, ppc, eps);
  return smx_launch_status();
}

extern "C" int smx_groupnorm_apply_f32(const float* x, int ldx, const float* ss, float* y, int ldy, int B, int HW, int C,
                                       int swish, void* stream) {
  return gn_apply_launch<float>(x, ldx, ss, y, ldy, B, HW, C, swish, stream);
}
extern "C" int smx_groupnorm_apply_bf16(const void* x, int ldx, const float* ss, void* y, int ldy, int B, int HW, int C,
                                        int swish, void* stream) {
  return gn_apply_launch<bf16_t>((const bf16_t*)x, ldx, ss, (bf16_t*)y, ldy, B, HW, C, swish, stream);
}

extern "C" int smx_layernorm_pos_f32(const float* x, const float* gamma, const float* beta, const float* pos,
                                     float* y, float* y_pos, int T, int E, int npos, float eps, void* stream) {
  return layernorm_launch<float>(x, gamma, beta, pos, y, y_pos, T, E, npos, eps, stream);
}
extern "C" int smx_layernorm_pos_bf16(const void* x, const float* gamma, const float* beta, const float* pos,
                                      void* y, void* y_pos, int T, int E, int npos, float eps, void* stream) {
  return layernorm_launch<bf16_t>((const bf16_t*)x, gamma, beta, pos, (bf16_t*)y, (bf16_t*)y_pos, T, E, npos, eps, stream);
}

extern "C" int smx_softmax_rows_f32(float* s, int ld, int R, int S, float scale, const uint8_t* mask,
                                    int rows_per_mask, void* stream) {
  return softmax_launch<float>(s, ld, R, S, scale, mask, rows_per_mask, stream);
}
extern "C" int smx_softmax_rows_bf16(void* s, int ld, int R, int S, float scale, const uint8_t* mask,
                                     int rows_per_mask, void* stream) {
  return softmax_launch<bf16_t>((bf16_t*)s, ld, R, S, scale, mask, rows_per_mask, stream);
}


/* ---- training entry points (GroupNorm / LayerNorm / softmax backward) ------------------------------------------------ */
extern "C" int smx_groupnorm_stats_train_f32(const float* x, int ldx, const float* gamma, const float* beta, float* ss, float* mr,
                                             int B, int HW, int C, int groups, float eps, float* ws, void* stream) {
  if (!x || !ss || !mr || !gamma || !beta || !ws || B <= 0 || HW <= 0) return SMX_EINVAL;
  if (C < 4 || C > 1024 || (C & (C - 1)) != 0 || C % groups != 0 || ldx % 4 != 0 || ldx < C) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int ppc = HW / 64; if (ppc < 32) ppc = 32; if (ppc > 256) ppc = 256;
  const int nch = (HW + ppc - 1) / ppc;
  const int rows = 256 / (C / 4);
  SMX_LAUNCH(gn_partial_kernel<float>, dim3(nch, B), dim3(256), (size_t)rows * C * 3 * sizeof(float), st, x, ldx, ws, HW, C, ppc, nch);
  SMX_LAUNCH(gn_finalize_kernel, dim3(B * groups), dim3(64), 0, st, ws, gamma, beta, ss, HW, C, groups, nch, ppc, eps, mr);
  return smx_launch_status();
}

/* ws: smx_groupnorm_ws_floats(B, HW, C) + 2*B*groups floats.  dgamma / dbeta are ACCUMULATED into. */
extern "C" int smx_groupnorm_bwd_f32(const float* x, int ldx, const float* dz, int ldz, const float* ss, const float* mr,
                                     const float* gamma, float* dx, int ldo, float* dgamma, float* dbeta,
                                     int B, int HW, int C, int groups, int swish, float* ws, void* stream) {
  if (!x || !dz || !ss || !mr || !gamma || !dx || !dgamma || !dbeta || !ws || B <= 0 || HW <= 0) return SMX_EINVAL;
  if (C < 4 || C > 1024 || (C & (C - 1)) != 0 || C % groups != 0 || ldx % 4 || ldz % 4 || ldo % 4 || ldx < C || ldz < C || ldo < C) return SMX_EINVAL;
  if ((((uintptr_t)x) | ((uintptr_t)dz) | ((uintptr_t)dx)) & 15) return SMX_EINVAL;
  { const int cpg = C / groups; if (cpg > 64 || (cpg & (cpg - 1)) != 0) return SMX_EINVAL; }     // the finalize kernel's (channel, chunk slice) lane map
  hipStream_t st = (hipStream_t)stream;
  int ppc = HW / 64; if (ppc < 32) ppc = 32; if (ppc > 256) ppc = 256;
  const int nch = (HW + ppc - 1) / ppc;
  const int rows = 256 / (C / 4);
  float* part = ws; float* bc = ws + (int64_t)B * nch * C * 2; float* coef = bc + (int64_t)B * C * 2;
  SMX_LAUNCH(gn_bwd_partial_kernel, dim3(nch, B), dim3(256), (size_t)rows * C * 2 * sizeof(float), st, x, ldx, dz, ldz, ss, mr, part, HW, C, groups, ppc, nch, swish);
  SMX_LAUNCH(gn_bwd_finalize_kernel, dim3(B * groups), dim3(64), 0, st, part, gamma, coef, bc, HW, C, groups, nch);
  const long long total4 = (long long)B * HW * (C / 4);
  int blocks = smx_cdiv(total4, 256); if (blocks > 8192) blocks = 8192;
  SMX_LAUNCH(gn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, x, ldx, dz, ldz, dx, ldo, ss, mr, coef, gamma, total4, HW, C, groups, swish);
  SMX_LAUNCH(gn_bwd_params_kernel, dim3(smx_cdiv(C, 256)), dim3(256), 0, st, bc, dgamma, dbeta, B, C);
  return smx_launch_status();
}

extern "C" int64_t smx_layernorm_bwd_ws_floats(int T, int E) { return (int64_t)smx_cdiv(T, 16) * 2 * E; }

/* gy / gypos: gradients of LN(x) and of LN(x)+pos (either may be null); dgamma / dbeta / dpos are ACCUMULATED into (dpos may be null). */
extern "C" int smx_layernorm_bwd_f32(const float* x, const float* gamma, const float* gy, const float* gypos, float* dx,
                                     float* dgamma, float* dbeta, float* dpos, int T, int E, int npos, float eps, float* ws, void* stream) {
  if (!x || !gamma || (!gy && !gypos) || !dx || !dgamma || !dbeta || !ws || T <= 0 || E <= 0 || E > 512) return SMX_EINVAL;
  if (dpos && (!gypos || npos <= 0 || T % npos != 0)) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = smx_cdiv(T, 16);
  if (E <= 64) SMX_LAUNCH(layernorm_bwd_kernel<1>, dim3(nblk), dim3(256), 0, st, x, gamma, gy, gypos, dx, ws, T, E, eps, 4);
  else if (E <= 256) SMX_LAUNCH(layernorm_bwd_kernel<4>, dim3(nblk), dim3(256), 0, st, x, gamma, gy, gypos, dx, ws, T, E, eps, 4);
  else SMX_LAUNCH(layernorm_bwd_kernel<8>, dim3(nblk), dim3(256), 0, st, x, gamma, gy, gypos, dx, ws, T, E, eps, 4);
  // ws rows are [dbeta(E) | dgamma(E)]: two strided column reductions in a fixed order
  SMX_LAUNCH(ln_param_reduce_kernel, dim3(smx_cdiv(2 * E, 32)), dim3(256), 0, st, ws, nblk, E, dbeta, dgamma);
  if (dpos) {
    const long long per = (long long)npos * E;
    int blocks = smx_cdiv(per, 256); if (blocks > 4096) blocks = 4096;
    SMX_LAUNCH(batch_sum_kernel, dim3(blocks), dim3(256), 0, st, gypos, dpos, T / npos, per);
  }
  return smx_launch_status();
}

extern "C" int smx_batch_sum_f32(const float* g, float* out, int B, int64_t per, void* stream) {
  if (!g || !out || B <= 0 || per <= 0) return SMX_EINVAL;
  int blocks = smx_cdiv(per, 256); if (blocks > 4096) blocks = 4096;
  SMX_LAUNCH(batch_sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, out, B, (long long)per);
  return smx_launch_status();
}

extern "C" int smx_softmax_rows_bwd_f32(const float* P, float* dP, int64_t R, int S, float scale, void* stream) {
  if (!P || !dP || R <= 0 || S <= 0 || S > 1024) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(smx_cdiv(R, 4)), block(256);
  if (S <= 256) SMX_LAUNCH(softmax_bwd_kernel<4>, grid, block, 0, st, P, dP, (long long)R, S, scale);
  else if (S <= 512) SMX_LAUNCH(softmax_bwd_kernel<8>, grid, block, 0, st, P, dP, (long long)R, S, scale);
  else SMX_LAUNCH(softmax_bwd_kernel<16>, grid, block, 0, st, P, dP, (long long)R, S, scale);
  return smx_launch_status();
}
