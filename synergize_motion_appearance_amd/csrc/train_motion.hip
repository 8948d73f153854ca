// Training: the motion estimator's training-mode pieces for gfx950 -- BatchNorm with BATCH statistics
// (`motion_estimator.train()`, models/appmotioncomp_model.py:162; sync_batchnorm/batchnorm.py:48-53 falls through to
// F.batch_norm(training=True)) fused with the ReLU that always follows it in the hourglass blocks
// (utils/motion_estimator_util.py:214-231, 363-380), its backward, and the backward of the keypoint head
// (archs/keypoint_detector_arch.py:48-86), of the fused heatmap / sparse-motion / sparse-warp stage
// (archs/dense_motion_arch.py:65-116) and of the mask-softmax flow blend (:140-158).
// NHWC fp32; per-channel reductions are two deterministic stages (per-chunk partials, then a fixed-order finish in double).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "smx.h"
#include "smx_common.h"

namespace {

inline int grid_for(long long total) { int g = smx_cdiv(total, 256); return g > 16384 ? 16384 : (g < 1 ? 1 : g); }

// ---- per-channel pair sums over rows: part[chunk][C][2] = { sum a, sum b } with (a, b) from a functor --------------------
// MODE 0: (x, x^2)                                              (BatchNorm statistics)
// MODE 1: (dy', dy' * xh), dy' = g * (y > 0), xh = (x - mean) rstd  (BatchNorm+ReLU backward)
template <int MODE>
__global__ __launch_bounds__(256) void chan_pair_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg,
                                                                const float* __restrict__ y, int ldy, const float* __restrict__ mr,
                                                                long long P, int C, int rows_per_chunk, float* __restrict__ part) {
  __shared__ float red[256 * 2];
  int cw = 1; while (cw < C && cw < 256) cw <<= 1;
  const int tx = threadIdx.x % cw, ty = threadIdx.x / cw, rpar = 256 / cw;
  const long long r_begin = (long long)blockIdx.x * rows_per_chunk, r_end = min(P, r_begin + rows_per_chunk);
  for (int c0 = 0; c0 < C; c0 += cw) {
    const int c = c0 + tx;
    float a = 0.f, b = 0.f;
    if (c < C) {
      const float mu = MODE ? mr[2 * c] : 0.f, rs = MODE ? mr[2 * c + 1] : 0.f;
      for (long long r = r_begin + ty; r < r_end; r += rpar) {
        const float xv = x[r * ldx + c];
        if (MODE == 0) { a += xv; b += xv * xv; }
        else { const float d = (!y || y[r * ldy + c] > 0.f) ? g[r * ldg + c] : 0.f; a += d; b += d * (xv - mu) * rs; }   // y null: no ReLU behind the BatchNorm
      }
    }
    __syncthreads();
    red[threadIdx.x * 2] = a; red[threadIdx.x * 2 + 1] = b;
    __syncthreads();
    if (ty == 0 && c < C) {
      float sa = 0.f, sb = 0.f;
      for (int k = 0; k < rpar; ++k) { sa += red[(k * cw + tx) * 2]; sb += red[(k * cw + tx) * 2 + 1]; }
      part[((long long)blockIdx.x * C + c) * 2] = sa; part[((long long)blockIdx.x * C + c) * 2 + 1] = sb;
    }
  }
}

// float4 form of the pair sums: 16 lanes x 4 channels = a 64-channel column block (blockIdx.y), 16 rows in flight per block, rows split into
// chunks along blockIdx.x -- the scalar kernel above ran P / 256 blocks of 4-byte loads (64 blocks for a 64 x 64 x B=4 hourglass level: 140 us
// for 16 MB).  Same partial layout [chunk][C][2]: the finish kernels do not change.
template <int MODE>
__global__ __launch_bounds__(256) void chan_pair_partial_v4_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg,
                                                                   const float* __restrict__ y, int ldy, const float* __restrict__ mr,
                                                                   long long P, int C, int rows_per_chunk, float* __restrict__ part) {
  __shared__ float4 red[2][16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = (blockIdx.y * 16 + tx) * 4;
  const long long r_begin = (long long)blockIdx.x * rows_per_chunk, r_end = min(P, r_begin + rows_per_chunk);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (c < C) {
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { mu[e] = mr[2 * (c + e)]; rs[e] = mr[2 * (c + e) + 1]; }
    }
    for (long long r = r_begin + ty; r < r_end; r += 16) {
      const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + c);
      if (MODE == 0) {
        a.x += xv.x; a.y += xv.y; a.z += xv.z; a.w += xv.w;
        b.x += xv.x * xv.x; b.y += xv.y * xv.y; b.z += xv.z * xv.z; b.w += xv.w * xv.w;
      } else {
        const float4 yv = y ? *reinterpret_cast<const float4*>(y + r * ldy + c) : make_float4(1.f, 1.f, 1.f, 1.f);   // y null: no ReLU gate
        const float4 gv = *reinterpret_cast<const float4*>(g + r * ldg + c);
        const float d0 = yv.x > 0.f ? gv.x : 0.f, d1 = yv.y > 0.f ? gv.y : 0.f, d2 = yv.z > 0.f ? gv.z : 0.f, d3 = yv.w > 0.f ? gv.w : 0.f;
        a.x += d0; a.y += d1; a.z += d2; a.w += d3;
        b.x += d0 * (xv.x - mu[0]) * rs[0]; b.y += d1 * (xv.y - mu[1]) * rs[1]; b.z += d2 * (xv.z - mu[2]) * rs[2]; b.w += d3 * (xv.w - mu[3]) * rs[3];
      }
    }
  }
  red[0][ty][tx] = a; red[1][ty][tx] = b;
  __syncthreads();
  if (ty == 0 && c < C) {
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
    for (int k = 0; k < 16; ++k) {                          // fixed order
      const float4 u = red[0][k][tx], v = red[1][k][tx];
      sa.x += u.x; sa.y += u.y; sa.z += u.z; sa.w += u.w; sb.x += v.x; sb.y += v.y; sb.z += v.z; sb.w += v.w;
    }
    float* o = part + ((long long)blockIdx.x * C + c) * 2;
    o[0] = sa.x; o[1] = sb.x; o[2] = sa.y; o[3] = sb.y; o[4] = sa.z; o[5] = sb.z; o[6] = sa.w; o[7] = sb.w;
  }
}

// BatchNorm statistics finish: mean, biased variance -> mr[c] = {mean, rstd}; running stats with momentum (unbiased variance), like
// F.batch_norm(training=True)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nchunk, int C, long long n, float eps, float momentum,
                                                          float* __restrict__ mr, float* __restrict__ run_mean, float* __restrict__ run_var) {
  // 32 channels per block, 8 lanes per channel walking the (up to 1024) chunk partials interleaved, fixed-order finish in double through
  // LDS: one thread per channel walked them alone (66 us per launch)
  __shared__ double red[2][8][32];
  const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + l;
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int k = g; k < nchunk; k += 8) { s += part[((long long)k * C + c) * 2]; q += part[((long long)k * C + c) * 2 + 1]; }
  red[0][g][l] = s; red[1][g][l] = q;
  __syncthreads();
  if (g != 0 || c >= C) return;
  s = ((red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l])) + ((red[0][4][l] + red[0][5][l]) + (red[0][6][l] + red[0][7][l]));
  q = ((red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l])) + ((red[1][4][l] + red[1][5][l]) + (red[1][6][l] + red[1][7][l]));
  const double mean = s / (double)n;
  double var = q / (double)n - mean * mean; if (var < 0.0) var = 0.0;
  mr[2 * c] = (float)mean; mr[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
  if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(n > 1 ? var * (double)n / (double)(n - 1) : var);
}

__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, const float* __restrict__ mr,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, long long P, int C, int relu) {
  const long long total = P * C;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C; const int c = (int)(i - r * C);
    float v = (x[r * ldx + c] - mr[2 * c]) * mr[2 * c + 1] * gamma[c] + beta[c];
    if (relu && v < 0.f) v = 0.f;
    y[r * ldy + c] = v;
  }
}

// backward finish: sums[c] = {sum dy', sum dy' xh}; dgamma += sum dy' xh; dbeta += sum dy'
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nchunk, int C, float* __restrict__ sums,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double red[2][8][32];
  const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + l;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int k = g; k < nchunk; k += 8) { a += part[((long long)k * C + c) * 2]; b += part[((long long)k * C + c) * 2 + 1]; }
  red[0][g][l] = a; red[1][g][l] = b;
  __syncthreads();
  if (g != 0 || c >= C) return;
  a = ((red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l])) + ((red[0][4][l] + red[0][5][l]) + (red[0][6][l] + red[0][7][l]));
  b = ((red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l])) + ((red[1][4][l] + red[1][5][l]) + (red[1][6][l] + red[1][7][l]));
  sums[2 * c] = (float)a; sums[2 * c + 1] = (float)b;
  dgamma[c] += (float)b; dbeta[c] += (float)a;
}

__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg,
                                                                const float* __restrict__ y, int ldy, const float* __restrict__ mr,
                                                                const float* __restrict__ sums, const float* __restrict__ gamma,
                                                                float* __restrict__ dx, int ldo, long long P, int C) {
  const long long total = P * C;
  const float inv_n = 1.f / (float)P;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C; const int c = (int)(i - r * C);
    const float d = (!y || y[r * ldy + c] > 0.f) ? g[r * ldg + c] : 0.f;
    const float xh = (x[r * ldx + c] - mr[2 * c]) * mr[2 * c + 1];
    dx[r * ldo + c] = mr[2 * c + 1] * gamma[c] * (d - sums[2 * c] * inv_n - xh * sums[2 * c + 1] * inv_n);
  }
}

// ---- 2x2 average pool backward -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ g, int ldg, float* __restrict__ dx, int ldx, long long total, int H, int W, int C) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long p = i / C;
    const int x = (int)(p % W); p /= W; const int y = (int)(p % H); const int b = (int)(p / H);
    dx[(((long long)b * H + y) * W + x) * ldx + c] = 0.25f * g[(((long long)b * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) * ldg + c];
  }
}

// ---- A3 backward: value_k = sum_p h_kp grid_p, jac_kj = sum_p h_kp J_kjp, h = softmax(l / T) ---------------------------------
//   dh_kp = dvalue_k . grid_p + sum_j djac_kj J_kjp;  dl_kp = h_kp (dh_kp - sum_q h_kq dh_kq) / T;  dJ_kjp = djac_kj h_kp
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void kp_head_bwd_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ jm, int ldj,
                                                          const float* __restrict__ dvalue, const float* __restrict__ djac,
                                                          float* __restrict__ dlogits, int lddl, float* __restrict__ djm, int lddj,
                                                          int H, int W, int K, float temperature) {
  __shared__ float red[4];
  const int b = blockIdx.x / K, k = blockIdx.x % K, HW = H * W;
  const float* lb = logits + (long long)b * HW * ldl + k;
  const float* jb = jm + (long long)b * HW * ldj + 4 * k;
  const float gvx = dvalue ? dvalue[(b * K + k) * 2] : 0.f, gvy = dvalue ? dvalue[(b * K + k) * 2 + 1] : 0.f;
  float gj[4] = {0.f, 0.f, 0.f, 0.f};
  if (djac) for (int j = 0; j < 4; ++j) gj[j] = djac[(b * K + k) * 4 + j];
  float mx = -INFINITY;
  for (int p = threadIdx.x; p < HW; p += 256) mx = fmaxf(mx, lb[(long long)p * ldl] / temperature);
  mx = block_max(mx, red);
  float s = 0.f, sd = 0.f;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const float e = expf(lb[(long long)p * ldl] / temperature - mx);
    const int y = p / W, x = p - y * W;
    const float gx = 2.f * ((float)x / (float)(W - 1)) - 1.f, gy = 2.f * ((float)y / (float)(H - 1)) - 1.f;
    const float4 j = *reinterpret_cast<const float4*>(jb + (long long)p * ldj);
    const float dh = gvx * gx + gvy * gy + gj[0] * j.x + gj[1] * j.y + gj[2] * j.z + gj[3] * j.w;
    s += e; sd += e * dh;
  }
  s = block_sum(s, red); sd = block_sum(sd, red);
  const float mean_dh = sd / s;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const float h = expf(lb[(long long)p * ldl] / temperature - mx) / s;
    const int y = p / W, x = p - y * W;
    const float gx = 2.f * ((float)x / (float)(W - 1)) - 1.f, gy = 2.f * ((float)y / (float)(H - 1)) - 1.f;
    const float4 j = *reinterpret_cast<const float4*>(jb + (long long)p * ldj);
    const float dh = gvx * gx + gvy * gy + gj[0] * j.x + gj[1] * j.y + gj[2] * j.z + gj[3] * j.w;
    dlogits[((long long)b * HW + p) * lddl + k] = h * (dh - mean_dh) / temperature;
    *reinterpret_cast<float4*>(djm + ((long long)b * HW + p) * lddj + 4 * k) = make_float4(gj[0] * h, gj[1] * h, gj[2] * h, gj[3] * h);
  }
}

// ---- A4-A6 backward: one block per (sample, keypoint); reduces over the H*W grid to the 12 keypoint gradients ---------------------
// forward per grid point z (k >= 1): heat = G(z; kp_d) - G(z; kp_s), G(z; m) = exp(-|z-m|^2 / (2 var));  drv_heat = G(z; kp_d)
//   T(z) = J (z - kp_d) + kp_s, J = J_s inv(J_d);  deformed = bilinear(src, T(z)) (align_corners=False, zeros)
// upstream: g_hg[b,y,x,4k+{0,1,2,3}] (heat, r, g, b), g_sparse[b,k,y,x,2] (T), g_dheat[b,y,x,k-1]
__global__ __launch_bounds__(256) void sparse_motion_bwd_kernel(const float* __restrict__ src, long long src_bs,
                                                                const float* __restrict__ kdv, const float* __restrict__ kdj,
                                                                const float* __restrict__ ksv, const float* __restrict__ ksj,
                                                                const float* __restrict__ g_hg, int ldh, const float* __restrict__ g_sparse,
                                                                const float* __restrict__ g_dheat,
                                                                float* __restrict__ d_kdv, float* __restrict__ d_kdj,
                                                                float* __restrict__ d_ksv, float* __restrict__ d_ksj,
                                                                int H, int W, int K, float var) {
  __shared__ float red[4];
  const int b = blockIdx.x / K, kk = blockIdx.x % K, k = kk + 1, K1 = K + 1, HW = H * W;
  const float* dv = kdv + ((long long)b * K + kk) * 2; const float* dj = kdj + ((long long)b * K + kk) * 4;
  const float* sv = ksv + ((long long)b * K + kk) * 2; const float* sj = ksj + ((long long)b * K + kk) * 4;
  const float a = dj[0], bb = dj[1], c = dj[2], d = dj[3];
  const float det = a * d - bb * c;
  const float i00 = d / det, i01 = -bb / det, i10 = -c / det, i11 = a / det;
  const float J00 = sj[0] * i00 + sj[1] * i10, J01 = sj[0] * i01 + sj[1] * i11;
  const float J10 = sj[2] * i00 + sj[3] * i10, J11 = sj[2] * i01 + sj[3] * i11;
  // accumulators: d kp_d value (2), d kp_s value (2), d J (4)
  float gdx = 0.f, gdy = 0.f, gsx = 0.f, gsy = 0.f, gJ00 = 0.f, gJ01 = 0.f, gJ10 = 0.f, gJ11 = 0.f;
  const float* sb = src + (long long)b * src_bs;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / W, x = p - y * W;
    const float zx = 2.f * ((float)x / (float)(W - 1)) - 1.f, zy = 2.f * ((float)y / (float)(H - 1)) - 1.f;
    const float cx = zx - dv[0], cy = zy - dv[1];
    const float tx = J00 * cx + J01 * cy + sv[0], ty = J10 * cx + J11 * cy + sv[1];
    const float sdx = zx - sv[0], sdy = zy - sv[1];
    const float gd = expf(-0.5f * (cx * cx + cy * cy) / var), gs = expf(-0.5f * (sdx * sdx + sdy * sdy) / var);
    const float4 gh = *reinterpret_cast<const float4*>(g_hg + ((long long)b * HW + p) * ldh + 4 * k);
    // heat = gd - gs (gradient gh.x), drv_heat = gd (gradient g_dheat)
    const float ghd = gh.x + (g_dheat ? g_dheat[((long long)b * HW + p) * K + kk] : 0.f);
    gdx += ghd * gd * cx / var; gdy += ghd * gd * cy / var;          // d G / d m = G (z - m) / var
    gsx -= gh.x * gs * sdx / var; gsy -= gh.x * gs * sdy / var;
    // bilinear gradient of the sparse warp wrt (tx, ty)
    const float ix = ((tx + 1.f) * W - 1.f) / 2.f, iy = ((ty + 1.f) * H - 1.f) / 2.f;
    float gtx = 0.f, gty = 0.f;
    if (ix > -2.f && ix < (float)W + 1.f && iy > -2.f && iy < (float)H + 1.f) {
      const float fx = floorf(ix), fy = floorf(iy); const int x0 = (int)fx, y0 = (int)fy;
      const float ax = ix - fx, ay = iy - fy;
      auto tapdot = [&](int yy, int xx) {
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const float* q = sb + ((long long)yy * W + xx) * 3;
          return q[0] * gh.y + q[1] * gh.z + q[2] * gh.w;
        }
        return 0.f;
      };
      const float d00 = tapdot(y0, x0), d01 = tapdot(y0, x0 + 1), d10 = tapdot(y0 + 1, x0), d11 = tapdot(y0 + 1, x0 + 1);
      gtx = ((d01 - d00) * (1.f - ay) + (d11 - d10) * ay) * 0.5f * (float)W;
      gty = ((d10 - d00) * (1.f - ax) + (d11 - d01) * ax) * 0.5f * (float)H;
    }
    if (g_sparse) {
      const float2 t = *reinterpret_cast<const float2*>(g_sparse + (((long long)b * K1 + k) * HW + p) * 2);
      gtx += t.x; gty += t.y;
    }
    // T = J c + kp_s, c = z - kp_d
    gsx += gtx; gsy += gty;
    gdx -= J00 * gtx + J10 * gty; gdy -= J01 * gtx + J11 * gty;
    gJ00 += gtx * cx; gJ01 += gtx * cy; gJ10 += gty * cx; gJ11 += gty * cy;
  }
  gdx = block_sum(gdx, red); gdy = block_sum(gdy, red); gsx = block_sum(gsx, red); gsy = block_sum(gsy, red);
  gJ00 = block_sum(gJ00, red); gJ01 = block_sum(gJ01, red); gJ10 = block_sum(gJ10, red); gJ11 = block_sum(gJ11, red);
  if (threadIdx.x == 0) {
    float* o = d_kdv + ((long long)b * K + kk) * 2; o[0] = gdx; o[1] = gdy;
    o = d_ksv + ((long long)b * K + kk) * 2; o[0] = gsx; o[1] = gsy;
    // J = S I, I = inv(D):  dS = gJ I^T;  dI = S^T gJ;  dD = -I^T dI I^T
    o = d_ksj + ((long long)b * K + kk) * 4;
    o[0] = gJ00 * i00 + gJ01 * i01; o[1] = gJ00 * i10 + gJ01 * i11; o[2] = gJ10 * i00 + gJ11 * i01; o[3] = gJ10 * i10 + gJ11 * i11;
    const float e00 = sj[0] * gJ00 + sj[2] * gJ10, e01 = sj[0] * gJ01 + sj[2] * gJ11;      // dI = S^T gJ
    const float e10 = sj[1] * gJ00 + sj[3] * gJ10, e11 = sj[1] * gJ01 + sj[3] * gJ11;
    // M = I^T dI: M[r][c] = sum_m I[m][r] dI[m][c]
    const float m00 = i00 * e00 + i10 * e10, m01 = i00 * e01 + i10 * e11, m10 = i01 * e00 + i11 * e10, m11 = i01 * e01 + i11 * e11;
    // dD = -M I^T: dD[r][c] = -sum_m M[r][m] I[c][m]
    o = d_kdj + ((long long)b * K + kk) * 4;
    o[0] = -(m00 * i00 + m01 * i01); o[1] = -(m00 * i10 + m01 * i11); o[2] = -(m10 * i00 + m11 * i01); o[3] = -(m10 * i10 + m11 * i11);
  }
}

// ---- A6b backward: mask = softmax_k(l), deformation = sum_k mask_k T_k, occ = sigmoid(l_K1) -------------------------------------------
__global__ __launch_bounds__(256) void mask_deformation_bwd_kernel(const float* __restrict__ ml, int ldm, const float* __restrict__ sparse,
                                                                   const float* __restrict__ g_def, const float* __restrict__ g_occ,
                                                                   float* __restrict__ d_ml, int lddm, float* __restrict__ d_sparse,
                                                                   long long npix, int HW, int K1) {
  for (long long p = blockIdx.x * 256LL + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
    const int b = (int)(p / HW); const int rem = (int)(p - (long long)b * HW);
    const float* l = ml + p * ldm;
    const float gx = g_def ? g_def[p * 2] : 0.f, gy = g_def ? g_def[p * 2 + 1] : 0.f;
    float mx = -INFINITY;
    for (int k = 0; k < K1; ++k) mx = fmaxf(mx, l[k]);
    float s = 0.f;
    for (int k = 0; k < K1; ++k) s += expf(l[k] - mx);
    float dot = 0.f;
    for (int k = 0; k < K1; ++k) {
      const float m = expf(l[k] - mx) / s;
      const float2 t = *reinterpret_cast<const float2*>(sparse + (((long long)b * K1 + k) * HW + rem) * 2);
      dot += m * (gx * t.x + gy * t.y);
    }
    for (int k = 0; k < K1; ++k) {
      const float m = expf(l[k] - mx) / s;
      const float2 t = *reinterpret_cast<const float2*>(sparse + (((long long)b * K1 + k) * HW + rem) * 2);
      d_ml[p * lddm + k] = m * (gx * t.x + gy * t.y - dot);
      *reinterpret_cast<float2*>(d_sparse + (((long long)b * K1 + k) * HW + rem) * 2) = make_float2(m * gx, m * gy);
    }
    const float o = 1.f / (1.f + expf(-l[K1]));
    d_ml[p * lddm + K1] = (g_occ ? g_occ[p] : 0.f) * o * (1.f - o);
  }
}

// ---- equivariance transform of a frame (models/appmotioncomp_model.py:50-104 `Transform.transform_frame`): per output pixel the
// coordinate z = make_coordinate_grid(H, W) goes through the random affine theta [B][2][3] and the thin-plate term
// sum_c U(|z - p_c|_1) w_bc, U(d) = d^2 log(d + 1e-6), added to BOTH coordinates; the frame is sampled there with
// F.grid_sample(mode bilinear, padding_mode reflection, align_corners False) ----------------------------------------------------------
__device__ __forceinline__ float reflect_coord(float in, float twice_low, float twice_high) {
  if (twice_low == twice_high) return 0.f;
  const float mn = twice_low * 0.5f, span = (twice_high - twice_low) * 0.5f;
  in = fabsf(in - mn);
  const float extra = fmodf(in, span);
  const int flips = (int)floorf(in / span);
  return (flips & 1) ? span - extra + mn : extra + mn;
}

__global__ __launch_bounds__(256) void tps_frame_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ theta,
                                                        const float* __restrict__ cpts, const float* __restrict__ cpar, int ncp,
                                                        long long total, int C, int H, int W) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % W); long long t = i / W; const int oy = (int)(t % H); const int b = (int)(t / H);
    const float zx = 2.f * ((float)ox / (float)(W - 1)) - 1.f, zy = 2.f * ((float)oy / (float)(H - 1)) - 1.f;
    const float* th = theta + b * 6;
    float gx = th[0] * zx + th[1] * zy + th[2], gy = th[3] * zx + th[4] * zy + th[5];
    if (cpts) {
      float r = 0.f;
      for (int c = 0; c < ncp; ++c) {
        const float d = fabsf(zx - cpts[2 * c]) + fabsf(zy - cpts[2 * c + 1]);
        r += d * d * logf(d + 1e-6f) * cpar[b * ncp + c];
      }
      gx += r; gy += r;
    }
    // grid_sample: unnormalise (align_corners=False), reflect about the pixel-area borders, clip, bilinear with in-bounds taps
    float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
    ix = fminf(fmaxf(reflect_coord(ix, -1.f, 2.f * W - 1.f), 0.f), (float)(W - 1));
    iy = fminf(fmaxf(reflect_coord(iy, -1.f, 2.f * H - 1.f), 0.f), (float)(H - 1));
    const float fx = floorf(ix), fy = floorf(iy); const int x0 = (int)fx, y0 = (int)fy;
    const float ax = ix - fx, ay = iy - fy;
    for (int c = 0; c < C; ++c) {
      const float* p = x + ((long long)b * C + c) * H * W;
      float v = 0.f;
      if (y0 >= 0 && y0 < H) {
        if (x0 >= 0 && x0 < W) v += p[y0 * W + x0] * (1.f - ax) * (1.f - ay);
        if (x0 + 1 < W) v += p[y0 * W + x0 + 1] * ax * (1.f - ay);
      }
      if (y0 + 1 < H) {
        if (x0 >= 0 && x0 < W) v += p[(y0 + 1) * W + x0] * (1.f - ax) * ay;
        if (x0 + 1 < W) v += p[(y0 + 1) * W + x0 + 1] * ax * ay;
      }
      y[((long long)b * C + c) * H * W + (long long)oy * W + ox] = v;
    }
  }
}

// rows per chunk: enough (chunk, 64-channel column) blocks to fill the chip (~1024), 16..256 rows each, at most 1024 chunks (the finish
// kernels walk them per channel)
static void chunks(long long P, int C, long long* rows, long long* nchunk) {
  const long long cols = (C + 63) / 64;
  long long r = P * cols / 1024;
  if (r < 16) r = 16;
  if (r > 256) r = 256;
  *rows = r; *nchunk = (P + r - 1) / r;
  if (*nchunk > 1024) { *rows = (P + 1023) / 1024; *nchunk = (P + *rows - 1) / *rows; }
}

static bool pair_v4(const float* x, int ldx, const float* g, int ldg, const float* y, int ldy, int C) {
  return C % 4 == 0 && ldx % 4 == 0 && (!g || ldg % 4 == 0) && (!y || ldy % 4 == 0) && ((((uintptr_t)x) | ((uintptr_t)g) | ((uintptr_t)y)) & 15) == 0;
}

}  // namespace

extern "C" int64_t smx_batchnorm_ws_floats(int64_t P, int C) {
  if (P <= 0 || C <= 0) return 0;
  long long rows, nchunk; chunks(P, C, &rows, &nchunk);
  return nchunk * C * 2 + 2 * (int64_t)C;
}

/* y = relu?(BN_train(x)): batch statistics over the P rows; mr [C][2] = {mean, rstd} saved for the backward; running_mean / running_var
 * updated with `momentum` (unbiased variance) when given */
extern "C" int smx_batchnorm_train_f32(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta, float* mr,
                                       float* running_mean, float* running_var, int64_t P, int C, float eps, float momentum, int relu,
                                       float* ws, void* stream) {
  if (!x || !y || !gamma || !beta || !mr || !ws || P <= 0 || C <= 0 || ldx < C || ldy < C) return SMX_EINVAL;
  long long rows, nchunk; chunks(P, C, &rows, &nchunk);
  hipStream_t st = (hipStream_t)stream;
  if (pair_v4(x, ldx, nullptr, 0, nullptr, 0, C))
    SMX_LAUNCH(chan_pair_partial_v4_kernel<0>, dim3((unsigned)nchunk, (C + 63) / 64), dim3(256), 0, st, x, ldx, (const float*)nullptr, 0, (const float*)nullptr, 0,
               (const float*)nullptr, (long long)P, C, (int)rows, ws);
  else
    SMX_LAUNCH(chan_pair_partial_kernel<0>, dim3((unsigned)nchunk), dim3(256), 0, st, x, ldx, (const float*)nullptr, 0, (const float*)nullptr, 0,
               (const float*)nullptr, (long long)P, C, (int)rows, ws);
  SMX_LAUNCH(bn_finalize_kernel, dim3(smx_cdiv(C, 32)), dim3(256), 0, st, ws, (int)nchunk, C, (long long)P, eps, momentum, mr, running_mean, running_var);
  SMX_LAUNCH(bn_relu_apply_kernel, dim3(grid_for((long long)P * C)), dim3(256), 0, st, x, ldx, y, ldy, mr, gamma, beta, (long long)P, C, relu);
  return smx_launch_status();
}

/* backward of y = relu(BN_train(x)) (relu: y > 0 gates g; y NULL: plain BatchNorm, no gate): dx; dgamma / dbeta ACCUMULATED.  ws: smx_batchnorm_ws_floats(P, C) */
extern "C" int smx_batchnorm_train_bwd_f32(const float* x, int ldx, const float* g, int ldg, const float* y, int ldy, const float* mr,
                                           const float* gamma, float* dx, int ldo, float* dgamma, float* dbeta, int64_t P, int C, float* ws,
                                           void* stream) {
  if (!x || !g || !mr || !gamma || !dx || !dgamma || !dbeta || !ws || P <= 0 || C <= 0 || ldx < C || ldg < C || (y && ldy < C) || ldo < C) return SMX_EINVAL;
  long long rows, nchunk; chunks(P, C, &rows, &nchunk);
  hipStream_t st = (hipStream_t)stream;
  float* sums = ws + nchunk * C * 2;
  if (pair_v4(x, ldx, g, ldg, y, ldy, C))
    SMX_LAUNCH(chan_pair_partial_v4_kernel<1>, dim3((unsigned)nchunk, (C + 63) / 64), dim3(256), 0, st, x, ldx, g, ldg, y, ldy, mr, (long long)P, C, (int)rows, ws);
  else
    SMX_LAUNCH(chan_pair_partial_kernel<1>, dim3((unsigned)nchunk), dim3(256), 0, st, x, ldx, g, ldg, y, ldy, mr, (long long)P, C, (int)rows, ws);
  SMX_LAUNCH(bn_bwd_finalize_kernel, dim3(smx_cdiv(C, 32)), dim3(256), 0, st, ws, (int)nchunk, C, sums, dgamma, dbeta);
  SMX_LAUNCH(bn_relu_bwd_apply_kernel, dim3(grid_for((long long)P * C)), dim3(256), 0, st, x, ldx, g, ldg, y, ldy, mr, sums, gamma, dx, ldo, (long long)P, C);
  return smx_launch_status();
}

extern "C" int smx_avgpool2_bwd_f32(const float* g, int ldg, float* dx, int ldx, int B, int H, int W, int C, void* stream) {
  if (!g || !dx || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || ldg < C || ldx < C) return SMX_EINVAL;
  const long long total = (long long)B * H * W * C;
  SMX_LAUNCH(avgpool2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, g, ldg, dx, ldx, total, H, W, C);
  return smx_launch_status();
}

/* dvalue [B][K][2], djac [B][K][4] (either may be null) -> dlogits [B][H][W][lddl] (channels 0..K-1), djmaps [B][H][W][lddj] (4k+j) */
extern "C" int smx_kp_head_bwd_f32(const float* logits, int ldl, const float* jmaps, int ldj, const float* dvalue, const float* djac,
                                   float* dlogits, int lddl, float* djmaps, int lddj, int B, int H, int W, int K, float temperature, void* stream) {
  if (!logits || !jmaps || !dlogits || !djmaps || B <= 0 || H <= 1 || W <= 1 || K <= 0 || ldl < K || lddl < K || ldj < 4 * K || lddj < 4 * K) return SMX_EINVAL;
  if (ldj % 4 || lddj % 4 || (((uintptr_t)jmaps) | ((uintptr_t)djmaps)) & 15) return SMX_EINVAL;
  SMX_LAUNCH(kp_head_bwd_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, logits, ldl, jmaps, ldj, dvalue, djac, dlogits, lddl, djmaps, lddj,
             H, W, K, temperature);
  return smx_launch_status();
}

/* gradients of the 64-channel hourglass input (g_hg, NHWC ld ldh), of the sparse motions (g_sparse [B][K+1][H][W][2], may be null) and of
 * the driving heatmaps (g_dheat [B][H][W][K], may be null) w.r.t. the keypoints: d_kpd_value / d_kps_value [B][K][2], d_kpd_jac / d_kps_jac
 * [B][K][4] (written).  src NHWC [B][H][W][3]: one source per sample (training pairs). */
extern "C" int smx_sparse_motion_bwd_f32(const float* src, const float* kpd_value, const float* kpd_jac, const float* kps_value, const float* kps_jac,
                                         const float* g_hg, int ldh, const float* g_sparse, const float* g_dheat, float* d_kpd_value,
                                         float* d_kpd_jac, float* d_kps_value, float* d_kps_jac, int B, int H, int W, int K, float kp_variance,
                                         void* stream) {
  if (!src || !kpd_value || !kpd_jac || !kps_value || !kps_jac || !g_hg || !d_kpd_value || !d_kpd_jac || !d_kps_value || !d_kps_jac) return SMX_EINVAL;
  if (B <= 0 || H <= 1 || W <= 1 || K <= 0 || ldh < 4 * (K + 1) || ldh % 4 != 0 || (((uintptr_t)g_hg) & 15)) return SMX_EINVAL;
  SMX_LAUNCH(sparse_motion_bwd_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, src, (long long)H * W * 3, kpd_value, kpd_jac, kps_value, kps_jac,
             g_hg, ldh, g_sparse, g_dheat, d_kpd_value, d_kpd_jac, d_kps_value, d_kps_jac, H, W, K, kp_variance);
  return smx_launch_status();
}

/* g_def [B][H][W][2], g_occ [B][H][W] (either may be null) -> d_logits [B][H][W][lddm] (K1 mask logits + the occlusion logit at K1),
 * d_sparse [B][K1][H][W][2] */
extern "C" int smx_mask_deformation_bwd_f32(const float* mask_logits, int ldm, const float* sparse, const float* g_def, const float* g_occ,
                                            float* d_logits, int lddm, float* d_sparse, int B, int H, int W, int K1, void* stream) {
  if (!mask_logits || !sparse || !d_logits || !d_sparse || B <= 0 || H <= 0 || W <= 0 || K1 <= 0 || ldm < K1 + 1 || lddm < K1 + 1) return SMX_EINVAL;
  const long long npix = (long long)B * H * W;
  SMX_LAUNCH(mask_deformation_bwd_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, mask_logits, ldm, sparse, g_def, g_occ, d_logits,
             lddm, d_sparse, npix, H * W, K1);
  return smx_launch_status();
}

/* x, y NCHW [B][C][H][W]; theta [B][2][3]; control_points [ncp][2] and control_params [B][ncp] (both null: affine only) */
extern "C" int smx_tps_transform_frame_f32(const float* x, float* y, const float* theta, const float* control_points, const float* control_params,
                                           int ncp, int B, int C, int H, int W, void* stream) {
  if (!x || !y || !theta || B <= 0 || C <= 0 || H <= 1 || W <= 1 || ((control_points == nullptr) != (control_params == nullptr)) || (control_points && ncp <= 0)) return SMX_EINVAL;
  const long long total = (long long)B * H * W;
  SMX_LAUNCH(tps_frame_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, theta, control_points, control_params, ncp, total, C, H, W);
  return smx_launch_status();
}
