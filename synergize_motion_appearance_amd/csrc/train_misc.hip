// Training: backward of the HBM-bound stages of the hot path for gfx950 -- the backward warp
// (grid_sampler_2d_backward x 9 per reference step, fused with the align_corners=True flow / occlusion resize),
// bilinear-resize adjoint, VectorQuantizer straight-through + codebook losses, the flow / occlusion update chain, the SFT
// modulation, L1 losses -- and the optimiser side (fused Adam over a flat parameter buffer, EMA).
//
// Scatter-type adjoints (warp -> source features, resize -> coarse grid, VQ -> codebook rows) use fp32 atomicAdd on
// pre-zeroed buffers (return-less global atomics run at L2 rate on gfx950); everything else is a gather.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "smx.h"
#include "smx_common.h"

namespace {

inline int grid_for(long long total) { int g = smx_cdiv(total, 256); return g > 16384 ? 16384 : (g < 1 ? 1 : g); }

// align_corners=True source index + weights -- the SAME arithmetic as warp_resize.hip's ac_src (and ATen)
__device__ __forceinline__ void ac_src(int o, int in, int out, int& i0, int& i1, float& l0, float& l1) {
#pragma clang fp contract(off)
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float real = scale * o;
  i0 = (int)real; if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = fminf(fmaxf(real - i0, 0.f), 1.f); l0 = 1.f - l1;
}

// ---- A7 backward ------------------------------------------------------------------------------------------------------
// forward: out[b,y,x,:] = occ_s * sum_taps w_t feat[bf, y_t, x_t, :],   (gx, gy, occ_s) = resize_ac(flow | occ)(y, x),
//          ix = (gx + 1)/2 (W - 1), iy = (gy + 1)/2 (H - 1), zeros padding.
// One pixel = LPP = C/4 consecutive lanes x float4 (the forward's layout: every tap is one coalesced run).  Per pixel:
//   dfeat[taps] += w_t occ_s g           (atomics, 16 B per lane per tap)
//   gsm[b,y,x] = { d gx, d gy, d occ_s } (lane-group shuffle reduction over the channels) -- resized back by resize_bwd.
template <int LPP>
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ feat, long long feat_bs, const float* __restrict__ flow,
                                                       const float* __restrict__ occ, const float* __restrict__ g,
                                                       float* __restrict__ dfeat, float* __restrict__ gsm, long long npix,
                                                       int H, int W, int C, int Hf, int Wf) {
  constexpr int PPB = 256 / LPP;
  const int sub = threadIdx.x % LPP;
  const long long pix = (long long)blockIdx.x * PPB + threadIdx.x / LPP;
  const bool live = pix < npix;
  const long long pp = live ? pix : npix - 1;
  const int HW = H * W;
  const int b = (int)(pp / HW); const int rem = (int)(pp - (long long)b * HW);
  const int y = rem / W, x = rem - y * W;
  const float* fb = flow + (long long)b * Hf * Wf * 2;
  const float* ob = occ ? occ + (long long)b * Hf * Wf : nullptr;
  float gxv, gyv, ov = 1.f;
  if (Hf == H && Wf == W) {
    gxv = fb[(y * Wf + x) * 2]; gyv = fb[(y * Wf + x) * 2 + 1];
    if (ob) ov = ob[y * Wf + x];
  } else {
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    ac_src(y, Hf, H, y0, y1, ly0, ly1); ac_src(x, Wf, W, x0, x1, lx0, lx1);
    const float2 f00 = *reinterpret_cast<const float2*>(fb + (y0 * Wf + x0) * 2);
    const float2 f01 = *reinterpret_cast<const float2*>(fb + (y0 * Wf + x1) * 2);
    const float2 f10 = *reinterpret_cast<const float2*>(fb + (y1 * Wf + x0) * 2);
    const float2 f11 = *reinterpret_cast<const float2*>(fb + (y1 * Wf + x1) * 2);
    gxv = ly0 * (lx0 * f00.x + lx1 * f01.x) + ly1 * (lx0 * f10.x + lx1 * f11.x);
    gyv = ly0 * (lx0 * f00.y + lx1 * f01.y) + ly1 * (lx0 * f10.y + lx1 * f11.y);
    if (ob) ov = ly0 * (lx0 * ob[y0 * Wf + x0] + lx1 * ob[y0 * Wf + x1]) + ly1 * (lx0 * ob[y1 * Wf + x0] + lx1 * ob[y1 * Wf + x1]);
  }
  const float ix = (gxv + 1.f) * 0.5f * (float)(W - 1), iy = (gyv + 1.f) * 0.5f * (float)(H - 1);
  const float fx = floorf(ix), fy = floorf(iy);
  const float tx = ix - fx, ty = iy - fy;
  const bool sane = ix > -2.f && ix < (float)W + 1.f && iy > -2.f && iy < (float)H + 1.f;
  const int x0 = sane ? (int)fx : -4, y0 = sane ? (int)fy : -4;
  const float w[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
  const float* fbase = feat + (long long)b * feat_bs + sub * 4;
  float* dbase = dfeat ? dfeat + (long long)b * feat_bs + sub * 4 : nullptr;
  float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) gv = *reinterpret_cast<const float4*>(g + pp * C + sub * 4);
  float4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const long long off = ((long long)yy * W + xx) * C;
      v[k] = *reinterpret_cast<const float4*>(fbase + off);
      if (dbase && live) {
        const float s = w[k] * ov;
        atomicAdd(dbase + off, s * gv.x); atomicAdd(dbase + off + 1, s * gv.y);
        atomicAdd(dbase + off + 2, s * gv.z); atomicAdd(dbase + off + 3, s * gv.w);
      }
    }
  }
  auto dot = [&](const float4& a) { return a.x * gv.x + a.y * gv.y + a.z * gv.z + a.w * gv.w; };
  const float d00 = dot(v[0]), d01 = dot(v[1]), d10 = dot(v[2]), d11 = dot(v[3]);
  float dix = ((d01 - d00) * (1.f - ty) + (d11 - d10) * ty) * ov;
  float diy = ((d10 - d00) * (1.f - tx) + (d11 - d01) * tx) * ov;
  float doc = w[0] * d00 + w[1] * d01 + w[2] * d10 + w[3] * d11;
#pragma unroll
  for (int o = LPP >> 1; o > 0; o >>= 1) { dix += __shfl_xor(dix, o, 64); diy += __shfl_xor(diy, o, 64); doc += __shfl_xor(doc, o, 64); }
  if (live && sub == 0 && gsm) {
    gsm[pp * 3] = dix * 0.5f * (float)(W - 1);
    gsm[pp * 3 + 1] = diy * 0.5f * (float)(H - 1);
    gsm[pp * 3 + 2] = occ ? doc : 0.f;
  }
}

// adjoint of resize_ac: dx[b, y_t, x_t, c] += l_y l_x g[b, oy, ox, c] over the 4 taps of every OUTPUT-grid element (atomics)
__global__ __launch_bounds__(256) void resize_ac_bwd_kernel(const float* __restrict__ g, int ldg, float* __restrict__ dx, int ldx,
                                                            long long total, int Hin, int Win, int Hout, int Wout, int C) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long p = i / C;
    const int ox = (int)(p % Wout); p /= Wout; const int oy = (int)(p % Hout); const int b = (int)(p / Hout);
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    ac_src(oy, Hin, Hout, y0, y1, ly0, ly1); ac_src(ox, Win, Wout, x0, x1, lx0, lx1);
    const float gv = g[(((long long)b * Hout + oy) * Wout + ox) * ldg + c];
    float* db = dx + (long long)b * Hin * Win * ldx + c;
    atomicAdd(db + ((long long)y0 * Win + x0) * ldx, ly0 * lx0 * gv);
    atomicAdd(db + ((long long)y0 * Win + x1) * ldx, ly0 * lx1 * gv);
    atomicAdd(db + ((long long)y1 * Win + x0) * ldx, ly1 * lx0 * gv);
    atomicAdd(db + ((long long)y1 * Win + x1) * ldx, ly1 * lx1 * gv);
  }
}

// ---- A12 backward (archs/vqgan_arch.py:60-76) -----------------------------------------------------------------------------
// loss = beta * mean((sg(e) - z)^2) + mean((e - sg(z))^2);  z_q = z + sg(e - z)
//   dz = g_zq + gl * 2 beta / numel * (z - e);   de[idx] += gl * 2 / numel * (e - z)   (atomics; gl = d total / d loss, device scalar)
__global__ __launch_bounds__(256) void vq_bwd_kernel(const float* __restrict__ z, const float* __restrict__ cb, const int64_t* __restrict__ idx,
                                                     const float* __restrict__ g_zq, const float* __restrict__ g_loss, float beta,
                                                     float* __restrict__ dz, float* __restrict__ dcb, long long N, int D) {
  const float gl = g_loss ? *g_loss : 0.f;
  const float inv = 2.f / (float)(N * D);
  const long long total = N * D;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / D; const int d = (int)(i - n * D);
    const long long row = idx[n];
    const float e = cb[row * D + d], zv = z[i];
    const float diff = zv - e;
    dz[i] = (g_zq ? g_zq[i] : 0.f) + gl * beta * inv * diff;
    if (dcb && gl != 0.f) atomicAdd(dcb + row * D + d, -gl * inv * diff);
  }
}

// ---- flow / occlusion update chain (appmotioncodebook_arch.py:587-601) ----------------------------------------------------
// forward (flow_occ_update): m_com = flow + r[:2]/hs, occ = sigmoid(occ_prev + r[2]);  hs = (Hf - 1)/2
// backward: d flow += g_mcom; d r[:2] = g_mcom / hs; d occ_prev = d r[2] = g_occ * occ (1 - occ)
__global__ __launch_bounds__(256) void flow_occ_update_bwd_kernel(const float* __restrict__ g_mcom, const float* __restrict__ g_occ,
                                                                  const float* __restrict__ occ, float* __restrict__ d_flow,
                                                                  float* __restrict__ d_r, float* __restrict__ d_occ_prev, long long npix, float inv_hs) {
  for (long long p = blockIdx.x * 256LL + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
    const float gx = g_mcom ? g_mcom[p * 2] : 0.f, gy = g_mcom ? g_mcom[p * 2 + 1] : 0.f;
    const float o = occ[p]; const float go = (g_occ ? g_occ[p] : 0.f) * o * (1.f - o);
    d_flow[p * 2] = gx; d_flow[p * 2 + 1] = gy;
    d_r[p * 3] = gx * inv_hs; d_r[p * 3 + 1] = gy * inv_hs; d_r[p * 3 + 2] = go;
    d_occ_prev[p] = go;
  }
}

// ---- SFT modulation backward (appmotioncodebook_arch.py:49-51): out = dec + w (dec * scale + shift) ------------------------
__global__ __launch_bounds__(256) void sft_bwd_kernel(const float* __restrict__ g, const float* __restrict__ dec, int ldd, const float* __restrict__ scale,
                                                      float* __restrict__ d_dec, float* __restrict__ d_scale, float* __restrict__ d_shift,
                                                      float w, long long P, int C) {
  const long long total = P * C;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C; const int c = (int)(i - r * C);
    const float gv = g[i];
    d_dec[i] = gv * (1.f + w * scale[i]);
    d_scale[i] = gv * w * dec[r * ldd + c];
    d_shift[i] = gv * w;
  }
}

// ---- L1 loss (reduction = mean): forward partial sums, then a fixed-order finish; backward sign(a - b) * gl * weight / n ----
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float* __restrict__ part) {
  __shared__ float red[256];
  float s = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += fabsf(a[i] - b[i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void scalar_finish_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *out = red[0] * scale;
}

__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g_loss,
                                                     float coef, float* __restrict__ da, long long n, int accumulate) {
  const float c = (g_loss ? *g_loss : 1.f) * coef;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float d = a[i] - b[i];
    const float v = d > 0.f ? c : (d < 0.f ? -c : 0.f);
    da[i] = accumulate ? da[i] + v : v;
  }
}

// y = alpha * x (+ y)   (dense)
__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float alpha, int accumulate) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = accumulate ? y[i] + alpha * x[i] : alpha * x[i];
}

// ---- optimiser: torch.optim.Adam (no amsgrad, weight_decay folded into the gradient) over a flat buffer ------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float gr = g[i] * gscale;
    if (wd != 0.f) gr += wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    // torch: step_size = lr / bias_correction1; denom = sqrt(v) / sqrt(bias_correction2) + eps
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

// space to depth: y[b][h][w][(p1*p + p2)*C + c] = x[b][h*p + p1][w*p + p2][c]  (the patch order of `to_app_feat_*`'s un-patchify store,
// appmotioncodebook_arch.py:222-240: its backward sees the output gradient as tokens x (p1 p2 c) columns)
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, long long total, int Ho, int Wo, int C, int p) {
  const int N = p * p * C;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i % N); long long t = i / N;
    const int w = (int)(t % Wo); t /= Wo; const int h = (int)(t % Ho); const int b = (int)(t / Ho);
    const int c = n % C, pp = n / C, p1 = pp / p, p2 = pp - p1 * p;
    y[i] = x[(((long long)b * Ho * p + h * p + p1) * (Wo * p) + w * p + p2) * ldx + c];
  }
}

// ema = decay * ema + (1 - decay) * p   (models/sr_model.py model_ema)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, long long n, float decay) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) ema[i] = decay * ema[i] + (1.f - decay) * p[i];
}

}  // namespace

/* dfeat (may be null: features without gradient) must be zeroed by the caller; gsm [B][H][W][3] = {d gx, d gy, d occ_s} at the
 * FEATURE resolution (may be null) -- bring it to the flow grid with smx_resize_ac_bwd_f32. */
extern "C" int smx_warp_bwd_f32(const float* feat, int feat_batch, const float* flow, const float* occ, const float* g,
                                float* dfeat, float* gsm, int B, int H, int W, int C, int Hf, int Wf, void* stream) {
  if (!feat || !flow || !g || B <= 0 || H <= 1 || W <= 1 || C <= 0 || Hf <= 0 || Wf <= 0) return SMX_EINVAL;
  if (C % 4 != 0 || (feat_batch != B)) return SMX_EINVAL;          // training: one source per sample (no broadcast source here)
  const int lpp = C / 4;
  if (lpp > 64 || (lpp & (lpp - 1)) != 0) return SMX_EINVAL;
  const long long npix = (long long)B * H * W;
  const long long feat_bs = (long long)H * W * C;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(smx_cdiv(npix * lpp, 256)), block(256);
#define SMX_WB(L) SMX_LAUNCH(warp_bwd_kernel<L>, grid, block, 0, st, feat, feat_bs, flow, occ, g, dfeat, gsm, npix, H, W, C, Hf, Wf)
  switch (lpp) {
    case 1: SMX_WB(1); break; case 2: SMX_WB(2); break; case 4: SMX_WB(4); break; case 8: SMX_WB(8); break;
    case 16: SMX_WB(16); break; case 32: SMX_WB(32); break; default: SMX_WB(64); break;
  }
#undef SMX_WB
  return smx_launch_status();
}

/* g [B][Hout][Wout][ldg] (C channels) -> dx [B][Hin][Win][ldx] += adjoint of the align_corners=True bilinear resize (dx pre-zeroed or accumulating) */
extern "C" int smx_resize_ac_bwd_f32(const float* g, int ldg, float* dx, int ldx, int B, int Hin, int Win, int Hout, int Wout, int C, void* stream) {
  if (!g || !dx || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C <= 0 || ldg < C || ldx < C) return SMX_EINVAL;
  const long long total = (long long)B * Hout * Wout * C;
  SMX_LAUNCH(resize_ac_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, g, ldg, dx, ldx, total, Hin, Win, Hout, Wout, C);
  return smx_launch_status();
}

extern "C" int smx_space_to_depth_f32(const float* x, int ldx, float* y, int B, int Ho, int Wo, int C, int p, void* stream) {
  if (!x || !y || B <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || p <= 0 || ldx < C) return SMX_EINVAL;
  const long long total = (long long)B * Ho * Wo * p * p * C;
  SMX_LAUNCH(s2d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, total, Ho, Wo, C, p);
  return smx_launch_status();
}

extern "C" int smx_vq_bwd_f32(const float* z, const float* codebook, const int64_t* idx, const float* g_zq, const float* g_loss, float beta,
                              float* dz, float* dcodebook, int64_t N, int D, void* stream) {
  if (!z || !codebook || !idx || !dz || N <= 0 || D <= 0) return SMX_EINVAL;
  SMX_LAUNCH(vq_bwd_kernel, dim3(grid_for((long long)N * D)), dim3(256), 0, (hipStream_t)stream, z, codebook, idx, g_zq, g_loss, beta, dz, dcodebook, (long long)N, D);
  return smx_launch_status();
}

extern "C" int smx_flow_occ_update_bwd_f32(const float* g_mcom, const float* g_occ, const float* occ, float* d_flow, float* d_r,
                                           float* d_occ_prev, int B, int H, int W, void* stream) {
  if (!occ || !d_flow || !d_r || !d_occ_prev || B <= 0 || H <= 1 || W <= 1) return SMX_EINVAL;
  const long long npix = (long long)B * H * W;
  SMX_LAUNCH(flow_occ_update_bwd_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, g_mcom, g_occ, occ, d_flow, d_r, d_occ_prev, npix,
             2.f / (float)(H - 1));
  return smx_launch_status();
}

extern "C" int smx_sft_combine_bwd_f32(const float* g, const float* dec, int ld_dec, const float* scale, float* d_dec, float* d_scale,
                                       float* d_shift, float w, int64_t P, int C, void* stream) {
  if (!g || !dec || !scale || !d_dec || !d_scale || !d_shift || P <= 0 || C <= 0 || ld_dec < C) return SMX_EINVAL;
  SMX_LAUNCH(sft_bwd_kernel, dim3(grid_for((long long)P * C)), dim3(256), 0, (hipStream_t)stream, g, dec, ld_dec, scale, d_dec, d_scale, d_shift, w, (long long)P, C);
  return smx_launch_status();
}

/* loss = weight * mean |a - b| -> out[0];  part: 1024 floats scratch */
extern "C" int smx_l1_loss_f32(const float* a, const float* b, int64_t n, float weight, float* part, float* out, void* stream) {
  if (!a || !b || !part || !out || n <= 0) return SMX_EINVAL;
  int blocks = smx_cdiv(n, 256 * 8); if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
  hipStream_t st = (hipStream_t)stream;
  SMX_LAUNCH(l1_partial_kernel, dim3(blocks), dim3(256), 0, st, a, b, (long long)n, part);
  SMX_LAUNCH(scalar_finish_kernel, dim3(1), dim3(256), 0, st, part, blocks, weight / (float)n, out);
  return smx_launch_status();
}

/* da (+)= g_loss[0] * weight / n * sign(a - b)   (g_loss: device scalar, null = 1) */
extern "C" int smx_l1_loss_bwd_f32(const float* a, const float* b, const float* g_loss, int64_t n, float weight, float* da, int accumulate, void* stream) {
  if (!a || !b || !da || n <= 0) return SMX_EINVAL;
  SMX_LAUNCH(l1_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, g_loss, weight / (float)n, da, (long long)n, accumulate);
  return smx_launch_status();
}

extern "C" int smx_scale_f32(const float* x, float* y, int64_t n, float alpha, int accumulate, void* stream) {
  if (!x || !y || n <= 0) return SMX_EINVAL;
  SMX_LAUNCH(scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)n, alpha, accumulate);
  return smx_launch_status();
}

/* one torch.optim.Adam step (step count t >= 1) over n contiguous parameters; gscale multiplies the gradient first (1/world for an
 * all-reduce SUM). */
extern "C" int smx_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int t, float gscale, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || t < 1) return SMX_EINVAL;
  const float bc1 = 1.f - powf(beta1, (float)t), bc2 = 1.f - powf(beta2, (float)t);
  SMX_LAUNCH(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr, beta1, beta2, eps, weight_decay, bc1,
             sqrtf(bc2), gscale);
  return smx_launch_status();
}

extern "C" int smx_ema_f32(float* ema, const float* p, int64_t n, float decay, void* stream) {
  if (!ema || !p || n <= 0) return SMX_EINVAL;
  SMX_LAUNCH(ema_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, ema, p, (long long)n, decay);
  return smx_launch_status();
}
