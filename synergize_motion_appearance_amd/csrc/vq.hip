// A12: fused nearest-codebook lookup (VectorQuantizer.forward, archs/vqgan_arch.py:33-93)
// for gfx950:  d = (|z|^2 + |e|^2) - 2 z.e  ->  first-minimum argmin  ->  z_q = z + (e - z).
//
// The reference materialises two [N][K] fp32 matrices (d and the one-hot); here the z.e
// contraction runs on the fp32 MFMA (32 tokens x 32 codes per wave-tile, exact fp32), the
// running (min, index) lives in registers in the MFMA C layout, the L2 reduction over the
// codebook axis finishes with wavefront shuffles, and nothing but z, the codebook, the
// indices and z_q touches HBM.  z fragments stay in registers for the whole codebook sweep;
// code tiles (32 x D) are double-buffered in LDS (row stride D+4 dwords: conflict-free ds_read_b128),
// the next tile streamed through a few registers in chunks inside the current tile's MFMA loop (one barrier per tile);
// sum (z_q - z)^2 leaves as per-block partials summed in a fixed order (bit-reproducible loss).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "smx.h"
#include "smx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

template <int D>
__global__ __launch_bounds__(256) void vq_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                 int64_t* __restrict__ idx_out, float* __restrict__ zq,
                                                 float* __restrict__ dmin_out, float* __restrict__ sq_part, int N, int Ks) {
  // 32 codes per tile, two LDS buffers: the next tile's global loads are issued DURING the current tile's MFMAs and land in
  // registers while the matrix pipe works; they go to the other buffer a quarter tile later -> ONE barrier per tile and no
  // exposed HBM/L2 round trip (the single-buffered form had a load -> barrier -> compute -> barrier sequence per 64 codes)
  constexpr int LD = D + 4, KS = D / 8, CT = 32;
  constexpr int PF = CT * (D / 4) / 256;               // float4 per thread per tile (D=256: 8, D=32: 1)
  static_assert(CT * (D / 4) % 256 == 0, "tile must split evenly over the block");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cs = sm;                       // [2][CT][LD] code tiles
  float* ee = sm + 2 * CT * LD;         // [Ks rounded up to CT] all code norms, computed once per block
  const int ks_pad = (Ks + CT - 1) / CT * CT;
  float* zzs = ee + ks_pad;             // [4][32] token norms per wave
  int* bis = reinterpret_cast<int*>(zzs + 128);   // [4][32] best index per wave
  float* wsum = reinterpret_cast<float*>(bis + 128);   // [4] per-wave squared error
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 128 + wave * 32;
  const int myrow = row0 + (lane & 31);
  const bool rok = myrow < N;

  // z fragments: lane l holds token (l&31), k = 8*kk + 4*(l>>5) + j
  float4 af[KS]; float zz = 0.f;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    af[kk] = rok ? *reinterpret_cast<const float4*>(z + (long long)myrow * D + kk * 8 + (lane >> 5) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    zz += af[kk].x * af[kk].x + af[kk].y * af[kk].y + af[kk].z * af[kk].z + af[kk].w * af[kk].w;
  }
  zz += __shfl_xor(zz, 32, 64);
  if (lane < 32) zzs[wave * 32 + lane] = zz;
  // all code norms once per block (Ks*D MACs: <1% of the block's 128*Ks*D): 8 lanes per code
  for (int c0 = 0; c0 < ks_pad; c0 += 32) {
    const int code = c0 + (threadIdx.x >> 3), part = threadIdx.x & 7; float s = 0.f;
    if (code < Ks) {
      const float* cp = cb + (long long)code * D;
      for (int c = part * 4; c < D; c += 32) { const float4 v = *reinterpret_cast<const float4*>(cp + c); s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (part == 0) ee[code] = s;
  }
  // the next tile travels global -> registers -> LDS in NCH chunks INSIDE the current tile's MFMA loop (chunk q is requested at
  // kk = q*KS/NCH and written to the other buffer KS/NCH steps later): only PF/NCH float4 are live at a time -- the whole-tile
  // prefetch cost 32 VGPRs at D = 256 and dropped the kernel to one wave per SIMD
  constexpr int NCH = PF >= 4 ? 4 : 1, PC = PF / NCH, KSTEP = KS / NCH;
  float4 pf[PC];
  auto fetch = [&](int t, int q0) {
#pragma unroll
    for (int q = 0; q < PC; ++q) {
      const int i = threadIdx.x + (q0 + q) * 256;
      const int r = i / (D / 4), c4 = i % (D / 4); const int code = t * CT + r;
      pf[q] = code < Ks ? *reinterpret_cast<const float4*>(cb + (long long)code * D + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash = [&](int buf, int q0) {
#pragma unroll
    for (int q = 0; q < PC; ++q) {
      const int i = threadIdx.x + (q0 + q) * 256;
      const int r = i / (D / 4), c4 = i % (D / 4);
      *reinterpret_cast<float4*>(cs + (buf * CT + r) * LD + c4 * 4) = pf[q];
    }
  };
#pragma unroll
  for (int c = 0; c < NCH; ++c) { fetch(0, c * PC); stash(0, c * PC); }
  __syncthreads();
  float zzr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) zzr[r] = zzs[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];

  float bestd[16]; int besti[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { bestd[r] = INFINITY; besti[r] = 0; }

  const int ntiles = ks_pad / CT;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* bp = cs + ((t & 1) * CT + (lane & 31)) * LD + (lane >> 5) * 4;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      if (kk % KSTEP == 0) {                           // compile-time positions (the loop is fully unrolled)
        if (kk > 0 && more) stash((t + 1) & 1, (kk / KSTEP - 1) * PC);   // the buffer tile t-1 used: free since the barrier that ended t-1
        if (more) fetch(t + 1, (kk / KSTEP) * PC);
      }
      const float4 bf = *reinterpret_cast<const float4*>(bp + kk * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].x, bf.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].y, bf.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].z, bf.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].w, bf.w, acc, 0, 0, 0);
    }
    if (more) stash((t + 1) & 1, (NCH - 1) * PC);
    const int code = t * CT + (lane & 31);
    if (code < Ks) {
      const float e2 = ee[code];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = (zzr[r] + e2) - 2.f * acc[r];
        if (d < bestd[r]) { bestd[r] = d; besti[r] = code; }
      }
    }
    __syncthreads();
  }
  // reduce over the 32 code lanes (same lane>>5 half): smaller d, ties -> smaller index
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float d = bestd[r]; int i = besti[r];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float d2 = __shfl_xor(d, o, 64); const int i2 = __shfl_xor(i, o, 64);
      if (d2 < d || (d2 == d && i2 < i)) { d = d2; i = i2; }
    }
    if ((lane & 31) == 0) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      bis[wave * 32 + row] = i;
      if (row0 + row < N) {
        idx_out[row0 + row] = (int64_t)i;
        if (dmin_out) dmin_out[row0 + row] = d;
      }
    }
  }
  __syncthreads();
  // gather: z_q = z + (e[idx] - z)  (straight-through form, vqgan_arch.py:76); 16 B per lane, D/4 lanes per token
  constexpr int LPR = D / 4, RPI = 64 / LPR;           // lanes per row, rows per iteration (D=256: 64 / 1, D=32: 8 / 8)
  float err = 0.f;
#pragma unroll 4
  for (int r = lane / LPR; r < 32; r += RPI) {
    const int row = row0 + r; if (row >= N) break;
    const int c = (lane % LPR) * 4;
    const float4 e = *reinterpret_cast<const float4*>(cb + (long long)bis[wave * 32 + r] * D + c);
    const float4 zv = *reinterpret_cast<const float4*>(z + (long long)row * D + c);
    const float4 df = make_float4(e.x - zv.x, e.y - zv.y, e.z - zv.z, e.w - zv.w);
    if (zq) *reinterpret_cast<float4*>(zq + (long long)row * D + c) = make_float4(zv.x + df.x, zv.y + df.y, zv.z + df.z, zv.w + df.w);
    err += (df.x * df.x + df.y * df.y) + (df.z * df.z + df.w * df.w);
  }
  if (sq_part) {
    // deterministic: fixed shuffle tree per wave, the block's four waves added in order, one partial per block; the host-visible
    // scalar is the fixed-order sum of the partials (vq_finalize_kernel) -- no atomics, bit-reproducible run to run
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) err += __shfl_xor(err, o, 64);
    if (lane == 0) wsum[wave] = err;
    __syncthreads();
    if (threadIdx.x == 0) sq_part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  }
}

// sqerr = sum of the per-block partials in a FIXED order (one block: strided per-thread sums, then a fixed LDS tree)
__global__ __launch_bounds__(256) void vq_finalize_kernel(const float* __restrict__ part, int n, float* __restrict__ sqerr) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *sqerr = red[0];
}

template <int D>
int launch_vq(const float* z, const float* cb, int64_t* idx, float* zq, float* dmin, float* sqerr, float* sq_ws, int N, int Ks, hipStream_t st) {
  const size_t lds = (size_t)(2 * 32 * (D + 4) + (Ks + 31) / 32 * 32 + 128 + 128 + 4) * sizeof(float);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)vq_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int nblk = smx_cdiv(N, 128);
  SMX_LAUNCH(vq_kernel<D>, dim3(nblk), dim3(256), lds, st, z, cb, idx, zq, dmin, sqerr ? sq_ws : nullptr, N, Ks);
  if (sqerr) SMX_LAUNCH(vq_finalize_kernel, dim3(1), dim3(256), 0, st, sq_ws, nblk, sqerr);
  return smx_launch_status();
}

}  // namespace

extern "C" int64_t smx_vq_ws_floats(int N) { return N > 0 ? (int64_t)smx_cdiv(N, 128) : 0; }

extern "C" int smx_vq_nearest_f32(const float* z, const float* codebook, int64_t* idx, float* zq, float* dmin,
                                  float* sqerr, float* sq_ws, int N, int D, int Ks, void* stream) {
  if (!z || !codebook || !idx || N <= 0 || Ks <= 0 || (sqerr && !sq_ws)) return SMX_EINVAL;
  if ((((uintptr_t)z) | ((uintptr_t)codebook) | ((uintptr_t)zq)) & 15) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  switch (D) {
    case 32: return launch_vq<32>(z, codebook, idx, zq, dmin, sqerr, sq_ws, N, Ks, st);
    case 64: return launch_vq<64>(z, codebook, idx, zq, dmin, sqerr, sq_ws, N, Ks, st);
    case 128: return launch_vq<128>(z, codebook, idx, zq, dmin, sqerr, sq_ws, N, Ks, st);
    case 256: return launch_vq<256>(z, codebook, idx, zq, dmin, sqerr, sq_ws, N, Ks, st);
    default: return SMX_EINVAL;
  }
}
