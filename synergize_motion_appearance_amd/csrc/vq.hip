// A12: fused nearest-codebook lookup (VectorQuantizer.forward, archs/vqgan_arch.py:33-93)
// for gfx950:  d = (|z|^2 + |e|^2) - 2 z.e  ->  first-minimum argmin  ->  z_q = z + (e - z).
//
// The reference materialises two [N][K] fp32 matrices (d and the one-hot); here the z.e
// contraction runs on the fp32 MFMA (32 tokens x 32 codes per wave-tile, exact fp32), the
// running (min, index) lives in registers in the MFMA C layout, the L2 reduction over the
// codebook axis finishes with wavefront shuffles, and nothing but z, the codebook, the
// indices and z_q touches HBM.  z fragments stay in registers for the whole codebook sweep;
// code tiles (32 x D) are staged in LDS (row stride D+4 dwords: conflict-free ds_read_b128).
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
                                                 float* __restrict__ dmin_out, float* __restrict__ sqerr, int N, int Ks) {
  constexpr int LD = D + 4, KS = D / 8, CT = 64;     // 64 codes staged per barrier pair (two 32-wide MFMA tiles)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cs = sm;                       // [CT][LD] code tile
  float* ee = sm + CT * LD;             // [Ks rounded up to CT] all code norms, computed once per block
  const int ks_pad = (Ks + CT - 1) / CT * CT;
  float* zzs = ee + ks_pad;             // [4][32] token norms per wave
  int* bis = reinterpret_cast<int*>(zzs + 128);   // [4][32] best index per wave
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
  __syncthreads();
  float zzr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) zzr[r] = zzs[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];

  float bestd[16]; int besti[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { bestd[r] = INFINITY; besti[r] = 0; }

  const int ntiles = ks_pad / CT;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();                    // previous tile fully consumed
    for (int i = threadIdx.x; i < CT * (D / 4); i += 256) {
      const int r = i / (D / 4), c4 = i % (D / 4); const int code = t * CT + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (code < Ks) v = *reinterpret_cast<const float4*>(cb + (long long)code * D + c4 * 4);
      *reinterpret_cast<float4*>(cs + r * LD + c4 * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* bp = cs + (h2 * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const float4 bf = *reinterpret_cast<const float4*>(bp + kk * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].x, bf.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].y, bf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].z, bf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].w, bf.w, acc, 0, 0, 0);
      }
      const int code = t * CT + h2 * 32 + (lane & 31);
      if (code < Ks) {
        const float e2 = ee[code];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = (zzr[r] + e2) - 2.f * acc[r];
          if (d < bestd[r]) { bestd[r] = d; besti[r] = code; }
        }
      }
    }
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
  // gather: z_q = z + (e[idx] - z)  (straight-through form, vqgan_arch.py:76)
  float err = 0.f;
  for (int r = 0; r < 32; ++r) {
    const int row = row0 + r; if (row >= N) break;
    const float* e = cb + (long long)bis[wave * 32 + r] * D;
    for (int c = lane; c < D; c += 64) {
      const float zv = z[(long long)row * D + c];
      const float df = e[c] - zv;
      if (zq) zq[(long long)row * D + c] = zv + df;
      err += df * df;
    }
  }
  if (sqerr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) err += __shfl_xor(err, o, 64);
    if (lane == 0) atomicAdd(sqerr, err);
  }
}

template <int D>
int launch_vq(const float* z, const float* cb, int64_t* idx, float* zq, float* dmin, float* sqerr, int N, int Ks, hipStream_t st) {
  const size_t lds = (size_t)(64 * (D + 4) + (Ks + 63) / 64 * 64 + 128 + 128) * sizeof(float);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)vq_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  SMX_LAUNCH(vq_kernel<D>, dim3(smx_cdiv(N, 128)), dim3(256), lds, st, z, cb, idx, zq, dmin, sqerr, N, Ks);
  return smx_launch_status();
}

}  // namespace

extern "C" int smx_vq_nearest_f32(const float* z, const float* codebook, int64_t* idx, float* zq, float* dmin,
                                  float* sqerr, int N, int D, int Ks, void* stream) {
  if (!z || !codebook || !idx || N <= 0 || Ks <= 0) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  switch (D) {
    case 32: return launch_vq<32>(z, codebook, idx, zq, dmin, sqerr, N, Ks, st);
    case 64: return launch_vq<64>(z, codebook, idx, zq, dmin, sqerr, N, Ks, st);
    case 128: return launch_vq<128>(z, codebook, idx, zq, dmin, sqerr, N, Ks, st);
    case 256: return launch_vq<256>(z, codebook, idx, zq, dmin, sqerr, N, Ks, st);
    default: return SMX_EINVAL;
  }
}
