// A12: fused nearest-codebook lookup (VectorQuantizer.forward, archs/vqgan_arch.py:33-93)
// for gfx950:  d = (|z|^2 + |e|^2) - 2 z.e  ->  first-minimum argmin  ->  z_q = z + (e - z).
//
// The reference materialises two [N][K] fp32 matrices (d and the one-hot); here the z.e
// contraction runs on the fp32 MFMA (32 tokens x 32 codes per wave-tile, exact fp32), the
// running (min, index) lives in registers in the MFMA C layout, and nothing but z, the codebook,
// the indices and z_q touches HBM.  z fragments stay in registers for the whole codebook sweep;
// code tiles (32 x D) are double-buffered in LDS (row stride D+4 dwords: conflict-free ds_read_b128),
// the next tile streamed through a few registers in chunks inside the current tile's MFMA loop (one barrier per tile);
// sum (z_q - z)^2 leaves as per-block partials summed in a fixed order (bit-reproducible loss).
//
// Everything outside the MFMA loop is kept short: a CU holds two blocks, and whenever one is in its prologue or epilogue the other
// has the SIMDs to itself but nobody to hide its own latencies (s_memtime stamps, tools/vq_phase.py: the first form of this kernel
// spent 37 % of a block's life outside the loop, and a block alone ran its loop at 0.71 of the pipe).  So: the code norms |e|^2 come
// from a tiny kernel of their own (every block used to recompute all Ks of them: 1 MB of L2 reads and 32 dependent round trips per
// block, 23 % of its life); the argmin over the 32 code lanes goes through an LDS transpose (8 ds_read_b128 per lane instead of 160
// dependent ds_bpermute); the gather epilogue takes z from the MFMA operand fragments every lane still HOLDS (through LDS, so its
// HBM lines are whole) -- z is read exactly once; and the loop reads each B slice one slice ahead of its MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include "smx.h"
#include "smx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int VQ_MAX_CODES = 4096;     // the norms' share of the workspace (and of LDS: 16 KB)

// |e|^2 of every code, once per call: 8 lanes per code, fixed summation order
__global__ __launch_bounds__(256) void vq_code_norms_kernel(const float* __restrict__ cb, float* __restrict__ norms, int D, int Ks, int ks_pad) {
  const int code = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7; float s = 0.f;
  if (code < Ks) {
    const float* cp = cb + (long long)code * D;
    for (int c = part * 4; c < D; c += 32) { const float4 v = *reinterpret_cast<const float4*>(cp + c); s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
  }
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
  if (part == 0 && code < ks_pad) norms[code] = s;
}

template <int D, bool PROF, bool SPLIT>
__global__ __launch_bounds__(256, 2) void vq_kernel(const float* __restrict__ z, const float* __restrict__ cb, const float* __restrict__ norms,
                                                 int64_t* __restrict__ idx_out, float* __restrict__ zq,
                                                 float* __restrict__ dmin_out, float* __restrict__ sq_part, int N, int Ks,
                                                 int tiles_per_split, float* __restrict__ pd, int* __restrict__ pi) {
  // SPLIT (few tokens: the launch would leave most CUs idle): blockIdx.y sweeps only its share of the code tiles and leaves each token's
  // (min distance, index) over that share in pd / pi [split][N]; vq_combine_kernel folds the shares and does the gather.
  // 32 codes per tile, two LDS buffers: the next tile's global loads are issued DURING the current tile's MFMAs and land in
  // registers while the matrix pipe works; they go to the other buffer a quarter tile later -> ONE barrier per tile and no
  // exposed HBM/L2 round trip (the single-buffered form had a load -> barrier -> compute -> barrier sequence per 64 codes)
  constexpr int LD = D + 4, KS = D / 8, CT = 32, TLD = 36;   // TLD: row stride of the argmin transpose (16-B aligned rows)
  constexpr int PF = CT * (D / 4) / 256;               // float4 per thread per tile (D=256: 8, D=32: 1)
  static_assert(CT * (D / 4) % 256 == 0, "tile must split evenly over the block");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cs = sm;                       // [2][CT][LD] code tiles
  float* ee = sm + (2 * CT * LD > 4 * 32 * TLD * 2 ? 2 * CT * LD : 4 * 32 * TLD * 2);   // [Ks rounded up to CT] the code norms
  const int ks_pad = (Ks + CT - 1) / CT * CT;
  float* zzs = ee + ks_pad;             // [4][32] token norms per wave
  int* bis = reinterpret_cast<int*>(zzs + 128);   // [4][32] best index per wave
  float* wsum = reinterpret_cast<float*>(bis + 128);   // [4] per-wave squared error
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 128 + wave * 32;
  const int myrow = row0 + (lane & 31);
  const bool rok = myrow < N;

  unsigned long long ts[6], wall0 = 0;
  if (PROF) { ts[0] = __builtin_readcyclecounter(); wall0 = wall_clock64(); }
  // z fragments: lane l holds token (l&31), k = 8*kk + 4*(l>>5) + j
  float4 af[KS]; float zz = 0.f;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    af[kk] = rok ? *reinterpret_cast<const float4*>(z + (long long)myrow * D + kk * 8 + (lane >> 5) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    zz += af[kk].x * af[kk].x + af[kk].y * af[kk].y + af[kk].z * af[kk].z + af[kk].w * af[kk].w;
  }
  zz += __shfl_xor(zz, 32, 64);
  if (lane < 32) zzs[wave * 32 + lane] = zz;
  for (int i = threadIdx.x; i < ks_pad; i += 256) ee[i] = norms[i];
  if (PROF) ts[1] = __builtin_readcyclecounter();
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
  const int ntiles = ks_pad / CT;
  const int tb = SPLIT ? (int)blockIdx.y * tiles_per_split : 0;
  const int te = SPLIT ? (tb + tiles_per_split < ntiles ? tb + tiles_per_split : ntiles) : ntiles;
#pragma unroll
  for (int c = 0; c < NCH; ++c) { fetch(tb, c * PC); stash(0, c * PC); }
  __syncthreads();
  float zzr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) zzr[r] = zzs[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];

  float bestd[16]; int besti[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { bestd[r] = INFINITY; besti[r] = 0; }

  if (PROF) ts[2] = __builtin_readcyclecounter();
  for (int t = tb; t < te; ++t) {
    const bool more = t + 1 < te;
    const int cur = (t - tb) & 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* bp = cs + (cur * CT + (lane & 31)) * LD + (lane >> 5) * 4;
    float4 bq[2];                                      // the slice after this one is read while this one's MFMAs run: a block whose
    bq[0] = *reinterpret_cast<const float4*>(bp);      // partner is in its prologue / epilogue has nobody to hide the LDS latency
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      if (kk + 1 < KS) bq[(kk + 1) & 1] = *reinterpret_cast<const float4*>(bp + (kk + 1) * 8);
      if (kk % KSTEP == 0) {                           // compile-time positions (the loop is fully unrolled)
        if (kk > 0 && more) stash(cur ^ 1, (kk / KSTEP - 1) * PC);   // the buffer tile t-1 used: free since the barrier that ended t-1
        if (more) fetch(t + 1, (kk / KSTEP) * PC);
      }
      const float4 bf = bq[kk & 1];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].x, bf.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].y, bf.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].z, bf.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].w, bf.w, acc, 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // the next slice's ds_read first ...
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // ... then this slice's four MFMAs
    }
    if (more) stash(cur ^ 1, (NCH - 1) * PC);
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
  if (PROF) ts[3] = __builtin_readcyclecounter();
  // argmin over the 32 code lanes: (d, index) pairs transposed through LDS (the code tiles are dead: the loop's last barrier is behind
  // every wave), then each lane folds 16 codes of one token -- smaller d, ties -> smaller index -- and meets its neighbour once
  float* td = cs + wave * (2 * 32 * TLD);
  int* ti = reinterpret_cast<int*>(td + 32 * TLD);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    td[row * TLD + (lane & 31)] = bestd[r];
    ti[row * TLD + (lane & 31)] = besti[r];
  }
  __syncthreads();
  {
    const int row = lane >> 1, hc = lane & 1;
    const float4* dp = reinterpret_cast<const float4*>(td + row * TLD + hc * 16);
    const int4* ip = reinterpret_cast<const int4*>(ti + row * TLD + hc * 16);
    float d = INFINITY; int i = 0x7fffffff;
    auto fold = [&](float d2, int i2) { if (d2 < d || (d2 == d && i2 < i)) { d = d2; i = i2; } };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 dv = dp[j]; const int4 iv = ip[j];
      fold(dv.x, iv.x); fold(dv.y, iv.y); fold(dv.z, iv.z); fold(dv.w, iv.w);
    }
    fold(__shfl_xor(d, 1, 64), __shfl_xor(i, 1, 64));
    if (i == 0x7fffffff) i = 0;                            // a row of NaN distances: index 0, as torch.argmin of an all-NaN row is not relied on
    if (SPLIT) {
      if (hc == 0 && row0 + row < N) { pd[(long long)blockIdx.y * N + row0 + row] = d; pi[(long long)blockIdx.y * N + row0 + row] = i; }
      return;
    }
    if (hc == 0) {
      bis[wave * 32 + row] = i;
      if (row0 + row < N) {
        idx_out[row0 + row] = (int64_t)i;
        if (dmin_out) dmin_out[row0 + row] = d;
      }
    }
  }
  __syncthreads();
  if (PROF) ts[4] = __builtin_readcyclecounter();
  // gather: z_q = z + (e[idx] - z)  (straight-through form, vqgan_arch.py:76).  z is NOT read again: every lane still holds its operand
  // fragments, which go through LDS CP columns at a time ([32 tokens][CP + 4] per wave, in the dead code tiles) and come back a token
  // row per 64 / (CP/4) lanes, so the codebook reads and the z_q stores are whole 128-byte lines (16 B per lane, CP*4 bytes per row)
  constexpr int CP = D < 64 ? D : 64, TL = CP + 4, LPRW = CP / 4, RPI = 64 / LPRW, KP = CP / 8;
  float* tz = cs + wave * (32 * TL);
  float err = 0.f;
#pragma unroll
  for (int ps = 0; ps < D / CP; ++ps) {
#pragma unroll
    for (int k = 0; k < KP; ++k)
      *reinterpret_cast<float4*>(tz + (lane & 31) * TL + k * 8 + (lane >> 5) * 4) = af[ps * KP + k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rr = 0; rr < 32 / RPI; ++rr) {
      const int r = rr * RPI + lane / LPRW, c = (lane % LPRW) * 4;
      const int row = row0 + r;
      const float4 zv = *reinterpret_cast<const float4*>(tz + r * TL + c);
      const float4 e = *reinterpret_cast<const float4*>(cb + (long long)bis[wave * 32 + r] * D + ps * CP + c);
      const float4 df = make_float4(e.x - zv.x, e.y - zv.y, e.z - zv.z, e.w - zv.w);
      if (row < N) {
        if (zq) *reinterpret_cast<float4*>(zq + (long long)row * D + ps * CP + c) = make_float4(zv.x + df.x, zv.y + df.y, zv.z + df.z, zv.w + df.w);
        err += (df.x * df.x + df.y * df.y) + (df.z * df.z + df.w * df.w);
      }
    }
    __builtin_amdgcn_wave_barrier();
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
  if (PROF) {                                              // tools/vq_phase.py: phase lengths in shader clocks, in place of eight dmin values
    ts[5] = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0 && dmin_out && row0 + 16 <= N) {
#pragma unroll
      for (int k = 1; k < 6; ++k) dmin_out[row0 + k - 1] = (float)(ts[k] - ts[k - 1]);
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      dmin_out[row0 + 6] = (float)(((xcc & 7) << 8) | ((hw >> 8) & 0xff));          // CU key
      dmin_out[row0 + 7] = (float)(ts[0] & 0xffffff); dmin_out[row0 + 8] = (float)((ts[0] >> 24) & 0xffffff);   // start (shader clock)
      dmin_out[row0 + 9] = (float)(ts[5] & 0xffffff); dmin_out[row0 + 10] = (float)((ts[5] >> 24) & 0xffffff);  // end
      dmin_out[row0 + 11] = (float)(wall0 & 0xffffff); dmin_out[row0 + 12] = (float)(wall_clock64() & 0xffffff); // 100 MHz wall clock
    }
  }
}

// the second half of the SPLIT form: per token fold the splits' (distance, index) pairs -- smaller d, ties -> smaller index, the order the
// one-sweep kernel resolves them in -- then index / distance / z_q = z + (e - z) / the block's share of sum (z_q - z)^2.  128 tokens per block
// (the partial count of the one-sweep form), D/4 lanes per token.
template <int D>
__global__ __launch_bounds__(256) void vq_combine_kernel(const float* __restrict__ z, const float* __restrict__ cb, const float* __restrict__ pd,
                                                         const int* __restrict__ pi, int splits, int64_t* __restrict__ idx_out,
                                                         float* __restrict__ zq, float* __restrict__ dmin_out, float* __restrict__ sq_part, int N) {
  constexpr int LPR = D / 4, TPP = 256 / LPR;             // lanes per token, tokens per pass
  __shared__ float wsum[4];
  const int g = threadIdx.x / LPR, c = (threadIdx.x % LPR) * 4;
  float err = 0.f;
  for (int t0 = 0; t0 < 128; t0 += TPP) {
    const int row = blockIdx.x * 128 + t0 + g;
    if (row >= N) break;                                   // rows ascend with t0 and g: nothing valid follows for this thread
    float d = pd[row]; int i = pi[row];
    for (int sidx = 1; sidx < splits; ++sidx) {
      const float d2 = pd[(long long)sidx * N + row]; const int i2 = pi[(long long)sidx * N + row];
      if (d2 < d || (d2 == d && i2 < i)) { d = d2; i = i2; }
    }
    if (c == 0) { idx_out[row] = (int64_t)i; if (dmin_out) dmin_out[row] = d; }
    const float4 e = *reinterpret_cast<const float4*>(cb + (long long)i * D + c);
    const float4 zv = *reinterpret_cast<const float4*>(z + (long long)row * D + c);
    const float4 df = make_float4(e.x - zv.x, e.y - zv.y, e.z - zv.z, e.w - zv.w);
    if (zq) *reinterpret_cast<float4*>(zq + (long long)row * D + c) = make_float4(zv.x + df.x, zv.y + df.y, zv.z + df.z, zv.w + df.w);
    err += (df.x * df.x + df.y * df.y) + (df.z * df.z + df.w * df.w);
  }
  if (sq_part) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) err += __shfl_xor(err, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = err;
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

constexpr int VQ_SPLIT_MAX = 8, VQ_SPLIT_TOKENS = 16384;   // the split form's share of the workspace: 2 x 8 x min(N, 16384) floats

template <int D, bool PROF>
int launch_vq_p(const float* z, const float* cb, int64_t* idx, float* zq, float* dmin, float* sqerr, float* ws, int N, int Ks, hipStream_t st) {
  const int ks_pad = (Ks + 31) / 32 * 32, ntiles = ks_pad / 32;
  const int tile_floats = 2 * 32 * (D + 4) > 4 * 32 * 36 * 2 ? 2 * 32 * (D + 4) : 4 * 32 * 36 * 2;
  const size_t lds = (size_t)(tile_floats + ks_pad + 128 + 128 + 4) * sizeof(float);
  const int nblk = smx_cdiv(N, 128);
  float* norms = ws; float* part = ws + VQ_MAX_CODES;
  static int cus = 0;
  if (!cus) { int dev = 0, n = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; } cus = n; }
  SMX_LAUNCH(vq_code_norms_kernel, dim3(ks_pad / 32), dim3(256), 0, st, cb, norms, D, Ks, ks_pad);
  // few tokens (one frame is 8 blocks, a training batch of four 32): split the codebook sweep over blockIdx.y so the launch covers the chip
  int splits = 1;
  if (!PROF && smx_tune(SMX_TUNE_VQ_SPLIT) && nblk * 2 <= cus && N <= VQ_SPLIT_TOKENS && ntiles >= 8) {
    splits = cus / nblk; if (splits > VQ_SPLIT_MAX) splits = VQ_SPLIT_MAX; if (splits > ntiles / 4) splits = ntiles / 4;   // >= 4 tiles per share: below that
  }                                                                                                                        // the second launch costs what the split saves
  if (splits > 1) {
    const int tps = smx_cdiv(ntiles, splits); splits = smx_cdiv(ntiles, tps);
    float* pd = part + nblk; int* pi = reinterpret_cast<int*>(pd + (size_t)VQ_SPLIT_MAX * (N < VQ_SPLIT_TOKENS ? N : VQ_SPLIT_TOKENS));
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)vq_kernel<D, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SMX_LAUNCH((vq_kernel<D, false, true>), dim3(nblk, splits), dim3(256), lds, st, z, cb, (const float*)norms, idx, zq, dmin, (float*)nullptr, N, Ks, tps, pd, pi);
    SMX_LAUNCH(vq_combine_kernel<D>, dim3(nblk), dim3(256), 0, st, z, cb, (const float*)pd, (const int*)pi, splits, idx, zq, dmin, sqerr ? part : nullptr, N);
  } else {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)vq_kernel<D, PROF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SMX_LAUNCH((vq_kernel<D, PROF, false>), dim3(nblk), dim3(256), lds, st, z, cb, (const float*)norms, idx, zq, dmin, sqerr ? part : nullptr, N, Ks, ntiles,
               (float*)nullptr, (int*)nullptr);
  }
  if (sqerr) SMX_LAUNCH(vq_finalize_kernel, dim3(1), dim3(256), 0, st, part, nblk, sqerr);
  return smx_launch_status();
}

template <int D>
int launch_vq(const float* z, const float* cb, int64_t* idx, float* zq, float* dmin, float* sqerr, float* ws, int N, int Ks, hipStream_t st) {
#ifdef SMX_TOOLS
  static const bool prof = getenv("SMX_VQ_PROF") != nullptr;
  if (prof) return launch_vq_p<D, true>(z, cb, idx, zq, dmin, sqerr, ws, N, Ks, st);
#endif
  return launch_vq_p<D, false>(z, cb, idx, zq, dmin, sqerr, ws, N, Ks, st);
}

}  // namespace

extern "C" int64_t smx_vq_ws_floats(int N) { return N > 0 ? (int64_t)VQ_MAX_CODES + smx_cdiv(N, 128) + 2LL * VQ_SPLIT_MAX * (N < VQ_SPLIT_TOKENS ? N : VQ_SPLIT_TOKENS) : 0; }

extern "C" int smx_vq_nearest_f32(const float* z, const float* codebook, int64_t* idx, float* zq, float* dmin,
                                  float* sqerr, float* sq_ws, int N, int D, int Ks, void* stream) {
  if (!z || !codebook || !idx || !sq_ws || N <= 0 || Ks <= 0 || Ks > VQ_MAX_CODES) return SMX_EINVAL;
  if ((((uintptr_t)z) | ((uintptr_t)codebook) | ((uintptr_t)zq) | ((uintptr_t)sq_ws)) & 15) return SMX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  switch (D) {
    case 32: return launch_vq<32>(z, codebook, idx, zq, dmin, sqerr, sq_ws, N, Ks, st);
    case 64: return launch_vq<64>(z, codebook, idx, zq, dmin, sqerr, sq_ws, N, Ks, st);
    case 128: return launch_vq<128>(z, codebook, idx, zq, dmin, sqerr, sq_ws, N, Ks, st);
    case 256: return launch_vq<256>(z, codebook, idx, zq, dmin, sqerr, sq_ws, N, Ks, st);
    default: return SMX_EINVAL;
  }
}
