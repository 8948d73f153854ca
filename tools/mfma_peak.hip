// What does the fp32 matrix pipe actually sustain on this chip?  Pure v_mfma_f32_32x32x2_f32 loops (no memory traffic), for
// 1/2/4 independent accumulator chains per wave and 1/2/4 waves per SIMD, a full-chip launch held for ~1 ms per point.
// Prints TF/s, the fraction of the 157.3 TF/s datasheet peak and the shader clock the run averaged (s_memtime ticks per
// wall_clock64 tick) -- the calibration every fp32-MFMA roofline fraction in DESIGN.md is read against.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void spin(float* out, unsigned long long* clk, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  const float a = a0 + threadIdx.x * 1e-9f, b = b0;
  const unsigned long long s0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  const unsigned long long s1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 12345.678f) out[0] = s;                      // keeps the chains alive
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1 - s0; clk[1] = w1 - w0; }
}

// the same loop on DATA: 32 A registers and B either from 8 registers (LDSB = 0) or from ds_read_b128 of a random LDS tile, one read per
// four MFMAs issued a group ahead (LDSB = 1: the code-tile walk of vq_kernel) -- switching activity is what the power controller sees
template <int LDSB>
__global__ __launch_bounds__(256) void spin_data(const float* __restrict__ src, float* out, unsigned long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int LD = 260;
  for (int i = threadIdx.x; i < 32 * LD; i += 256) sm[i] = src[i % 8192] * 0.01f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float4 a[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) a[k] = *reinterpret_cast<const float4*>(src + ((threadIdx.x * 32 + k) * 4) % 8192);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* bp = sm + (lane & 31) * LD + (lane >> 5) * 4;
  const unsigned long long s0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    float4 bq[2];
    bq[0] = LDSB ? *reinterpret_cast<const float4*>(bp) : a[31];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (k + 1 < 32) bq[(k + 1) & 1] = LDSB ? *reinterpret_cast<const float4*>(bp + (k + 1) * 8) : a[(k + 7) & 31];
      const float4 b = bq[k & 1];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].w, b.w, acc, 0, 0, 0);
      if (LDSB) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] *= 0.001f;      // keeps the values finite; 16 VALU per 128 MFMAs
  }
  const unsigned long long s1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 12345.678f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1 - s0; clk[1] = w1 - w0; }
}

template <int LDSB>
void run_data(int waves_per_simd, const float* src, float* out, unsigned long long* clk) {
  const int lds = waves_per_simd == 1 ? 96 * 1024 : waves_per_simd == 2 ? 64 * 1024 : 36 * 1024;
  hipFuncSetAttribute((const void*)spin_data<LDSB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int nblk = 256 * waves_per_simd, iters = 2048 / waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  spin_data<LDSB><<<nblk, 256, lds>>>(src, out, clk, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) spin_data<LDSB><<<nblk, 256, lds>>>(src, out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double flop = 2.0 * 32 * 32 * 2 * 128.0 * iters * 4.0 * nblk * reps;
  const double tf = flop / (ms * 1e-3) / 1e12;
  printf("random data, B from %s  waves/SIMD %d : %7.1f TF/s  (%.3f of 157.3)   %.3f ms/launch   shader clock %.0f MHz\n",
         LDSB ? "LDS (ds_read_b128 per 4 MFMAs)" : "registers", waves_per_simd, tf, tf / 157.3, ms / reps, 100.0 * (double)h[0] / (double)h[1]);
}


// vq_kernel's tile loop rebuilt a piece at a time: which of its non-MFMA parts keeps a wave that is ALONE on its SIMD from filling the pipe?
//   EPI: the per-tile epilogue (16 x {add, fma, compare, 2 selects} on the accumulator, accumulator reset)
//   BAR: one block barrier per tile        STREAM: the next tile's 32 KB travel L2 -> registers -> LDS in four chunks inside the loop
template <bool EPI, bool BAR, bool STREAM>
__global__ __launch_bounds__(256) void spin_tile(const float* __restrict__ src, const float* __restrict__ cbk, float* out, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int LD = 260;
  for (int i = threadIdx.x; i < 2 * 32 * LD; i += 256) sm[i] = src[i % 8192] * 0.01f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float4 a[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) a[k] = *reinterpret_cast<const float4*>(src + ((threadIdx.x * 32 + k) * 4) % 8192);
  float best[16]; int bi[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { best[r] = 1e30f; bi[r] = 0; }
  float4 pf[2];
  for (int t = 0; t < tiles; ++t) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* bp = sm + ((t & 1) * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
    float* wp = sm + (((t + 1) & 1) * 32) * LD;
    const float* gp = cbk + (size_t)((t + 1) & 31) * 8192;
    float4 bq[2];
    bq[0] = *reinterpret_cast<const float4*>(bp);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (k + 1 < 32) bq[(k + 1) & 1] = *reinterpret_cast<const float4*>(bp + (k + 1) * 8);
      if (STREAM && k % 8 == 0) {
        if (k > 0) {
#pragma unroll
          for (int q = 0; q < 2; ++q) { const int i = threadIdx.x + ((k / 8 - 1) * 2 + q) * 256; *reinterpret_cast<float4*>(wp + (i / 64) * LD + (i % 64) * 4) = pf[q]; }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int i = threadIdx.x + ((k / 8) * 2 + q) * 256; pf[q] = *reinterpret_cast<const float4*>(gp + i * 4); }
      }
      const float4 b = bq[k & 1];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].w, b.w, acc, 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    if (STREAM) {
#pragma unroll
      for (int q = 0; q < 2; ++q) { const int i = threadIdx.x + (6 + q) * 256; *reinterpret_cast<float4*>(wp + (i / 64) * LD + (i % 64) * 4) = pf[q]; }
    }
    if (EPI) {
      const float e2 = sm[(t * 32 + lane) % 4096];
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = (a[r].x + e2) - 2.f * acc[r]; if (d < best[r]) { best[r] = d; bi[r] = t * 32 + (lane & 31); } }
    } else {
      if (acc[0] == 1234.5f) best[0] = acc[5];
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += best[r] + bi[r];
  if (s == 12345.678f) out[0] = s;
}

template <bool EPI, bool BAR, bool STREAM>
void run_tile(int waves_per_simd, const float* src, const float* cbk, float* out) {
  const int lds = waves_per_simd == 1 ? 96 * 1024 : 72 * 1024;
  hipFuncSetAttribute((const void*)spin_tile<EPI, BAR, STREAM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int nblk = 256 * waves_per_simd, tiles = 1024 / waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) spin_tile<EPI, BAR, STREAM><<<nblk, 256, lds>>>(src, cbk, out, tiles);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) spin_tile<EPI, BAR, STREAM><<<nblk, 256, lds>>>(src, cbk, out, tiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double tf = 2.0 * 32 * 32 * 2 * 128.0 * tiles * 4.0 * nblk * reps / (ms * 1e-3) / 1e12;
  printf("tile loop%s%s%s  waves/SIMD %d : %7.1f TF/s  (%.3f of 157.3)\n", EPI ? " +epilogue" : "", BAR ? " +barrier" : "", STREAM ? " +code stream" : "",
         waves_per_simd, tf, tf / 157.3);
}

template <int CHAINS>
void run(int waves_per_simd, float* out, unsigned long long* clk) {
  // one 256-thread block = one wave on each of a CU's four SIMDs; LDS is used to cap residency at `waves_per_simd` blocks per CU
  const int lds = waves_per_simd == 1 ? 96 * 1024 : waves_per_simd == 2 ? 64 * 1024 : 32 * 1024;
  hipFuncSetAttribute((const void*)spin<CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int nblk = 256 * waves_per_simd, iters = 4096 / CHAINS / waves_per_simd * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  spin<CHAINS><<<nblk, 256, lds>>>(out, clk, iters, 1.f, 1e-6f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) spin<CHAINS><<<nblk, 256, lds>>>(out, clk, iters, 1.f, 1e-6f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double flop = 2.0 * 32 * 32 * 2 * 16.0 * CHAINS * iters * 4.0 * nblk * reps;
  const double tf = flop / (ms * 1e-3) / 1e12;
  // wall_clock64 ticks at 100 MHz
  printf("chains %d  waves/SIMD %d : %7.1f TF/s  (%.3f of 157.3)   %.3f ms/launch   s_memtime/wall = %.2f -> %.0f MHz if s_memtime is the shader clock\n",
         CHAINS, waves_per_simd, tf, tf / 157.3, ms / reps, (double)h[0] / (double)h[1], 100.0 * (double)h[0] / (double)h[1]);
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 64); hipMalloc(&clk, 64);
  for (int w : {1, 2, 4}) { run<1>(w, out, clk); run<2>(w, out, clk); run<4>(w, out, clk); }
  // a long hold: does the rate sag once the power controller reacts?  (20 launches of the best shape back to back)
  for (int r = 0; r < 4; ++r) run<2>(2, out, clk);
  float* src; hipMalloc(&src, 8192 * 4);
  { float h[8192]; unsigned x = 12345u; for (int i = 0; i < 8192; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xffff) / 32768.f - 1.f; }
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice); }
  for (int w : {1, 2, 4}) { run_data<0>(w, src, out, clk); run_data<1>(w, src, out, clk); }
  float* cbk; hipMalloc(&cbk, 32 * 8192 * 4); hipMemset(cbk, 0, 32 * 8192 * 4);
  for (int w : {1, 2}) {
    run_tile<false, false, false>(w, src, cbk, out);
    run_tile<true, false, false>(w, src, cbk, out);
    run_tile<false, true, false>(w, src, cbk, out);
    run_tile<false, false, true>(w, src, cbk, out);
    run_tile<true, true, false>(w, src, cbk, out);
    run_tile<true, true, true>(w, src, cbk, out);
  }
  return 0;
}
