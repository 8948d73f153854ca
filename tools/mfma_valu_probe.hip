// What does a wave's VALU (or SALU / LDS) instruction cost the fp32 matrix pipe of the SIMD it shares with an MFMA wave?
// One 512-thread block per CU: waves 0-3 (one per SIMD) run a pure MFMA loop on 8 independent accumulator chains, waves 4-7 (their SIMD partners) run a
// loop of ONE kind of instruction until the MFMA waves are done.  Printed per (MFMA kind, partner kind): cycles per MFMA (alone: 64 for 32x32x2, 32 for
// 16x16x4), the partner's instructions per 1000 cycles, and the pipe cycles one partner instruction costs = (cycles - cycles alone) / partner instructions.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o /tmp/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { P_NONE, P_FMA, P_PKFMA, P_EXP, P_MOV, P_SALU, P_DSREAD, P_SLEEP, P_FMA_PRIO, P_COUNT };
static const char* MFNAME[] = {"32x32x2 f32", "16x16x4 f32", "32x32x16 bf16", "16x16x32 bf16"};
static const char* PNAME[] = {"nothing", "v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_mov_b32", "s_add_u32", "ds_read_b128", "s_sleep 1", "v_fma_f32 @prio3"};

template <int MF, int PK, int OWN>   // OWN: the MFMA wave itself issues OWN v_fma per MFMA (no partner needed)
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, float a0, float b0) {
  __shared__ int done_flag;
  __shared__ float4 pad[512];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (threadIdx.x == 0) done_flag = 0;
  pad[threadIdx.x] = make_float4(a0, b0, a0, b0);
  __syncthreads();
  const float a = a0 + lane * 1e-9f, b = b0;
  if (wave < 4) {
    f32x16 acc[8];
    f32x4 acq[8];
    float x[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { x[c] = a + c; for (int r = 0; r < 16; ++r) acc[c][r] = 0.f; for (int r = 0; r < 4; ++r) acq[c][r] = 0.f; }
    bf16x8 ha, hb;
#pragma unroll
    for (int r = 0; r < 8; ++r) { ha[r] = (__bf16)(a + r); hb[r] = (__bf16)(b * r); }
    float2 y2[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
    f32x4 dd[4] = {{a, b, a, b}, {b, a, b, a}, {a, a, b, b}, {b, b, a, a}};
    unsigned sacc2 = 0;
    const unsigned ldsaddr2 = (unsigned)(size_t)(&pad[threadIdx.x]);
    const unsigned goff = lane * 16;
    const float* gsrc = reinterpret_cast<const float*>(out) + 8192 + wave * 256;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (MF == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
          else if (MF == 1) acq[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acq[c], 0, 0, 0);
          else if (MF == 2) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc[c], 0, 0, 0);
          else acq[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acq[c], 0, 0, 0);
#pragma unroll
          for (int k = 0; k < OWN; ++k) {
            if (PK == P_NONE || PK == P_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(c + k) & 7]) : "v"(a), "v"(b));
            if (PK == P_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y2[(c + k) & 3]) : "v"(y2[(c + k + 1) & 3]));
            if (PK == P_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(c + k) & 7]));
            if (PK == P_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[(c + k) & 7]) : "v"(a));
            if (PK == P_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc2));
            if (PK == P_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(dd[(c + k) & 3]) : "v"(ldsaddr2));
            if (PK == P_SLEEP) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dd[(c + k) & 3]) : "v"(goff), "s"(gsrc));
          }
          if (OWN && PK == P_DSREAD && (c & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (OWN && PK == P_SLEEP && (c & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) { s += x[c]; for (int r = 0; r < 16; ++r) s += acc[c][r]; for (int r = 0; r < 4; ++r) s += acq[c][r]; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    s += y2[0].x + y2[1].y + y2[2].x + y2[3].y + dd[0][0] + dd[1][1] + dd[2][2] + dd[3][3] + sacc2;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { __hip_atomic_fetch_add(&done_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)iters * 64; }
    if (s == 12345.678f) out[0] = 0;
    return;
  }
  float x[8];
  f32x4 d[4] = {{a, b, a, b}, {b, a, b, a}, {a, a, b, b}, {b, b, a, a}};
#pragma unroll
  for (int c = 0; c < 8; ++c) x[c] = a + c;
  float2 y[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
  unsigned sacc = 0;
  unsigned long long n = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned ldsaddr = (unsigned)(size_t)(&pad[threadIdx.x]);
  if (PK == P_FMA_PRIO) __builtin_amdgcn_s_setprio(3);
  if (PK != P_NONE && OWN == 0) {
    while (__hip_atomic_load(&done_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
      for (int r = 0; r < 32; ++r) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (PK == P_FMA || PK == P_FMA_PRIO) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
          if (PK == P_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[c & 3]) : "v"(y[(c + 1) & 3]));
          if (PK == P_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[c]));
          if (PK == P_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[c]) : "v"(a));
          if (PK == P_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
          if (PK == P_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(d[c & 3]) : "v"(ldsaddr));
          if (PK == P_SLEEP) asm volatile("s_sleep 1");
        }
        if (PK == P_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      n += 256;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = sacc;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += x[c] + y[c & 3].x + y[c & 3].y + d[c & 3][0];
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = n; }
  if (s == 12345.678f) out[0] = 0;
}

template <int MF, int PK, int OWN>
double run(unsigned long long* dout, double alone) {
  const int nblk = 256, iters = (MF == 0 || MF == 2) ? 2000 : 4000;
  for (int r = 0; r < 2; ++r) probe<MF, PK, OWN><<<nblk, 512>>>(dout, iters, 1.f, 1e-6f);
  hipDeviceSynchronize();
  static unsigned long long h[256 * 8 * 2];
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  double mc = 0, mn = 0, pc = 0, pn = 0;
  for (int b = 0; b < nblk; ++b) for (int w = 0; w < 8; ++w) { (w < 4 ? mc : pc) += h[(b * 8 + w) * 2]; (w < 4 ? mn : pn) += h[(b * 8 + w) * 2 + 1]; }
  const double cpm = mc / mn;
  if (OWN) printf("  %-12s + %d own %-20s per MFMA : %6.1f cycles per MFMA  -> %5.1f pipe cycles per own instruction\n", MFNAME[MF], OWN,
                  PK == P_NONE ? "v_fma_f32" : PK == P_SLEEP ? "global_load_dwordx4" : PNAME[PK], cpm, (cpm - alone) / OWN);
  else if (PK == P_NONE) printf("  %-12s alone                 : %6.1f cycles per MFMA\n", MFNAME[MF], cpm);
  else printf("  %-12s next to %-13s: %6.1f cycles per MFMA, partner %6.1f instr / 1000 cycles -> %5.1f pipe cycles per partner instruction\n",
              MFNAME[MF], PNAME[PK], cpm, 1000.0 * pn / pc, pn > 0 ? (cpm - alone) * (mn / mc) * pc / pn : 0.0);
  return cpm;
}

int main() {
  unsigned long long* dout; hipMalloc(&dout, 1 << 20);
  hipMemset(dout, 0, 1 << 20);
  const double a0 = run<0, P_NONE, 0>(dout, 0);
  run<0, P_FMA, 0>(dout, a0); run<0, P_PKFMA, 0>(dout, a0); run<0, P_EXP, 0>(dout, a0); run<0, P_MOV, 0>(dout, a0);
  run<0, P_SALU, 0>(dout, a0); run<0, P_DSREAD, 0>(dout, a0); run<0, P_SLEEP, 0>(dout, a0); run<0, P_FMA_PRIO, 0>(dout, a0);
  run<0, P_NONE, 1>(dout, a0); run<0, P_NONE, 2>(dout, a0); run<0, P_NONE, 4>(dout, a0); run<0, P_NONE, 8>(dout, a0);
  run<0, P_PKFMA, 1>(dout, a0); run<0, P_PKFMA, 4>(dout, a0); run<0, P_EXP, 1>(dout, a0); run<0, P_EXP, 4>(dout, a0); run<0, P_MOV, 4>(dout, a0);
  run<0, P_SALU, 1>(dout, a0); run<0, P_SALU, 4>(dout, a0); run<0, P_SALU, 8>(dout, a0); run<0, P_DSREAD, 1>(dout, a0); run<0, P_DSREAD, 2>(dout, a0);
  run<0, P_SLEEP, 1>(dout, a0); run<0, P_SLEEP, 2>(dout, a0);
  const double a1 = run<1, P_NONE, 0>(dout, 0);
  run<1, P_FMA, 0>(dout, a1); run<1, P_PKFMA, 0>(dout, a1); run<1, P_EXP, 0>(dout, a1); run<1, P_MOV, 0>(dout, a1);
  run<1, P_SALU, 0>(dout, a1); run<1, P_DSREAD, 0>(dout, a1);
  run<1, P_NONE, 1>(dout, a1); run<1, P_NONE, 2>(dout, a1); run<1, P_NONE, 4>(dout, a1);
  const double a2 = run<2, P_NONE, 0>(dout, 0);
  run<2, P_FMA, 0>(dout, a2); run<2, P_NONE, 1>(dout, a2); run<2, P_NONE, 2>(dout, a2); run<2, P_NONE, 4>(dout, a2); run<2, P_NONE, 8>(dout, a2);
  run<2, P_PKFMA, 4>(dout, a2); run<2, P_SALU, 4>(dout, a2); run<2, P_DSREAD, 1>(dout, a2); run<2, P_DSREAD, 2>(dout, a2); run<2, P_SLEEP, 1>(dout, a2);
  const double a3 = run<3, P_NONE, 0>(dout, 0);
  run<3, P_FMA, 0>(dout, a3); run<3, P_NONE, 1>(dout, a3); run<3, P_NONE, 2>(dout, a3); run<3, P_NONE, 4>(dout, a3);
  run<3, P_DSREAD, 1>(dout, a3); run<3, P_SLEEP, 1>(dout, a3);
  return 0;
}
