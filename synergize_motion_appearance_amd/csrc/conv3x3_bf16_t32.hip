// 3x3 / stride 1 / pad 1 convolution on the bf16 MFMA, region-direct, 16x32-pixel tiles ("t32"): the big-launch form of
// conv3x3_bf16.hip (same call sites: ResBlock / Upsample / SFT / conv-FFN / RefineFlow 3x3 convolutions of configs[2];
// reference archs/vqgan_arch.py:168-191, archs/appmotioncodebook_arch.py:28-52).
//
// The 16x16-tile kernel reads 1 KB of LDS per v_mfma_f32_32x32x16_bf16 (2 A + 2 B fragments per 4 MFMAs), stages its weights through
// registers with ds_write_b128 and pays two barriers and a store phase per 72 MFMAs per wave: at full matrix rate its LDS traffic
// (reads + staging writes) would need ~80 % of the CU's LDS cycles, and it sits at 0.30 of the bf16 pipe.  This form cuts the LDS
// bytes per MFMA in half and takes the staging out of the instruction stream:
//   * a block owns a 16-row x 32-column output tile x 64 output channels; a wave owns FOUR consecutive tile rows (4 x 32 pixels) x 64
//     channels = 8 accumulator tiles (128 VGPRs).  An MFMA pixel fragment is one ROW of 32 pixels, so the A operand of tap (ky, kx) for
//     tile row r is the region row r + ky at column offset kx -- the same fragment serves (r, ky), (r+1, ky-1), (r+2, ky-2): per
//     16-channel step a wave reads 3 kx x 6 region rows = 18 A fragments + 9 taps x 2 = 18 weight fragments for 72 MFMAs
//     = 0.5 KB of LDS reads per MFMA (25 % of the LDS read rate at full matrix rate);
//   * the K loop walks 16-channel slices (one MFMA k-step) through a DOUBLE-buffered LDS stage (region 18x34 px x 32 B = 19.1 KB +
//     the slice's nine taps' weights 18 KB; 2 x 37.1 KB per block, two blocks per CU): slice s+1 is staged while slice s is
//     multiplied, ONE barrier per 72 MFMAs per wave and no store phase between barriers;
//   * weights are pre-packed in FRAGMENT order ([n-block][slice][tap][n-half][lane][8 bf16], smx_conv3x3_bf16_t32_pack) so a slice's
//     18 KB are one contiguous run that goes global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction): no VGPRs,
//     no ds_write, and the weight fragment reads are lane-linear (conflict-free by construction).  The DMA is issued from inline asm:
//     hipcc guards every ds_read that follows a __builtin_amdgcn_global_load_lds with s_waitcnt vmcnt(0) (it cannot prove the read
//     does not alias the DMA's destination), which would serialise the weight stream with the MFMA loop; the asm statement is older
//     than the region's own (compiler-counted) loads, so the compiler's wait for those also covers it, and an explicit vmcnt(0)
//     stands before the barrier;
//   * the region keeps 32 B per pixel UNPADDED; the 16-B half a lane reads is XOR-swizzled by bit 3 of the pixel's column: the
//     16-lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) then touch 16 distinct bank quads for every tap offset;
//   * region staging: global -> registers (issued before the slice's MFMAs) -> GroupNorm(+swish) of the producer applied
//     branch-free -> ds_write_b128 into the OTHER stage after the MFMAs (nearest-x2 upsampling folded into the addresses);
//   * epilogue: wave-private exchange through LDS (no block barrier inside), 16-B stores of 8 channels, bias / activation /
//     residual / SFT fused, Welford partials {mean, M2} per 16x32-pixel tile for the next GroupNorm.
// (n-block, tile) pairs are adjacent in the XCD-ordered block walk, so the second n-block of a tile finds the region in L2.
// Measured and left out (profiles/r04_t32_*.txt): s_setprio 1 around the MFMA groups (0 to -7 %), nt / sc1 cache policies on the region or
// weight requests and non-temporal output stores (nt on the region: -20 %: the 32-B-per-pixel slices of one 128-B line live off L1 hits), persistent blocks with and
// without the next tile's first requests issued ahead of the epilogue (-5 % ... 0: see DESIGN section 4b).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "smx.h"
#include "smx_common.h"
#include "bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TH = 16, TW = 32, RH = TH + 2, RW = TW + 2, RPX = RH * RW;        // 612 region pixels
constexpr int PXB = 32, ROWB = RW * PXB;                                          // bytes per region pixel (16 ch) / region row (1088)
constexpr int REGION_B = RPX * PXB;                                               // 19,584
constexpr int WSL_B = 9 * 2 * 1024;                                               // 18,432: [tap][n-half][64 lanes][16 B]
constexpr int STAGE_B = REGION_B + WSL_B;                                         // 38,016
constexpr int DUMMY_B = 1024;                                                     // sink for the 56 threads without a 5th region chunk (keeps the store branch-free)
constexpr int SS_B = 4096;                                                        // the image's GroupNorm {scale, shift} table: C_in <= 512 channels x 8 B
constexpr int SS_OFF = 2 * STAGE_B + DUMMY_B;
constexpr int LDS_B = SS_OFF + SS_B;                                              // 81,152 -> 2 blocks per CU (162,304 of 163,840 B)
constexpr int BN = 64, NT = 256, CSL = 16;
constexpr int CLD = BN + 4;                                                       // epilogue pitch (floats): conflict-free b128
constexpr int NIT = (RPX * 2 + NT - 1) / NT;                                      // 16-B region chunks per thread and slice: 5
constexpr int TAIL_T = RPX * 2 - (NIT - 1) * NT;                                  // threads with a 5th chunk: 200
static_assert(4 * 32 * CLD * 4 <= LDS_B, "epilogue exchange fits");

struct CP {
  const bf16_t* x; const bf16_t* wp; const float* bias; const void* res; bf16_t* y; float* stats; const float* in_ss;
  int in_swish, res_f32;
  const bf16_t* mul; int ldmul; float sft_w;
  int lda, ldc, ldres;
  int B, H, W, Cin, Cout, up2, act;
  int tiles_y, tiles_x, nnb;
};

__device__ __forceinline__ float c_act(float v, int act) {
  switch (act) {
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    case SMX_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case SMX_ACT_SWISH: return v / (1.f + expf(-v));
    case SMX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case SMX_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The K loop's memory requests are issued from inline asm and counted BY HAND (cdna_hip_programming guide 5.7: hipcc neither counts nor
// waits for them).  hipcc's own bookkeeping cannot express this pipeline: beside a pending LDS-DMA it turns every wait into vmcnt(0)
// (and guards each ds_read with one unless scoped-noalias information proves the read cannot alias the DMA), which drains the
// region requests that are meant to stay in flight for a whole step.
// One LDS-DMA: 64 lanes x 16 B from `gsrc` (per lane) to LDS [lds_dst, lds_dst + 1 KB) (wave-uniform).  M0 carries the LDS base and is
// compiler-reserved: saved and restored inside the statement.
#ifndef T32_RPOL
#define T32_RPOL ""                 // cache policy of the REGION requests (tools/conv_t32_ablate.py builds "nt" / "sc1" variants)
#endif
__device__ __forceinline__ void glds16r(const void* gsrc, unsigned lds_dst) {   // the same for a region chunk
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off " T32_RPOL "\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
#ifndef T32_WPOL
#define T32_WPOL ""                 // cache policy of the WEIGHT requests
#endif
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off " T32_WPOL "\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// a 16-B global load hipcc does not know about: the destination holds garbage until a `claim` whose count covers it
__device__ __forceinline__ void gload16(u32x4& dst, const void* gsrc) {
  asm volatile("global_load_dwordx4 %0, %1, off " T32_RPOL " ; t32-request" : "=v"(dst) : "v"(gsrc) : "memory");
}
template <int N>
__device__ __forceinline__ void claim(u32x4& r) {                          // wait until at most N requests are in flight; r is valid (and opaque) from here
  asm volatile("s_waitcnt vmcnt(%1) ; t32-claim %0" : "+v"(r) : "i"(N) : "memory");
}

// everything the K loop carries from step to step
struct KState {
  f32x16 acc[4][2];            // [tile row of the wave][n half]: [n][pixel] tiles
  u32x4 rreg[NIT];             // this thread's region chunks of the slice being staged (asm-loaded: valid only behind their `claim`)
  int goff[NIT], loff[NIT];    // their global element offsets (slice 0) / LDS byte offsets inside a stage
  unsigned okmask;             // bit k: chunk k lies inside the image (else it is conv padding: stored as zeros)
  int aoff[3];                 // A fragment offsets of the three kx (wave's first region row, lane's column, swizzled half)
  int boff;                    // lane * 16: weight fragments are lane-linear
  int sink;                    // (tid - TAIL_T) * 16: where a thread without a 5th chunk parks its store
};

// GroupNorm(+swish) of one region chunk and its store into the stage being filled.  The chunk's registers are opaque up to the asm
// statement: hipcc otherwise hoists this (register-only) transform up to the load's issue point, a step earlier and across the
// barrier, and parks the wave on the HBM round trip.  MODE: 0 raw, 1 scale / shift, 2 scale / shift + swish.
template <int K, int MODE, int NWAIT>
__device__ __forceinline__ void t32_stage_chunk(KState& st, unsigned char* __restrict__ Rn, const float* __restrict__ SSl, int c0, int tid, int sink_base) {
  claim<NWAIT>(st.rreg[K]);
  uint4 v = make_uint4(st.rreg[K].x, st.rreg[K].y, st.rreg[K].z, st.rreg[K].w);
  if (MODE >= 1) {
    float4 ssv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ssv[e] = *reinterpret_cast<const float4*>(SSl + 2 * c0 + 4 * e);
    float f[8]; unpack8(v, f);
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const float4 s4 = ssv[e >> 1];
      f[e] = fmaf(f[e], s4.x, s4.y); f[e + 1] = fmaf(f[e + 1], s4.z, s4.w);
    }
    if (MODE == 2) {
      constexpr float L2E = 1.44269504088896340736f;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * f[e]));
    }
    v = pack8(f);
  }
  if (!((st.okmask >> K) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);             // the conv's zero padding stays exactly 0
  // (a store under `if (tid < TAIL_T)` leaves its vmcnt wait inside the branch and hipcc re-waits at the next reuse of the register:
  // every thread stores, the 56 surplus ones of the last round into a sink behind the stages)
  const int off = (K < NIT - 1 || tid < TAIL_T) ? st.loff[K] : sink_base + st.sink;
  *reinterpret_cast<uint4*>(Rn + off) = v;
}

// Without a GroupNorm loader (MODE 0: SFT / fuse / Upsample / conv-FFN convolutions) the region needs no registers at all: the raw slice
// goes global -> LDS by DMA like the weights (5 more wave-instructions per wave and step), straight into the stage.  Thread t's DMA lane
// of round k fills LDS slot t + 256 k; the slot order carries the swizzle, so KState::goff is the offset of THAT slot's chunk (same
// pixel as the chunk the thread owns, possibly the other 16-B half).  Conv padding: the owner of a chunk outside the image overwrites
// it with zeros once the wave's DMAs have landed (the partner lane that may have filled the slot is in the same wave).
__device__ __forceinline__ void t32_region_dma(KState& st, const bf16_t* __restrict__ Xs, unsigned dst_stage, int wave, int tid) {
  const unsigned rdst = dst_stage + wave * 1024;
#pragma unroll
  for (int k = 0; k < NIT; ++k)
    if (k < NIT - 1 || tid < TAIL_T) glds16r(Xs + st.goff[k], rdst + k * (NT * 16));
}
__device__ __forceinline__ void t32_zero_padding(KState& st, unsigned char* __restrict__ Rn, int tid, int sink_base) {
  if (st.okmask == 0x1fu) return;                                           // (interior tiles: nothing to do)
#pragma unroll
  for (int k = 0; k < NIT; ++k)
    if (!((st.okmask >> k) & 1u) && (k < NIT - 1 || tid < TAIL_T)) *reinterpret_cast<uint4*>(Rn + st.loff[k]) = make_uint4(0u, 0u, 0u, 0u);
}

// One step = one 16-channel slice: 72 MFMAs out of stage Rd, while the other stage is filled for slice s+1 --
//   * weights: five LDS-DMAs per wave, issued first: a whole step to land;
//   * region: chunk-wise software pipeline.  Chunk k of slice s+1 was requested exactly one step ago; behind MFMA unit U_k it is
//     claimed, normalised, stored, and its registers are re-requested for slice s+2.  Every request has a full step (72 MFMAs) of
//     latency budget, the transform's VALU work is spread over the step in five pieces, and the registers are never held twice.
// Request bookkeeping (vmcnt retires in order).  Issue order of a steady-state step: D x5, L0, L1, L2, L3, L4.  In flight at unit U_k,
// oldest first: Lk..L4 of the previous step, D x5, L0..L(k-1) of this one -- Lk has landed once at most 4 + 5 = 9 remain (every k); at
// the end of the step the DMAs have landed once at most the 5 L's behind them remain.  A step that requests no region (the block's
// last slices) has 9 - k in flight behind Lk instead, and drains to 0 at its end.
// NEXT / NEXT2: slices s+1 / s+2 exist (compile time: a claim under a run-time condition makes hipcc copy the chunk's registers into the
// claim's operand BEFORE the wait -- i.e. read them while the request is in flight; tools/t32_isa_audit.py checks the generated ISA for
// exactly that).  The stage is a run-time value (s & 1) folded into four base registers; every other LDS offset is an immediate.
template <int MODE, bool NEXT, bool NEXT2, int ABL>
__device__ __forceinline__ void t32_step(KState& st, int s, const bf16_t* __restrict__ X, const unsigned char* __restrict__ wsrc,
                                         unsigned char* smem, unsigned lds0, const float* __restrict__ SSl, int wave, int tid) {
  const int cur = (s & 1) * STAGE_B, nxt = STAGE_B - cur;                    // byte offsets of the stage multiplied out of / being filled
  const unsigned char* Rd = smem + cur;
  unsigned char* Rn = smem + nxt;
  if (NEXT && !(ABL & 2)) {
    const unsigned char* src = wsrc + (long long)(s + 1) * WSL_B;
    const unsigned dst = lds0 + nxt + REGION_B;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int q = min(wave + 4 * k, 17);                                   // wave w moves the KB chunks w, w+4, ... of the slice's 18 (chunk 17 three times: same count on every wave)
      glds16(src + q * 1024, dst + q * 1024);
    }
  }
  if (MODE == 0 && NEXT && !(ABL & 1)) t32_region_dma(st, X + (s + 1) * CSL, lds0 + nxt, wave, tid);   // no loader transform: the raw region IS the stage
  const unsigned char* Wt = Rd + REGION_B + st.boff;
  const bf16x8 abl_frag = __builtin_bit_cast(bf16x8, make_uint4(0x3c003c00u + tid, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u));
  auto chunk = [&](auto kc) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    if (MODE == 0) return;
    // the sink of a thread without a 5th chunk sits behind both stages: 2 * STAGE_B from smem
    if (NEXT && !(ABL & 4)) t32_stage_chunk<K, MODE, (ABL & 3) ? 0 : (NEXT2 ? 9 : 9 - K)>(st, Rn, SSl, (s + 1) * CSL, tid, 2 * STAGE_B - nxt);
    if (NEXT && (ABL & 4)) claim<(ABL & 3) ? 0 : (NEXT2 ? 9 : 9 - K)>(st.rreg[K]);   // (a request without its claim would land in registers hipcc has given to something else)
    if (NEXT2 && !(ABL & 1)) gload16(st.rreg[K], X + st.goff[K] + (s + 2) * CSL);
  };
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    bf16x8 a[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if (ABL & 16) { a[r] = abl_frag; asm volatile("" : "+v"(a[r])); }
      else a[r] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Rd + st.aoff[kx] + r * ROWB));
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      bf16x8 b0, b1;
      if (ABL & 16) { b0 = abl_frag; b1 = abl_frag; asm volatile("" : "+v"(b0), "+v"(b1)); }
      else {
        b0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Wt + ((ky * 3 + kx) * 2 + 0) * 1024));
        b1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Wt + ((ky * 3 + kx) * 2 + 1) * 1024));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (ABL & 8) { asm volatile("" :: "v"(b0), "v"(b1), "v"(a[i + ky])); continue; }
        st.acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a[i + ky], st.acc[i][0], 0, 0, 0);   // A = weights: the tile is [n][pixel]
        st.acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a[i + ky], st.acc[i][1], 0, 0, 0);
      }

      const int u = kx * 3 + ky;                                             // MFMA unit (8 MFMAs) just issued
      if (u == 0) chunk(std::integral_constant<int, 0>{});
      if (u == 2) chunk(std::integral_constant<int, 1>{});
      if (u == 4) chunk(std::integral_constant<int, 2>{});
      if (u == 6) chunk(std::integral_constant<int, 3>{});
      if (u == 7) chunk(std::integral_constant<int, 4>{});
    }
  }
  if (MODE != 0 && NEXT2 && !(ABL & 3)) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // this wave's share of the weights of slice s+1 has landed
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE == 0 && NEXT && !(ABL & 4)) t32_zero_padding(st, Rn, tid, 2 * STAGE_B - nxt);
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);                                         // nothing of the next step above this line
}

template <int MODE, int ABL>
__device__ __forceinline__ void t32_k_loop(KState& st, const CP& p, int img, int nsl, const bf16_t* __restrict__ X, const unsigned char* __restrict__ wsrc,
                                           unsigned char* smem, int wave, int tid) {
  const unsigned lds0 = (unsigned)(uintptr_t)((lds_void*)smem);
  const float* SSl = reinterpret_cast<const float*>(smem + SS_OFF) + (tid & 1) * 16;
  if (MODE == 0) t32_region_dma(st, X, lds0, wave, tid);
  else {
#pragma unroll
    for (int k = 0; k < NIT; ++k) gload16(st.rreg[k], X + st.goff[k]);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int q = min(wave + 4 * k, 17);
    glds16(wsrc + q * 1024, lds0 + REGION_B + q * 1024);
  }
  if (MODE) {                                                                // the scale / shift table: 2 * C_in floats, one float2 per thread and round
    const float2* sp = reinterpret_cast<const float2*>(p.in_ss) + (long long)img * p.Cin;
    for (int c = tid; c < p.Cin; c += NT) reinterpret_cast<float2*>(smem + SS_OFF)[c] = sp[c];
    __syncthreads();
  }
  if (MODE == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t32_zero_padding(st, smem, tid, 2 * STAGE_B);
  } else {
    t32_stage_chunk<0, MODE, 0>(st, smem, SSl, 0, tid, 2 * STAGE_B);         // vmcnt(0): the first slice's region AND weights
    t32_stage_chunk<1, MODE, 0>(st, smem, SSl, 0, tid, 2 * STAGE_B);
    t32_stage_chunk<2, MODE, 0>(st, smem, SSl, 0, tid, 2 * STAGE_B);
    t32_stage_chunk<3, MODE, 0>(st, smem, SSl, 0, tid, 2 * STAGE_B);
    t32_stage_chunk<4, MODE, 0>(st, smem, SSl, 0, tid, 2 * STAGE_B);
    if (nsl > 1) {
#pragma unroll
      for (int k = 0; k < NIT; ++k) gload16(st.rreg[k], X + st.goff[k] + CSL);  // in flight across the barrier
    }
  }
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  int s = 0;
  for (; s + 2 < nsl; ++s) t32_step<MODE, true, true, ABL>(st, s, X, wsrc, smem, lds0, SSl, wave, tid);
  if (nsl > 1) { t32_step<MODE, true, false, ABL>(st, s, X, wsrc, smem, lds0, SSl, wave, tid); ++s; }
  t32_step<MODE, false, false, ABL>(st, s, X, wsrc, smem, lds0, SSl, wave, tid);
}

// ABL (tools builds only, -DSMX_TOOLS; tools/conv_t32_ablate.py): timing-only variants with one phase compiled out -- 1: no region global
// loads after the first slices, 2: no weight DMA after the first slice, 4: no region transform / LDS store after the first slice, 8: no MFMAs
// (fragment reads kept), 16: no fragment reads (MFMAs on stale registers), 32: no epilogue global traffic, 64: no epilogue.  Results are garbage for ABL != 0.
template <int ABL>
__global__ __launch_bounds__(NT, 2) void conv3x3_t32_kernel(CP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order: block b runs on XCD b % 8; every XCD walks a contiguous range of (image, tile, n-block) with the n-block fastest
  int logical;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int nb = logical % p.nnb; logical /= p.nnb;
  const int bx = logical % p.tiles_x; logical /= p.tiles_x;
  const int by = logical % p.tiles_y; const int img = logical / p.tiles_y;
  const int n0 = nb * BN;
  const int Hs = p.up2 ? p.H >> 1 : p.H, Ws_ = p.up2 ? p.W >> 1 : p.W;
  const bf16_t* __restrict__ X = p.x + (long long)img * Hs * Ws_ * p.lda;
  const int nsl = p.Cin / CSL;

  KState st;
  // ---- region staging: 1224 chunks of 16 B per slice; thread t owns chunks t + 256 k (its 16-B half = t & 1 for all of them) ----
  const int half = tid & 1;
  st.okmask = 0;
  st.sink = (tid - TAIL_T) * 16;
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int item = tid + NT * k, px = item >> 1;
    const int ry = px / RW, rx = px - ry * RW;
    int iy = by * TH - 1 + ry, ix = bx * TW - 1 + rx;
    const bool ok = item < RPX * 2 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    if (p.up2) { iy >>= 1; ix >>= 1; }
    // a padding chunk loads a valid address and is zeroed at store time; without a loader (DMA'd region) the chunk is the one of LDS slot
    // tid + 256 k: same pixel, 16-B half taken from the slot's parity
    st.goff[k] = ok ? (iy * Ws_ + ix) * p.lda + (p.in_ss ? half : half ^ ((rx >> 3) & 1)) * 8 : 0;
    st.loff[k] = ry * ROWB + rx * PXB + ((half ^ ((rx >> 3) & 1)) << 4);
    st.okmask |= (ok ? 1u : 0u) << k;
  }
  const int loader = p.in_ss ? (p.in_swish ? 2 : 1) : 0;
  // the GroupNorm {scale, shift} pairs of the image's C_in channels sit in LDS for the block's life (16 VGPRs per lane and 4 loads per
  // step when they travelled with every slice's region request: the kernel is at the 256-VGPR limit of two waves per SIMD)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) st.acc[i][j][r] = 0.f;
  // A fragment of (region row R, kx): lane l <-> pixel column kx + (l & 31), k half l >> 5
  const int hh = lane >> 5, pcol = lane & 31;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int col = pcol + kx;
    st.aoff[kx] = 4 * wave * ROWB + col * PXB + ((hh ^ ((col >> 3) & 1)) << 4);
  }
  st.boff = lane * 16;
  // weights: slice s of this n-block = 18 contiguous KB in the fragment-ordered pack
  const unsigned char* __restrict__ WP = reinterpret_cast<const unsigned char*>(p.wp) + (long long)nb * nsl * WSL_B + lane * 16;
  if (loader == 2) t32_k_loop<2, ABL>(st, p, img, nsl, X, WP, smem, wave, tid);
  else if (loader == 1) t32_k_loop<1, ABL>(st, p, img, nsl, X, WP, smem, wave, tid);
  else t32_k_loop<0, ABL>(st, p, img, nsl, X, WP, smem, wave, tid);
  f32x16 (&acc)[4][2] = st.acc;

  // ---- epilogue: wave-private exchange through LDS (the stages are dead: every wave is past the loop's last barrier) --------------
  // The accumulator tile is [n][pixel]: lane (pixel = l & 31, hh) holds channels 8g + 4hh + (0..3) of its pixel in registers 4g..4g+3.
  // Code size matters here (the first version unrolled every activation x alignment case into both passes: 174 KB of code, and the
  // instruction cache misses of a block's prologue + epilogue cost 20-30 us per block): ONE copy of the per-pixel code per
  // activation, selected per block; blocks with a ragged channel tail or unaligned rows take a compact generic loop.
  float* Cw = reinterpret_cast<float*>(smem) + wave * (32 * CLD);          // [32 px = one tile row][CLD]
  const int cq = lane & 7;                                                 // this thread's 8-channel chunk (same for all its pixels)
  const int nc = n0 + cq * 8;
  const bool al = (p.ldc % 8 == 0) && ((((uintptr_t)p.y) & 15) == 0) &&
                  (!p.res || (p.res_f32 ? ((p.ldres % 4 == 0) && ((((uintptr_t)p.res) & 15) == 0)) : ((p.ldres % 8 == 0) && ((((uintptr_t)p.res) & 15) == 0))));
  const bool fast = al && n0 + BN <= p.Cout;                               // block-uniform: whole 16-B chunks everywhere
  const long long ipix = (long long)img * p.H * p.W;
  const bf16_t* __restrict__ R16 = reinterpret_cast<const bf16_t*>(p.res) + ipix * p.ldres;
  const float* __restrict__ R32 = reinterpret_cast<const float*>(p.res) + ipix * p.ldres;
  const bf16_t* __restrict__ M16 = p.mul + ipix * p.ldmul;
  bf16_t* __restrict__ Y16 = p.y + ipix * p.ldc;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = (p.bias && nc + e < p.Cout) ? p.bias[nc + e] : 0.f;
  float piv[8], sm[8], sq[8];                                              // Welford partials of what is stored (shifted by the first value)
#pragma unroll
  for (int e = 0; e < 8; ++e) { piv[e] = 0.f; sm[e] = 0.f; sq[e] = 0.f; }
  // pixel of (row pass r, q): tile row 4*wave + r, column (lane >> 3) + 8 * q; one pass = one tile row of this wave (32 pixels x 64 channels)
  const int opix0 = (by * TH + 4 * wave) * p.W + bx * TW + (lane >> 3);
  auto to_lds = [&](const f32x16 (&A0)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(Cw + pcol * CLD + j * 32 + 8 * g + 4 * hh) = make_float4(A0[j][4 * g], A0[j][4 * g + 1], A0[j][4 * g + 2], A0[j][4 * g + 3]);
  };
  auto row_to_lds = [&](int r) __attribute__((always_inline)) {           // acc[] is only ever indexed by constants (a run-time index = scratch)
    if (r == 0) to_lds(acc[0]); else if (r == 1) to_lds(acc[1]); else if (r == 2) to_lds(acc[2]); else to_lds(acc[3]);
  };
  // the fast path of one row pass: ACT resolved at compile time; RES: 0 none, 1 bf16 residual, 2 SFT (res + mul), 3 fp32 residual
  auto pass_fast = [&](int r, auto act_c, auto res_c) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_c)::value, RES = decltype(res_c)::value;
    const int ob = opix0 + r * p.W;
    uint4 rq[4], mq[4];
    if (RES == 1 || RES == 2) {                                            // one HBM round trip per pass, overlapped with the exchange
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!(ABL & 32)) rq[q] = *reinterpret_cast<const uint4*>(R16 + (ob + 8 * q) * p.ldres + nc);
        if (RES == 2 && !(ABL & 32)) mq[q] = *reinterpret_cast<const uint4*>(M16 + (ob + 8 * q) * p.ldmul + nc);
      }
    }
    row_to_lds(r);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int px = (lane >> 3) + 8 * q;
      const int opix = ob + 8 * q;
      const float4 v0 = *reinterpret_cast<const float4*>(Cw + px * CLD + cq * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(Cw + px * CLD + cq * 8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] += bv[e];
        if (ACT == SMX_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
        else if (ACT == SMX_ACT_LRELU02) v[e] = fmaxf(v[e], 0.f) + 0.2f * fminf(v[e], 0.f);
        else if (ACT != SMX_ACT_NONE) v[e] = c_act(v[e], ACT);
      }
      if (RES == 2) {
        float r_[8], m[8]; unpack8(rq[q], r_); unpack8(mq[q], m);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = r_[e] + p.sft_w * (r_[e] * m[e] + v[e]);
      } else if (RES == 1) {
        float r_[8]; unpack8(rq[q], r_);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r_[e];
      } else if (RES == 3) {
        const float4 q0 = *reinterpret_cast<const float4*>(R32 + opix * p.ldres + nc), q1 = *reinterpret_cast<const float4*>(R32 + opix * p.ldres + nc + 4);
        v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w; v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
      }
      const uint4 pk = pack8(v);
      if (!(ABL & 32) || pk.x == 0x12345678u) *reinterpret_cast<uint4*>(Y16 + opix * p.ldc + nc) = pk;
      if (p.stats) {
        unpack8(pk, v);                                                    // statistics of the values as STORED (bf16-rounded)
        if (r == 0 && q == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) piv[e] = v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[e] - piv[e]; sm[e] += d; sq[e] += d * d; }
      }
    }
    __builtin_amdgcn_wave_barrier();                                       // this wave's reads of Cw are issued before the next pass overwrites it
  };
  // ragged channel tail / unaligned rows: per-element stores, activation by run-time switch, not unrolled over the pixels
  auto pass_generic = [&](int r) __attribute__((always_inline)) {
    const int ob = opix0 + r * p.W;
    row_to_lds(r);
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const int px = (lane >> 3) + 8 * q;
      const int opix = ob + 8 * q;
      const float4 v0 = *reinterpret_cast<const float4*>(Cw + px * CLD + cq * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(Cw + px * CLD + cq * 8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = c_act(v[e] + bv[e], p.act);
        if (nc + e < p.Cout) {
          if (p.res) t += p.res_f32 ? R32[opix * p.ldres + nc + e] : bf2f(R16[opix * p.ldres + nc + e]);
          const bf16_t h = f2bf(t);
          Y16[opix * p.ldc + nc + e] = h;
          t = bf2f(h);
        } else t = 0.f;
        v[e] = t;
      }
      if (p.stats) {
        if (r == 0 && q == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) piv[e] = v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[e] - piv[e]; sm[e] += d; sq[e] += d * d; }
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto all_fast = [&](auto act_c, auto res_c) __attribute__((always_inline)) {
#pragma unroll 1
    for (int r = 0; r < 4; ++r) pass_fast(r, act_c, res_c);
  };
  auto by_res = [&](auto act_c) __attribute__((always_inline)) {
    if (!p.res) all_fast(act_c, std::integral_constant<int, 0>{});
    else if (p.res_f32) all_fast(act_c, std::integral_constant<int, 3>{});
    else all_fast(act_c, std::integral_constant<int, 1>{});
  };
  if (ABL & 64) {                                                          // (tools) no epilogue at all; every accumulator stays live
#pragma unroll
    for (int i = 0; i < 4; ++i) { asm volatile("" :: "v"(acc[i][0])); asm volatile("" :: "v"(acc[i][1])); }
  }
  else if (fast && p.mul) all_fast(std::integral_constant<int, SMX_ACT_NONE>{}, std::integral_constant<int, 2>{});      // SFT: the launcher guarantees the fast layout
  else if (fast) {
    switch (p.act) {
      case SMX_ACT_NONE: by_res(std::integral_constant<int, SMX_ACT_NONE>{}); break;
      case SMX_ACT_RELU: by_res(std::integral_constant<int, SMX_ACT_RELU>{}); break;
      case SMX_ACT_LRELU02: by_res(std::integral_constant<int, SMX_ACT_LRELU02>{}); break;
      case SMX_ACT_SWISH: by_res(std::integral_constant<int, SMX_ACT_SWISH>{}); break;
      case SMX_ACT_GELU: by_res(std::integral_constant<int, SMX_ACT_GELU>{}); break;
      default: by_res(std::integral_constant<int, SMX_ACT_SIGMOID>{}); break;
    }
  } else {
#pragma unroll 1
    for (int r = 0; r < 4; ++r) pass_generic(r);
  }
  if (p.stats) {
    // per channel: Chan-merge the 32 threads (16 values each) that share this channel chunk
    __syncthreads();                                                       // every wave's Cw is dead: reuse as [32 thread groups][64 ch][2]
    float* red = reinterpret_cast<float*>(smem);
    constexpr float NV = 16.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float ms = sm[e] * (1.f / NV);
      red[((tid >> 3) * BN + cq * 8 + e) * 2] = piv[e] + ms;
      red[((tid >> 3) * BN + cq * 8 + e) * 2 + 1] = fmaxf(sq[e] - sm[e] * ms, 0.f);
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      float mean = 0.f;
#pragma unroll 8
      for (int t = 0; t < 32; ++t) mean += red[(t * BN + tid) * 2];
      mean *= (1.f / 32.f);
      float m2 = 0.f;
#pragma unroll 8
      for (int t = 0; t < 32; ++t) { const float d = red[(t * BN + tid) * 2] - mean; m2 += red[(t * BN + tid) * 2 + 1] + NV * d * d; }
      const long long chunk = ((long long)img * p.tiles_y + by) * p.tiles_x + bx;
      float* o = p.stats + (chunk * p.Cout + n0 + tid) * 2;
      o[0] = mean; o[1] = m2;
    }
  }
}

// fragment-ordered weight pack: wp[nb][s][tap][j][lane][e] = w[n = 64 nb + 32 j + (lane & 31)][tap * Cin + 16 s + 8 (lane >> 5) + e]
__global__ void conv3x3_t32_pack_kernel(const bf16_t* __restrict__ w, int ldw, uint4* __restrict__ wp, int Cin, int Cout, long long nchunks) {
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const int lane = (int)(c & 63);
  long long t = c >> 6;
  const int j = (int)(t & 1); t >>= 1;
  const int tap = (int)(t % 9); t /= 9;
  const int nsl = Cin / CSL;
  const int s = (int)(t % nsl); const int nb = (int)(t / nsl);
  const int n = nb * BN + j * 32 + (lane & 31);
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (n < Cout) {
    const bf16_t* src = w + (long long)n * ldw + tap * Cin + s * CSL + (lane >> 5) * 8;
    if ((((uintptr_t)src) & 15) == 0) v = *reinterpret_cast<const uint4*>(src);
    else {
      uint16_t h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = src[e];
      v = make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16), h[4] | ((uint32_t)h[5] << 16), h[6] | ((uint32_t)h[7] << 16));
    }
  }
  wp[c] = v;
}

}  // namespace

extern "C" long long smx_conv3x3_bf16_t32_pack_elems(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % CSL) return -1;
  return (long long)((Cout + BN - 1) / BN) * BN * 9 * Cin;
}

extern "C" int smx_conv3x3_bf16_t32_pack(const void* w, int ldw, void* wp, int Cin, int Cout, void* stream) {
  if (!w || !wp || Cin <= 0 || Cout <= 0 || Cin % CSL || ldw < 9 * Cin || (((uintptr_t)wp) & 15)) return SMX_EINVAL;
  const long long nchunks = smx_conv3x3_bf16_t32_pack_elems(Cin, Cout) / 8;
  SMX_LAUNCH(conv3x3_t32_pack_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, ldw, (uint4*)wp, Cin, Cout, nchunks);
  return smx_launch_status();
}

static int conv3x3_t32_launch(const void* x, int lda, const void* wp, const float* bias, const void* res, int res_f32, int ldres, const void* mul,
                              int ldmul, float sft_w, void* y, int ldc, int B, int H, int W, int Cin, int Cout, int up2, int act,
                              const float* in_ss, int in_swish, float* stats_part, void* stream, int abl = 0) {
  if (!x || !wp || !y || B <= 0 || Cin <= 0 || Cout <= 0) return SMX_EINVAL;
  if (mul && (!res || res_f32 || Cout % 8 || ldc % 8 || ldres % 8 || ldmul % 8 || ldmul < Cout || act != SMX_ACT_NONE ||
              ((((uintptr_t)y) | ((uintptr_t)res) | ((uintptr_t)mul)) & 15))) return SMX_EINVAL;
  if (H % TH != 0 || W % TW != 0 || Cin % CSL != 0 || lda % 8 != 0 || lda < Cin || ldc < Cout) return SMX_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)wp & 15) || (in_ss && (((uintptr_t)in_ss & 15) || Cin * 8 > SS_B)) || (res && ldres < Cout)) return SMX_EINVAL;
  if (up2 && ((H & 1) || (W & 1))) return SMX_EINVAL;
  CP p;
  p.x = (const bf16_t*)x; p.wp = (const bf16_t*)wp; p.bias = bias; p.res = res; p.res_f32 = res_f32; p.y = (bf16_t*)y; p.stats = stats_part;
  p.mul = (const bf16_t*)mul; p.ldmul = mul ? ldmul : 0; p.sft_w = sft_w;
  p.in_ss = in_ss; p.in_swish = in_swish; p.lda = lda; p.ldc = ldc; p.ldres = res ? ldres : 0;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.up2 = up2 ? 1 : 0; p.act = act;
  p.tiles_y = H / TH; p.tiles_x = W / TW; p.nnb = (Cout + BN - 1) / BN;
  const long long blocks = (long long)B * p.tiles_y * p.tiles_x * p.nnb;
  if (blocks > 2147483647LL || (long long)(up2 ? H / 2 : H) * (up2 ? W / 2 : W) * lda > 2147483647LL) return SMX_EINVAL;
  if ((long long)H * W * ldc > 2147483647LL || (long long)H * W * (res ? ldres : 0) > 2147483647LL || (long long)H * W * (mul ? ldmul : 0) > 2147483647LL) return SMX_EINVAL;
#ifdef SMX_TOOLS
#define T32_ABL_CASE(A) case A: { static bool at = false; if (!at) { SMX_HIP(hipFuncSetAttribute((const void*)conv3x3_t32_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B)); at = true; } \
    SMX_LAUNCH(conv3x3_t32_kernel<A>, dim3((unsigned)blocks), dim3(NT), LDS_B, (hipStream_t)stream, p); return smx_launch_status(); }
  switch (abl) { T32_ABL_CASE(1) T32_ABL_CASE(2) T32_ABL_CASE(3) T32_ABL_CASE(4) T32_ABL_CASE(7) T32_ABL_CASE(8) T32_ABL_CASE(16) T32_ABL_CASE(24) T32_ABL_CASE(32) T32_ABL_CASE(39) T32_ABL_CASE(63) T32_ABL_CASE(31) T32_ABL_CASE(127) T32_ABL_CASE(64) T32_ABL_CASE(103)
    case 0: break; default: return SMX_EINVAL; }
#else
  if (abl) return SMX_EINVAL;
#endif
  SMX_HIP(smx_max_dynamic_lds((const void*)conv3x3_t32_kernel<0>, LDS_B));
  SMX_LAUNCH(conv3x3_t32_kernel<0>, dim3((unsigned)blocks), dim3(NT), LDS_B, (hipStream_t)stream, p);
  return smx_launch_status();
}

#ifdef SMX_TOOLS
extern "C" int smx_conv3x3_bf16_t32_abl(const void* x, int lda, const void* wp, const float* bias, const void* res, int res_f32, int ldres,
                                        void* y, int ldc, int B, int H, int W, int Cin, int Cout, int up2, int act, const float* in_ss,
                                        int in_swish, float* stats_part, void* stream, int abl) {
  return conv3x3_t32_launch(x, lda, wp, bias, res, res_f32, ldres, nullptr, 0, 0.f, y, ldc, B, H, W, Cin, Cout, up2, act, in_ss, in_swish, stats_part, stream, abl);
}
#endif

extern "C" int smx_conv3x3_bf16_t32(const void* x, int lda, const void* wp, const float* bias, const void* res, int res_f32, int ldres,
                                    void* y, int ldc, int B, int H, int W, int Cin, int Cout, int up2, int act, const float* in_ss,
                                    int in_swish, float* stats_part, void* stream) {
  return conv3x3_t32_launch(x, lda, wp, bias, res, res_f32, ldres, nullptr, 0, 0.f, y, ldc, B, H, W, Cin, Cout, up2, act, in_ss, in_swish, stats_part, stream);
}

extern "C" int smx_conv3x3_sft_bf16_t32(const void* x, int lda, const void* wp, const float* bias, const void* dec, int lddec, const void* scale,
                                        int ldscale, float sft_w, void* y, int ldc, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!dec || !scale) return SMX_EINVAL;
  return conv3x3_t32_launch(x, lda, wp, bias, dec, 0, lddec, scale, ldscale, sft_w, y, ldc, B, H, W, Cin, Cout, 0, SMX_ACT_NONE, nullptr, 0, nullptr, stream);
}
