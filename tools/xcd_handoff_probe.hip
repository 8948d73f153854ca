// Cross-kernel hand-off probe (the shared-device anomaly).  tools/preempt_repro.py --trace names the FIRST launch whose
// output differs in a corrupted pass: `sparse_motion_kernel` -- a pure element-wise kernel whose inputs fingerprint equal.  What it reads
// was written by the launch right before it on the same stream: `normalize_kp_kernel`, ONE workgroup (one XCD), a few hundred bytes;
// sparse_motion runs on all 8 XCDs.  This program is that pattern and nothing else, no torch, no libsmx:
//
//     producer<<<1, 128>>>   writes  buf[i] = tag(iteration, i)            (540 words, like kp value + jacobian of 6 frames)
//     consumer<<<2048, 256>>> every workgroup reads the whole buffer and compares with tag(iteration, i)
//
// back to back on ONE stream (HIP orders them: barrier bit + agent-scope release / acquire), thousands of iterations, the buffer rotating
// over a few addresses like a caching allocator's.  A mismatch = the consumer saw bytes of an EARLIER iteration although the producer had
// completed: a stale line in the reading XCD's L2 / a producer write-back that had not reached memory.
//   xcd_handoff_probe <seconds> <mode> [tag]
//   mode 0 plain | 1 producer ends with __threadfence_system() | 2 consumer reads with sc1 (agent-scope relaxed atomic loads)
//        3 hipStreamSynchronize between the two launches | 4 producer on 8 workgroups (one per XCD), each writing the whole buffer
//        5 consumer starts with an agent-scope acquire fence | 6 producer stores sc1 (agent-scope relaxed atomic stores) | 7 = 2 + 6
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int N = 540;

struct Report {
  unsigned long long bad_words, bad_blocks;
  unsigned long long by_xcd[8];        // mismatching words by the reader's XCC_ID
  unsigned long long by_age[6];        // stale by 1, 2, 3, 4, >= 5 iterations, [5] = not a tag of this buffer at all
  unsigned long long launches;
};

__device__ __forceinline__ unsigned tag(unsigned it, unsigned i) { return (it << 10) | i; }

template <int MODE>
__global__ void producer(unsigned* buf, unsigned it) {
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    if (MODE == 6 || MODE == 7) __hip_atomic_store(buf + i, tag(it, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else buf[i] = tag(it, i);
  }
  if (MODE == 1) __threadfence_system();
}

template <int MODE>
__global__ __launch_bounds__(256) void consumer(const unsigned* buf, unsigned it, Report* r) {
  if (MODE == 5) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  unsigned bad = 0;
  for (int i = threadIdx.x; i < N; i += 256) {
    const unsigned got = (MODE == 2 || MODE == 7) ? __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : buf[i];
    if (got != tag(it, i)) {
      ++bad;
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      atomicAdd(&r->by_xcd[xcc & 7], 1ull);
      const unsigned git = got >> 10;
      const int age = ((got & 1023u) == (unsigned)i && git < it) ? (int)(it - git) : 0;
      atomicAdd(&r->by_age[age == 0 ? 5 : (age > 5 ? 4 : age - 1)], 1ull);
    }
  }
  if (bad) { atomicAdd(&r->bad_words, (unsigned long long)bad); }
  const unsigned any = __syncthreads_or(bad);
  if (threadIdx.x == 0 && any) atomicAdd(&r->bad_blocks, 1ull);
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int MODE>
static void run(double secs, const char* tg) {
  constexpr int NBUF = 3;
  unsigned* bufs[NBUF];
  for (int k = 0; k < NBUF; ++k) { CK(hipMalloc(&bufs[k], 4096)); CK(hipMemset(bufs[k], 0, 4096)); }
  Report* d; CK(hipMalloc(&d, sizeof(Report))); CK(hipMemset(d, 0, sizeof(Report)));
  hipStream_t s; CK(hipStreamCreate(&s));
  const double t_end = now_s() + secs;
  unsigned it = 1; unsigned long long launches = 0;
  while (now_s() < t_end) {
    for (int k = 0; k < 64; ++k, ++it) {
      unsigned* b = bufs[it % NBUF];
      producer<MODE><<<MODE == 4 ? 8 : 1, 128, 0, s>>>(b, it);
      if (MODE == 3) CK(hipStreamSynchronize(s));
      consumer<MODE><<<2048, 256, 0, s>>>(b, it, d);
      ++launches;
    }
    CK(hipStreamSynchronize(s));
  }
  Report h; CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("[%s] mode %d: %llu producer->consumer pairs, mismatching words %llu in %llu workgroups | by reader XCD:", tg, MODE, launches, h.bad_words, h.bad_blocks);
  for (int x = 0; x < 8; ++x) printf(" %llu", h.by_xcd[x]);
  printf(" | stale by 1/2/3/4/>=5 iterations: %llu %llu %llu %llu %llu, foreign: %llu\n", h.by_age[0], h.by_age[1], h.by_age[2], h.by_age[3], h.by_age[4], h.by_age[5]);
  fflush(stdout);
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: xcd_handoff_probe <seconds> <mode 0..7> [tag]\n"); return 1; }
  CK(hipSetDevice(0));
  const double secs = atof(argv[1]); const int mode = atoi(argv[2]); const char* tg = argc > 3 ? argv[3] : "probe";
  switch (mode) {
    case 0: run<0>(secs, tg); break; case 1: run<1>(secs, tg); break; case 2: run<2>(secs, tg); break; case 3: run<3>(secs, tg); break;
    case 4: run<4>(secs, tg); break; case 5: run<5>(secs, tg); break; case 6: run<6>(secs, tg); break; case 7: run<7>(secs, tg); break;
    default: return 1;
  }
  return 0;
}
