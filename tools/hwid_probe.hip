// Where do the 4 waves of a 256-thread workgroup land?  Prints, for a few workgroups, each wave's (XCC, SE, CU, SIMD, slot).
// Build: hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o gpurun_out/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
  extern __shared__ float smem[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;      // keep the block resident for a while
  smem[threadIdx.x] = a;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (smem[3] == 1234.5f);
  }
}
int main() {
  const int nb = 2048;
  unsigned* d; hipMalloc(&d, nb * 4 * 2 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  probe<<<nb, 256, 65536>>>(d, 20000);
  std::vector<unsigned> h(nb * 8);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int same_simd_pairs = 0, spread = 0;
  for (int b = 0; b < nb; ++b) {
    unsigned simd[4];
    for (int w = 0; w < 4; ++w) simd[w] = (h[(b * 4 + w) * 2] >> 4) & 3;
    bool distinct = simd[0] != simd[1] && simd[0] != simd[2] && simd[0] != simd[3] && simd[1] != simd[2] && simd[1] != simd[3] && simd[2] != simd[3];
    distinct ? ++spread : ++same_simd_pairs;
    if (b < 12) {
      printf("block %4d:", b);
      for (int w = 0; w < 4; ++w) {
        const unsigned v = h[(b * 4 + w) * 2];
        printf("  [xcc %u se %u cu %2u simd %u slot %u]", h[(b * 4 + w) * 2 + 1] & 15, (v >> 13) & 7, (v >> 8) & 15, (v >> 4) & 3, v & 15);
      }
      printf("\n");
    }
  }
  printf("blocks with 4 waves on 4 distinct SIMDs: %d; with two waves sharing a SIMD: %d\n", spread, same_simd_pairs);
  return 0;
}
