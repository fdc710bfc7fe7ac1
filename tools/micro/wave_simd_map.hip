// Which SIMD does wave w of a 512-thread block run on?  (gfx9 HW_ID: bits 5:4 = SIMD_ID, 3:0 = WAVE_ID, 11:8 = CU_ID)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/wave_simd_map.hip -o tools/micro/wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 8 * 4);
  hipLaunchKernelGGL(k, dim3(4), dim3(512), 0, 0, d);
  unsigned h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 8; ++w) printf("  w%d->simd%u(cu%u)", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15);
    printf("\n");
  }
  return 0;
}
