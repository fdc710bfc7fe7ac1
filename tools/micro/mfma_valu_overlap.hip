// Micro-benchmark (GPU box): do bf16 / fp32 MFMAs of one wave overlap with VALU work of ANOTHER wave on the same SIMD?
// Block = 8 waves (2 per SIMD).  mode 0: all waves MFMA chain; 1: all waves VALU chain; 2: waves 0-3 MFMA, 4-7 VALU
// (each SIMD hosts one of each).  If the pipes overlap, t(2) ~ max(t_mfma/2-ish, t_valu/2-ish) rather than their sum.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF>
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
  const bool do_valu = mode == 1 || (mode == 2 && wave >= 4);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float v0 = threadIdx.x * 1e-3f, v1 = 1.0001f, v2 = 0.5f, v3 = 0.25f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * e); }
  if (do_mfma) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if constexpr (BF) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, v2, acc, 0, 0, 0);
      }
    }
  }
  if (do_valu) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {     // 4 independent chains of dependent FMAs: 128 VALU per iteration
        v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1);
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[7] + v0 + v1 + v2 + v3;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int bf = 0; bf < 2; ++bf)
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(s);
        if (bf) hipLaunchKernelGGL(k<true>, dim3(256), dim3(512), 0, 0, mode, 2000, out);
        else hipLaunchKernelGGL(k<false>, dim3(256), dim3(512), 0, 0, mode, 2000, out);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
      }
      printf("%s mode %d (%s): %.3f ms\n", bf ? "bf16 32x32x16" : "fp32 32x32x2", mode,
             mode == 0 ? "8 waves MFMA: 16 per iter" : mode == 1 ? "8 waves VALU: 128 per iter" : "4 waves MFMA + 4 waves VALU", best);
    }
  return 0;
}
