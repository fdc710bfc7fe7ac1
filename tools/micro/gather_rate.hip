// Micro-benchmark (GPU box): how many 128-B lines per second can the chip gather when 32 lanes read one random
// 512-B row (float4 per lane) -- the access shape of agg_wide_kernel -- as a function of the table size (L2-resident,
// MALL-resident, HBM) and of the rows in flight per lane group.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/gather_rate.hip -o tools/micro/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void gather(const float* __restrict__ T, const int* __restrict__ idx, int64_t n_idx, float* out) {
  const int lane = threadIdx.x & 31;
  const int64_t group = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5), ngroups = (int64_t)gridDim.x * 8;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = group * U; i + U <= n_idx; i += ngroups * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(T + (int64_t)idx[i + u] * 128 + lane * 4);
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 12345.678f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  const int64_t n_idx = 1 << 24;      // 16.7M row gathers = 67M lines
  float* out; hipMalloc(&out, 4096);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int64_t rows : {4096ll, 32768ll, 262144ll, 1048576ll}) {
    float* T; hipMalloc(&T, rows * 512); hipMemset(T, 0, rows * 512);
    std::vector<int> h(n_idx);
    unsigned long long x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (int)(x % (unsigned long long)rows); }
    int* idx; hipMalloc(&idx, n_idx * 4); hipMemcpy(idx, h.data(), n_idx * 4, hipMemcpyHostToDevice);
    for (int blocks_per_cu : {4, 8}) {
      for (int U : {2, 4, 8}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          hipEventRecord(s);
          const dim3 grid(256 * blocks_per_cu);
          if (U == 2) hipLaunchKernelGGL(gather<2>, grid, dim3(256), 0, 0, T, idx, n_idx, out);
          else if (U == 4) hipLaunchKernelGGL(gather<4>, grid, dim3(256), 0, 0, T, idx, n_idx, out);
          else hipLaunchKernelGGL(gather<8>, grid, dim3(256), 0, 0, T, idx, n_idx, out);
          hipEventRecord(e); hipEventSynchronize(e);
          float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
        }
        printf("table %7.1f MB  blocks/CU %d  U %d : %.3f ms  %.1f G lines/s  %.2f TB/s\n", rows * 512 / 1e6, blocks_per_cu, U, best,
               n_idx * 4 / best / 1e6, n_idx * 512.0 / best / 1e9);
      }
    }
    hipFree(T); hipFree(idx);
  }
  return 0;
}
