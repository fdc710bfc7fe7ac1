// Shared helpers for the gfx950 kernels (wave64, DPP cross-lane, error plumbing).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bgnn.h"

#define BGNN_LAUNCH_CHECK()                         \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

// hipFuncSetAttribute is per (function, device): cached per device id so that a second device driven by the same
// process gets its own call (one process normally drives one GPU; residency / CU-count caches elsewhere assume that all
// devices of the process are the same model).
constexpr int BGNN_MAX_DEVICES = 16;
static inline hipError_t bgnn_set_max_dynamic_lds(const void* fn, int bytes, int* done /*[BGNN_MAX_DEVICES], zero-initialised*/) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const bool cached = dev >= 0 && dev < BGNN_MAX_DEVICES;
  if (cached && __atomic_load_n(&done[dev], __ATOMIC_ACQUIRE) == bytes + 1) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && cached) __atomic_store_n(&done[dev], bytes + 1, __ATOMIC_RELEASE);
  return e;
}

// Zero-fill of device scratch as a kernel of our own, not hipMemsetAsync: inside a captured HIP graph (ROCm 7.2) the memset node
// of a small region replayed correctly back to back, but after eager work between two replays it filled the region with a stale
// pattern (pointer-like words) instead of zeros -- the aggregation's tile counters then started from garbage
// (tools/graph_agg_probe2.py).  `bytes` % 4 == 0, `p` 4-byte aligned.
static __global__ void bgnn_zero_words_kernel(uint32_t* __restrict__ p, size_t nwords) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t bgnn_zero_async(void* p, size_t bytes, hipStream_t st) {
  const size_t nwords = bytes / 4;
  if (nwords == 0) return hipSuccess;
  const size_t blocks = (nwords + 255) / 256;
  hipLaunchKernelGGL(bgnn_zero_words_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, (uint32_t*)p, nwords);
  return hipGetLastError();
}

static inline bool bgnn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t bgnn_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

namespace bgnn {

// ---- cross-lane moves without LDS traffic (gfx9 DPP controls) ----------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float swz_xor16(float x) {  // lane ^ 16 inside each 32-lane half
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));
}

// all-reduce (sum) over aligned groups of LF consecutive lanes, LF in {1,2,4,...,64}
template <int LF>
__device__ __forceinline__ float group_sum(float x) {
  if constexpr (LF >= 2) x += dpp_mov<0xB1>(x);    // quad_perm [1,0,3,2]  : lane^1
  if constexpr (LF >= 4) x += dpp_mov<0x4E>(x);    // quad_perm [2,3,0,1]  : lane^2
  if constexpr (LF >= 8) x += dpp_mov<0x141>(x);   // row_half_mirror      : other quad of the 8
  if constexpr (LF >= 16) x += dpp_mov<0x140>(x);  // row_mirror           : other 8 of the 16
  if constexpr (LF >= 32) x += swz_xor16(x);       // other 16 of the 32
  if constexpr (LF >= 64) x += __shfl_xor(x, 32);
  return x;
}
template <int LF>
__device__ __forceinline__ float group_max(float x) {
  if constexpr (LF >= 2) x = fmaxf(x, dpp_mov<0xB1>(x));
  if constexpr (LF >= 4) x = fmaxf(x, dpp_mov<0x4E>(x));
  if constexpr (LF >= 8) x = fmaxf(x, dpp_mov<0x141>(x));
  if constexpr (LF >= 16) x = fmaxf(x, dpp_mov<0x140>(x));
  if constexpr (LF >= 32) x = fmaxf(x, swz_xor16(x));
  if constexpr (LF >= 64) x = fmaxf(x, __shfl_xor(x, 32));
  return x;
}

// XCD-aware work split: block b runs on XCD (b % 8) (round-robin dispatch, speed only); give each
// XCD one contiguous eighth of the tile range so neighbouring tiles share that XCD's L2.
struct XcdRange {
  int64_t begin, end, step;
};
__device__ __forceinline__ XcdRange xcd_tile_range(int64_t ntiles) {
  const int nx = 8;
  int64_t per = (ntiles + nx - 1) / nx;
  int x = blockIdx.x % nx;
  int64_t slot = blockIdx.x / nx, nslots = (gridDim.x + nx - 1 - x) / nx;  // blocks landing on this XCD
  XcdRange r;
  r.begin = x * per + slot;
  r.end = min((int64_t)(x + 1) * per, ntiles);
  r.step = nslots;
  return r;
}

// XCD balance.  One contiguous eighth of the rows per XCD starves half the chip when the degree is not uniform along the
// row order (bridged graph: target rows have ~3x the in-edges of source rows: hidden aggregation 1.27 -> 1.02 ms once
// balanced).  The rows are therefore cut into 8 * XCD_NSEG contiguous segments dealt round-robin to the XCDs.  A
// kernel walks POSITIONS of its XCD's own sequence (xcd_pos_range: begin / end / step like XcdRange) and maps a
// position to a tile with xcd_tile_of (< 0: padding of the last segments).
constexpr int XCD_NSEG = 4;
__device__ __forceinline__ XcdRange xcd_pos_range(int64_t ntiles) {
  const int nx = 8;
  const int64_t per = (ntiles + nx - 1) / nx, seg_len = (per + XCD_NSEG - 1) / XCD_NSEG;
  const int x = blockIdx.x % nx;
  const int64_t slot = blockIdx.x / nx, nslots = (gridDim.x + nx - 1 - x) / nx;
  XcdRange r;
  r.begin = x * per + slot;
  r.end = x * per + XCD_NSEG * seg_len;
  r.step = nslots;
  return r;
}
__device__ __forceinline__ int64_t xcd_tile_of(int64_t pos, int64_t ntiles) {
  const int nx = 8;
  const int64_t per = (ntiles + nx - 1) / nx, seg_len = (per + XCD_NSEG - 1) / XCD_NSEG;
  const int x = blockIdx.x % nx;
  const int64_t j = pos - x * per, sg = j / seg_len;
  const int64_t t = (sg * nx + x) * seg_len + (j - sg * seg_len);
  return t < ntiles ? t : -1;
}

}  // namespace bgnn
