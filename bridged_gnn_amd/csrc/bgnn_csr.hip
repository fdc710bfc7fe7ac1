// Edge-list plumbing on the GPU: by-destination CSR (graph_partition), coalesce, top-k -> edges.
// Reference: Bridged-GNN/models/KTGNN.py:385-398 (+ PyG remove_self_loops/add_self_loops),
// torch_geometric.utils.coalesce call sites main_bridged_graph.py:75,:113,:193,
// main_bridged_graph.py:61-63 / :105-107 (edge assembly from top-k indices).
// Integer work, HBM-bound; sorts use rocPRIM's device radix sort (stable), everything else is
// hand-written.  All results are deterministic (no atomics decide an order).
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "bgnn_common.h"

namespace {

struct CsrWs {
  int32_t *keys_in, *keys_out, *vals_in, *vals_out, *deg, *pos;
  void* tmp;
  size_t tmp_bytes;
};

static size_t csr_tmp_bytes(int64_t M, int64_t N) {
  size_t a = 0, b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                            (size_t)M, 0, 32, (hipStream_t)0);
  (void)rocprim::exclusive_scan(nullptr, b, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)(M > N + 1 ? M : N + 1),
                          rocprim::plus<int32_t>(), (hipStream_t)0);
  return a > b ? a : b;
}

static CsrWs carve(void* ws, int64_t M, int64_t N) {
  char* p = (char*)ws;
  CsrWs w;
  auto take = [&](size_t bytes) { char* q = p; p += bgnn_align_up(bytes, 256); return q; };
  w.keys_in = (int32_t*)take(sizeof(int32_t) * M);
  w.keys_out = (int32_t*)take(sizeof(int32_t) * M);
  w.vals_in = (int32_t*)take(sizeof(int32_t) * M);
  w.vals_out = (int32_t*)take(sizeof(int32_t) * M);
  w.deg = (int32_t*)take(sizeof(int32_t) * (N + 2));
  w.pos = (int32_t*)take(sizeof(int32_t) * (M + 1));
  w.tmp_bytes = csr_tmp_bytes(M, N);
  w.tmp = take(w.tmp_bytes);
  return w;
}

// keys: destination (N = "dropped" bucket for removed self loops); values: position in the input list
__global__ void csr_keys_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t N, int rewrite,
                                int32_t* __restrict__ keys, int32_t* __restrict__ vals,
                                int32_t* __restrict__ keep, int32_t* __restrict__ deg) {
  const int64_t M = E + (rewrite ? N : 0);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < M; t += (int64_t)gridDim.x * blockDim.x) {
    int32_t k, kp;
    if (t < E) {
      const int64_t s = ei[t], d = ei[E + t];
      kp = !(rewrite && s == d);
      k = kp ? (int32_t)d : (int32_t)N;
    } else {
      kp = 1;
      k = (int32_t)(t - E);
    }
    keys[t] = k;
    vals[t] = (int32_t)t;
    keep[t] = kp;
    if (kp) atomicAdd(&deg[k], 1);   // integer histogram: order-independent
  }
}

__global__ void csr_fill_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t Eout,
                                const int32_t* __restrict__ sorted_vals, const int32_t* __restrict__ pos,
                                int32_t* __restrict__ col, int32_t* __restrict__ eperm) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < Eout; t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = sorted_vals[t];
    col[t] = v < E ? (int32_t)ei[v] : (int32_t)(v - E);
    if (eperm) eperm[t] = pos[v];
  }
}

__global__ void csr_finish_kernel(const int32_t* __restrict__ rowptr, int64_t N, int64_t* __restrict__ E_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *E_out = rowptr[N];
}

static inline unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" size_t bgnn_csr_workspace_bytes(int64_t N, int64_t E) {
  const int64_t M = E + N;
  size_t b = 0;
  b += 4 * bgnn_align_up(sizeof(int32_t) * M, 256);
  b += bgnn_align_up(sizeof(int32_t) * (N + 2), 256);
  b += bgnn_align_up(sizeof(int32_t) * (M + 1), 256);
  b += bgnn_align_up(csr_tmp_bytes(M, N), 256);
  return b + 256;
}

extern "C" int bgnn_build_dst_csr(const int64_t* edge_index, int64_t E, int64_t N, int rewrite_self_loops,
                                  int32_t* rowptr, int32_t* col, int32_t* eperm_opt, int64_t* E_out_dev,
                                  void* ws, size_t ws_bytes, void* stream) {
  if (!rowptr || !col || !E_out_dev || !ws || (E > 0 && !edge_index)) return BGNN_E_NULL;
  if (N <= 0 || E < 0 || N >= (int64_t)1 << 31 || E + N >= (int64_t)1 << 31) return BGNN_E_SHAPE;
  if (ws_bytes < bgnn_csr_workspace_bytes(N, E)) return BGNN_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t M = E + (rewrite_self_loops ? N : 0);
  CsrWs w = carve(ws, E + N, N);
  hipError_t e;
  if ((e = hipMemsetAsync(w.deg, 0, sizeof(int32_t) * (N + 2), st)) != hipSuccess) return (int)e;
  if (M == 0) {
    if ((e = hipMemsetAsync(rowptr, 0, sizeof(int32_t) * (N + 1), st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(E_out_dev, 0, sizeof(int64_t), st)) != hipSuccess) return (int)e;
    return 0;
  }
  int32_t* keep = w.keys_out;   // reused before the sort overwrites it
  hipLaunchKernelGGL(csr_keys_kernel, dim3(grid_for(M)), dim3(256), 0, st, edge_index, E, N, rewrite_self_loops,
                     w.keys_in, w.vals_in, keep, w.deg);
  BGNN_LAUNCH_CHECK();
  // pos[t] = rank of input slot t among the kept edges (position in the rewritten list)
  size_t tb = w.tmp_bytes;
  if ((e = rocprim::exclusive_scan(w.tmp, tb, keep, w.pos, 0, (size_t)M, rocprim::plus<int32_t>(), st)) != hipSuccess)
    return (int)e;
  // rowptr = exclusive scan of the in-degree histogram (N+1 entries; deg[N] is the dropped bucket -> ignored)
  tb = w.tmp_bytes;
  if ((e = rocprim::exclusive_scan(w.tmp, tb, w.deg, rowptr, 0, (size_t)(N + 1), rocprim::plus<int32_t>(), st)) != hipSuccess)
    return (int)e;
  // stable sort by destination keeps input order inside a row; appended self loops come last
  int bits = 1;
  while (((int64_t)1 << bits) <= N) ++bits;
  tb = w.tmp_bytes;
  if ((e = rocprim::radix_sort_pairs(w.tmp, tb, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)M, 0, bits, st)) != hipSuccess)
    return (int)e;
  hipLaunchKernelGGL(csr_finish_kernel, dim3(1), dim3(64), 0, st, rowptr, N, E_out_dev);
  // kept edges occupy sorted slots [0, E') because the dropped bucket N sorts last; E' <= M
  hipLaunchKernelGGL(csr_fill_kernel, dim3(grid_for(M)), dim3(256), 0, st, edge_index, E, M, w.vals_out, w.pos,
                     col, eperm_opt);
  BGNN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------ coalesce
namespace {
__global__ void coalesce_keys_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t n, int64_t* __restrict__ keys) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < E; t += (int64_t)gridDim.x * blockDim.x)
    keys[t] = ei[t] * n + ei[E + t];
}
__global__ void coalesce_flag_kernel(const int64_t* __restrict__ keys, int64_t E, int32_t* __restrict__ flag) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < E; t += (int64_t)gridDim.x * blockDim.x)
    flag[t] = (t == 0 || keys[t] != keys[t - 1]) ? 1 : 0;
}
__global__ void coalesce_write_kernel(const int64_t* __restrict__ keys, const int32_t* __restrict__ flag,
                                      const int32_t* __restrict__ pos, int64_t E, int64_t n,
                                      int64_t* __restrict__ ei, int64_t* __restrict__ E_out) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < E; t += (int64_t)gridDim.x * blockDim.x) {
    if (flag[t]) {
      const int64_t k = keys[t];
      ei[pos[t]] = k / n;
      ei[E + pos[t]] = k % n;
    }
    if (t == E - 1) *E_out = (int64_t)pos[t] + flag[t];
  }
}
static size_t coalesce_tmp_bytes(int64_t E) {
  size_t a = 0, b = 0;
  (void)rocprim::radix_sort_keys(nullptr, a, (int64_t*)nullptr, (int64_t*)nullptr, (size_t)E, 0, 64, (hipStream_t)0);
  (void)rocprim::exclusive_scan(nullptr, b, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)E, rocprim::plus<int32_t>(), (hipStream_t)0);
  return a > b ? a : b;
}
}  // namespace

extern "C" size_t bgnn_coalesce_workspace_bytes(int64_t E) {
  if (E <= 0) return 256;
  return 2 * bgnn_align_up(sizeof(int64_t) * E, 256) + 2 * bgnn_align_up(sizeof(int32_t) * E, 256) +
         bgnn_align_up(coalesce_tmp_bytes(E), 256) + 256;
}

extern "C" int bgnn_coalesce_i64(int64_t* edge_index, int64_t E, int64_t num_nodes, int64_t* E_out_dev,
                                 void* ws, size_t ws_bytes, void* stream) {
  if (!E_out_dev || !ws || (E > 0 && !edge_index)) return BGNN_E_NULL;
  if (E < 0 || num_nodes <= 0 || E >= (int64_t)1 << 31) return BGNN_E_SHAPE;
  if (num_nodes > (int64_t)3037000499LL) return BGNN_E_RANGE;   // row*n+col must fit int64
  if (ws_bytes < bgnn_coalesce_workspace_bytes(E)) return BGNN_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (E == 0) {
    if ((e = hipMemsetAsync(E_out_dev, 0, sizeof(int64_t), st)) != hipSuccess) return (int)e;
    return 0;
  }
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { char* q = p; p += bgnn_align_up(bytes, 256); return q; };
  int64_t* k0 = (int64_t*)take(sizeof(int64_t) * E);
  int64_t* k1 = (int64_t*)take(sizeof(int64_t) * E);
  int32_t* flag = (int32_t*)take(sizeof(int32_t) * E);
  int32_t* pos = (int32_t*)take(sizeof(int32_t) * E);
  size_t tmp_bytes = coalesce_tmp_bytes(E);
  void* tmp = take(tmp_bytes);
  hipLaunchKernelGGL(coalesce_keys_kernel, dim3(grid_for(E)), dim3(256), 0, st, edge_index, E, num_nodes, k0);
  int bits = 1;
  while (bits < 63 && ((int64_t)1 << bits) < num_nodes * num_nodes) ++bits;
  size_t tb = tmp_bytes;
  if ((e = rocprim::radix_sort_keys(tmp, tb, k0, k1, (size_t)E, 0, bits, st)) != hipSuccess) return (int)e;
  hipLaunchKernelGGL(coalesce_flag_kernel, dim3(grid_for(E)), dim3(256), 0, st, k1, E, flag);
  tb = tmp_bytes;
  if ((e = rocprim::exclusive_scan(tmp, tb, flag, pos, 0, (size_t)E, rocprim::plus<int32_t>(), st)) != hipSuccess) return (int)e;
  hipLaunchKernelGGL(coalesce_write_kernel, dim3(grid_for(E)), dim3(256), 0, st, k1, flag, pos, E, num_nodes, edge_index, E_out_dev);
  BGNN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------ top-k -> edges
namespace {
__global__ void topk_edges_kernel(const int64_t* __restrict__ idx, int64_t Nq, int32_t k, int64_t cand_base,
                                  int64_t query_base, int64_t* __restrict__ out) {
  const int64_t total = Nq * k;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    out[t] = idx[t] + cand_base;             // from = selected candidate
    out[total + t] = t / k + query_base;     // to   = query
  }
}
}  // namespace

extern "C" int bgnn_topk_edges_i64(const int64_t* idx, int64_t Nq, int32_t k, int64_t cand_base, int64_t query_base,
                                   int64_t* edge_index_out, void* stream) {
  if (!idx || !edge_index_out) return BGNN_E_NULL;
  if (Nq < 0 || k <= 0) return BGNN_E_SHAPE;
  if (Nq == 0) return 0;
  hipLaunchKernelGGL(topk_edges_kernel, dim3(grid_for(Nq * k)), dim3(256), 0, (hipStream_t)stream, idx, Nq, k,
                     cand_base, query_base, edge_index_out);
  BGNN_LAUNCH_CHECK();
  return 0;
}

// top-k table -> COALESCED edge list in one go.  The k candidates of a query are distinct, so the list has no duplicates and
// `coalesce` (sort by from * n + to) is a STABLE sort of the candidate ids alone when the pairs are visited query-major: 32-bit
// key/value pairs over ceil(log2 Nc) bits (3 radix passes for 1e5 candidates) instead of 64-bit keys over 2 log2 n bits (5
// passes), no flag / scan / compaction, and the edge count is known on the host (no device-to-host read).
namespace {
__global__ void topk_pairs_kernel(const int64_t* __restrict__ idx, int64_t total, int32_t k, int32_t* __restrict__ keys,
                                  int32_t* __restrict__ vals) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    keys[t] = (int32_t)idx[t];
    vals[t] = (int32_t)(t / k);
  }
}
__global__ void topk_pairs_write_kernel(const int32_t* __restrict__ keys, const int32_t* __restrict__ vals, int64_t total,
                                        int64_t cand_base, int64_t query_base, int64_t* __restrict__ out) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    out[t] = (int64_t)keys[t] + cand_base;
    out[total + t] = (int64_t)vals[t] + query_base;
  }
}
static size_t topk_sort_tmp_bytes(int64_t total) {
  size_t a = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                  (size_t)total, 0, 32, (hipStream_t)0);
  return a;
}
}  // namespace

extern "C" size_t bgnn_topk_edges_coalesced_workspace_bytes(int64_t Nq, int32_t k) {
  const int64_t total = Nq > 0 && k > 0 ? Nq * k : 0;
  if (total == 0) return 256;
  return 4 * bgnn_align_up(sizeof(int32_t) * (size_t)total, 256) + bgnn_align_up(topk_sort_tmp_bytes(total), 256) + 256;
}

extern "C" int bgnn_topk_edges_coalesced_i64(const int64_t* idx, int64_t Nq, int32_t k, int64_t Nc, int64_t cand_base,
                                             int64_t query_base, int64_t* edge_index_out, void* ws, size_t ws_bytes, void* stream) {
  if (!idx || !edge_index_out || !ws) return BGNN_E_NULL;
  if (Nq < 0 || k <= 0 || Nc < k || Nc >= (int64_t)1 << 31 || Nq >= (int64_t)1 << 31 || Nq * k >= (int64_t)1 << 31) return BGNN_E_SHAPE;
  if (ws_bytes < bgnn_topk_edges_coalesced_workspace_bytes(Nq, k)) return BGNN_E_WORKSPACE;
  if (Nq == 0) return 0;
  const int64_t total = Nq * k;
  hipStream_t st = (hipStream_t)stream;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { char* q = p; p += bgnn_align_up(bytes, 256); return q; };
  int32_t* k0 = (int32_t*)take(sizeof(int32_t) * total);
  int32_t* k1 = (int32_t*)take(sizeof(int32_t) * total);
  int32_t* v0 = (int32_t*)take(sizeof(int32_t) * total);
  int32_t* v1 = (int32_t*)take(sizeof(int32_t) * total);
  size_t tb = topk_sort_tmp_bytes(total);
  void* tmp = take(tb);
  hipLaunchKernelGGL(topk_pairs_kernel, dim3(grid_for(total)), dim3(256), 0, st, idx, total, k, k0, v0);
  int bits = 1;
  while (bits < 31 && ((int64_t)1 << bits) < Nc) ++bits;
  hipError_t e;
  if ((e = rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, (size_t)total, 0, bits, st)) != hipSuccess) return (int)e;
  hipLaunchKernelGGL(topk_pairs_write_kernel, dim3(grid_for(total)), dim3(256), 0, st, k1, v1, total, cand_base, query_base,
                     edge_index_out);
  BGNN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------ row packing (halo send lists)
namespace {
// one thread per 16-byte chunk: a wave reads whole rows (or several narrow rows) and writes one contiguous span
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t ld_src,
                                                          const int64_t* __restrict__ idx, int64_t n, int32_t c4,
                                                          int64_t src_rows, float* __restrict__ dst, int64_t ld_dst) {
  const int64_t total = n * c4;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / c4;
    const int c = (int)(t - r * c4);
    int64_t s = idx[r];
    s = s < 0 ? 0 : (s >= src_rows ? src_rows - 1 : s);     // an index outside the table must not become a fault
    *reinterpret_cast<float4*>(dst + r * ld_dst + 4 * c) = *reinterpret_cast<const float4*>(src + s * ld_src + 4 * c);
  }
}
}  // namespace

extern "C" int bgnn_gather_rows_f32(const float* src, int64_t src_rows, int64_t ld_src, const int64_t* idx, int64_t n,
                                    int32_t row_floats, float* dst, int64_t ld_dst, void* stream) {
  if (n == 0) return 0;                        // an empty send list has no storage behind its pointers
  if (!src || !idx || !dst) return BGNN_E_NULL;
  if (n < 0 || src_rows <= 0 || row_floats <= 0 || (row_floats & 3) || ld_src < row_floats || ld_dst < row_floats) return BGNN_E_SHAPE;
  if ((ld_src & 3) || (ld_dst & 3) || !bgnn_aligned16(src) || !bgnn_aligned16(dst)) return BGNN_E_ALIGN;
  if (n == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n * (row_floats / 4))), dim3(256), 0, (hipStream_t)stream,
                     src, ld_src, idx, n, row_floats / 4, src_rows, dst, ld_dst);
  BGNN_LAUNCH_CHECK();
  return 0;
}
