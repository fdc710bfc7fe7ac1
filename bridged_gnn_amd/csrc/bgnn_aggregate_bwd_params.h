// Parameters of the atomic-free ("pull") aggregation backward, shared by the general kernels (bgnn_aggregate_bwd.hip) and the
// 32-bit-addressed pair for the plain D <= 128 launch (bgnn_aggregate_bwd_fast.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace bgnn_bwd {

struct PullParams {
  const float* h_t2s; const float* h_s2t; int64_t ldh;
  const float* a_t2s; const float* a_s2t;
  const int32_t* rowptr; const int32_t* col; const uint8_t* mask;
  int64_t N; int32_t D; float slope;
  const float* out; int64_t ldo; const float* alpha; const float* gout; int64_t ldg;
  const int32_t* t_rowptr; const int32_t* t_eid; const int32_t* t_dst;
  uint4* rec;            // [E'][2]
  unsigned int* queue;   // [16] per-XCD dynamic tile counters (pass A: 0..7, pass B: 8..15), zeroed per call
  float* dstside;        // [N][ldh]
  float* dh_t2s; float* dh_s2t; float* da_t2s; float* da_s2t;
  // Hub rows (wide kernels; the forward's scheme, bgnn_aggregate.hip AggParams): a row is walked by one lane group, so a row of
  // ~750 edges (the Twitter_Graph stand-in's source nodes) is a chain of ~190 dependent steps.  Rows with >= hub_threshold edges
  // are skipped as rows and walked as <= 64-edge segments that ride behind the real rows of the same launch; a segment leaves
  // its partial row sums in scratch and a merge launch adds them in a fixed order (deterministic like the rest).  Pass A
  // segments destinations by in-degree (d_*), pass B sources by out-degree (s_*, offsets into the by-source arrays).
  int32_t hub_threshold;
  const int32_t* d_vnode; const int32_t* d_vbounds; int64_t d_nv; float* d_vpart;                 // [d_nv][ldh]
  const int32_t* s_vnode; const int32_t* s_vbounds; int64_t s_nv; float* s_vpartS; float* s_vpartT;   // [s_nv][ldh] each
  // agg_bwd_*_fast_kernel only (filled by pull_fast_plan): both tables inside ONE window of < 4 GB (bgnn_aggregate.hip: fast_plan)
  int64_t E;
  const char* tbl_base;
  uint32_t tbl_bytes, off_t2s, off_s2t, dead_off;
};

// Can the launch run the fast pair (no hub rows, 64 < D <= 128, windows below 4 GB, N <= 2^24)?  Fills the window fields.
bool pull_fast_plan(PullParams& p);
// pass A (by destination) + pass B (by source) on `st`; p.queue zeroed by the caller
int pull_fast_launch(const PullParams& p, hipStream_t st);

}  // namespace bgnn_bwd
