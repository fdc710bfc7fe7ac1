// Fused AdaptedConv attention + aggregation for gfx950 (wave64).
//
// Replaces (reference, Bridged-GNN/models/KTGNN.py): the per-edge GATv2 pre-activation gathers
// :292-293, the attention GEMVs :294-295, torch_geometric.utils.softmax :298-299 (scatter-max,
// exp, scatter-add, divide) and the two MessagePassing.propagate scatter-adds :303-305 / message
// :317-319 -- about eight passes over [E',D] temporaries -- by ONE pass over a by-destination CSR
// with an online softmax.  HBM-bound: algorithmic bytes = E'(4D+4) + N(8D+4).
//
// Mapping: a "group" of GL = LF*EP consecutive lanes owns one destination row.  LF lanes span the
// feature dimension (float4 per lane, LF*4 >= D); EP sub-groups walk different in-edges of the
// row (edge-parallel, for narrow rows); each sub-group keeps U neighbour rows in flight.
//   D=128: LF=32, EP=1 -> 2 rows per wave, 8 x 512 B gathers in flight per wave.
//   D=4  : LF=1,  EP=8 -> 8 rows per wave, each lane owns whole (16-B) neighbour rows.
// Blocks are persistent over an XCD-contiguous range of row tiles so neighbouring destination
// rows (which share in-neighbours in a bridged / kNN graph) hit the same XCD's L2.
#include <cstdlib>
#include "bgnn_common.h"

namespace {

struct AggParams {
  const float* h_t2s;
  const float* h_s2t;
  int64_t ldh;
  const float* a_t2s;
  const float* a_s2t;
  const int32_t* rowptr;
  const int32_t* col;
  const uint8_t* mask;
  int64_t row_begin;
  int64_t row_end;
  int32_t D;
  float slope;
  float* out;
  int64_t ldo;
  float* alpha;
  const float* ep_scale;
  const float* ep_shift;
  int ep_relu;
  unsigned int* tile_queue;   // optional [8], zeroed per launch: per-XCD dynamic tile counters.  Static striding lets the
                     // blocks of an XCD drift apart (the set of rows in flight outgrows the 4 MB L2); pulling the next
                     // CHUNK of 4 tiles from a shared counter keeps them on neighbouring rows (HBM traffic 7.3 -> 2.4 GB;
                     // one atomic per tile was itself the bound: same-address device atomics retire at ~11 M/s)
  double* colsum;    // optional [2*ldo + 2]: per-domain column sums (+ node counts) of the finished output rows, i.e. the
                     // domain sums the NEXT conv needs (KTGNN.py:275) without another pass over the activations
  int32_t heads;     // H convs evaluated together: tables/out are [N, H*ldh'] interleaved, a_* are [H][D]; a (row, head)
                     // pair is one virtual row of the kernel (ldh/ldo below are the strides of a VIRTUAL row)
  float* state_ms;   // [rows][2] running (max, sum) of a row whose edges are visited in two launches
  int mode;          // 0: one launch; 1: first part -> leave (m, s) in state_ms and the raw accumulator in out;
                     // 2: second part -> resume from them, then normalise + epilogue; 3 (narrow heads kernel): one launch
                     // that also leaves the finished rows' (m, s) in state_ms
  int64_t park_begin;  // mode 1: nodes below it have no second part and are finished (normalise + epilogue) right away
  int tq_chunk;      // tiles per queue claim
  int tq_interleave; // see agg_wide_kernel
  int xcd_segments;  // see agg_wide_kernel
  // Hub rows (agg_wide_kernel / agg_heads_lanes_kernel only).  A row is walked by ONE lane group, so a row of ~750 in-edges
  // (the 581 source nodes of the Twitter_Graph stand-in) is a chain of ~190 dependent gather steps that outlives every other row:
  // hub_threshold > 0 makes the launch skip rows with that many edges; they are cut into segments ("virtual rows": vrow_node[v] =
  // the real node, `rowptr` = the segment bounds inside `col`) that a second launch parks like two-part rows (mode 1), and a merge
  // kernel finishes them (bgnn_adaptedconv_aggregate_hub_f32).
  // The segments ride in the SAME launch: rows [row_end, row_end + n_vrows) of the launch are virtual (vrow_node[v] = the real
  // node, vrow_bounds[2v], [2v+1] = the segment's edges inside `col`); a virtual row always parks (m, s) in vms[v] and its raw
  // accumulator in vout[v], whatever `mode` says for the real rows.
  const int32_t* vrow_node;
  const int32_t* vrow_bounds;
  int64_t n_vrows;
  float* vout; float* vms;
  int32_t hub_threshold;
  // agg_wide_fast_kernel only (filled by fast_plan): both tables inside ONE window of < 4 GB, addressed by 32-bit byte offsets
  const char* tbl_base;
  uint32_t tbl_bytes, off_t2s, off_s2t;
  int32_t gather_hint; // 0 unknown, 1 neighbouring rows share neighbours (L2-resident gathers), 2 scattered (HBM-bound): resident blocks per CU
  uint32_t dead_off;   // row offset of a dead slot: lane base + dead_off is past the window and below 2^32, or 0 (row 0) when that does not fit
};

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

template <int LF, int EP, int U>
__global__ __launch_bounds__(256) void agg_kernel(AggParams p) {
  constexpr int GL = LF * EP;            // lanes per destination row
  constexpr int GPW = 64 / GL;           // rows per wave
  constexpr int RPB = 4 * GPW;           // rows per block iteration (4 waves)
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / GL;               // group inside the wave
  const int lg = lane % GL;              // lane inside the group
  const int sub = lg / LF;               // edge-parallel sub-group
  const int f0 = (lg % LF) * 4;          // first feature column of this lane
  const bool fvalid = f0 < p.D;          // pad lanes (LF*4 > ldh) never touch memory

  const int64_t ntiles = ((p.row_end - p.row_begin) * p.heads + RPB - 1) / RPB;
  bgnn::XcdRange tr = bgnn::xcd_pos_range(ntiles);          // positions of this XCD's segment sequence (XCD balance)
  // per-block column sums of the finished rows (per domain) live in LDS so the hot loop keeps its register budget
  __shared__ float red[2][LF * 4 + 1];
  if (p.colsum != nullptr) {
    for (int t = threadIdx.x; t < 2 * (LF * 4 + 1); t += 256) (&red[0][0])[t] = 0.f;
    __syncthreads();
  }

  __shared__ unsigned int dyn_tile;
  const int64_t xbase = tr.begin - (blockIdx.x / 8);        // first tile of this XCD's range
  int64_t tile = tr.begin - tr.step;
  int64_t chunk_left = 0;
  const int TQ_CHUNK = p.tq_chunk;                          // tiles per queue fetch (see agg_wide_kernel)
  for (;;) {
    if (p.tile_queue != nullptr) {                          // kernel-uniform
      if (chunk_left == 0) {
        __syncthreads();
        if (threadIdx.x == 0) dyn_tile = atomicAdd(&p.tile_queue[blockIdx.x % 8], 1u);
        __syncthreads();
        tile = xbase + (int64_t)dyn_tile * TQ_CHUNK;
        chunk_left = TQ_CHUNK;
      } else {
        tile += 1;
      }
      --chunk_left;
    } else {
      tile += tr.step;
    }
    if (tile >= tr.end) break;
    const int64_t gt = bgnn::xcd_tile_of(tile, ntiles);     // `tile` is a position in the XCD's sequence
    if (gt < 0) continue;
    // virtual row = (destination node, head); heads == 1: virtual row == node
    const int64_t i = p.row_begin * p.heads + gt * RPB + wave * GPW + g;
    const bool rvalid = i < p.row_end * p.heads;
    const int64_t ic = rvalid ? i : p.row_begin * p.heads;
    const int64_t node = ic / p.heads;
    const int head = (int)(ic - node * p.heads);
    const bool dom_s = p.mask[node] != 0;
    const float* __restrict__ H = dom_s ? p.h_t2s : p.h_s2t;
    const float* __restrict__ av = (dom_s ? p.a_t2s : p.a_s2t) + head * p.D;
    const int32_t beg = rvalid ? p.rowptr[node] : 0;
    const int32_t end = rvalid ? p.rowptr[node + 1] : 0;

    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), hi = a4;
    if (fvalid) {
      // attention vector is [D] dense: guard the tail when D % 4 != 0
      a4.x = av[f0];
      a4.y = f0 + 1 < p.D ? av[f0 + 1] : 0.f;
      a4.z = f0 + 2 < p.D ? av[f0 + 2] : 0.f;
      a4.w = f0 + 3 < p.D ? av[f0 + 3] : 0.f;
      hi = *reinterpret_cast<const float4*>(H + ic * p.ldh + f0);
    }

    float m = -INFINITY, s = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.mode == 2 && rvalid && sub == 0) {      // resume: sub-group 0 carries the state of the first launch
      m = p.state_ms[2 * i];
      s = p.state_ms[2 * i + 1];
      if (f0 < p.ldo) acc = *reinterpret_cast<const float4*>(p.out + i * p.ldo + f0);
    }
    const int32_t deg = end - beg;
    const int32_t niter = (deg + EP * U - 1) / (EP * U);   // uniform inside the group

    // software pipeline: neighbour ids for the next chunk are fetched while this chunk's rows fly
    int32_t nid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int32_t e = beg + sub + u * EP;
      nid[u] = e < end ? p.col[e] : -1;
    }
    for (int32_t it = 0; it < niter; ++it) {
      const int32_t e0 = beg + it * (EP * U) + sub;
      int32_t id[U];
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        id[u] = nid[u];
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id[u] >= 0 && fvalid) v[u] = *reinterpret_cast<const float4*>(H + ((int64_t)id[u] * p.heads + head) * p.ldh + f0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int32_t e = e0 + (U + u) * EP;
        nid[u] = e < end ? p.col[e] : -1;
      }
      float lg_[U];
      float cm = -INFINITY;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float t = a4.x * leaky(v[u].x + hi.x, p.slope);
        t = fmaf(a4.y, leaky(v[u].y + hi.y, p.slope), t);
        t = fmaf(a4.z, leaky(v[u].z + hi.z, p.slope), t);
        t = fmaf(a4.w, leaky(v[u].w + hi.w, p.slope), t);
        t = bgnn::group_sum<LF>(t);
        lg_[u] = id[u] >= 0 ? t : -INFINITY;
        cm = fmaxf(cm, lg_[u]);
      }
      if (p.alpha != nullptr && (lg % LF) == 0) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (id[u] >= 0) p.alpha[e0 + u * EP] = lg_[u];       // raw logit, normalised below
      }
      const float mn = fmaxf(m, cm);
      // m == mn covers the (-inf,-inf) start of an empty sub-group without producing NaN
      const float sc = (m == mn) ? 1.f : __expf(m - mn);
      s *= sc;
      acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float pe = (lg_[u] == -INFINITY) ? 0.f : __expf(lg_[u] - mn);
        s += pe;
        acc.x = fmaf(pe, v[u].x, acc.x);
        acc.y = fmaf(pe, v[u].y, acc.y);
        acc.z = fmaf(pe, v[u].z, acc.z);
        acc.w = fmaf(pe, v[u].w, acc.w);
      }
      m = mn;
    }

    // merge the EP partial (m, s, acc) states of the row
#pragma unroll
    for (int off = LF; off < GL; off <<= 1) {
      const float m2 = __shfl_xor(m, off);
      const float s2 = __shfl_xor(s, off);
      float4 b;
      b.x = __shfl_xor(acc.x, off); b.y = __shfl_xor(acc.y, off);
      b.z = __shfl_xor(acc.z, off); b.w = __shfl_xor(acc.w, off);
      const float mn = fmaxf(m, m2);
      const float c1 = (m == mn) ? 1.f : __expf(m - mn);
      const float c2 = (m2 == mn) ? 1.f : __expf(m2 - mn);
      s = s * c1 + s2 * c2;
      acc.x = acc.x * c1 + b.x * c2; acc.y = acc.y * c1 + b.y * c2;
      acc.z = acc.z * c1 + b.z * c2; acc.w = acc.w * c1 + b.w * c2;
      m = mn;
    }

    if (p.mode == 1 && node >= p.park_begin) {    // first part only: park the online-softmax state
      if (rvalid && sub == 0) {
        if (lg == 0) { p.state_ms[2 * i] = m; p.state_ms[2 * i + 1] = s; }
        if (f0 < p.ldo) *reinterpret_cast<float4*>(p.out + i * p.ldo + f0) = fvalid ? acc : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      continue;
    }
    const float inv = 1.f / (s + 1e-16f);   // PyG softmax denominator (KTGNN.py:299)
    if (p.alpha != nullptr && (lg % LF) == 0) {
      for (int32_t e = beg + sub; e < end; e += EP) p.alpha[e] = __expf(p.alpha[e] - m) * inv;
    }
    if (rvalid && sub == 0 && f0 < p.ldo) {
      float4 o = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
      if (p.ep_scale != nullptr) {
        float sc4[4], sh4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool ok = f0 + c < p.D;
          sc4[c] = ok ? p.ep_scale[f0 + c] : 0.f;
          sh4[c] = ok ? p.ep_shift[f0 + c] : 0.f;
        }
        o.x = fmaf(o.x, sc4[0], sh4[0]); o.y = fmaf(o.y, sc4[1], sh4[1]);
        o.z = fmaf(o.z, sc4[2], sh4[2]); o.w = fmaf(o.w, sc4[3], sh4[3]);
      }
      if (p.ep_relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      if (!fvalid) o = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(p.out + i * p.ldo + f0) = o;
      if (p.colsum != nullptr) {
        float* r = red[dom_s ? 0 : 1];
        unsafeAtomicAdd(&r[f0], o.x); unsafeAtomicAdd(&r[f0 + 1], o.y);     // ds_add_f32, no return value
        unsafeAtomicAdd(&r[f0 + 2], o.z); unsafeAtomicAdd(&r[f0 + 3], o.w);
        if (lg == 0) unsafeAtomicAdd(&r[LF * 4], 1.f);
      }
    }
  }
  if (p.colsum != nullptr) {
    // one hardware fp64 atomic per (block, column, domain)
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * (LF * 4 + 1); t += 256) {
      const int d = t / (LF * 4 + 1), c = t % (LF * 4 + 1);
      const double sum = (double)red[d][c];
      if (c == LF * 4) unsafeAtomicAdd(&p.colsum[2 * p.ldo + d], sum);
      else if (c < p.ldo) unsafeAtomicAdd(&p.colsum[d * p.ldo + c], sum);
    }
  }
}

// ---- wide rows (LF >= 16 lanes per row, i.e. D > 32) ------------------------------------------------------------
// Same mapping as agg_kernel<LF,1,U>, rebuilt around the VALU budget (the ISA of agg_kernel<32,1,4> spends ~220 VALU
// issues + ~30 quarter-rate integer multiplies per 4 edges; with the gathers mostly L2 hits that, not HBM, is the bound):
//  * the U per-edge logit partials are reduced with a TRANSPOSING butterfly: each xor step halves the number of live
//    values (lane keeps the partial of the edge whose bit matches its lane bit and sends the other), so after log2(U)
//    steps lane l holds the quad-sum of edge (l & (U-1)) -- U-1 DPP adds instead of U*log2(LF); the remaining
//    log2(LF/U) steps run on ONE value.  exp() is then evaluated once per lane for "its" edge (2 transcendental
//    issues per U edges instead of U+1) and the weights return to all lanes as the DPP operand of the FMAs;
//  * leaky_relu(z) = max(z, slope*z) (0 <= slope <= 1; exact) on packed pairs: v_pk_add / v_pk_mul / v_pk_fma;
//  * one neighbour id per lane per step (edge l & (U-1)), broadcast by quad_perm, instead of U guarded loads;
//  * neighbour address = base + id * stride as ONE v_mad_u64_u32; tail edges gather row 0 (valid memory) with
//    weight 0 instead of branching around the load.
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ int dpp_movi(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }

template <int K>
__device__ __forceinline__ float quad_bcast(float x) { return bgnn::dpp_mov<K * 0x55>(x); }
template <int K>
__device__ __forceinline__ int quad_bcasti(int x) { return dpp_movi<K * 0x55>(x); }

// butterfly step over lane bit `CTRL`: keep the value of my side, add the partner's partial of the same edge
template <int CTRL>
__device__ __forceinline__ float bfly(bool hi_side, float t_lo, float t_hi) {
  const float keep = hi_side ? t_hi : t_lo;
  const float send = hi_side ? t_lo : t_hi;
  return keep + bgnn::dpp_mov<CTRL>(send);
}

template <int LF, int U>
__global__ __launch_bounds__(256) void agg_wide_kernel(AggParams p) {
  static_assert(LF == 16 || LF == 32 || LF == 64, "wide rows only");
  static_assert(U == 4 || U == 8, "U");
  constexpr int GPW = 64 / LF;           // rows per wave
  constexpr int RPB = 4 * GPW;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / LF;
  const int lg = lane % LF;
  const int f0 = lg * 4;
  const bool fvalid = f0 < p.D;
  const int f0c = fvalid ? f0 : 0;       // pad lanes read column 0 (valid memory); their attention weights are 0
  const int k = lane & (U - 1);          // the edge slot of a step this lane scores
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;

  const int64_t nrows_all = (p.row_end - p.row_begin) * p.heads + p.n_vrows;      // real (row, head) pairs, then hub segments
  const int64_t ntiles = (nrows_all + RPB - 1) / RPB;
  bgnn::XcdRange tr = bgnn::xcd_tile_range(ntiles);
  __shared__ float red[2][LF * 4 + 1];
  if (p.colsum != nullptr) {
    for (int t = threadIdx.x; t < 2 * (LF * 4 + 1); t += 256) (&red[0][0])[t] = 0.f;
    __syncthreads();
  }
  __shared__ unsigned int dyn_tile;
  const int64_t xbase = tr.begin - (blockIdx.x / 8);
  const uint32_t nstride = (uint32_t)(p.heads * p.ldh * 4);   // bytes between neighbour rows of one head (host-checked < 2^32)
  const f2 sl = {p.slope, p.slope};
  // epilogue affine of this lane's four columns: loaded once (eight dependent loads per ROW sat on every row's critical path)
  float sc4[4] = {0.f, 0.f, 0.f, 0.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.ep_scale != nullptr) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (f0 + c < p.D) { sc4[c] = p.ep_scale[f0 + c]; sh4[c] = p.ep_shift[f0 + c]; }
  }
  float4 cs_s = make_float4(0.f, 0.f, 0.f, 0.f), cs_t = cs_s;   // column sums of the rows this lane wrote, per domain
  float n_s = 0.f, n_t = 0.f;
  f2 aS01 = {0.f, 0.f}, aS23 = aS01, aT01 = aS01, aT23 = aS01;  // heads == 1: this lane's columns of a_t2s / a_s2t
  if (p.heads == 1 && fvalid) {
    aS01.x = p.a_t2s[f0]; aS01.y = f0 + 1 < p.D ? p.a_t2s[f0 + 1] : 0.f;
    aS23.x = f0 + 2 < p.D ? p.a_t2s[f0 + 2] : 0.f; aS23.y = f0 + 3 < p.D ? p.a_t2s[f0 + 3] : 0.f;
    aT01.x = p.a_s2t[f0]; aT01.y = f0 + 1 < p.D ? p.a_s2t[f0 + 1] : 0.f;
    aT23.x = f0 + 2 < p.D ? p.a_s2t[f0 + 2] : 0.f; aT23.y = f0 + 3 < p.D ? p.a_s2t[f0 + 3] : 0.f;
  }
  // The queue hands out CHUNKS of TQ_CHUNK consecutive tiles: device-scope atomics on ONE address retire at only
  // ~11 M/s (90 ns each, measured), so one atomic per 8-row tile put a floor of 125k tiles x 90 ns / 8 queues = 1.4 ms
  // under the C4 launch -- the kernel took 1.45 ms for 1 to 42 in-edges per row alike.
  const int TQ_CHUNK = p.tq_chunk;
  // p.tq_interleave: claim k of an XCD does not take TQ_CHUNK CONSECUTIVE tiles but tiles r, r+G, r+2G, .. of super-chunk
  // s (k = s*G + r, G = blocks on this XCD): the blocks then sweep the same G-tile window together, phase by phase --
  // one atomic per TQ_CHUNK tiles with the L2 footprint of single-tile claims
  const int64_t G = tr.step;
  const int64_t tstride = p.tq_interleave ? G : 1;
  // XCD balance: a contiguous eighth of the rows per XCD hands the XCDs that own the high-degree domain (the bridged
  // graph's target rows have ~3x the in-edges of its source rows) most of the edges while the others idle.  With
  // p.xcd_segments = S > 1 the rows are cut into 8*S contiguous segments dealt round-robin to the XCDs: position j of
  // an XCD's own sequence is tile (j / seg_len * 8 + xcd) * seg_len + j % seg_len -- still long contiguous runs per L2.
  const int NSEG = p.xcd_segments > 1 ? p.xcd_segments : 1;
  const int64_t per_x = (ntiles + 7) / 8, seg_len = (per_x + NSEG - 1) / NSEG;
  const int64_t xend = NSEG > 1 ? xbase + (int64_t)NSEG * seg_len : tr.end;
  const int xcd = blockIdx.x % 8;
  int64_t tile = tr.begin - tr.step;
  int64_t chunk_left = 0;
  for (;;) {
    if (p.tile_queue != nullptr) {
      if (chunk_left == 0) {                     // block-uniform
        __syncthreads();
        if (threadIdx.x == 0) dyn_tile = atomicAdd(&p.tile_queue[blockIdx.x % 8], 1u);
        __syncthreads();
        if (p.tq_interleave) {
          const int64_t k = dyn_tile, sc = k / G, r = k - sc * G;
          tile = xbase + sc * (TQ_CHUNK * G) + r;
        } else {
          tile = xbase + (int64_t)dyn_tile * TQ_CHUNK;
        }
        chunk_left = TQ_CHUNK;
        if (tile >= xend) break;                 // first tile of a claim beyond the range: so is every later claim
      } else {
        tile += tstride;
      }
      --chunk_left;
      if (tile >= xend) { chunk_left = 0; continue; }     // (interleaved: a later phase of the last super-chunk)
    } else {
      tile += tr.step;
      if (tile >= xend) break;
    }
    int64_t gt = tile;
    if (NSEG > 1) {
      const int64_t j = tile - xbase, sg = j / seg_len;
      gt = (sg * 8 + xcd) * seg_len + (j - sg * seg_len);
      if (gt >= ntiles) continue;                // padding of the last segments (block-uniform)
    }
    const int64_t i0 = p.row_begin * p.heads + gt * RPB + wave * GPW + g;
    const int64_t vfirst = p.row_end * p.heads;                                         // first virtual row of the launch
    const bool in_range = i0 < vfirst + p.n_vrows;
    const bool virt = in_range && i0 >= vfirst;                                         // a hub segment (heads == 1)
    const int64_t ic = in_range ? i0 : p.row_begin * p.heads;
    const int64_t vix = virt ? i0 - vfirst : 0;
    const int64_t node = virt ? (int64_t)p.vrow_node[vix] : ic / p.heads;
    const int head = virt ? 0 : (int)(ic - node * p.heads);
    const int64_t hrow = virt ? node : ic;                                              // the row's own table row
    const bool dom_s = p.mask[node] != 0;
    const float* __restrict__ H = dom_s ? p.h_t2s : p.h_s2t;
    const float* __restrict__ av = (dom_s ? p.a_t2s : p.a_s2t) + head * p.D;
    // (virtual rows carry their own (begin, end) pair: hub rows are not adjacent in `col`)
    int32_t beg = in_range ? (virt ? p.vrow_bounds[2 * vix] : p.rowptr[node]) : 0;
    int32_t end = in_range ? (virt ? p.vrow_bounds[2 * vix + 1] : p.rowptr[node + 1]) : 0;
    const bool hub = !virt && p.hub_threshold > 0 && end - beg >= p.hub_threshold;      // left to its segments
    const bool rvalid = in_range && !hub;
    const int64_t i = virt ? vix : i0;                                                  // index into the output / state arrays
    float* __restrict__ obase = virt ? p.vout : p.out;
    float* __restrict__ msbase = virt ? p.vms : p.state_ms;
    if (hub) { beg = 0; end = 0; }
    const char* __restrict__ Hb = reinterpret_cast<const char*>(H + (int64_t)head * p.ldh + f0c);

    f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
    if (p.heads == 1) {                    // both attention vectors live in registers (loaded once per kernel)
      a01 = dom_s ? aS01 : aT01; a23 = dom_s ? aS23 : aT23;
    } else if (fvalid) {
      a01.x = av[f0];
      a01.y = f0 + 1 < p.D ? av[f0 + 1] : 0.f;
      a23.x = f0 + 2 < p.D ? av[f0 + 2] : 0.f;
      a23.y = f0 + 3 < p.D ? av[f0 + 3] : 0.f;
    }
    const float4 hi4 = *reinterpret_cast<const float4*>(H + hrow * p.ldh + f0c);
    const f2 h01 = {hi4.x, hi4.y}, h23 = {hi4.z, hi4.w};

    float m = -INFINITY, s = 0.f;          // s: per-lane partial (sum over the steps of "my" edge slot)
    f2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
    if (p.mode == 2 && rvalid && !virt) {
      m = p.state_ms[2 * i];
      if (k == 0) s = p.state_ms[2 * i + 1];
      if (f0 < p.ldo) {
        const float4 t = *reinterpret_cast<const float4*>(p.out + i * p.ldo + f0);
        acc01 = f2{t.x, t.y}; acc23 = f2{t.z, t.w};
      }
    }
    const int32_t niter = (end - beg + U - 1) / U;
    // issue the U row gathers of one step from the per-lane ids (lane l holds the id of slot l & (U-1))
    auto issue = [&](float4 (&v)[U], int32_t myid) {
      int32_t id[U];
      if constexpr (U == 4) {
        id[0] = quad_bcasti<0>(myid); id[1] = quad_bcasti<1>(myid);
        id[2] = quad_bcasti<2>(myid); id[3] = quad_bcasti<3>(myid);
      } else {
        const int32_t oth = dpp_movi<0x124>(myid);          // row_ror:4 -> the other quad of my 8 (an id of slot k^4)
        const int32_t lo = b2 ? oth : myid, hi = b2 ? myid : oth;
        id[0] = quad_bcasti<0>(lo); id[1] = quad_bcasti<1>(lo); id[2] = quad_bcasti<2>(lo); id[3] = quad_bcasti<3>(lo);
        id[4] = quad_bcasti<0>(hi); id[5] = quad_bcasti<1>(hi); id[6] = quad_bcasti<2>(hi); id[7] = quad_bcasti<3>(hi);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t r = (uint32_t)max(id[u], 0);
        v[u] = *reinterpret_cast<const float4*>(Hb + (uint64_t)r * nstride);
      }
    };
    // score + online-softmax update for one step whose rows are in v (myid: this lane's slot id, e_cur: its edge)
    auto update = [&](const float4 (&v)[U], int32_t myid, int32_t e_cur) {
      float t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        f2 z0 = f2{v[u].x, v[u].y} + h01, z1 = f2{v[u].z, v[u].w} + h23;
        z0 = __builtin_elementwise_max(z0, z0 * sl);
        z1 = __builtin_elementwise_max(z1, z1 * sl);
        f2 q = a01 * z0;
        q = __builtin_elementwise_fma(a23, z1, q);
        t[u] = q.x + q.y;
      }
      float r;
      if constexpr (U == 4) {
        const float rA = bfly<0xB1>(b0, t[0], t[1]), rB = bfly<0xB1>(b0, t[2], t[3]);
        r = bfly<0x4E>(b1, rA, rB);
        r += bgnn::dpp_mov<0x124>(r);                       // row_ror:4 (keeps lane & 3)
      } else {
        const float r0 = bfly<0xB1>(b0, t[0], t[1]), r1 = bfly<0xB1>(b0, t[2], t[3]);
        const float r2 = bfly<0xB1>(b0, t[4], t[5]), r3 = bfly<0xB1>(b0, t[6], t[7]);
        const float rA = bfly<0x4E>(b1, r0, r1), rB = bfly<0x4E>(b1, r2, r3);
        r = bfly<0x124>(b2, rA, rB);                        // rotation by 4: every partial is received exactly once
      }
      r += bgnn::dpp_mov<0x128>(r);                         // row_ror:8
      if constexpr (LF >= 32) r += bgnn::swz_xor16(r);
      if constexpr (LF >= 64) r += __shfl_xor(r, 32);
      const float l = myid >= 0 ? r : -INFINITY;
      if (p.alpha != nullptr && lg < U && myid >= 0) p.alpha[e_cur] = l;     // raw logit, normalised below

      float cm = fmaxf(l, bgnn::dpp_mov<0xB1>(l));
      cm = fmaxf(cm, bgnn::dpp_mov<0x4E>(cm));
      if constexpr (U == 8) cm = fmaxf(cm, bgnn::dpp_mov<0x124>(cm));
      const float mn = fmaxf(m, cm);
      const float sc = (m == mn) ? 1.f : __expf(m - mn);
      const float pe = (l == -INFINITY) ? 0.f : __expf(l - mn);
      s = fmaf(s, sc, pe);
      const f2 sc2 = {sc, sc};
      acc01 *= sc2; acc23 *= sc2;
      float pl = pe, ph = pe;
      if constexpr (U == 8) {
        const float oth = bgnn::dpp_mov<0x124>(pe);
        pl = b2 ? oth : pe; ph = b2 ? pe : oth;
      }
#define BGNN_ACC(u, w)                                               \
      {                                                              \
        acc01.x = fmaf(w, v[u].x, acc01.x); acc01.y = fmaf(w, v[u].y, acc01.y); \
        acc23.x = fmaf(w, v[u].z, acc23.x); acc23.y = fmaf(w, v[u].w, acc23.y); \
      }
      { const float w = quad_bcast<0>(pl); BGNN_ACC(0, w) }
      { const float w = quad_bcast<1>(pl); BGNN_ACC(1, w) }
      { const float w = quad_bcast<2>(pl); BGNN_ACC(2, w) }
      { const float w = quad_bcast<3>(pl); BGNN_ACC(3, w) }
      if constexpr (U == 8) {
        { const float w = quad_bcast<0>(ph); BGNN_ACC(4, w) }
        { const float w = quad_bcast<1>(ph); BGNN_ACC(5, w) }
        { const float w = quad_bcast<2>(ph); BGNN_ACC(6, w) }
        { const float w = quad_bcast<3>(ph); BGNN_ACC(7, w) }
      }
#undef BGNN_ACC
      m = mn;
    };
    auto slot_id = [&](int32_t e) { return e < end ? p.col[e] : -1; };

    // (issuing step it+1's gathers before scoring step it -- two steps in flight -- measured no gain on MI355X at
    //  any of U = 4/8, occupancy 4..6: the gather rate, not the latency per wave, is the bound; profiles/r01/README.md)
    int32_t e = beg + k;
    int32_t myid = slot_id(e);
    for (int32_t it = 0; it < niter; ++it) {
      float4 v[U];
      issue(v, myid);
      const int32_t nextid = slot_id(e + U);
      update(v, myid, e);
      e += U;
      myid = nextid;
    }
    // the row's denominator: sum of the U per-slot partials
    s += bgnn::dpp_mov<0xB1>(s);
    s += bgnn::dpp_mov<0x4E>(s);
    if constexpr (U == 8) s += bgnn::dpp_mov<0x124>(s);

    if (virt || (p.mode == 1 && node >= p.park_begin)) {
      if (rvalid) {
        if (lg == 0) { msbase[2 * i] = m; msbase[2 * i + 1] = s; }
        if (f0 < p.ldo)
          *reinterpret_cast<float4*>(obase + i * p.ldo + f0) =
              fvalid ? make_float4(acc01.x, acc01.y, acc23.x, acc23.y) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      continue;
    }
    const float inv = 1.f / (s + 1e-16f);   // PyG softmax denominator (KTGNN.py:299)
    if (p.alpha != nullptr && lg < U) {     // the lane that wrote a raw logit normalises it (program order, no fence)
      for (int32_t e2 = beg + lg; e2 < end; e2 += U) p.alpha[e2] = __expf(p.alpha[e2] - m) * inv;
    }
    if (rvalid && f0 < p.ldo) {
      float4 o = make_float4(acc01.x * inv, acc01.y * inv, acc23.x * inv, acc23.y * inv);
      if (p.ep_scale != nullptr) {
        o.x = fmaf(o.x, sc4[0], sh4[0]); o.y = fmaf(o.y, sc4[1], sh4[1]);
        o.z = fmaf(o.z, sc4[2], sh4[2]); o.w = fmaf(o.w, sc4[3], sh4[3]);
      }
      if (p.ep_relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      if (!fvalid) o = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(p.out + i * p.ldo + f0) = o;
      if (p.colsum != nullptr) {            // per-lane fp32 partials over the <= few hundred rows this lane finishes
        if (dom_s) { cs_s.x += o.x; cs_s.y += o.y; cs_s.z += o.z; cs_s.w += o.w; n_s += 1.f; }
        else       { cs_t.x += o.x; cs_t.y += o.y; cs_t.z += o.z; cs_t.w += o.w; n_t += 1.f; }
      }
    }
  }
  if (p.colsum != nullptr) {
    // lanes -> LDS (ds_add_f32) -> one hardware fp64 atomic per (block, column, domain)
    unsafeAtomicAdd(&red[0][f0], cs_s.x); unsafeAtomicAdd(&red[0][f0 + 1], cs_s.y);
    unsafeAtomicAdd(&red[0][f0 + 2], cs_s.z); unsafeAtomicAdd(&red[0][f0 + 3], cs_s.w);
    unsafeAtomicAdd(&red[1][f0], cs_t.x); unsafeAtomicAdd(&red[1][f0 + 1], cs_t.y);
    unsafeAtomicAdd(&red[1][f0 + 2], cs_t.z); unsafeAtomicAdd(&red[1][f0 + 3], cs_t.w);
    if (lg == 0) { unsafeAtomicAdd(&red[0][LF * 4], n_s); unsafeAtomicAdd(&red[1][LF * 4], n_t); }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * (LF * 4 + 1); t += 256) {
      const int d = t / (LF * 4 + 1), c = t % (LF * 4 + 1);
      const double sum = (double)red[d][c];
      if (c == LF * 4) unsafeAtomicAdd(&p.colsum[2 * p.ldo + d], sum);
      else if (c < p.ldo) unsafeAtomicAdd(&p.colsum[d * p.ldo + c], sum);
    }
  }
}

// ---- wide rows, the plain launch (heads == 1, mode 0, no hub segments) with 32-bit addressing -----------------------------------
// agg_wide_kernel is VALU-bound on gfx950 (PMC, C4: VALU busy 78 % of the launch, 440 M wave instructions: ~105 per step of
// U = 4 edges x 2 rows plus ~230 per row pair of set-up / epilogue), not gather-bound.  This variant drops what the plain
// launch does not need, in the SAME arithmetic order (results are bit-identical to agg_wide_kernel, tests pin that):
//  * addresses: both tables lie in one window of < 4 GB (fast_plan checks), so a neighbour row is `window + 32-bit byte
//    offset` through ONE buffer descriptor.  The lane that owns a slot multiplies its id once (v_mul_u32_u24) and the
//    four gathers of a step take their offset as `lane base + quad-broadcast(offset)`, one v_add_u32 with a DPP operand
//    each -- instead of per edge: a DPP move, a clamp and a quarter-rate v_mad_u64_u32 (4 issue slots);
//  * a slot past the row's end carries offset 0 and a `valid` bit instead of id -1 + clamp;
//  * row set-up in 32-bit integers, no (row, head) division, no virtual-row / two-part branches (ALPHA: the training forward's
//    attention coefficients, written exactly as agg_wide_kernel writes them);
//  * the per-domain column sums take the row with weight 1 / 0 (packed FMAs) instead of two divergent branches.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_movu(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true); }

template <int LF, bool ALPHA>
__global__ __launch_bounds__(256) void agg_wide_fast_kernel(AggParams p) {
  static_assert(LF == 16 || LF == 32 || LF == 64, "wide rows only");
  constexpr int U = 4;
  constexpr int GPW = 64 / LF;
  constexpr int RPB = 4 * GPW;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / LF;
  const int lg = lane % LF;
  const int f0 = lg * 4;
  const bool fvalid = f0 < p.D;
  const int f0c = fvalid ? f0 : 0;
  const int k = lane & (U - 1);
  const bool b0 = lane & 1, b1 = lane & 2;

  const int32_t row0 = (int32_t)p.row_begin, row1 = (int32_t)p.row_end;
  const int32_t ntiles = (row1 - row0 + RPB - 1) / RPB;
  __shared__ float red[2][LF * 4 + 1];
  if (p.colsum != nullptr) {
    for (int t = threadIdx.x; t < 2 * (LF * 4 + 1); t += 256) (&red[0][0])[t] = 0.f;
    __syncthreads();
  }
  __shared__ unsigned int dyn_tile;
  const uint32_t nstride = (uint32_t)(p.ldh * 4);
  const uint32_t ostride = (uint32_t)(p.ldo * 4);
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.tbl_base), (short)0, (int)p.tbl_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, (short)0, (int)((uint32_t)row1 * ostride), 0x00020000);
  const f2 sl = {p.slope, p.slope};
  float sc4[4] = {0.f, 0.f, 0.f, 0.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.ep_scale != nullptr) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (f0 + c < p.D) { sc4[c] = p.ep_scale[f0 + c]; sh4[c] = p.ep_shift[f0 + c]; }
  }
  f2 cs_s01 = {0.f, 0.f}, cs_s23 = cs_s01, cs_t01 = cs_s01, cs_t23 = cs_s01;   // column sums of the rows this lane wrote, per domain
  float n_s = 0.f, n_t = 0.f;
  f2 aS01 = {0.f, 0.f}, aS23 = aS01, aT01 = aS01, aT23 = aS01;
  if (fvalid) {
    aS01.x = p.a_t2s[f0]; aS01.y = f0 + 1 < p.D ? p.a_t2s[f0 + 1] : 0.f;
    aS23.x = f0 + 2 < p.D ? p.a_t2s[f0 + 2] : 0.f; aS23.y = f0 + 3 < p.D ? p.a_t2s[f0 + 3] : 0.f;
    aT01.x = p.a_s2t[f0]; aT01.y = f0 + 1 < p.D ? p.a_s2t[f0 + 1] : 0.f;
    aT23.x = f0 + 2 < p.D ? p.a_s2t[f0 + 2] : 0.f; aT23.y = f0 + 3 < p.D ? p.a_s2t[f0 + 3] : 0.f;
  }
  // tile scheduling: agg_wide_kernel's (per-XCD queue, interleaved claims, XCD segments), in 32-bit integers
  const int32_t TQ_CHUNK = p.tq_chunk;
  const int32_t xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const int32_t G = ((int32_t)gridDim.x + 7 - xcd) / 8;           // blocks on this XCD
  const int32_t tstride = p.tq_interleave ? G : 1;
  const int32_t NSEG = p.xcd_segments > 1 ? p.xcd_segments : 1;
  const int32_t per_x = (ntiles + 7) / 8, seg_len = (per_x + NSEG - 1) / NSEG;
  const int32_t xbase = xcd * per_x;
  const int32_t xend = NSEG > 1 ? xbase + NSEG * seg_len : min((xcd + 1) * per_x, ntiles);
  int32_t tile = xbase + slot - G;
  int32_t chunk_left = 0;
  for (;;) {
    if (p.tile_queue != nullptr) {
      if (chunk_left == 0) {                     // block-uniform
        __syncthreads();
        if (threadIdx.x == 0) dyn_tile = atomicAdd(&p.tile_queue[xcd], 1u);
        __syncthreads();
        const uint32_t kq = dyn_tile;
        if (p.tq_interleave) {
          const int32_t sc = (int32_t)kq / G, r = (int32_t)kq - sc * G;
          tile = xbase + sc * (TQ_CHUNK * G) + r;
        } else {
          tile = xbase + (int32_t)kq * TQ_CHUNK;
        }
        chunk_left = TQ_CHUNK;
        if (tile >= xend) break;
      } else {
        tile += tstride;
      }
      --chunk_left;
      if (tile >= xend) { chunk_left = 0; continue; }
    } else {
      tile += G;
      if (tile >= xend) break;
    }
    int32_t gt = tile;
    if (NSEG > 1) {
      const int32_t j = tile - xbase, sg = j / seg_len;
      gt = (sg * 8 + xcd) * seg_len + (j - sg * seg_len);
      if (gt >= ntiles) continue;                // padding of the last segments (block-uniform)
    }
    const int32_t i0 = row0 + gt * RPB + wave * GPW + g;
    const bool rvalid = i0 < row1;
    const int32_t ic = rvalid ? i0 : row0;
    const bool dom_s = p.mask[ic] != 0;
    const int32_t beg = rvalid ? p.rowptr[ic] : 0;
    const int32_t end = rvalid ? p.rowptr[ic + 1] : 0;
    const uint32_t lbase = (dom_s ? p.off_t2s : p.off_s2t) + (uint32_t)f0c * 4u;     // this lane's columns of row 0 of its table
    const f2 a01 = dom_s ? aS01 : aT01, a23 = dom_s ? aS23 : aT23;
    const float4 hi4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rt, lbase + __umul24((uint32_t)ic, nstride), 0, 0));
    f2 h01 = {hi4.x, hi4.y}, h23 = {hi4.z, hi4.w};

    float m = -INFINITY, s = 0.f;          // s: per-lane partial (sum over the steps of "my" edge slot)
    f2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
    const int32_t niter = (end - beg + U - 1) / U;
    // the four gathers of a step: offset = this lane's column base + the quad-broadcast row offset of slot u
    auto issue = [&](float4 (&v)[U], uint32_t off) {
      v[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rt, lbase + dpp_movu<0x00>(off), 0, 0));
      v[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rt, lbase + dpp_movu<0x55>(off), 0, 0));
      v[2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rt, lbase + dpp_movu<0xAA>(off), 0, 0));
      v[3] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rt, lbase + dpp_movu<0xFF>(off), 0, 0));
      // Issued HERE.  Instruction selection orders pure arithmetic freely against loads and barriers (and likes to push four
      // 16-byte loads behind the other buffer's arithmetic to save registers): the empty statement below clobbers memory, so the
      // loads above stay in front of it, and redefines the row's own h, which every score of update() reads, so that arithmetic
      // stays behind it.  No instruction, no wait (it names no register a load is writing).
      asm volatile("" : "+v"(h01), "+v"(h23) : : "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    // take the id requested by load_id.  The step's state rides through the statement so that all of update() is in front of the wait.
    auto take_id = [&](uint32_t& id) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(id), "+v"(m), "+v"(s), "+v"(acc01), "+v"(acc23) : : "memory");
    };
    // The id of a later step's slot, as INLINE ASSEMBLY: written as `p.col[..]` the compiler sinks the load (its only use is a later
    // iteration's offset) to the loop top, in front of the gathers that need it -- two dependent round trips per step.  The
    // compiler does not count this load in its s_waitcnt vmcnt(n): it then waits for one load more than it needs, never less
    // (loads return in order); the value is taken behind an explicit wait (take_id).  A slot past the end re-reads the row's
    // last id (valid memory) and is dead by its `ok` bit.
    auto load_id = [&](int32_t ee) {
      uint32_t id;
      const uint32_t eoff = (uint32_t)max(min(ee, end - 1), 0) * 4u;     // (an empty row rides along with its wave: col[0])
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("global_load_dword %0, %1, %2" : "=v"(id) : "v"(eoff), "s"(p.col) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      return id;
    };
    // dead slots gather nothing: their offset lies past the window (the buffer's range check returns zeros), see fast_plan
    auto row_off = [&](uint32_t id, bool alive) { return alive ? __umul24(id, nstride) : p.dead_off; };
    // score + online-softmax update of one step whose rows are in v
    auto update = [&](const float4 (&v)[U], bool ok, int32_t e_cur) {
      float t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        f2 z0 = f2{v[u].x, v[u].y} + h01, z1 = f2{v[u].z, v[u].w} + h23;
        z0 = __builtin_elementwise_max(z0, z0 * sl);
        z1 = __builtin_elementwise_max(z1, z1 * sl);
        f2 q = a01 * z0;
        q = __builtin_elementwise_fma(a23, z1, q);
        t[u] = q.x + q.y;
      }
      const float rA = bfly<0xB1>(b0, t[0], t[1]), rB = bfly<0xB1>(b0, t[2], t[3]);
      float r = bfly<0x4E>(b1, rA, rB);
      r += bgnn::dpp_mov<0x124>(r);                         // row_ror:4 (keeps lane & 3)
      r += bgnn::dpp_mov<0x128>(r);                         // row_ror:8
      if constexpr (LF >= 32) r += bgnn::swz_xor16(r);
      if constexpr (LF >= 64) r += __shfl_xor(r, 32);
      const float l = ok ? r : -INFINITY;
      if constexpr (ALPHA) {
        if (lg < U && ok) p.alpha[e_cur] = l;                // raw logit of my slot, normalised below by the same lane
      }
      float cm = fmaxf(l, bgnn::dpp_mov<0xB1>(l));
      cm = fmaxf(cm, bgnn::dpp_mov<0x4E>(cm));
      const float mn = fmaxf(m, cm);
      const float sc = (m == mn) ? 1.f : __expf(m - mn);
      const float pe = (l == -INFINITY) ? 0.f : __expf(l - mn);
      s = fmaf(s, sc, pe);
      const f2 sc2 = {sc, sc};
      acc01 *= sc2; acc23 *= sc2;
#define BGNN_ACC(u, w)                                               \
      {                                                              \
        acc01.x = fmaf(w, v[u].x, acc01.x); acc01.y = fmaf(w, v[u].y, acc01.y); \
        acc23.x = fmaf(w, v[u].z, acc23.x); acc23.y = fmaf(w, v[u].w, acc23.y); \
      }
#ifdef AGGF_X_NOACC      /* ablation (tools/exp_libs): a step without its 12 accumulate instructions -- is the launch VALU-bound? */
      acc01.x += pe * v[0].x + v[1].y + v[2].z + v[3].w;
#else
      { const float w = quad_bcast<0>(pe); BGNN_ACC(0, w) }
      { const float w = quad_bcast<1>(pe); BGNN_ACC(1, w) }
      { const float w = quad_bcast<2>(pe); BGNN_ACC(2, w) }
      { const float w = quad_bcast<3>(pe); BGNN_ACC(3, w) }
#endif
#undef BGNN_ACC
      m = mn;
    };
    // The step loop is WAVE-UNIFORM: every row of the wave runs the longest row's step count.  A step past a row's end is dead
    // in all four slots -- no traffic (dead_off), l = -inf, weight 0: m, s and the accumulator pass through unchanged -- and its
    // lanes would idle in lockstep anyway.  A scalar trip count keeps exec-mask bookkeeping out of the loop.
    // (Two steps in flight -- buffers A / B, the next step's gathers issued before this step is scored -- measured 0.953 vs
    //  0.933 ms at 111 VGPRs / 4 waves per SIMD; a step without its 12 accumulate instructions 0.918 vs 0.942 ms: neither the
    //  latency of a wave nor the VALU is the bound, the gather rate is.  profiles/r03/README.md)
    int32_t nw = __builtin_amdgcn_readlane(niter, 0);
    if constexpr (LF <= 32) nw = max(nw, __builtin_amdgcn_readlane(niter, 32));
    if constexpr (LF <= 16) nw = max(nw, max(__builtin_amdgcn_readlane(niter, 16), __builtin_amdgcn_readlane(niter, 48)));
    int32_t e = beg + k;
    bool ok = e < end;
    uint32_t myoff = p.dead_off;           // byte offset of my slot's neighbour row inside its table
    if (nw > 0) myoff = row_off((uint32_t)p.col[max(min(e, end - 1), 0)], ok);
    for (int32_t it = 0; it < nw; ++it) {
      float4 v[U];
      issue(v, myoff);
      const int32_t e_cur = e;
      e += U;
      const bool ok2 = e < end;
      uint32_t nextid = load_id(e);        // behind the gathers
      update(v, ok, e_cur);
      ok = ok2;
      take_id(nextid);
      myoff = row_off(nextid, ok);
    }
    s += bgnn::dpp_mov<0xB1>(s);
    s += bgnn::dpp_mov<0x4E>(s);

    const float inv = 1.f / (s + 1e-16f);   // PyG softmax denominator (KTGNN.py:299)
    if constexpr (ALPHA) {
      if (lg < U) {                          // the lane that wrote a raw logit normalises it (program order, no fence)
        // four of the lane's slots per trip: the reads of a trip are in flight together (one read-modify-write per trip was a chain
        // of dependent global round trips per row: 1.25 -> see profiles/r03/README.md)
        for (int32_t e2 = beg + lg; e2 < end; e2 += 4 * U) {
          float a[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] = e2 + j * U < end ? p.alpha[e2 + j * U] : 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (e2 + j * U < end) p.alpha[e2 + j * U] = __expf(a[j] - m) * inv;
        }
      }
    }
    if (rvalid && f0 < p.ldo) {
      float4 o = make_float4(acc01.x * inv, acc01.y * inv, acc23.x * inv, acc23.y * inv);
      if (p.ep_scale != nullptr) {
        o.x = fmaf(o.x, sc4[0], sh4[0]); o.y = fmaf(o.y, sc4[1], sh4[1]);
        o.z = fmaf(o.z, sc4[2], sh4[2]); o.w = fmaf(o.w, sc4[3], sh4[3]);
      }
      if (p.ep_relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      if (!fvalid) o = make_float4(0.f, 0.f, 0.f, 0.f);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), ro, __umul24((uint32_t)i0, ostride) + (uint32_t)f0 * 4u, 0, 0);
      if (p.colsum != nullptr) {            // weight 1 / 0 per domain: o * 1 + cs == cs + o, o * 0 + cs == cs (finite o)
        const float ws = dom_s ? 1.f : 0.f, wt = dom_s ? 0.f : 1.f;
        const f2 ws2 = {ws, ws}, wt2 = {wt, wt}, o01 = {o.x, o.y}, o23 = {o.z, o.w};
        cs_s01 = __builtin_elementwise_fma(o01, ws2, cs_s01); cs_s23 = __builtin_elementwise_fma(o23, ws2, cs_s23);
        cs_t01 = __builtin_elementwise_fma(o01, wt2, cs_t01); cs_t23 = __builtin_elementwise_fma(o23, wt2, cs_t23);
        n_s += ws; n_t += wt;
      }
    }
  }
  if (p.colsum != nullptr) {
    unsafeAtomicAdd(&red[0][f0], cs_s01.x); unsafeAtomicAdd(&red[0][f0 + 1], cs_s01.y);
    unsafeAtomicAdd(&red[0][f0 + 2], cs_s23.x); unsafeAtomicAdd(&red[0][f0 + 3], cs_s23.y);
    unsafeAtomicAdd(&red[1][f0], cs_t01.x); unsafeAtomicAdd(&red[1][f0 + 1], cs_t01.y);
    unsafeAtomicAdd(&red[1][f0 + 2], cs_t23.x); unsafeAtomicAdd(&red[1][f0 + 3], cs_t23.y);
    if (lg == 0) { unsafeAtomicAdd(&red[0][LF * 4], n_s); unsafeAtomicAdd(&red[1][LF * 4], n_t); }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * (LF * 4 + 1); t += 256) {
      const int d = t / (LF * 4 + 1), c = t % (LF * 4 + 1);
      const double sum = (double)red[d][c];
      if (c == LF * 4) unsafeAtomicAdd(&p.colsum[2 * p.ldo + d], sum);
      else if (c < p.ldo) unsafeAtomicAdd(&p.colsum[d * p.ldo + c], sum);
    }
  }
}

// Narrow multi-head variant (D <= 4, i.e. one float4 per head): ONE walk of a destination's in-edges serves all HEADS convs
// (KT-GNN's classifier stage: clf_base(x), clf_target(x), clf_target(T(x)) share the graph; tables interleaved per node).
// One lane per (edge slot, head): lanes lg = sub * HEADS + h of a row's group.  The HEADS lanes of an edge read the
// neighbour's HEADS consecutive 16-byte pieces -- one cache line per edge; a single lane walking all heads (round 1's
// agg_heads_kernel) issued HEADS separate line requests per edge, and the narrow gathers are bound by the L1's line-request
// rate, not by bytes (C4: 0.240 -> 0.188 ms).  The neighbour id is a broadcast load, and a lane carries one head's state.
template <int HEADS, int EP, int U>
__global__ __launch_bounds__(256) void agg_heads_lanes_kernel(AggParams p) {
  constexpr int GL = EP * HEADS, GPW = 64 / GL, RPB = 4 * GPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / GL, lg = lane % GL;
  const int sub = lg / HEADS, h = lg % HEADS;
  const bool lane_on = g < GPW;                       // 64 % GL lanes idle
  const int64_t ntiles = (p.row_end - p.row_begin + p.n_vrows + RPB - 1) / RPB;      // real rows, then hub segments
  bgnn::XcdRange tr = bgnn::xcd_pos_range(ntiles);    // positions of this XCD's segment sequence (XCD balance)
  const int64_t rs = (int64_t)HEADS * p.ldh;         // floats between consecutive nodes of a table
  for (int64_t pos = tr.begin; pos < tr.end; pos += tr.step) {
    const int64_t tile = bgnn::xcd_tile_of(pos, ntiles);
    if (tile < 0) continue;
    const int64_t i = p.row_begin + tile * RPB + wave * GPW + g;
    const bool in_range = lane_on && i < p.row_end + p.n_vrows;
    const bool virt = in_range && i >= p.row_end;                          // a hub segment (AggParams)
    const int64_t vix = virt ? i - p.row_end : 0;
    const int64_t ic = in_range ? i : p.row_begin;
    const int64_t node = virt ? (int64_t)p.vrow_node[vix] : ic;
    const bool dom_s = p.mask[node] != 0;
    const float* __restrict__ H = dom_s ? p.h_t2s : p.h_s2t;
    const float* __restrict__ av = dom_s ? p.a_t2s : p.a_s2t;
    int32_t beg = in_range ? (virt ? p.vrow_bounds[2 * vix] : p.rowptr[ic]) : 0;
    int32_t end = in_range ? (virt ? p.vrow_bounds[2 * vix + 1] : p.rowptr[ic + 1]) : 0;
    const bool hub = !virt && p.hub_threshold > 0 && end - beg >= p.hub_threshold;
    const bool rvalid = in_range && !hub;
    const int64_t oi = virt ? vix : i;                                     // index into the output / state arrays
    float* __restrict__ obase = virt ? p.vout : p.out;
    float* __restrict__ msbase = virt ? p.vms : p.state_ms;
    if (hub) { beg = 0; end = 0; }
    float4 a4;
    a4.x = av[h * p.D];
    a4.y = p.D > 1 ? av[h * p.D + 1] : 0.f;
    a4.z = p.D > 2 ? av[h * p.D + 2] : 0.f;
    a4.w = p.D > 3 ? av[h * p.D + 3] : 0.f;
    const float4 hi = *reinterpret_cast<const float4*>(H + node * rs + h * p.ldh);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float m = -INFINITY, s = 0.f;
    if (p.mode == 2 && rvalid && !virt && sub == 0) {
      m = p.state_ms[2 * (i * HEADS + h)];
      s = p.state_ms[2 * (i * HEADS + h) + 1];
      acc = *reinterpret_cast<const float4*>(p.out + (i * HEADS + h) * p.ldo);
    }
    const int32_t niter = (end - beg + EP * U - 1) / (EP * U);
    for (int32_t it = 0; it < niter; ++it) {
      int32_t id[U];
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t e = beg + it * (EP * U) + sub + u * EP;
        id[u] = e < end ? p.col[e] : -1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = id[u] >= 0 ? *reinterpret_cast<const float4*>(H + (int64_t)id[u] * rs + h * p.ldh) : make_float4(0.f, 0.f, 0.f, 0.f);
      float lg_[U], cm = -INFINITY;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float t = a4.x * leaky(v[u].x + hi.x, p.slope);
        t = fmaf(a4.y, leaky(v[u].y + hi.y, p.slope), t);
        t = fmaf(a4.z, leaky(v[u].z + hi.z, p.slope), t);
        t = fmaf(a4.w, leaky(v[u].w + hi.w, p.slope), t);
        lg_[u] = id[u] >= 0 ? t : -INFINITY;
        cm = fmaxf(cm, lg_[u]);
      }
      const float mn = fmaxf(m, cm);
      const float sc = (m == mn) ? 1.f : __expf(m - mn);
      s *= sc;
      acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float pe = (lg_[u] == -INFINITY) ? 0.f : __expf(lg_[u] - mn);
        s += pe;
        acc.x = fmaf(pe, v[u].x, acc.x); acc.y = fmaf(pe, v[u].y, acc.y);
        acc.z = fmaf(pe, v[u].z, acc.z); acc.w = fmaf(pe, v[u].w, acc.w);
      }
      m = mn;
    }
    // merge the EP edge slots of a (row, head): partner lanes are HEADS * off apart (every lane of the wave takes part)
#pragma unroll
    for (int off = 1; off < EP; off <<= 1) {
      const int src = (sub ^ off) * HEADS + h + g * GL;
      const int from = lane_on ? src : lane;
      const float m2 = __shfl(m, from), s2 = __shfl(s, from);
      float4 b;
      b.x = __shfl(acc.x, from); b.y = __shfl(acc.y, from); b.z = __shfl(acc.z, from); b.w = __shfl(acc.w, from);
      const float mn = fmaxf(m, m2);
      const float c1 = (m == mn) ? 1.f : __expf(m - mn), c2 = (m2 == mn) ? 1.f : __expf(m2 - mn);
      s = s * c1 + s2 * c2;
      acc.x = acc.x * c1 + b.x * c2; acc.y = acc.y * c1 + b.y * c2;
      acc.z = acc.z * c1 + b.z * c2; acc.w = acc.w * c1 + b.w * c2;
      m = mn;
    }
    if (rvalid && sub == 0) {
      float* o = obase + (oi * HEADS + h) * p.ldo;
      if (virt || (p.mode == 1 && i >= p.park_begin)) {
        msbase[2 * (oi * HEADS + h)] = m;
        msbase[2 * (oi * HEADS + h) + 1] = s;
        *reinterpret_cast<float4*>(o) = acc;
      } else {
        const float inv = 1.f / (s + 1e-16f);
        float4 r = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        if (p.mode == 3) {       // training: the softmax state of the finished row, for the heads backward
          p.state_ms[2 * (i * HEADS + h)] = m;
          p.state_ms[2 * (i * HEADS + h) + 1] = s;
        }
        if (p.ep_relu == 2) {    // log_softmax over the head's D classes (KTGNN.py:435), row-local
          float mx = r.x;
          if (p.D > 1) mx = fmaxf(mx, r.y);
          if (p.D > 2) mx = fmaxf(mx, r.z);
          if (p.D > 3) mx = fmaxf(mx, r.w);
          float se = expf(r.x - mx);
          if (p.D > 1) se += expf(r.y - mx);
          if (p.D > 2) se += expf(r.z - mx);
          if (p.D > 3) se += expf(r.w - mx);
          const float lse = logf(se);
          r.x = r.x - mx - lse;
          r.y = p.D > 1 ? r.y - mx - lse : 0.f;
          r.z = p.D > 2 ? r.z - mx - lse : 0.f;
          r.w = p.D > 3 ? r.w - mx - lse : 0.f;
        }
        *reinterpret_cast<float4*>(o) = r;
      }
    }
  }
}

// Tiles per queue claim.  4 amortises the same-address atomic (~90 ns each); measured on a rank's share of C4
// (tools/rank_of_8_time.py): 1 -> 0.80 ms, 2 -> 0.69, 4 -> 0.67, 8 -> 0.70 per forward, also for the short launches.
inline int tq_chunk_for(int64_t ntiles, int64_t grid) {
  static const int forced = [] { const char* e = getenv("BGNN_AGG_CHUNK"); return e ? atoi(e) : 0; }();
  (void)ntiles; (void)grid;
  return forced > 0 ? forced : 4;
}

template <int HEADS, int EP, int U>
int launch_heads(const AggParams& p, hipStream_t st) {
  // (EP, U) = (2, 4) from a sweep on C4 (three heads): 4/4 0.222, 4/2 0.247, 8/2 0.378, 2/2 0.223, 2/4 0.188, 2/8 0.192, 1/4 0.189,
  // 1/8 0.184 ms (EP = 1 leaves a long row's whole edge list to one lane triple); tools/heads_fwd_time.py times this launch
  constexpr int RPB = 4 * (64 / (EP * HEADS));
  static const int cap = [] {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2048;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, agg_heads_lanes_kernel<HEADS, EP, U>, 256, 0) != hipSuccess || per_cu < 1) return 2048;
    if (per_cu > 8) per_cu = 8;
    return per_cu * prop.multiProcessorCount / 8 * 8;
  }();
  const int64_t ntiles = (p.row_end - p.row_begin + p.n_vrows + RPB - 1) / RPB;
  int64_t grid = ntiles < cap ? (ntiles + 7) / 8 * 8 : cap;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL((agg_heads_lanes_kernel<HEADS, EP, U>), dim3((unsigned)grid), dim3(256), 0, st, p);
  BGNN_LAUNCH_CHECK();
  return 0;
}

// Persistent grid = exactly the blocks that are co-resident (occupancy x CUs): a larger grid would
// leave late-starting blocks walking their strided tiles alone, long after their neighbours' rows
// left the L2.  The hardware query is immutable, so it is cached per instantiation.
template <int LF, int EP, int U>
int resident_blocks() {
  static const int nb = [] {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2048;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, agg_kernel<LF, EP, U>, 256, 0) != hipSuccess || per_cu < 1)
      return 2048;
    if (per_cu > 8) per_cu = 8;
    return per_cu * prop.multiProcessorCount / 8 * 8;
  }();
  return nb;
}

template <int LF, int EP, int U>
int launch(const AggParams& p, hipStream_t st) {
  constexpr int RPB = 4 * (64 / (LF * EP));
  int64_t ntiles = ((p.row_end - p.row_begin) * p.heads + RPB - 1) / RPB;
  const int64_t cap = resident_blocks<LF, EP, U>();
  int64_t grid = ntiles < cap ? (ntiles + 7) / 8 * 8 : cap;   // multiple of 8 (XCD split)
  if (grid < 8) grid = 8;
  AggParams q = p;
  q.tq_chunk = tq_chunk_for(ntiles, grid);
  hipLaunchKernelGGL((agg_kernel<LF, EP, U>), dim3((unsigned)grid), dim3(256), 0, st, q);
  BGNN_LAUNCH_CHECK();
  return 0;
}

template <int LF, int U>
int launch_wide(const AggParams& p, hipStream_t st) {
  constexpr int RPB = 4 * (64 / LF);
  static const int cap = [] {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2048;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, agg_wide_kernel<LF, U>, 256, 0) != hipSuccess || per_cu < 1) return 2048;
    if (per_cu > 8) per_cu = 8;
    const char* e = getenv("BGNN_AGG_BLOCKS_PER_CU");        // sweep knob (tools): fewer resident blocks = a smaller window of rows per XCD
    if (e && atoi(e) > 0 && atoi(e) < per_cu) per_cu = atoi(e);
    return per_cu * prop.multiProcessorCount / 8 * 8;
  }();
  const int64_t ntiles = ((p.row_end - p.row_begin) * p.heads + p.n_vrows + RPB - 1) / RPB;
  int64_t grid = ntiles < cap ? (ntiles + 7) / 8 * 8 : cap;
  if (grid < 8) grid = 8;
  AggParams q = p;
  q.tq_chunk = tq_chunk_for(ntiles, grid);
  static const int il = [] { const char* e = getenv("BGNN_AGG_INTERLEAVE"); return e ? atoi(e) : 1; }();
  q.tq_interleave = il;
  static const int nseg = [] { const char* e = getenv("BGNN_XCD_SEGMENTS"); return e ? atoi(e) : 8; }();
  q.xcd_segments = nseg;
  hipLaunchKernelGGL((agg_wide_kernel<LF, U>), dim3((unsigned)grid), dim3(256), 0, st, q);
  BGNN_LAUNCH_CHECK();
  return 0;
}

// Can this launch run agg_wide_fast_kernel?  table_rows = rows of the two tables (every id in `col` is below it; 0 = unknown).
// Fills the window fields of `p`.  BGNN_AGG_FAST=0 keeps the general kernel (tests compare the two bit for bit).
static bool fast_plan(AggParams& p, int64_t table_rows) {
  static const bool on = [] { const char* e = getenv("BGNN_AGG_FAST"); return !(e && atoi(e) == 0); }();
  if (!on || table_rows <= 0) return false;
  if (p.heads != 1 || p.mode != 0 || p.n_vrows != 0 || p.hub_threshold != 0) return false;
  const int64_t lim24 = (int64_t)1 << 24, lim32 = ((int64_t)1 << 32) - 1;       // v_mul_u32_u24 operands / 32-bit byte offsets
  if (table_rows > lim24 || p.row_end > table_rows || p.ldh * 4 >= lim24 || p.ldo * 4 >= lim24) return false;
  const char* a = reinterpret_cast<const char*>(p.h_t2s);
  const char* b = reinterpret_cast<const char*>(p.h_s2t);
  const char* lo = a < b ? a : b;
  const int64_t span = (int64_t)((a < b ? b : a) - lo) + table_rows * p.ldh * 4;
  if (span > lim32 || p.row_end * p.ldo * 4 > lim32) return false;
  p.tbl_base = lo;
  p.tbl_bytes = (uint32_t)span;
  p.off_t2s = (uint32_t)(a - lo);
  p.off_s2t = (uint32_t)(b - lo);
  // lane base < (offset of the upper table) + (row stride); a dead slot's address must stay in [window, 2^32)
  const int64_t lane_max = (span - table_rows * p.ldh * 4) + p.ldh * 4, dead = (((int64_t)1 << 32) - 16) - lane_max;
  p.dead_off = dead >= span ? (uint32_t)dead : 0u;
  return true;
}

template <int LF, bool ALPHA>
int launch_wide_fast(const AggParams& p, hipStream_t st) {
  constexpr int RPB = 4 * (64 / LF);
  // resident blocks per CU: everything that fits (5 at 82 VGPRs) when the gathers live off the L2s -- more waves hide more L2 latency (C4:
  // 0.938 / 0.972 / 1.062 ms at 5 / 4 / 3) --, three when the caller knows the graph has no neighbour reuse: the launch is then HBM-bound and
  // fewer streams of random rows do better (uniform-random C4: 1.660 / 1.619 / 1.585 ms)
  static const int per_cu_max = [] {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, agg_wide_fast_kernel<LF, ALPHA>, 256, 0) != hipSuccess || per_cu < 1) return 0;
    if (per_cu > 8) per_cu = 8;
    const char* e = getenv("BGNN_AGG_BLOCKS_PER_CU");
    if (e && atoi(e) > 0 && atoi(e) < per_cu) per_cu = atoi(e);
    return per_cu;
  }();
  static const int n_cu = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return prop.multiProcessorCount;
  }();
  static const bool forced = getenv("BGNN_AGG_BLOCKS_PER_CU") != nullptr;
  int per_cu = per_cu_max;
  if (p.gather_hint == 2 && !forced && per_cu > 3) per_cu = 3;
  const int64_t cap = (per_cu_max > 0 && n_cu > 0) ? (int64_t)per_cu * n_cu / 8 * 8 : 2048;
  const int64_t ntiles = (p.row_end - p.row_begin + RPB - 1) / RPB;
  int64_t grid = ntiles < cap ? (ntiles + 7) / 8 * 8 : cap;
  if (grid < 8) grid = 8;
  AggParams q = p;
  q.tq_chunk = tq_chunk_for(ntiles, grid);
  static const int il = [] { const char* e = getenv("BGNN_AGG_INTERLEAVE"); return e ? atoi(e) : 1; }();
  q.tq_interleave = il;
  static const int nseg = [] { const char* e = getenv("BGNN_XCD_SEGMENTS"); return e ? atoi(e) : 8; }();
  q.xcd_segments = nseg;
  hipLaunchKernelGGL((agg_wide_fast_kernel<LF, ALPHA>), dim3((unsigned)grid), dim3(256), 0, st, q);
  BGNN_LAUNCH_CHECK();
  return 0;
}

// ---- hub rows: merge of the parked segment states (see AggParams::hub_threshold) -------------------------------------------
struct HubMergeParams {
  const int32_t* hub_rows; const int32_t* seg_ptr; int64_t n_hubs;
  const float* part_acc;   // [segments][heads * ldo]  raw accumulators (mode 1 parks them in `out`)
  const float* part_ms;    // [segments * heads][2]    (max, sum)
  const uint8_t* mask;
  int32_t D; int64_t ldo; int32_t heads;
  float* out; const float* ep_scale; const float* ep_shift; int ep_relu;
  double* colsum; float* state_ms;
  float* alpha; const int32_t* rowptr;   // training forward: the segments left RAW logits in alpha[e]; normalised here with the row's (m, s)
};

// wide rows (heads == 1): LF lanes own a hub row, one float4 of columns each
template <int LF>
__global__ __launch_bounds__(256) void hub_merge_wide_kernel(HubMergeParams p) {
  const int lane = threadIdx.x % LF, grp = threadIdx.x / LF;
  constexpr int GPB = 256 / LF;
  const int f0 = lane * 4;
  const bool fvalid = f0 < p.D;
  __shared__ float red[2][LF * 4 + 1];
  if (p.colsum != nullptr) {
    for (int t = threadIdx.x; t < 2 * (LF * 4 + 1); t += 256) (&red[0][0])[t] = 0.f;
    __syncthreads();
  }
  float sc4[4] = {1.f, 1.f, 1.f, 1.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.ep_scale != nullptr) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (f0 + c < p.D) { sc4[c] = p.ep_scale[f0 + c]; sh4[c] = p.ep_shift[f0 + c]; }
  }
  float4 cs_s = make_float4(0.f, 0.f, 0.f, 0.f), cs_t = cs_s;
  float n_s = 0.f, n_t = 0.f;
  for (int64_t hrow = (int64_t)blockIdx.x * GPB + grp; hrow < p.n_hubs; hrow += (int64_t)gridDim.x * GPB) {
    const int64_t node = p.hub_rows[hrow];
    float m = -INFINITY, s = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int32_t v1 = p.seg_ptr[hrow + 1];
    for (int32_t v0 = p.seg_ptr[hrow]; v0 < v1; v0 += 4) {          // four segments' loads in flight
      float m2[4], s2[4];
      float4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t v = v0 + u < v1 ? v0 + u : v1 - 1;
        m2[u] = v0 + u < v1 ? p.part_ms[2 * v] : -INFINITY;
        s2[u] = v0 + u < v1 ? p.part_ms[2 * v + 1] : 0.f;
        b[u] = f0 < p.ldo ? *reinterpret_cast<const float4*>(p.part_acc + v * p.ldo + f0) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float mn = fmaxf(m, m2[u]);
        const float c1 = (m == mn) ? 1.f : __expf(m - mn), c2 = (m2[u] == mn) ? 1.f : __expf(m2[u] - mn);
        s = s * c1 + s2[u] * c2;
        acc.x = acc.x * c1 + b[u].x * c2; acc.y = acc.y * c1 + b[u].y * c2;
        acc.z = acc.z * c1 + b[u].z * c2; acc.w = acc.w * c1 + b[u].w * c2;
        m = mn;
      }
    }
    if (p.alpha != nullptr) {                // the same expression as the aggregation kernel's own normalisation
      const float inv = 1.f / (s + 1e-16f);
      for (int32_t e = p.rowptr[node] + lane; e < p.rowptr[node + 1]; e += LF) p.alpha[e] = __expf(p.alpha[e] - m) * inv;
    }
    if (f0 < p.ldo) {
      const float inv = 1.f / (s + 1e-16f);
      float4 o = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
      if (p.ep_scale != nullptr) {
        o.x = fmaf(o.x, sc4[0], sh4[0]); o.y = fmaf(o.y, sc4[1], sh4[1]);
        o.z = fmaf(o.z, sc4[2], sh4[2]); o.w = fmaf(o.w, sc4[3], sh4[3]);
      }
      if (p.ep_relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      if (!fvalid) o = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(p.out + node * p.ldo + f0) = o;
      if (p.mask[node] != 0) { cs_s.x += o.x; cs_s.y += o.y; cs_s.z += o.z; cs_s.w += o.w; n_s += 1.f; }
      else                   { cs_t.x += o.x; cs_t.y += o.y; cs_t.z += o.z; cs_t.w += o.w; n_t += 1.f; }
    }
  }
  if (p.colsum != nullptr) {      // lanes -> LDS -> one fp64 atomic per (block, column, domain), as in agg_wide_kernel
    unsafeAtomicAdd(&red[0][f0], cs_s.x); unsafeAtomicAdd(&red[0][f0 + 1], cs_s.y);
    unsafeAtomicAdd(&red[0][f0 + 2], cs_s.z); unsafeAtomicAdd(&red[0][f0 + 3], cs_s.w);
    unsafeAtomicAdd(&red[1][f0], cs_t.x); unsafeAtomicAdd(&red[1][f0 + 1], cs_t.y);
    unsafeAtomicAdd(&red[1][f0 + 2], cs_t.z); unsafeAtomicAdd(&red[1][f0 + 3], cs_t.w);
    if (lane == 0) { unsafeAtomicAdd(&red[0][LF * 4], n_s); unsafeAtomicAdd(&red[1][LF * 4], n_t); }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * (LF * 4 + 1); t += 256) {
      const int d = t / (LF * 4 + 1), c = t % (LF * 4 + 1);
      const double sum = (double)red[d][c];
      if (c == LF * 4) unsafeAtomicAdd(&p.colsum[2 * p.ldo + d], sum);
      else if (c < p.ldo) unsafeAtomicAdd(&p.colsum[d * p.ldo + c], sum);
    }
  }
}

// interleaved narrow heads: one thread per (hub row, head)
__global__ __launch_bounds__(256) void hub_merge_heads_kernel(HubMergeParams p) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.n_hubs * p.heads) return;
  const int64_t hrow = t / p.heads;
  const int h = (int)(t - hrow * p.heads);
  const int64_t node = p.hub_rows[hrow];
  float m = -INFINITY, s = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int32_t v = p.seg_ptr[hrow]; v < p.seg_ptr[hrow + 1]; ++v) {
    const int64_t e = (int64_t)v * p.heads + h;
    const float m2 = p.part_ms[2 * e], s2 = p.part_ms[2 * e + 1];
    const float4 b = *reinterpret_cast<const float4*>(p.part_acc + e * p.ldo);
    const float mn = fmaxf(m, m2);
    const float c1 = (m == mn) ? 1.f : __expf(m - mn), c2 = (m2 == mn) ? 1.f : __expf(m2 - mn);
    s = s * c1 + s2 * c2;
    acc.x = acc.x * c1 + b.x * c2; acc.y = acc.y * c1 + b.y * c2;
    acc.z = acc.z * c1 + b.z * c2; acc.w = acc.w * c1 + b.w * c2;
    m = mn;
  }
  const float inv = 1.f / (s + 1e-16f);
  float4 r = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  if (p.state_ms != nullptr) {
    p.state_ms[2 * (node * p.heads + h)] = m;
    p.state_ms[2 * (node * p.heads + h) + 1] = s;
  }
  if (p.ep_relu == 2) {    // log_softmax over the head's D classes (KTGNN.py:435), as in the aggregation kernels
    float mx = r.x;
    if (p.D > 1) mx = fmaxf(mx, r.y);
    if (p.D > 2) mx = fmaxf(mx, r.z);
    if (p.D > 3) mx = fmaxf(mx, r.w);
    float se = expf(r.x - mx);
    if (p.D > 1) se += expf(r.y - mx);
    if (p.D > 2) se += expf(r.z - mx);
    if (p.D > 3) se += expf(r.w - mx);
    const float lse = logf(se);
    r.x = r.x - mx - lse;
    r.y = p.D > 1 ? r.y - mx - lse : 0.f;
    r.z = p.D > 2 ? r.z - mx - lse : 0.f;
    r.w = p.D > 3 ? r.w - mx - lse : 0.f;
  }
  *reinterpret_cast<float4*>(p.out + (node * p.heads + h) * p.ldo) = r;
}

static int dispatch_aggregate(const AggParams& p, hipStream_t st, int64_t table_rows = 0) {
  const int32_t D = p.D, heads = p.heads;
  if (heads == 3 && D <= 4 && p.ldh == 4 && p.ldo == 4) return launch_heads<3, 2, 4>(p, st);   // KT-GNN's classifier stage
  if (heads == 2 && D <= 4 && p.ldh == 4 && p.ldo == 4) return launch_heads<2, 2, 4>(p, st);
  const int nv = (D + 3) / 4;   // float4 slots per row
  // (LF, EP, U) picked from the tools/tune_agg.py sweep on MI355X (profiles/r01/tune_agg_v2.json):
  // one sub-group per row with deep unrolling beats edge-parallel sub-groups except for the narrowest rows.
  if (nv <= 1) return launch<1, 4, 4>(p, st);
  if (nv <= 2) return launch<2, 2, 4>(p, st);
  if (nv <= 4) return launch<4, 1, 8>(p, st);
  if (nv <= 8) return launch<8, 1, 4>(p, st);
  // wide rows: the VALU-lean kernel (needs max(z, slope*z) == leaky_relu and 32-bit row strides)
  const bool wide_ok = p.slope >= 0.f && p.slope <= 1.f && (int64_t)heads * p.ldh * 4 < (int64_t)1 << 32;
  if (wide_ok) {
    AggParams q = p;
    if (fast_plan(q, table_rows)) {
      if (q.alpha != nullptr) {
        if (nv <= 16) return launch_wide_fast<16, true>(q, st);
        if (nv <= 32) return launch_wide_fast<32, true>(q, st);
        return launch_wide_fast<64, true>(q, st);
      }
      if (nv <= 16) return launch_wide_fast<16, false>(q, st);
      if (nv <= 32) return launch_wide_fast<32, false>(q, st);
      return launch_wide_fast<64, false>(q, st);
    }
    if (nv <= 16) return launch_wide<16, 4>(p, st);
    if (nv <= 32) return launch_wide<32, 4>(p, st);
    return launch_wide<64, 4>(p, st);
  }
  if (nv <= 16) return launch<16, 1, 8>(p, st);
  if (nv <= 32) return launch<32, 1, 4>(p, st);
  return launch<64, 1, 4>(p, st);
}

// does dispatch_aggregate pick a kernel that understands hub_threshold / vrow_node for this shape?
static bool hub_capable(int32_t D, int64_t ldh, int64_t ldo, int32_t heads, float slope) {
  if ((heads == 3 || heads == 2) && D <= 4 && ldh == 4 && ldo == 4) return true;
  const int nv = (D + 3) / 4;
  return heads == 1 && nv > 8 && slope >= 0.f && slope <= 1.f && ldh * 4 < (int64_t)1 << 32;
}

}  // namespace

static int aggregate_impl(const float* h_t2s, const float* h_s2t, int64_t ldh,
                          const float* a_t2s, const float* a_s2t,
                          const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                          int64_t row_begin, int64_t row_end, int32_t D, float negative_slope,
                          float* out, int64_t ldo, float* alpha_opt,
                          const float* ep_scale_opt, const float* ep_shift_opt, int ep_relu,
                          float* state_ms_opt, int part, int64_t park_begin, int32_t heads,
                          double* colsum_opt, uint32_t* tile_queue_opt, int64_t table_rows, int32_t gather_hint, void* stream) {
  if (colsum_opt && heads != 1) return BGNN_E_SHAPE;
  if (part == 1 && (park_begin < row_begin || park_begin > row_end)) return BGNN_E_SHAPE;
  if (part < 0 || part > 3 || (part != 0 && (!state_ms_opt || alpha_opt))) return BGNN_E_NULL;
  if (part == 3 && !((heads == 3 || heads == 2) && D <= 4 && ldh == 4 && ldo == 4)) return BGNN_E_SHAPE;
  if (heads < 1 || heads > 8 || (heads > 1 && (alpha_opt || ep_scale_opt))) return BGNN_E_SHAPE;
  if (ep_relu < 0 || ep_relu > 2) return BGNN_E_SHAPE;
  // ep_relu == 2: log_softmax epilogue instead of ReLU -- only the interleaved narrow-heads kernel has it
  if (ep_relu == 2 && !((heads == 3 || heads == 2) && D <= 4 && ldh == 4 && ldo == 4)) return BGNN_E_SHAPE;
  if (!h_t2s || !h_s2t || !a_t2s || !a_s2t || !rowptr || !col || !mask || !out) return BGNN_E_NULL;
  if (row_begin < 0 || row_end < row_begin || D <= 0 || D > 256 || ldh < D || ldo < D) return BGNN_E_SHAPE;
  if ((ldh & 3) || (ldo & 3) || !bgnn_aligned16(h_t2s) || !bgnn_aligned16(h_s2t) || !bgnn_aligned16(out))
    return BGNN_E_ALIGN;
  if ((ep_scale_opt == nullptr) != (ep_shift_opt == nullptr)) return BGNN_E_NULL;
  if (row_end == row_begin) return 0;
  AggParams p{h_t2s, h_s2t, ldh, a_t2s, a_s2t, rowptr, col, mask, row_begin, row_end, D, negative_slope,
              out, ldo, alpha_opt, ep_scale_opt, ep_shift_opt, ep_relu, tile_queue_opt, colsum_opt, heads, state_ms_opt, part,
              park_begin, 4};
  p.gather_hint = gather_hint;
  hipStream_t st = (hipStream_t)stream;
  if (tile_queue_opt) {
    hipError_t e = bgnn_zero_async(tile_queue_opt, 8 * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
  }
  return dispatch_aggregate(p, st, table_rows);
}

extern "C" int bgnn_adaptedconv_aggregate_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
                                              const float* a_t2s, const float* a_s2t,
                                              const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                              int64_t row_begin, int64_t row_end, int32_t D, float negative_slope,
                                              float* out, int64_t ldo, float* alpha_opt,
                                              const float* ep_scale_opt, const float* ep_shift_opt, int ep_relu,
                                              float* state_ms_opt, int part, int64_t park_begin, int32_t heads,
                                              double* colsum_opt, uint32_t* tile_queue_opt, void* stream) {
  return aggregate_impl(h_t2s, h_s2t, ldh, a_t2s, a_s2t, rowptr, col, mask, row_begin, row_end, D, negative_slope, out, ldo, alpha_opt,
                        ep_scale_opt, ep_shift_opt, ep_relu, state_ms_opt, part, park_begin, heads, colsum_opt, tile_queue_opt, 0, 0, stream);
}

extern "C" int bgnn_adaptedconv_aggregate_bounded_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
                                                      const float* a_t2s, const float* a_s2t,
                                                      const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                                      int64_t row_begin, int64_t row_end, int32_t D, float negative_slope,
                                                      float* out, int64_t ldo, float* alpha_opt,
                                                      const float* ep_scale_opt, const float* ep_shift_opt, int ep_relu,
                                                      float* state_ms_opt, int part, int64_t park_begin, int32_t heads,
                                                      double* colsum_opt, uint32_t* tile_queue_opt, int64_t table_rows, int32_t gather_hint,
                                                      void* stream) {
  if (table_rows < row_end || gather_hint < 0 || gather_hint > 2) return BGNN_E_SHAPE;
  return aggregate_impl(h_t2s, h_s2t, ldh, a_t2s, a_s2t, rowptr, col, mask, row_begin, row_end, D, negative_slope, out, ldo, alpha_opt,
                        ep_scale_opt, ep_shift_opt, ep_relu, state_ms_opt, part, park_begin, heads, colsum_opt, tile_queue_opt, table_rows, gather_hint, stream);
}

extern "C" size_t bgnn_aggregate_hub_workspace_bytes(int64_t n_segments, int32_t heads, int64_t ldo) {
  const size_t nv = (size_t)(n_segments > 0 ? n_segments : 0), h = (size_t)(heads > 0 ? heads : 1);
  return bgnn_align_up(sizeof(float) * nv * h * (size_t)(ldo > 0 ? ldo : 0), 256) + bgnn_align_up(sizeof(float) * 2 * nv * h, 256) + 256;
}

extern "C" int bgnn_adaptedconv_aggregate_hub_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
                                                  const float* a_t2s, const float* a_s2t,
                                                  const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                                  int64_t N, int32_t D, float negative_slope, float* out, int64_t ldo,
                                                  const float* ep_scale_opt, const float* ep_shift_opt, int ep_relu,
                                                  float* state_ms_opt, int32_t heads, double* colsum_opt,
                                                  uint32_t* tile_queue_opt, int32_t hub_threshold, const int32_t* hub_rows,
                                                  int64_t n_hubs, const int32_t* hub_seg_ptr, const int32_t* seg_bounds,
                                                  const int32_t* seg_node, int64_t n_segments, float* alpha_opt,
                                                  void* ws, size_t ws_bytes, void* stream) {
  if (!h_t2s || !h_s2t || !a_t2s || !a_s2t || !rowptr || !col || !mask || !out) return BGNN_E_NULL;
  if (alpha_opt && heads != 1) return BGNN_E_SHAPE;
  if (n_hubs > 0 && (!hub_rows || !hub_seg_ptr || !seg_bounds || !seg_node || !ws)) return BGNN_E_NULL;
  if (N < 0 || D <= 0 || D > 256 || ldh < D || ldo < D || (ldh & 3) || (ldo & 3) || hub_threshold < 2 || n_hubs < 0 || n_segments < n_hubs)
    return BGNN_E_SHAPE;
  if (!hub_capable(D, ldh, ldo, heads, negative_slope)) return BGNN_E_SHAPE;
  if (colsum_opt && heads != 1) return BGNN_E_SHAPE;
  if (heads > 1 && ep_scale_opt) return BGNN_E_SHAPE;
  if (ep_relu < 0 || ep_relu > 2 || (ep_relu == 2 && heads == 1)) return BGNN_E_SHAPE;
  if ((ep_scale_opt == nullptr) != (ep_shift_opt == nullptr)) return BGNN_E_NULL;
  if (!bgnn_aligned16(h_t2s) || !bgnn_aligned16(h_s2t) || !bgnn_aligned16(out) || (ws && !bgnn_aligned16(ws))) return BGNN_E_ALIGN;
  if (n_hubs > 0 && ws_bytes < bgnn_aggregate_hub_workspace_bytes(n_segments, heads, ldo)) return BGNN_E_WORKSPACE;
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int part = (state_ms_opt && heads > 1) ? 3 : 0;
  AggParams p{h_t2s, h_s2t, ldh, a_t2s, a_s2t, rowptr, col, mask, 0, N, D, negative_slope,
              out, ldo, alpha_opt, ep_scale_opt, ep_shift_opt, ep_relu, tile_queue_opt, colsum_opt, heads, state_ms_opt, part,
              0, 4};
  p.hub_threshold = n_hubs > 0 ? hub_threshold : 0;
  if (tile_queue_opt) {
    hipError_t e = bgnn_zero_async(tile_queue_opt, 8 * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
  }
  float* part_acc = n_hubs > 0 ? (float*)ws : nullptr;
  float* part_ms = n_hubs > 0 ? (float*)((char*)ws + bgnn_align_up(sizeof(float) * (size_t)n_segments * (size_t)heads * (size_t)ldo, 256)) : nullptr;
  if (n_hubs > 0) {                                    // the hub rows' segments ride in the same launch as virtual rows
    p.vrow_node = seg_node; p.vrow_bounds = seg_bounds; p.n_vrows = n_segments; p.vout = part_acc; p.vms = part_ms;
  }
  int rc = dispatch_aggregate(p, st, N);              // (all N rows, square: every id in `col` is a row)
  if (rc || n_hubs == 0) return rc;
  HubMergeParams mp{hub_rows, hub_seg_ptr, n_hubs, part_acc, part_ms, mask, D, ldo, heads, out, ep_scale_opt, ep_shift_opt,
                    ep_relu, colsum_opt, part == 3 ? state_ms_opt : nullptr, alpha_opt, rowptr};
  if (heads > 1) {
    hipLaunchKernelGGL(hub_merge_heads_kernel, dim3((unsigned)((n_hubs * heads + 255) / 256)), dim3(256), 0, st, mp);
  } else {
    const int nvs = (D + 3) / 4;
    int64_t grid = n_hubs < 2048 ? n_hubs : 2048;
    if (nvs <= 16) hipLaunchKernelGGL(hub_merge_wide_kernel<16>, dim3((unsigned)((grid + 15) / 16)), dim3(256), 0, st, mp);
    else if (nvs <= 32) hipLaunchKernelGGL(hub_merge_wide_kernel<32>, dim3((unsigned)((grid + 7) / 8)), dim3(256), 0, st, mp);
    else hipLaunchKernelGGL(hub_merge_wide_kernel<64>, dim3((unsigned)((grid + 3) / 4)), dim3(256), 0, st, mp);
  }
  BGNN_LAUNCH_CHECK();
  return 0;
}

#ifdef BGNN_TUNING
// Tuning-only entry (compiled into tools/libbgnn_tune.so, never into libbgnn_hip.so): run one explicit
// (LF, EP, U) instantiation so a sweep can pick the dispatch table above from measurements.
extern "C" int bgnn_tune_reset_counters(void* stream) { (void)stream; return 0; }
extern "C" int bgnn_tune_aggregate(const float* h_t2s, const float* h_s2t, int64_t ldh, const float* a_t2s,
                                   const float* a_s2t, const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                   int64_t row_begin, int64_t row_end, int32_t D, float slope, float* out, int64_t ldo,
                                   int variant, uint32_t* tile_queue, void* stream) {
  AggParams p{h_t2s, h_s2t, ldh, a_t2s, a_s2t, rowptr, col, mask, row_begin, row_end, D, slope,
              out, ldo, nullptr, nullptr, nullptr, 0, tile_queue, nullptr, 1, nullptr, 0};
  hipStream_t st = (hipStream_t)stream;
  if (tile_queue && bgnn_zero_async(tile_queue, 32, st) != hipSuccess) return -1;
  switch (variant) {
    case 40: return launch_wide<32, 4>(p, st);
    case 41: return launch_wide<32, 8>(p, st);
    case 42: return launch_wide<16, 4>(p, st);
    case 43: return launch_wide<16, 8>(p, st);
    case 44: return launch_wide<64, 4>(p, st);
    case 45: return launch_wide<64, 8>(p, st);
    // D = 128 (LF = 32)
    case 0: return launch<32, 1, 4>(p, st);
    case 1: return launch<32, 1, 8>(p, st);
    case 2: return launch<32, 2, 4>(p, st);
    case 3: return launch<32, 2, 2>(p, st);
    case 4: return launch<32, 1, 2>(p, st);
    case 5: return launch<32, 2, 8>(p, st);
    case 6: return launch<64, 1, 4>(p, st);
    // D <= 4 (LF = 1)
    case 10: return launch<1, 8, 2>(p, st);
    case 11: return launch<1, 4, 4>(p, st);
    case 12: return launch<1, 2, 4>(p, st);
    case 13: return launch<1, 1, 4>(p, st);
    case 14: return launch<1, 1, 8>(p, st);
    case 15: return launch<1, 4, 2>(p, st);
    case 16: return launch<1, 2, 8>(p, st);
    case 17: return launch<1, 16, 2>(p, st);
    // D = 64 (LF = 16) / D = 32 (LF = 8)
    case 20: return launch<16, 2, 4>(p, st);
    case 21: return launch<16, 1, 4>(p, st);
    case 22: return launch<16, 4, 2>(p, st);
    case 23: return launch<16, 1, 8>(p, st);
    case 30: return launch<8, 2, 4>(p, st);
    case 31: return launch<8, 4, 2>(p, st);
    case 32: return launch<8, 1, 4>(p, st);
    case 33: return launch<8, 8, 2>(p, st);
    default: return BGNN_E_RANGE;
  }
}
#endif
