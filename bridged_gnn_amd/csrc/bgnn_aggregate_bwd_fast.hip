// The atomic-free ("pull") aggregation backward for the plain hidden-conv launch (64 < D <= 128, no hub rows), rebuilt around
// what bounds it on gfx950.  Same two passes and the same mathematics as agg_bwd_dst_kernel / agg_bwd_src_kernel
// (bgnn_aggregate_bwd.hip; reference: autograd through Bridged-GNN/models/KTGNN.py:292-305):
//   pass A (by destination i):  c_ji = g_i . h_j,  t_i = g_i . out_i,  de_ji = alpha_ji (c_ji - t_i),
//                               dstside[i] = a * sum_j de_ji leaky'(z_ji),  da += sum_ji de_ji leaky(z_ji),  z_ji = h_j + h_i,
//                               and a 32-byte record per edge {4 x 32 sign bits of z_ji | alpha_ji (sign bit = domain of i), de_ji};
//   pass B (by source j):       dH_X[j] = sum_{i in X} (alpha_ji g_i) + a_X * sum_{i in X} de_ji leaky'(z_ji) + dstside[j] (X = S, T).
// The general kernels spend 809 M / 652 M vector instructions per C4 launch (forward: 442 M) and twelve full-width memory
// instructions per step of pass B.  Here (the forward's agg_wide_fast_kernel is the model):
//  * one window of < 4 GB holds both tables, so a neighbour row is `window + 32-bit offset` through one buffer descriptor;
//    g, out, the records and the outputs get their own descriptors; a dead slot's offset lies past its buffer (no traffic, zeros);
//  * lane l scores the edge slot l & 3 of a step: the four c_ji partial sums are reduced by a transposing butterfly (13
//    instructions instead of 4 x 5), de is formed once per slot and returns to all lanes as a quad broadcast;
//  * leaky' enters as a SELECT between de and slope * de, so a * (..) leaves the edge loop (one multiply per row and column);
//  * the record's alpha / de are written by the slot's lane (one 8-byte store per step), its sign words by the group's first
//    lane from the compare masks themselves;
//  * pass B: every quad loads the record of its lane's slot once (two loads per step instead of eight), the sign words return
//    by quad broadcast, the per-domain split is a select of (alpha, de) by the record's domain bit;
//  * wave-uniform step loops, ids requested one step ahead by inline-assembly loads (see agg_wide_fast_kernel for why).
#include <cstdlib>
#include "bgnn_common.h"
#include "bgnn_aggregate_bwd_params.h"

namespace {

using bgnn_bwd::PullParams;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

constexpr uint32_t OOB = 0xFFFFFFF0u;         // an offset no buffer below 4 GB - 16 contains

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true); }
template <int K>
__device__ __forceinline__ float quad_bcast(float x) { return bgnn::dpp_mov<K * 0x55>(x); }
template <int K>
__device__ __forceinline__ uint32_t quad_bcastu(uint32_t x) { return dpp_u<K * 0x55>(x); }
template <int CTRL>
__device__ __forceinline__ float bfly(bool hi_side, float t_lo, float t_hi) {
  const float keep = hi_side ? t_hi : t_lo;
  const float send = hi_side ? t_lo : t_hi;
  return keep + bgnn::dpp_mov<CTRL>(send);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 ld128(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// per-XCD dynamic tile queue over positions of the XCD's segment sequence (the general kernels' scheduling: bgnn_common.h)
struct TileWalk {
  bgnn::XcdRange tr;
  int64_t xbase, tile, chunk_left, ntiles;
  unsigned int* q;
  __device__ __forceinline__ void init(int64_t ntiles_, unsigned int* queue) {
    ntiles = ntiles_;
    tr = bgnn::xcd_pos_range(ntiles);
    xbase = tr.begin - (blockIdx.x / 8);
    tile = 0; chunk_left = 0;
    q = queue + blockIdx.x % 8;
  }
  // -> global tile index, -1: skip, -2: done.  `slot` is a block-shared word.
  __device__ __forceinline__ int64_t next(unsigned int* slot) {
    constexpr int TQ_CHUNK = 4;
    if (chunk_left == 0) {
      __syncthreads();
      if (threadIdx.x == 0) *slot = atomicAdd(q, 1u);
      __syncthreads();
      tile = xbase + (int64_t)(*slot) * TQ_CHUNK;
      chunk_left = TQ_CHUNK;
    } else {
      tile += 1;
    }
    --chunk_left;
    if (tile >= tr.end) return -2;
    const int64_t gt = bgnn::xcd_tile_of(tile, ntiles);
    return gt < 0 ? -1 : gt;
  }
};

// ---- pass A: by destination ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void agg_bwd_dst_fast_kernel(PullParams p) {
  constexpr int LF = 32, U = 4, GPW = 2, RPB = 4 * GPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 5, l = lane & 31;
  const int f0 = l * 4;
  const bool fvalid = f0 < p.D;
  const int f0c = fvalid ? f0 : 0;
  const int k = lane & (U - 1);
  const bool b0 = lane & 1, b1 = lane & 2;
  const int32_t N = (int32_t)p.N;
  const uint32_t nstride = (uint32_t)(p.ldh * 4), gstride = (uint32_t)(p.ldg * 4), ostride = (uint32_t)(p.ldo * 4);
  const __amdgpu_buffer_rsrc_t rt = rsrc(p.tbl_base, p.tbl_bytes);
  const __amdgpu_buffer_rsrc_t rg = rsrc(p.gout, (uint32_t)N * gstride), ro = rsrc(p.out, (uint32_t)N * ostride);
  const __amdgpu_buffer_rsrc_t rrec = rsrc(p.rec, (uint32_t)(p.E * 32)), rds = rsrc(p.dstside, (uint32_t)N * nstride);
  f2 aS01 = {0.f, 0.f}, aS23 = aS01, aT01 = aS01, aT23 = aS01;
  if (fvalid) {
    aS01.x = p.a_t2s[f0]; aS01.y = f0 + 1 < p.D ? p.a_t2s[f0 + 1] : 0.f;
    aS23.x = f0 + 2 < p.D ? p.a_t2s[f0 + 2] : 0.f; aS23.y = f0 + 3 < p.D ? p.a_t2s[f0 + 3] : 0.f;
    aT01.x = p.a_s2t[f0]; aT01.y = f0 + 1 < p.D ? p.a_s2t[f0 + 1] : 0.f;
    aT23.x = f0 + 2 < p.D ? p.a_s2t[f0 + 2] : 0.f; aT23.y = f0 + 3 < p.D ? p.a_s2t[f0 + 3] : 0.f;
  }
  f2 daS01 = {0.f, 0.f}, daS23 = daS01, daT01 = daS01, daT23 = daS01;     // da partials of the rows this lane walked, per domain
  __shared__ unsigned int dyn_tile;
  TileWalk tw;
  tw.init(((int64_t)N + RPB - 1) / RPB, p.queue);
  for (;;) {
    const int64_t gt = tw.next(&dyn_tile);
    if (gt == -2) break;
    if (gt == -1) continue;
    const int32_t i0 = (int32_t)gt * RPB + wave * GPW + g;
    const bool rvalid = i0 < N;
    const int32_t ic = rvalid ? i0 : 0;
    const bool dom_s = p.mask[ic] != 0;
    const int32_t beg = rvalid ? p.rowptr[ic] : 0;
    const int32_t end = rvalid ? p.rowptr[ic + 1] : 0;
    const uint32_t lbase = (dom_s ? p.off_t2s : p.off_s2t) + (uint32_t)f0c * 4u;
    const uint32_t live = (rvalid && fvalid) ? 0u : OOB;                    // (pad lanes / pad rows read zeros)
    const float4 hi4 = ld128(rt, lbase + __umul24((uint32_t)ic, nstride));
    const float4 gi4 = ld128(rg, (__umul24((uint32_t)ic, gstride) + (uint32_t)f0c * 4u) | live);
    const float4 oi4 = ld128(ro, (__umul24((uint32_t)ic, ostride) + (uint32_t)f0c * 4u) | live);
    f2 h01 = {hi4.x, hi4.y}, h23 = {hi4.z, hi4.w};
    const f2 g01 = {gi4.x, gi4.y}, g23 = {gi4.z, gi4.w};
    const float ti = bgnn::group_sum<LF>(gi4.x * oi4.x + gi4.y * oi4.y + gi4.z * oi4.z + gi4.w * oi4.w);
    f2 w01 = {0.f, 0.f}, w23 = w01, z01s = w01, z23s = w01;               // sum_j sel,  sum_j sel * z   (sel = de * leaky'(z))
    const int32_t niter = (end - beg + U - 1) / U;
    const int32_t nw = max(__builtin_amdgcn_readlane(niter, 0), __builtin_amdgcn_readlane(niter, 32));

    auto issue = [&](float4 (&v)[U], uint32_t off) {
      v[0] = ld128(rt, lbase + dpp_u<0x00>(off));
      v[1] = ld128(rt, lbase + dpp_u<0x55>(off));
      v[2] = ld128(rt, lbase + dpp_u<0xAA>(off));
      v[3] = ld128(rt, lbase + dpp_u<0xFF>(off));
      asm volatile("" : "+v"(h01), "+v"(h23) : : "memory");            // gathers in front, the step's arithmetic behind (agg_wide_fast_kernel)
      __builtin_amdgcn_sched_barrier(0);
    };
    // id and alpha of a later step's slot (inline assembly: see agg_wide_fast_kernel::load_id)
    auto load_slot = [&](int32_t ee, uint32_t& id, float& al) {
      const uint32_t eoff = (uint32_t)max(min(ee, end - 1), 0) * 4u;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %4"
                   : "=&v"(id), "=&v"(al) : "v"(eoff), "s"(p.col), "s"(p.alpha) : "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    auto row_off = [&](uint32_t id, bool alive) { return alive ? __umul24(id, nstride) : p.dead_off; };

    int32_t e = beg + k;
    bool ok = e < end;
    uint32_t myoff = p.dead_off;
    float al = 0.f;
    if (nw > 0) {
      const int32_t ec = max(min(e, end - 1), 0);
      myoff = row_off((uint32_t)p.col[ec], ok);
      al = p.alpha[ec];
    }
    for (int32_t it = 0; it < nw; ++it) {
      float4 v[U];
      issue(v, myoff);
      const int32_t e_cur = e;
      e += U;
      const bool ok2 = e < end;
      uint32_t nextid; float nextal;
      load_slot(e, nextid, nextal);

      float t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        f2 q = g01 * f2{v[u].x, v[u].y};
        q = __builtin_elementwise_fma(g23, f2{v[u].z, v[u].w}, q);
        t[u] = q.x + q.y;
      }
      const float rA = bfly<0xB1>(b0, t[0], t[1]), rB = bfly<0xB1>(b0, t[2], t[3]);
      float r = bfly<0x4E>(b1, rA, rB);
      r += bgnn::dpp_mov<0x124>(r);
      r += bgnn::dpp_mov<0x128>(r);
      r += bgnn::swz_xor16(r);                              // c of my slot's edge
      const float de = ok ? al * (r - ti) : 0.f;
      const float des = de * p.slope;
      if (l < U) {                                          // the slot's lane in the group's first quad: {alpha (sign = domain of i), de}
        const float als = dom_s ? -al : al;
        u32x2_t hd = {__builtin_bit_cast(unsigned, als), __builtin_bit_cast(unsigned, de)};
        __builtin_amdgcn_raw_buffer_store_b64(hd, rrec, ok ? (uint32_t)e_cur * 32u + 16u : OOB, 0, 0);
      }
      const int32_t eb = e_cur - k;                         // slot 0 of this step
#define BGNN_BWD_EDGE(u)                                                                                         \
      {                                                                                                          \
        const float de_u = quad_bcast<u>(de), des_u = quad_bcast<u>(des);                                        \
        const f2 z01 = f2{v[u].x, v[u].y} + h01, z23 = f2{v[u].z, v[u].w} + h23;                                 \
        const bool p0 = z01.x > 0.f, p1 = z01.y > 0.f, p2 = z23.x > 0.f, p3 = z23.y > 0.f;                       \
        const f2 s01 = {p0 ? de_u : des_u, p1 ? de_u : des_u}, s23 = {p2 ? de_u : des_u, p3 ? de_u : des_u};     \
        w01 += s01; w23 += s23;                                                                                  \
        z01s = __builtin_elementwise_fma(s01, z01, z01s); z23s = __builtin_elementwise_fma(s23, z23, z23s);      \
        const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1), m2 = __ballot(p2), m3 = __ballot(p3);     \
        if (l == 0) { /* sign words of the edge (column 4*lane + c <-> bit lane of word c): group 0 owns the low words, group 1 the high */ \
          const u32x4_t m = {g ? (unsigned)(m0 >> 32) : (unsigned)m0, g ? (unsigned)(m1 >> 32) : (unsigned)m1,   \
                             g ? (unsigned)(m2 >> 32) : (unsigned)m2, g ? (unsigned)(m3 >> 32) : (unsigned)m3};  \
          __builtin_amdgcn_raw_buffer_store_b128(m, rrec, eb + u < end ? (uint32_t)(eb + u) * 32u : OOB, 0, 0);  \
        }                                                                                                        \
      }
      BGNN_BWD_EDGE(0) BGNN_BWD_EDGE(1) BGNN_BWD_EDGE(2) BGNN_BWD_EDGE(3)
#undef BGNN_BWD_EDGE
      ok = ok2;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(nextid), "+v"(nextal), "+v"(w01), "+v"(w23), "+v"(z01s), "+v"(z23s) : : "memory");
      myoff = row_off(nextid, ok);
      al = nextal;
    }
    if (rvalid) {
      const f2 a01 = dom_s ? aS01 : aT01, a23 = dom_s ? aS23 : aT23;
      const f2 d01 = a01 * w01, d23 = a23 * w23;                             // dstside: a * sum_j de leaky'
      if (f0 < p.ldh) {
        // (whole-vector bit_cast: __builtin_bit_cast of an ext-vector ELEMENT reads element 0 with this compiler)
        const u32x4_t o = __builtin_bit_cast(u32x4_t, make_float4(d01.x, d01.y, d23.x, d23.y));
        const u32x4_t zero = {0u, 0u, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(fvalid ? o : zero, rds, __umul24((uint32_t)i0, nstride) + (uint32_t)f0 * 4u, 0, 0);
      }
      if (fvalid) {
        const float ws = dom_s ? 1.f : 0.f, wt = dom_s ? 0.f : 1.f;
        const f2 ws2 = {ws, ws}, wt2 = {wt, wt};
        daS01 = __builtin_elementwise_fma(z01s, ws2, daS01); daS23 = __builtin_elementwise_fma(z23s, ws2, daS23);
        daT01 = __builtin_elementwise_fma(z01s, wt2, daT01); daT23 = __builtin_elementwise_fma(z23s, wt2, daT23);
      }
    }
  }
  // da: block reduction through LDS, one atomic per (block, column, domain)
  __shared__ float red[2][LF * 4];
  for (int t = threadIdx.x; t < 2 * LF * 4; t += 256) (&red[0][0])[t] = 0.f;
  __syncthreads();
  unsafeAtomicAdd(&red[0][f0], daS01.x); unsafeAtomicAdd(&red[0][f0 + 1], daS01.y);
  unsafeAtomicAdd(&red[0][f0 + 2], daS23.x); unsafeAtomicAdd(&red[0][f0 + 3], daS23.y);
  unsafeAtomicAdd(&red[1][f0], daT01.x); unsafeAtomicAdd(&red[1][f0 + 1], daT01.y);
  unsafeAtomicAdd(&red[1][f0 + 2], daT23.x); unsafeAtomicAdd(&red[1][f0 + 3], daT23.y);
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * LF * 4; t += 256) {
    const int d = t / (LF * 4), c = t % (LF * 4);
    if (c < p.D) unsafeAtomicAdd(d == 0 ? &p.da_t2s[c] : &p.da_s2t[c], red[d][c]);
  }
}

// ---- pass B: by source --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void agg_bwd_src_fast_kernel(PullParams p) {
  constexpr int U = 4, GPW = 2, RPB = 4 * GPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 5, l = lane & 31;
  const int f0 = l * 4;
  const bool fvalid = f0 < p.D;
  const int f0c = fvalid ? f0 : 0;
  const int k = lane & (U - 1);
  const uint32_t lanebit = 1u << l;
  const int32_t N = (int32_t)p.N;
  const uint32_t nstride = (uint32_t)(p.ldh * 4), gstride = (uint32_t)(p.ldg * 4);
  const __amdgpu_buffer_rsrc_t rg = rsrc(p.gout, (uint32_t)N * gstride), rrec = rsrc(p.rec, (uint32_t)(p.E * 32));
  const __amdgpu_buffer_rsrc_t rds = rsrc(p.dstside, (uint32_t)N * nstride);
  const __amdgpu_buffer_rsrc_t rdS = rsrc(p.dh_t2s, (uint32_t)N * nstride), rdT = rsrc(p.dh_s2t, (uint32_t)N * nstride);
  const uint32_t gbase = (uint32_t)f0c * 4u;
  // a dead slot's g row: past the buffer when that fits below 2^32 (no traffic), else row 0 (its alpha and de are zero)
  const uint32_t gdead = (uint64_t)N * gstride + gstride <= (uint64_t)OOB ? OOB - gstride : 0u;
  f2 aS01 = {0.f, 0.f}, aS23 = aS01, aT01 = aS01, aT23 = aS01;
  if (fvalid) {
    aS01.x = p.a_t2s[f0]; aS01.y = f0 + 1 < p.D ? p.a_t2s[f0 + 1] : 0.f;
    aS23.x = f0 + 2 < p.D ? p.a_t2s[f0 + 2] : 0.f; aS23.y = f0 + 3 < p.D ? p.a_t2s[f0 + 3] : 0.f;
    aT01.x = p.a_s2t[f0]; aT01.y = f0 + 1 < p.D ? p.a_s2t[f0 + 1] : 0.f;
    aT23.x = f0 + 2 < p.D ? p.a_s2t[f0 + 2] : 0.f; aT23.y = f0 + 3 < p.D ? p.a_s2t[f0 + 3] : 0.f;
  }
  __shared__ unsigned int dyn_tile;
  TileWalk tw;
  tw.init(((int64_t)N + RPB - 1) / RPB, p.queue + 8);
  for (;;) {
    const int64_t gt = tw.next(&dyn_tile);
    if (gt == -2) break;
    if (gt == -1) continue;
    const int32_t j0 = (int32_t)gt * RPB + wave * GPW + g;
    const bool rvalid = j0 < N;
    const int32_t jc = rvalid ? j0 : 0;
    const int32_t beg = rvalid ? p.t_rowptr[jc] : 0;
    const int32_t end = rvalid ? p.t_rowptr[jc + 1] : 0;
    f2 accS01 = {0.f, 0.f}, accS23 = accS01, accT01 = accS01, accT23 = accS01;   // sum alpha g, per domain of the destination
    f2 wS01 = accS01, wS23 = accS01, wT01 = accS01, wT23 = accS01;               // sum de leaky', per domain
    const int32_t niter = (end - beg + U - 1) / U;
    const int32_t nw = max(__builtin_amdgcn_readlane(niter, 0), __builtin_amdgcn_readlane(niter, 32));

    auto load_slot = [&](int32_t ss, uint32_t& eid, uint32_t& dst) {
      const uint32_t soff = (uint32_t)max(min(ss, end - 1), 0) * 4u;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %4"
                   : "=&v"(eid), "=&v"(dst) : "v"(soff), "s"(p.t_eid), "s"(p.t_dst) : "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    int32_t sidx = beg + k;
    bool ok = sidx < end;
    uint32_t goff = gdead, roff = OOB;
    if (nw > 0) {
      const int32_t sc = max(min(sidx, end - 1), 0);
      goff = ok ? __umul24((uint32_t)p.t_dst[sc], gstride) : gdead;
      roff = ok ? (uint32_t)p.t_eid[sc] * 32u : OOB;
    }
    for (int32_t it = 0; it < nw; ++it) {
      float4 v[U];
      v[0] = ld128(rg, gbase + dpp_u<0x00>(goff));
      v[1] = ld128(rg, gbase + dpp_u<0x55>(goff));
      v[2] = ld128(rg, gbase + dpp_u<0xAA>(goff));
      v[3] = ld128(rg, gbase + dpp_u<0xFF>(goff));
      // my slot's record (every quad holds its own copy of the step's four records; a dead slot reads zeros: alpha = de = 0)
      const u32x4_t mk = __builtin_amdgcn_raw_buffer_load_b128(rrec, roff, 0, 0);
      const u32x2_t hd = __builtin_amdgcn_raw_buffer_load_b64(rrec, roff == OOB ? OOB : roff + 16u, 0, 0);
      asm volatile("" : "+v"(accS01), "+v"(accT01) : : "memory");            // loads in front, the step's arithmetic behind
      __builtin_amdgcn_sched_barrier(0);
      sidx += U;
      const bool ok2 = sidx < end;
      uint32_t nexteid, nextdst;
      load_slot(sidx, nexteid, nextdst);

      const float als = __uint_as_float(hd.x), de = __uint_as_float(hd.y);       // (never __builtin_bit_cast on a vector ELEMENT: it reads element 0)
      const bool ds = (hd.x >> 31) != 0u;                                    // domain of the destination
      const float al = __builtin_fabsf(als);
      const float alS = ds ? al : 0.f, alT = ds ? 0.f : al;
      const float deS = ds ? de : 0.f, deT = ds ? 0.f : de;
      const float desS = deS * p.slope, desT = deT * p.slope;
#define BGNN_BWD_EDGE(u)                                                                                         \
      {                                                                                                          \
        const float alS_u = quad_bcast<u>(alS), alT_u = quad_bcast<u>(alT);                                      \
        const float deS_u = quad_bcast<u>(deS), deT_u = quad_bcast<u>(deT);                                      \
        const float desS_u = quad_bcast<u>(desS), desT_u = quad_bcast<u>(desT);                                  \
        const bool p0 = (quad_bcastu<u>(mk.x) & lanebit) != 0u, p1 = (quad_bcastu<u>(mk.y) & lanebit) != 0u;     \
        const bool p2 = (quad_bcastu<u>(mk.z) & lanebit) != 0u, p3 = (quad_bcastu<u>(mk.w) & lanebit) != 0u;     \
        const f2 g01 = {v[u].x, v[u].y}, g23 = {v[u].z, v[u].w};                                                 \
        const f2 alS2 = {alS_u, alS_u}, alT2 = {alT_u, alT_u};                                                   \
        accS01 = __builtin_elementwise_fma(alS2, g01, accS01); accS23 = __builtin_elementwise_fma(alS2, g23, accS23); \
        accT01 = __builtin_elementwise_fma(alT2, g01, accT01); accT23 = __builtin_elementwise_fma(alT2, g23, accT23); \
        wS01 += f2{p0 ? deS_u : desS_u, p1 ? deS_u : desS_u}; wS23 += f2{p2 ? deS_u : desS_u, p3 ? deS_u : desS_u}; \
        wT01 += f2{p0 ? deT_u : desT_u, p1 ? deT_u : desT_u}; wT23 += f2{p2 ? deT_u : desT_u, p3 ? deT_u : desT_u}; \
      }
      BGNN_BWD_EDGE(0) BGNN_BWD_EDGE(1) BGNN_BWD_EDGE(2) BGNN_BWD_EDGE(3)
#undef BGNN_BWD_EDGE
      ok = ok2;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(nexteid), "+v"(nextdst), "+v"(accS01), "+v"(accS23), "+v"(accT01), "+v"(accT23),
                   "+v"(wS01), "+v"(wS23), "+v"(wT01), "+v"(wT23) : : "memory");
      goff = ok ? __umul24(nextdst, gstride) : gdead;
      roff = ok ? nexteid * 32u : OOB;
    }
    if (rvalid && f0 < p.ldh) {
      const bool dom_j = p.mask[j0] != 0;
      const uint32_t ooff = __umul24((uint32_t)j0, nstride) + (uint32_t)f0 * 4u;
      const float4 ds4 = ld128(rds, ooff);
      f2 oS01 = __builtin_elementwise_fma(aS01, wS01, accS01), oS23 = __builtin_elementwise_fma(aS23, wS23, accS23);
      f2 oT01 = __builtin_elementwise_fma(aT01, wT01, accT01), oT23 = __builtin_elementwise_fma(aT23, wT23, accT23);
      if (!fvalid) { oS01 = f2{0.f, 0.f}; oS23 = oS01; oT01 = oS01; oT23 = oS01; }
      const f2 d01 = {ds4.x, ds4.y}, d23 = {ds4.z, ds4.w};
      if (dom_j) { oS01 += d01; oS23 += d23; }
      else       { oT01 += d01; oT23 += d23; }
      const u32x4_t uS = __builtin_bit_cast(u32x4_t, make_float4(oS01.x, oS01.y, oS23.x, oS23.y));
      const u32x4_t uT = __builtin_bit_cast(u32x4_t, make_float4(oT01.x, oT01.y, oT23.x, oT23.y));
      __builtin_amdgcn_raw_buffer_store_b128(uS, rdS, ooff, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(uT, rdT, ooff, 0, 0);
    }
  }
}

int resident_cap() {
  static const int cap = [] {
    int a = 0, b = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2048;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, agg_bwd_dst_fast_kernel, 256, 0) != hipSuccess || a < 1) return 2048;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, agg_bwd_src_fast_kernel, 256, 0) != hipSuccess || b < 1) return 2048;
    int per_cu = a < b ? a : b;
    if (per_cu > 8) per_cu = 8;
    return per_cu * prop.multiProcessorCount / 8 * 8;
  }();
  return cap;
}

}  // namespace

namespace bgnn_bwd {

bool pull_fast_plan(PullParams& p) {
  static const bool on = [] { const char* e = getenv("BGNN_AGG_FAST"); return !(e && atoi(e) == 0); }();
  if (!on || p.hub_threshold != 0 || p.d_nv != 0 || p.s_nv != 0) return false;
  const int nv = (p.D + 3) / 4;
  if (nv <= 16 || nv > 32 || !(p.slope >= 0.f && p.slope <= 1.f)) return false;     // (sel = de or slope * de needs nothing of slope; kept equal to the forward's envelope)
  const int64_t lim24 = (int64_t)1 << 24, lim32 = (int64_t)0xFFFFFFF0u;
  if (p.N > lim24 || p.ldh * 4 >= lim24 || p.ldg * 4 >= lim24 || p.ldo * 4 >= lim24) return false;
  if (p.N * p.ldh * 4 > lim32 || p.N * p.ldg * 4 > lim32 || p.N * p.ldo * 4 > lim32 || p.E * 32 > lim32) return false;
  const char* a = reinterpret_cast<const char*>(p.h_t2s);
  const char* b = reinterpret_cast<const char*>(p.h_s2t);
  const char* lo = a < b ? a : b;
  const int64_t span = (int64_t)((a < b ? b : a) - lo) + p.N * p.ldh * 4;
  if (span > lim32) return false;
  p.tbl_base = lo;
  p.tbl_bytes = (uint32_t)span;
  p.off_t2s = (uint32_t)(a - lo);
  p.off_s2t = (uint32_t)(b - lo);
  const int64_t lane_max = (span - p.N * p.ldh * 4) + p.ldh * 4, dead = (((int64_t)1 << 32) - 16) - lane_max;
  p.dead_off = dead >= span ? (uint32_t)dead : 0u;
  return true;
}

int pull_fast_launch(const PullParams& p, hipStream_t st) {
  constexpr int RPB = 8;
  const int64_t ntiles = (p.N + RPB - 1) / RPB;
  const int64_t cap = resident_cap();
  int64_t grid = ntiles < cap ? (ntiles + 7) / 8 * 8 : cap;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL(agg_bwd_dst_fast_kernel, dim3((unsigned)grid), dim3(256), 0, st, p);
  BGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(agg_bwd_src_fast_kernel, dim3((unsigned)grid), dim3(256), 0, st, p);
  BGNN_LAUNCH_CHECK();
  return 0;
}

}  // namespace bgnn_bwd
