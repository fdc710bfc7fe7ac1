// Backward of the fused AdaptedConv aggregation (reference: autograd through
// Bridged-GNN/models/KTGNN.py:292-305 as driven by main_graph_knowledge_transfer.py:39-68).
//
// Forward (per destination i, H = table of i's domain, a = its attention vector):
//   e_ji = a . leaky(h_j + h_i),  alpha = softmax_j(e_ji),  out_i = sum_j alpha_ji h_j
// Given g_i = dL/dout_i and the saved alpha / out:
//   c_ji  = g_i . h_j ;  t_i = g_i . out_i ;  de_ji = alpha_ji (c_ji - t_i)
//   dH[j] += alpha_ji g_i + de_ji (a * leaky'(h_j + h_i))        (source side, scattered)
//   dH[i] += sum_j de_ji (a * leaky'(h_j + h_i))                 (destination side, per row)
//   da    += sum_ji de_ji leaky(h_j + h_i)
// One pass over the by-destination CSR (same persistent XCD-contiguous tiling as the forward).  The
// scattered source-side sums use hardware fp32 atomics; to keep every atomic wave-instruction on
// contiguous 128-B row segments (MI355X_MICROARCH.md "Global float atomics": access shape) lane l of a
// row group owns the feature columns {l, l+LF, l+2LF, l+3LF} instead of a float4.
// Atomic sums are order-dependent in the last bits (like torch's scatter_add backward on GPUs).
#include "bgnn_common.h"
#include "bgnn_aggregate_bwd_params.h"

namespace {

struct BwdParams {
  const float* h_t2s; const float* h_s2t; int64_t ldh;
  const float* a_t2s; const float* a_s2t;
  const int32_t* rowptr; const int32_t* col; const uint8_t* mask;
  int64_t row_begin, row_end; int32_t D; float slope;
  const float* out; int64_t ldo; const float* alpha; const float* gout; int64_t ldg;
  float* dh_t2s; float* dh_s2t; float* da_t2s; float* da_s2t;
};

template <int LF, int U>
__global__ __launch_bounds__(256) void agg_bwd_kernel(BwdParams p) {
  constexpr int GPW = 64 / LF, RPB = 4 * GPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LF, l = lane % LF;
  int cidx[4];
  bool cok[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { cidx[c] = c * LF + l; cok[c] = cidx[c] < p.D; }
  float accS[4] = {0.f, 0.f, 0.f, 0.f}, accT[4] = {0.f, 0.f, 0.f, 0.f};   // da partials per domain

  const int64_t ntiles = (p.row_end - p.row_begin + RPB - 1) / RPB;
  bgnn::XcdRange tr = bgnn::xcd_pos_range(ntiles);   // positions of this XCD's segment sequence (XCD balance)
  for (int64_t pos = tr.begin; pos < tr.end; pos += tr.step) {
    const int64_t tile = bgnn::xcd_tile_of(pos, ntiles);
    if (tile < 0) continue;
    const int64_t i = p.row_begin + tile * RPB + wave * GPW + g;
    const bool rvalid = i < p.row_end;
    const int64_t ic = rvalid ? i : p.row_begin;
    const bool dom_s = p.mask[ic] != 0;
    const float* __restrict__ H = dom_s ? p.h_t2s : p.h_s2t;
    float* __restrict__ dH = dom_s ? p.dh_t2s : p.dh_s2t;
    const float* __restrict__ av = dom_s ? p.a_t2s : p.a_s2t;
    const int32_t beg = rvalid ? p.rowptr[ic] : 0, end = rvalid ? p.rowptr[ic + 1] : 0;
    float gi[4], hi[4], a[4], accd[4];
    float tpart = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      gi[c] = (cok[c] && rvalid) ? p.gout[ic * p.ldg + cidx[c]] : 0.f;
      hi[c] = cok[c] ? H[ic * p.ldh + cidx[c]] : 0.f;
      a[c] = cok[c] ? av[cidx[c]] : 0.f;
      const float oi = cok[c] ? p.out[ic * p.ldo + cidx[c]] : 0.f;
      tpart = fmaf(gi[c], oi, tpart);
      accd[c] = 0.f;
    }
    const float ti = bgnn::group_sum<LF>(tpart);
    const int32_t niter = (end - beg + U - 1) / U;
    for (int32_t it = 0; it < niter; ++it) {
      int32_t id[U];
      float al[U], hj[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t e = beg + it * U + u;
        id[u] = e < end ? p.col[e] : -1;
        al[u] = e < end ? p.alpha[e] : 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) hj[u][c] = (id[u] >= 0 && cok[c]) ? H[(int64_t)id[u] * p.ldh + cidx[c]] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float cp = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) cp = fmaf(gi[c], hj[u][c], cp);
        const float cdot = bgnn::group_sum<LF>(cp);
        const float de = al[u] * (cdot - ti);
        if (id[u] >= 0) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float z = hj[u][c] + hi[c];
            const bool pos = z > 0.f;
            const float w = de * a[c] * (pos ? 1.f : p.slope);
            accd[c] += w;
            const float lz = de * (pos ? z : z * p.slope);
            accS[c] += dom_s ? lz : 0.f;
            accT[c] += dom_s ? 0.f : lz;
            if (cok[c]) unsafeAtomicAdd(&dH[(int64_t)id[u] * p.ldh + cidx[c]], fmaf(al[u], gi[c], w));
          }
        }
      }
    }
    if (rvalid) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (cok[c]) unsafeAtomicAdd(&dH[i * p.ldh + cidx[c]], accd[c]);
    }
  }
  // da: fold the row groups of the wave (lanes with equal l), then one atomic per (wave, column, domain)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int off = LF; off < 64; off <<= 1) { accS[c] += __shfl_xor(accS[c], off); accT[c] += __shfl_xor(accT[c], off); }
    if (g == 0 && cok[c]) { unsafeAtomicAdd(&p.da_t2s[cidx[c]], accS[c]); unsafeAtomicAdd(&p.da_s2t[cidx[c]], accT[c]); }
  }
}

template <int LF, int U>
int launch_bwd(const BwdParams& p, hipStream_t st) {
  constexpr int RPB = 4 * (64 / LF);
  static const int cap = [] {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2048;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, agg_bwd_kernel<LF, U>, 256, 0) != hipSuccess || per_cu < 1) return 2048;
    if (per_cu > 8) per_cu = 8;
    return per_cu * prop.multiProcessorCount / 8 * 8;
  }();
  const int64_t ntiles = (p.row_end - p.row_begin + RPB - 1) / RPB;
  int64_t grid = ntiles < cap ? (ntiles + 7) / 8 * 8 : cap;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL((agg_bwd_kernel<LF, U>), dim3((unsigned)grid), dim3(256), 0, st, p);
  BGNN_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// Atomic-free ("pull") form (D <= 128; rows of one float4 with ldh = 4 take the narrow kernels further down).  The scatter above moves E' x 4D bytes through float
// atomics, which retire at ~1.3 TB/s on MI355X (8.96 ms for the C4 hidden conv).  Here the source-side sums are GATHERED
// over a by-source (transposed) CSR instead:
//   pass A (by destination, same walk as the forward): t_i, c_ji, de_ji; the destination-side sums go to a plain
//          per-row buffer; per edge a 32-byte record {alpha, de, domain(i), -, 4 x 32-bit sign masks of h_j + h_i}
//          is written in CSR order (so pass B needs neither h_i nor the attention recomputation);
//   pass B (by source): for every out-edge j -> i gather dL/dout_i (4 lines) and the record (1 line), rebuild
//          alpha g_i + de (a * leaky') in registers and write each dH row exactly once (no zero-fill, no atomics,
//          deterministic).
using bgnn_bwd::PullParams;

template <int LF>
__global__ __launch_bounds__(256) void agg_bwd_dst_kernel(PullParams p) {
  constexpr int GPW = 64 / LF, RPB = 4 * GPW, U = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LF, l = lane % LF;
  const int f0 = l * 4;
  const bool fvalid = f0 < p.D;
  const int f0c = fvalid ? f0 : 0;
  float4 accS = make_float4(0.f, 0.f, 0.f, 0.f), accT = accS;          // da partials per domain
  const int64_t ntiles = (p.N + p.d_nv + RPB - 1) / RPB;     // real rows, then the hub rows' segments
  bgnn::XcdRange tr = bgnn::xcd_pos_range(ntiles);   // positions of this XCD's segment sequence (XCD balance)
  __shared__ unsigned int dyn_tile;
  const int64_t xbase = tr.begin - (blockIdx.x / 8);
  constexpr int TQ_CHUNK = 4;                     // tiles per queue fetch (same-address atomics retire at ~11 M/s)
  int64_t tile = 0, chunk_left = 0;
  for (;;) {
    // per-XCD dynamic tile queue (same reason as in the forward: the blocks of an XCD stay on neighbouring rows)
    if (chunk_left == 0) {
      __syncthreads();
      if (threadIdx.x == 0) dyn_tile = atomicAdd(&p.queue[blockIdx.x % 8], 1u);
      __syncthreads();
      tile = xbase + (int64_t)dyn_tile * TQ_CHUNK;
      chunk_left = TQ_CHUNK;
    } else {
      tile += 1;
    }
    --chunk_left;
    if (tile >= tr.end) break;
    const int64_t gt = bgnn::xcd_tile_of(tile, ntiles);     // `tile` is a position in the XCD's sequence
    if (gt < 0) continue;
    const int64_t i0 = gt * RPB + wave * GPW + g;
    const bool in_range = i0 < p.N + p.d_nv;
    const bool virt = in_range && i0 >= p.N;                   // a segment of a hub destination
    const int64_t vix = virt ? i0 - p.N : 0;
    const int64_t i = virt ? (int64_t)p.d_vnode[vix] : i0;
    const int64_t ic = in_range ? i : 0;
    const bool dom_s = p.mask[ic] != 0;
    const float* __restrict__ H = dom_s ? p.h_t2s : p.h_s2t;
    const float* __restrict__ av = dom_s ? p.a_t2s : p.a_s2t;
    int32_t beg = in_range ? (virt ? p.d_vbounds[2 * vix] : p.rowptr[ic]) : 0;
    int32_t end = in_range ? (virt ? p.d_vbounds[2 * vix + 1] : p.rowptr[ic + 1]) : 0;
    const bool hub = !virt && p.hub_threshold > 0 && end - beg >= p.hub_threshold;
    const bool rvalid = in_range && !hub;
    if (hub) { beg = 0; end = 0; }
    float4 gi = make_float4(0.f, 0.f, 0.f, 0.f), a4 = gi, oi = gi;
    const float4 hi = *reinterpret_cast<const float4*>(H + ic * p.ldh + f0c);
    if (fvalid) {
      if (rvalid) gi = *reinterpret_cast<const float4*>(p.gout + ic * p.ldg + f0);
      oi = *reinterpret_cast<const float4*>(p.out + ic * p.ldo + f0);
      a4.x = av[f0]; a4.y = f0 + 1 < p.D ? av[f0 + 1] : 0.f; a4.z = f0 + 2 < p.D ? av[f0 + 2] : 0.f; a4.w = f0 + 3 < p.D ? av[f0 + 3] : 0.f;
    }
    const float ti = bgnn::group_sum<LF>(gi.x * oi.x + gi.y * oi.y + gi.z * oi.z + gi.w * oi.w);
    float4 accd = make_float4(0.f, 0.f, 0.f, 0.f), accz = accd;
    const int32_t niter = (end - beg + U - 1) / U;
    int32_t nid[U];
    float nal[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t e = beg + u;
      nid[u] = e < end ? p.col[e] : -1;
      nal[u] = e < end ? p.alpha[e] : 0.f;
    }
    for (int32_t it = 0; it < niter; ++it) {
      int32_t id[U];
      float al[U];
      float4 hj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        id[u] = nid[u]; al[u] = nal[u];
        hj[u] = *reinterpret_cast<const float4*>(H + (int64_t)max(id[u], 0) * p.ldh + f0c);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {                 // ids / alphas of the next step fly with this step's rows
        const int32_t e = beg + (it + 1) * U + u;
        nid[u] = e < end ? p.col[e] : -1;
        nal[u] = e < end ? p.alpha[e] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float cdot = bgnn::group_sum<LF>(gi.x * hj[u].x + gi.y * hj[u].y + gi.z * hj[u].z + gi.w * hj[u].w);
        const float de = id[u] >= 0 ? al[u] * (cdot - ti) : 0.f;
        const float zx = hj[u].x + hi.x, zy = hj[u].y + hi.y, zz = hj[u].z + hi.z, zw = hj[u].w + hi.w;
        const bool px = zx > 0.f, py = zy > 0.f, pz = zz > 0.f, pw = zw > 0.f;
        accd.x += de * a4.x * (px ? 1.f : p.slope); accd.y += de * a4.y * (py ? 1.f : p.slope);
        accd.z += de * a4.z * (pz ? 1.f : p.slope); accd.w += de * a4.w * (pw ? 1.f : p.slope);
        accz.x += de * (px ? zx : zx * p.slope); accz.y += de * (py ? zy : zy * p.slope);
        accz.z += de * (pz ? zz : zz * p.slope); accz.w += de * (pw ? zw : zw * p.slope);
        const unsigned long long bx = __ballot(px), by = __ballot(py), bz = __ballot(pz), bw = __ballot(pw);
        if (l == 0 && id[u] >= 0) {
          const int sh = g * LF;
          const uint32_t lm = LF == 32 ? 0xFFFFFFFFu : ((1u << (LF & 31)) - 1u);
          uint4 hd, mk;
          hd.x = __float_as_uint(al[u]); hd.y = __float_as_uint(de); hd.z = dom_s ? 1u : 0u; hd.w = 0u;
          mk.x = (uint32_t)(bx >> sh) & lm; mk.y = (uint32_t)(by >> sh) & lm;
          mk.z = (uint32_t)(bz >> sh) & lm; mk.w = (uint32_t)(bw >> sh) & lm;
          uint4* r = p.rec + (int64_t)(beg + it * U + u) * 2;
          r[0] = hd; r[1] = mk;
        }
      }
    }
    if (rvalid && f0 < p.ldh)
      *reinterpret_cast<float4*>((virt ? p.d_vpart + vix * p.ldh : p.dstside + i * p.ldh) + f0) = fvalid ? accd : make_float4(0.f, 0.f, 0.f, 0.f);
    if (rvalid && fvalid) {
      if (dom_s) { accS.x += accz.x; accS.y += accz.y; accS.z += accz.z; accS.w += accz.w; }
      else       { accT.x += accz.x; accT.y += accz.y; accT.z += accz.z; accT.w += accz.w; }
    }
  }
  // da: block reduction through LDS, one atomic per (block, column, domain)
  __shared__ float red[2][LF * 4];
  for (int t = threadIdx.x; t < 2 * LF * 4; t += 256) (&red[0][0])[t] = 0.f;
  __syncthreads();
  unsafeAtomicAdd(&red[0][f0], accS.x); unsafeAtomicAdd(&red[0][f0 + 1], accS.y); unsafeAtomicAdd(&red[0][f0 + 2], accS.z); unsafeAtomicAdd(&red[0][f0 + 3], accS.w);
  unsafeAtomicAdd(&red[1][f0], accT.x); unsafeAtomicAdd(&red[1][f0 + 1], accT.y); unsafeAtomicAdd(&red[1][f0 + 2], accT.z); unsafeAtomicAdd(&red[1][f0 + 3], accT.w);
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * LF * 4; t += 256) {
    const int d = t / (LF * 4), c = t % (LF * 4);
    if (c < p.D) unsafeAtomicAdd(d == 0 ? &p.da_t2s[c] : &p.da_s2t[c], red[d][c]);
  }
}

template <int LF>
__global__ __launch_bounds__(256) void agg_bwd_src_kernel(PullParams p) {
  constexpr int GPW = 64 / LF, RPB = 4 * GPW, U = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LF, l = lane % LF;
  const int f0 = l * 4;
  const bool fvalid = f0 < p.D;
  const int f0c = fvalid ? f0 : 0;
  float4 aS = make_float4(0.f, 0.f, 0.f, 0.f), aT = aS;
  if (fvalid) {
    aS.x = p.a_t2s[f0]; aS.y = f0 + 1 < p.D ? p.a_t2s[f0 + 1] : 0.f; aS.z = f0 + 2 < p.D ? p.a_t2s[f0 + 2] : 0.f; aS.w = f0 + 3 < p.D ? p.a_t2s[f0 + 3] : 0.f;
    aT.x = p.a_s2t[f0]; aT.y = f0 + 1 < p.D ? p.a_s2t[f0 + 1] : 0.f; aT.z = f0 + 2 < p.D ? p.a_s2t[f0 + 2] : 0.f; aT.w = f0 + 3 < p.D ? p.a_s2t[f0 + 3] : 0.f;
  }
  const int64_t ntiles = (p.N + p.s_nv + RPB - 1) / RPB;     // real rows, then the hub sources' segments
  bgnn::XcdRange tr = bgnn::xcd_pos_range(ntiles);   // positions of this XCD's segment sequence (XCD balance)
  __shared__ unsigned int dyn_tile;
  const int64_t xbase = tr.begin - (blockIdx.x / 8);
  constexpr int TQ_CHUNK = 4;
  int64_t tile = 0, chunk_left = 0;
  for (;;) {
    if (chunk_left == 0) {
      __syncthreads();
      if (threadIdx.x == 0) dyn_tile = atomicAdd(&p.queue[8 + blockIdx.x % 8], 1u);
      __syncthreads();
      tile = xbase + (int64_t)dyn_tile * TQ_CHUNK;
      chunk_left = TQ_CHUNK;
    } else {
      tile += 1;
    }
    --chunk_left;
    if (tile >= tr.end) break;
    const int64_t gt = bgnn::xcd_tile_of(tile, ntiles);     // `tile` is a position in the XCD's sequence
    if (gt < 0) continue;
    const int64_t j0 = gt * RPB + wave * GPW + g;
    const bool in_range = j0 < p.N + p.s_nv;
    const bool virt = in_range && j0 >= p.N;                   // a segment of a hub source
    const int64_t vix = virt ? j0 - p.N : 0;
    const int64_t j = virt ? (int64_t)p.s_vnode[vix] : j0;
    const int64_t jc = in_range ? j : 0;
    int32_t beg = in_range ? (virt ? p.s_vbounds[2 * vix] : p.t_rowptr[jc]) : 0;
    int32_t end = in_range ? (virt ? p.s_vbounds[2 * vix + 1] : p.t_rowptr[jc + 1]) : 0;
    const bool hub = !virt && p.hub_threshold > 0 && end - beg >= p.hub_threshold;
    const bool rvalid = in_range && !hub;
    if (hub) { beg = 0; end = 0; }
    float4 accS = make_float4(0.f, 0.f, 0.f, 0.f), accT = accS;
    const int32_t niter = (end - beg + U - 1) / U;
    int32_t ne[U], ni[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t k = beg + u;
      ne[u] = k < end ? p.t_eid[k] : -1;
      ni[u] = k < end ? p.t_dst[k] : 0;
    }
    for (int32_t it = 0; it < niter; ++it) {
      uint4 hd[U], mk[U];
      float4 g4[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = ne[u] >= 0;
        const uint4* r = p.rec + (int64_t)max(ne[u], 0) * 2;
        hd[u] = r[0]; mk[u] = r[1];
        if (!ok) { hd[u].x = 0u; hd[u].y = 0u; }                 // alpha = de = 0: no contribution
        g4[u] = *reinterpret_cast<const float4*>(p.gout + (int64_t)ni[u] * p.ldg + f0c);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {                 // edge ids / destinations of the next step fly with this step's gathers
        const int32_t k = beg + (it + 1) * U + u;
        ne[u] = k < end ? p.t_eid[k] : -1;
        ni[u] = k < end ? p.t_dst[k] : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float al = __uint_as_float(hd[u].x), de = __uint_as_float(hd[u].y);
        const bool ds = hd[u].z != 0u;
        const float4 a4 = ds ? aS : aT;
        float4 v;
        v.x = fmaf(al, g4[u].x, de * a4.x * (((mk[u].x >> l) & 1u) ? 1.f : p.slope));
        v.y = fmaf(al, g4[u].y, de * a4.y * (((mk[u].y >> l) & 1u) ? 1.f : p.slope));
        v.z = fmaf(al, g4[u].z, de * a4.z * (((mk[u].z >> l) & 1u) ? 1.f : p.slope));
        v.w = fmaf(al, g4[u].w, de * a4.w * (((mk[u].w >> l) & 1u) ? 1.f : p.slope));
        if (ds) { accS.x += v.x; accS.y += v.y; accS.z += v.z; accS.w += v.w; }
        else    { accT.x += v.x; accT.y += v.y; accT.z += v.z; accT.w += v.w; }
      }
    }
    if (virt && f0 < p.ldh) {                                   // a segment: its partial sums, merged afterwards
      if (!fvalid) { accS = make_float4(0.f, 0.f, 0.f, 0.f); accT = accS; }
      *reinterpret_cast<float4*>(p.s_vpartS + vix * p.ldh + f0) = accS;
      *reinterpret_cast<float4*>(p.s_vpartT + vix * p.ldh + f0) = accT;
    } else if (rvalid && f0 < p.ldh) {
      const bool dom_j = p.mask[j] != 0;
      const float4 ds4 = *reinterpret_cast<const float4*>(p.dstside + j * p.ldh + f0);
      if (!fvalid) { accS = make_float4(0.f, 0.f, 0.f, 0.f); accT = accS; }
      if (dom_j) { accS.x += ds4.x; accS.y += ds4.y; accS.z += ds4.z; accS.w += ds4.w; }
      else       { accT.x += ds4.x; accT.y += ds4.y; accT.z += ds4.z; accT.w += ds4.w; }
      *reinterpret_cast<float4*>(p.dh_t2s + j * p.ldh + f0) = accS;
      *reinterpret_cast<float4*>(p.dh_s2t + j * p.ldh + f0) = accT;
    }
  }
}

// Narrow rows (D <= 4, one float4 per row: KT-GNN's classifier convs).  Same two passes with EP lanes per row walking
// different edges (consecutive lanes -> consecutive 16-byte records); the record is {alpha, de, domain | sign bits, -}.
template <int EP>
__global__ __launch_bounds__(256) void agg_bwd_dst_narrow_kernel(PullParams p) {
  constexpr int GPW = 64 / EP, RPB = 4 * GPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / EP, sub = lane % EP;
  float4 accS = make_float4(0.f, 0.f, 0.f, 0.f), accT = accS;
  const int64_t ntiles = (p.N + RPB - 1) / RPB;
  bgnn::XcdRange tr = bgnn::xcd_pos_range(ntiles);   // positions of this XCD's segment sequence (XCD balance)
  for (int64_t pos = tr.begin; pos < tr.end; pos += tr.step) {
    const int64_t tile = bgnn::xcd_tile_of(pos, ntiles);
    if (tile < 0) continue;
    const int64_t i = tile * RPB + wave * GPW + g;
    const bool rvalid = i < p.N;
    const int64_t ic = rvalid ? i : 0;
    const bool dom_s = p.mask[ic] != 0;
    const float* __restrict__ H = dom_s ? p.h_t2s : p.h_s2t;
    const float* __restrict__ av = dom_s ? p.a_t2s : p.a_s2t;
    const int32_t beg = rvalid ? p.rowptr[ic] : 0, end = rvalid ? p.rowptr[ic + 1] : 0;
    float4 a4;
    a4.x = av[0]; a4.y = p.D > 1 ? av[1] : 0.f; a4.z = p.D > 2 ? av[2] : 0.f; a4.w = p.D > 3 ? av[3] : 0.f;
    const float4 hi = *reinterpret_cast<const float4*>(H + ic * 4);
    float4 gi = *reinterpret_cast<const float4*>(p.gout + ic * 4);
    if (!rvalid) gi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.D < 4) gi.w = 0.f;
    if (p.D < 3) gi.z = 0.f;
    if (p.D < 2) gi.y = 0.f;
    const float4 oi = *reinterpret_cast<const float4*>(p.out + ic * 4);
    const float ti = gi.x * oi.x + gi.y * oi.y + gi.z * oi.z + gi.w * oi.w;
    float4 accd = make_float4(0.f, 0.f, 0.f, 0.f), accz = accd;
    for (int32_t e = beg + sub; e < end; e += EP) {
      const int32_t j = p.col[e];
      const float al = p.alpha[e];
      const float4 hj = *reinterpret_cast<const float4*>(H + (int64_t)j * 4);
      const float de = al * (gi.x * hj.x + gi.y * hj.y + gi.z * hj.z + gi.w * hj.w - ti);
      const float zx = hj.x + hi.x, zy = hj.y + hi.y, zz = hj.z + hi.z, zw = hj.w + hi.w;
      const bool px = zx > 0.f, py = zy > 0.f, pz = zz > 0.f, pw = zw > 0.f;
      accd.x += de * a4.x * (px ? 1.f : p.slope); accd.y += de * a4.y * (py ? 1.f : p.slope);
      accd.z += de * a4.z * (pz ? 1.f : p.slope); accd.w += de * a4.w * (pw ? 1.f : p.slope);
      accz.x += de * (px ? zx : zx * p.slope); accz.y += de * (py ? zy : zy * p.slope);
      accz.z += de * (pz ? zz : zz * p.slope); accz.w += de * (pw ? zw : zw * p.slope);
      uint4 r;
      r.x = __float_as_uint(al); r.y = __float_as_uint(de);
      r.z = (dom_s ? 16u : 0u) | (px ? 1u : 0u) | (py ? 2u : 0u) | (pz ? 4u : 0u) | (pw ? 8u : 0u);
      r.w = 0u;
      p.rec[e] = r;
    }
#pragma unroll
    for (int off = 1; off < EP; off <<= 1) {
      accd.x += __shfl_xor(accd.x, off); accd.y += __shfl_xor(accd.y, off); accd.z += __shfl_xor(accd.z, off); accd.w += __shfl_xor(accd.w, off);
    }
    if (rvalid && sub == 0) *reinterpret_cast<float4*>(p.dstside + i * 4) = accd;
    if (rvalid) {
      if (dom_s) { accS.x += accz.x; accS.y += accz.y; accS.z += accz.z; accS.w += accz.w; }
      else       { accT.x += accz.x; accT.y += accz.y; accT.z += accz.z; accT.w += accz.w; }
    }
  }
  // da: wave reduction, then one atomic per (wave, column, domain)
  float v[8] = {accS.x, accS.y, accS.z, accS.w, accT.x, accT.y, accT.z, accT.w};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    v[c] = bgnn::group_sum<64>(v[c]);
    if (lane == 0 && (c & 3) < p.D) unsafeAtomicAdd(c < 4 ? &p.da_t2s[c] : &p.da_s2t[c - 4], v[c]);
  }
}

template <int EP>
__global__ __launch_bounds__(256) void agg_bwd_src_narrow_kernel(PullParams p) {
  constexpr int GPW = 64 / EP, RPB = 4 * GPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / EP, sub = lane % EP;
  float4 aS, aT;
  aS.x = p.a_t2s[0]; aS.y = p.D > 1 ? p.a_t2s[1] : 0.f; aS.z = p.D > 2 ? p.a_t2s[2] : 0.f; aS.w = p.D > 3 ? p.a_t2s[3] : 0.f;
  aT.x = p.a_s2t[0]; aT.y = p.D > 1 ? p.a_s2t[1] : 0.f; aT.z = p.D > 2 ? p.a_s2t[2] : 0.f; aT.w = p.D > 3 ? p.a_s2t[3] : 0.f;
  const int64_t ntiles = (p.N + RPB - 1) / RPB;
  bgnn::XcdRange tr = bgnn::xcd_pos_range(ntiles);   // positions of this XCD's segment sequence (XCD balance)
  for (int64_t pos = tr.begin; pos < tr.end; pos += tr.step) {
    const int64_t tile = bgnn::xcd_tile_of(pos, ntiles);
    if (tile < 0) continue;
    const int64_t j = tile * RPB + wave * GPW + g;
    const bool rvalid = j < p.N;
    const int64_t jc = rvalid ? j : 0;
    const int32_t beg = rvalid ? p.t_rowptr[jc] : 0, end = rvalid ? p.t_rowptr[jc + 1] : 0;
    float4 accS = make_float4(0.f, 0.f, 0.f, 0.f), accT = accS;
    for (int32_t k = beg + sub; k < end; k += EP) {
      const int32_t e = p.t_eid[k], i = p.t_dst[k];
      const uint4 r = p.rec[e];
      float4 g4 = *reinterpret_cast<const float4*>(p.gout + (int64_t)i * 4);
      if (p.D < 4) g4.w = 0.f;
      if (p.D < 3) g4.z = 0.f;
      if (p.D < 2) g4.y = 0.f;
      const float al = __uint_as_float(r.x), de = __uint_as_float(r.y);
      const bool ds = (r.z & 16u) != 0u;
      const float4 a4 = ds ? aS : aT;
      float4 v;
      v.x = fmaf(al, g4.x, de * a4.x * ((r.z & 1u) ? 1.f : p.slope));
      v.y = fmaf(al, g4.y, de * a4.y * ((r.z & 2u) ? 1.f : p.slope));
      v.z = fmaf(al, g4.z, de * a4.z * ((r.z & 4u) ? 1.f : p.slope));
      v.w = fmaf(al, g4.w, de * a4.w * ((r.z & 8u) ? 1.f : p.slope));
      if (ds) { accS.x += v.x; accS.y += v.y; accS.z += v.z; accS.w += v.w; }
      else    { accT.x += v.x; accT.y += v.y; accT.z += v.z; accT.w += v.w; }
    }
#pragma unroll
    for (int off = 1; off < EP; off <<= 1) {
      accS.x += __shfl_xor(accS.x, off); accS.y += __shfl_xor(accS.y, off); accS.z += __shfl_xor(accS.z, off); accS.w += __shfl_xor(accS.w, off);
      accT.x += __shfl_xor(accT.x, off); accT.y += __shfl_xor(accT.y, off); accT.z += __shfl_xor(accT.z, off); accT.w += __shfl_xor(accT.w, off);
    }
    if (rvalid && sub == 0) {
      const bool dom_j = p.mask[j] != 0;
      const float4 ds4 = *reinterpret_cast<const float4*>(p.dstside + j * 4);
      if (dom_j) { accS.x += ds4.x; accS.y += ds4.y; accS.z += ds4.z; accS.w += ds4.w; }
      else       { accT.x += ds4.x; accT.y += ds4.y; accT.z += ds4.z; accT.w += ds4.w; }
      *reinterpret_cast<float4*>(p.dh_t2s + j * 4) = accS;
      *reinterpret_cast<float4*>(p.dh_s2t + j * 4) = accT;
    }
  }
}

static int launch_pull_narrow(const PullParams& p, hipStream_t st) {
  constexpr int EP = 8, RPB = 4 * (64 / EP);
  const int64_t ntiles = (p.N + RPB - 1) / RPB;
  int64_t grid = ntiles < 2048 ? (ntiles + 7) / 8 * 8 : 2048;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL((agg_bwd_dst_narrow_kernel<EP>), dim3((unsigned)grid), dim3(256), 0, st, p);
  BGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL((agg_bwd_src_narrow_kernel<EP>), dim3((unsigned)grid), dim3(256), 0, st, p);
  BGNN_LAUNCH_CHECK();
  return 0;
}

// Interleaved narrow heads (KT-GNN's classifier stage under autograd: clf_base(x), clf_target(x), clf_target(T(x)) share the
// graph): the pull form for HEADS convs in ONE walk per pass.  Tables / out / grad_out / dH are [N][HEADS][4] (48-byte rows
// for three heads), the attention vectors [HEADS][D].  Nothing per EDGE is kept, neither by the forward nor between the passes:
// alpha is rebuilt from the finished rows' softmax state (m, s) that the forward leaves in `state_ms`
// (bgnn_adaptedconv_aggregate_f32, part 3), and pass B recomputes alpha / de of an edge j -> i from a 144-byte per-NODE record
// of i that pass A leaves ({h_i, gr_i, m, 1/s, t_i, domain}: 144 MB for C4, L2 / MALL friendly) instead of streaming a
// 32-byte record per edge through HBM twice (672 MB written in destination order, read in source order: measured 1.8 ms for
// the two passes, of which ~1 ms was that stream).  With `log_softmax` the incoming gradient is taken through the row-local
// log_softmax first: gr = g - exp(logp) * sum(g) -- then t_i = gr . out_i can be formed from the log-probabilities themselves
// (sum(gr) = 0 cancels the unknown shift).
struct HeadsBwdParams {
  const float* h_t2s; const float* h_s2t;
  const float* a_t2s; const float* a_s2t;
  const int32_t* rowptr; const int32_t* col; const uint8_t* mask;
  int64_t N; int32_t D; float slope;
  const float* out; const float* state_ms; const float* gout; int log_softmax;
  const int32_t* t_rowptr; const int32_t* t_dst;
  float4* node;          // [N][3 * HEADS]: h_i[HEADS] | gr_i[HEADS] | per head (m, 1/s, t_i, domain)
  float* dstside;        // [N][HEADS][4]
  float* dh_t2s; float* dh_s2t; float* da_t2s; float* da_s2t;
  // hub rows: as in PullParams (segments behind the real rows of each pass, partial rows of HEADS * 4 floats, fixed-order merges)
  int32_t hub_threshold;
  const int32_t* d_vnode; const int32_t* d_vbounds; int64_t d_nv; float* d_vpart;
  const int32_t* s_vnode; const int32_t* s_vbounds; int64_t s_nv; float* s_vpartS; float* s_vpartT;
};

__device__ __forceinline__ float4 mask_cols(float4 v, int D) {
  if (D < 4) v.w = 0.f;
  if (D < 3) v.z = 0.f;
  if (D < 2) v.y = 0.f;
  return v;
}

// alpha and de of one edge and head from the two end points (the forward's logit, same operation order)
struct EdgeTerm { float al, de; float4 lk, lz; };   // lk: leaky'(z) per column, lz: leaky(z)
__device__ __forceinline__ EdgeTerm edge_term(const float4& hj, const float4& hi, const float4& a4, const float4& gr, float m,
                                              float inv, float ti, float slope) {
  const float zx = hj.x + hi.x, zy = hj.y + hi.y, zz = hj.z + hi.z, zw = hj.w + hi.w;
  EdgeTerm r;
  r.lk = make_float4(zx > 0.f ? 1.f : slope, zy > 0.f ? 1.f : slope, zz > 0.f ? 1.f : slope, zw > 0.f ? 1.f : slope);
  r.lz = make_float4(zx * r.lk.x, zy * r.lk.y, zz * r.lk.z, zw * r.lk.w);
  float t = a4.x * r.lz.x;
  t = fmaf(a4.y, r.lz.y, t); t = fmaf(a4.z, r.lz.z, t); t = fmaf(a4.w, r.lz.w, t);
  r.al = __expf(t - m) * inv;
  r.de = r.al * (gr.x * hj.x + gr.y * hj.y + gr.z * hj.z + gr.w * hj.w - ti);
  return r;
}

// Both passes run one lane per (edge slot, head) like agg_heads_lanes_kernel (lanes lg = sub * HEADS + h of a row's group): the
// HEADS lanes of an edge read consecutive 16-byte pieces of one line.
template <int HEADS, int EP, int U>
__global__ __launch_bounds__(256) void agg_heads_bwd_dst_kernel(HeadsBwdParams p) {
  constexpr int GL = EP * HEADS, GPW = 64 / GL, RPB = 4 * GPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / GL, lg = lane % GL;
  const int sub = lg / HEADS, h = lg % HEADS;
  const bool lane_on = g < GPW;
  constexpr int64_t rs = HEADS * 4;
  float4 accS = make_float4(0.f, 0.f, 0.f, 0.f), accT = accS;
  const int64_t ntiles = (p.N + p.d_nv + RPB - 1) / RPB;     // real rows, then the hub destinations' segments
  bgnn::XcdRange tr = bgnn::xcd_pos_range(ntiles);
  for (int64_t pos = tr.begin; pos < tr.end; pos += tr.step) {
    const int64_t tile = bgnn::xcd_tile_of(pos, ntiles);
    if (tile < 0) continue;
    const int64_t i0 = tile * RPB + wave * GPW + g;
    const bool in_range = lane_on && i0 < p.N + p.d_nv;
    const bool virt = in_range && i0 >= p.N;                    // a segment of a hub destination
    const int64_t vix = virt ? i0 - p.N : 0;
    const int64_t i = virt ? (int64_t)p.d_vnode[vix] : i0;
    const bool rvalid = in_range;
    const int64_t ic = rvalid ? i : 0;
    const bool dom_s = p.mask[ic] != 0;
    const float* __restrict__ H = dom_s ? p.h_t2s : p.h_s2t;
    const float* __restrict__ av = dom_s ? p.a_t2s : p.a_s2t;
    int32_t beg = rvalid ? (virt ? p.d_vbounds[2 * vix] : p.rowptr[ic]) : 0;
    int32_t end = rvalid ? (virt ? p.d_vbounds[2 * vix + 1] : p.rowptr[ic + 1]) : 0;
    // a hub row still leaves its node record here (pass B reads it); its edges and its dstside row belong to the segments
    const bool hub = !virt && p.hub_threshold > 0 && end - beg >= p.hub_threshold;
    if (hub) { beg = 0; end = 0; }
    float4 a4;
    a4.x = av[h * p.D];
    a4.y = p.D > 1 ? av[h * p.D + 1] : 0.f;
    a4.z = p.D > 2 ? av[h * p.D + 2] : 0.f;
    a4.w = p.D > 3 ? av[h * p.D + 3] : 0.f;
    const float4 hi = *reinterpret_cast<const float4*>(H + ic * rs + 4 * h);
    float4 gi = mask_cols(*reinterpret_cast<const float4*>(p.gout + ic * rs + 4 * h), p.D);
    if (!rvalid) gi = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 oi = mask_cols(*reinterpret_cast<const float4*>(p.out + ic * rs + 4 * h), p.D);
    if (p.log_softmax) {           // adjoint of the row-local log_softmax (pad columns stay 0)
      const float sg = gi.x + gi.y + gi.z + gi.w;
      gi.x -= expf(oi.x) * sg;
      if (p.D > 1) gi.y -= expf(oi.y) * sg;
      if (p.D > 2) gi.z -= expf(oi.z) * sg;
      if (p.D > 3) gi.w -= expf(oi.w) * sg;
    }
    const float ti = gi.x * oi.x + gi.y * oi.y + gi.z * oi.z + gi.w * oi.w;
    const float mh = p.state_ms[2 * (ic * HEADS + h)];
    const float inv = 1.f / (p.state_ms[2 * (ic * HEADS + h) + 1] + 1e-16f);
    if (rvalid && !virt && sub == 0) {
      float4* nd = p.node + i * (3 * HEADS);
      nd[h] = hi;
      nd[HEADS + h] = gi;
      nd[2 * HEADS + h] = make_float4(mh, inv, ti, dom_s ? 1.f : 0.f);
    }
    float4 accd = make_float4(0.f, 0.f, 0.f, 0.f), accz = accd;
    for (int32_t e0 = beg + sub; e0 < end; e0 += EP * U) {
      int32_t jj[U];
      float4 hj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) jj[u] = e0 + u * EP < end ? p.col[e0 + u * EP] : -1;
#pragma unroll
      for (int u = 0; u < U; ++u) hj[u] = *reinterpret_cast<const float4*>(H + (int64_t)max(jj[u], 0) * rs + 4 * h);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (jj[u] < 0) continue;
        const EdgeTerm t = edge_term(hj[u], hi, a4, gi, mh, inv, ti, p.slope);
        accd.x += t.de * a4.x * t.lk.x; accd.y += t.de * a4.y * t.lk.y;
        accd.z += t.de * a4.z * t.lk.z; accd.w += t.de * a4.w * t.lk.w;
        accz.x += t.de * t.lz.x; accz.y += t.de * t.lz.y; accz.z += t.de * t.lz.z; accz.w += t.de * t.lz.w;
      }
    }
#pragma unroll
    for (int off = 1; off < EP; off <<= 1) {      // the EP edge slots of a (row, head): partner lanes are HEADS * off apart
      const int from = lane_on ? (sub ^ off) * HEADS + h + g * GL : lane;
      accd.x += __shfl(accd.x, from); accd.y += __shfl(accd.y, from);
      accd.z += __shfl(accd.z, from); accd.w += __shfl(accd.w, from);
    }
    if (rvalid && !hub && sub == 0) *reinterpret_cast<float4*>((virt ? p.d_vpart + vix * rs : p.dstside + i * rs) + 4 * h) = accd;
    if (rvalid) {
      if (dom_s) { accS.x += accz.x; accS.y += accz.y; accS.z += accz.z; accS.w += accz.w; }
      else       { accT.x += accz.x; accT.y += accz.y; accT.z += accz.z; accT.w += accz.w; }
    }
  }
  // da: block reduction per (head, column, domain) in LDS, one atomic each per block
  __shared__ float red[2][HEADS][4];
  if (threadIdx.x < 2 * HEADS * 4) (&red[0][0][0])[threadIdx.x] = 0.f;
  __syncthreads();
  if (lane_on) {
    unsafeAtomicAdd(&red[0][h][0], accS.x); unsafeAtomicAdd(&red[0][h][1], accS.y);
    unsafeAtomicAdd(&red[0][h][2], accS.z); unsafeAtomicAdd(&red[0][h][3], accS.w);
    unsafeAtomicAdd(&red[1][h][0], accT.x); unsafeAtomicAdd(&red[1][h][1], accT.y);
    unsafeAtomicAdd(&red[1][h][2], accT.z); unsafeAtomicAdd(&red[1][h][3], accT.w);
  }
  __syncthreads();
  if (threadIdx.x < 2 * HEADS * 4) {
    const int d = threadIdx.x / (HEADS * 4), hh = (threadIdx.x / 4) % HEADS, c = threadIdx.x & 3;
    if (c < p.D) unsafeAtomicAdd(d == 0 ? &p.da_t2s[hh * p.D + c] : &p.da_s2t[hh * p.D + c], red[d][hh][c]);
  }
}

template <int HEADS, int EP, int U>
__global__ __launch_bounds__(256) void agg_heads_bwd_src_kernel(HeadsBwdParams p) {
  constexpr int GL = EP * HEADS, GPW = 64 / GL, RPB = 4 * GPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / GL, lg = lane % GL;
  const int sub = lg / HEADS, h = lg % HEADS;
  const bool lane_on = g < GPW;
  constexpr int64_t rs = HEADS * 4;
  float4 aS, aT;
  aS.x = p.a_t2s[h * p.D]; aS.y = p.D > 1 ? p.a_t2s[h * p.D + 1] : 0.f;
  aS.z = p.D > 2 ? p.a_t2s[h * p.D + 2] : 0.f; aS.w = p.D > 3 ? p.a_t2s[h * p.D + 3] : 0.f;
  aT.x = p.a_s2t[h * p.D]; aT.y = p.D > 1 ? p.a_s2t[h * p.D + 1] : 0.f;
  aT.z = p.D > 2 ? p.a_s2t[h * p.D + 2] : 0.f; aT.w = p.D > 3 ? p.a_s2t[h * p.D + 3] : 0.f;
  const int64_t ntiles = (p.N + p.s_nv + RPB - 1) / RPB;     // real rows, then the hub sources' segments
  bgnn::XcdRange tr = bgnn::xcd_pos_range(ntiles);
  for (int64_t pos = tr.begin; pos < tr.end; pos += tr.step) {
    const int64_t tile = bgnn::xcd_tile_of(pos, ntiles);
    if (tile < 0) continue;
    const int64_t j0 = tile * RPB + wave * GPW + g;
    const bool in_range = lane_on && j0 < p.N + p.s_nv;
    const bool virt = in_range && j0 >= p.N;                    // a segment of a hub source
    const int64_t vix = virt ? j0 - p.N : 0;
    const int64_t j = virt ? (int64_t)p.s_vnode[vix] : j0;
    const int64_t jc = in_range ? j : 0;
    int32_t beg = in_range ? (virt ? p.s_vbounds[2 * vix] : p.t_rowptr[jc]) : 0;
    int32_t end = in_range ? (virt ? p.s_vbounds[2 * vix + 1] : p.t_rowptr[jc + 1]) : 0;
    const bool hub = !virt && p.hub_threshold > 0 && end - beg >= p.hub_threshold;
    const bool rvalid = in_range && !hub;
    if (hub) { beg = 0; end = 0; }
    // row j of both tables: the destination's domain picks one
    const float4 hS = *reinterpret_cast<const float4*>(p.h_t2s + jc * rs + 4 * h);
    const float4 hT = *reinterpret_cast<const float4*>(p.h_s2t + jc * rs + 4 * h);
    float4 accS = make_float4(0.f, 0.f, 0.f, 0.f), accT = accS;
    for (int32_t k0 = beg + sub; k0 < end; k0 += EP * U) {
      int32_t ii[U];
      float4 nh[U], ng[U], ns[U];
#pragma unroll
      for (int u = 0; u < U; ++u) ii[u] = k0 + u * EP < end ? p.t_dst[k0 + u * EP] : -1;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float4* nd = p.node + (int64_t)max(ii[u], 0) * (3 * HEADS);
        nh[u] = nd[h]; ng[u] = nd[HEADS + h]; ns[u] = nd[2 * HEADS + h];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ii[u] < 0) continue;
        const bool ds = ns[u].w != 0.f;              // (m, 1/s, t_i, domain of i)
        const float4 a4 = ds ? aS : aT;
        const EdgeTerm t = edge_term(ds ? hS : hT, nh[u], a4, ng[u], ns[u].x, ns[u].y, ns[u].z, p.slope);
        float4 v;
        v.x = fmaf(t.al, ng[u].x, t.de * a4.x * t.lk.x); v.y = fmaf(t.al, ng[u].y, t.de * a4.y * t.lk.y);
        v.z = fmaf(t.al, ng[u].z, t.de * a4.z * t.lk.z); v.w = fmaf(t.al, ng[u].w, t.de * a4.w * t.lk.w);
        if (ds) { accS.x += v.x; accS.y += v.y; accS.z += v.z; accS.w += v.w; }
        else    { accT.x += v.x; accT.y += v.y; accT.z += v.z; accT.w += v.w; }
      }
    }
    const bool dom_j = rvalid && p.mask[jc] != 0;
#pragma unroll
    for (int off = 1; off < EP; off <<= 1) {
      const int from = lane_on ? (sub ^ off) * HEADS + h + g * GL : lane;
      accS.x += __shfl(accS.x, from); accS.y += __shfl(accS.y, from); accS.z += __shfl(accS.z, from); accS.w += __shfl(accS.w, from);
      accT.x += __shfl(accT.x, from); accT.y += __shfl(accT.y, from); accT.z += __shfl(accT.z, from); accT.w += __shfl(accT.w, from);
    }
    if (virt && sub == 0) {                                      // a segment: its partial sums, merged afterwards
      *reinterpret_cast<float4*>(p.s_vpartS + vix * rs + 4 * h) = accS;
      *reinterpret_cast<float4*>(p.s_vpartT + vix * rs + 4 * h) = accT;
    } else if (rvalid && sub == 0) {
      const float4 ds4 = *reinterpret_cast<const float4*>(p.dstside + j * rs + 4 * h);
      if (dom_j) { accS.x += ds4.x; accS.y += ds4.y; accS.z += ds4.z; accS.w += ds4.w; }
      else       { accT.x += ds4.x; accT.y += ds4.y; accT.z += ds4.z; accT.w += ds4.w; }
      *reinterpret_cast<float4*>(p.dh_t2s + j * rs + 4 * h) = accS;
      *reinterpret_cast<float4*>(p.dh_s2t + j * rs + 4 * h) = accT;
    }
  }
}

// merges of the hub segments' partial rows (fixed order).  One thread per float4 of a hub row.
struct PullMergeParams {
  const int32_t* hub_rows; const int32_t* seg_ptr; int64_t n_hubs; int64_t ldh;
  const float* partA; const float* partB;      // pass A: partA = d_vpart; pass B: partA / partB = s_vpartS / s_vpartT
  const uint8_t* mask; float* dstside; float* dh_t2s; float* dh_s2t;
};
__global__ __launch_bounds__(256) void pull_merge_dst_kernel(PullMergeParams p) {
  const int64_t c4n = p.ldh / 4, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.n_hubs * c4n) return;
  const int64_t h = t / c4n, f0 = (t - h * c4n) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int32_t v = p.seg_ptr[h]; v < p.seg_ptr[h + 1]; ++v) {
    const float4 b = *reinterpret_cast<const float4*>(p.partA + (int64_t)v * p.ldh + f0);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  *reinterpret_cast<float4*>(p.dstside + (int64_t)p.hub_rows[h] * p.ldh + f0) = a;
}
__global__ __launch_bounds__(256) void pull_merge_src_kernel(PullMergeParams p) {
  const int64_t c4n = p.ldh / 4, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.n_hubs * c4n) return;
  const int64_t h = t / c4n, f0 = (t - h * c4n) * 4;
  const int64_t j = p.hub_rows[h];
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
  for (int32_t v = p.seg_ptr[h]; v < p.seg_ptr[h + 1]; ++v) {
    const float4 b = *reinterpret_cast<const float4*>(p.partA + (int64_t)v * p.ldh + f0);
    const float4 d = *reinterpret_cast<const float4*>(p.partB + (int64_t)v * p.ldh + f0);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    c.x += d.x; c.y += d.y; c.z += d.z; c.w += d.w;
  }
  const float4 ds4 = *reinterpret_cast<const float4*>(p.dstside + j * p.ldh + f0);
  if (p.mask[j] != 0) { a.x += ds4.x; a.y += ds4.y; a.z += ds4.z; a.w += ds4.w; }
  else                { c.x += ds4.x; c.y += ds4.y; c.z += ds4.z; c.w += ds4.w; }
  *reinterpret_cast<float4*>(p.dh_t2s + j * p.ldh + f0) = a;
  *reinterpret_cast<float4*>(p.dh_s2t + j * p.ldh + f0) = c;
}
struct PullHubs {        // host side of the hub tables (device pointers)
  const int32_t* d_rows; const int32_t* d_seg_ptr; int64_t d_nh;
  const int32_t* s_rows; const int32_t* s_seg_ptr; int64_t s_nh;
};

template <int HEADS>
int launch_heads_bwd(const HeadsBwdParams& p, hipStream_t st, const PullHubs* hubs = nullptr) {
  // (EP, U) = (2, 4) from a sweep on C4, both passes together: lane per head 2/4 0.69, 1/4 0.70, 1/2 0.73, 2/2 0.74, 4/4 0.74,
  // 4/2 0.80 ms (one lane for all heads: 1.33 ms); tools/heads_bwd_time.py times this pair of launches
  constexpr int EPV = 2, UV = 4, RPB = 4 * (64 / (EPV * HEADS));
  const int64_t nmax = p.N + (p.d_nv > p.s_nv ? p.d_nv : p.s_nv);
  const int64_t ntiles = (nmax + RPB - 1) / RPB;
  int64_t grid = ntiles < 2048 ? (ntiles + 7) / 8 * 8 : 2048;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL((agg_heads_bwd_dst_kernel<HEADS, EPV, UV>), dim3((unsigned)grid), dim3(256), 0, st, p);
  BGNN_LAUNCH_CHECK();
  const int64_t ld = HEADS * 4;
  if (hubs && hubs->d_nh > 0) {            // the hub destinations' dstside rows, before pass B reads them
    PullMergeParams m{hubs->d_rows, hubs->d_seg_ptr, hubs->d_nh, ld, p.d_vpart, nullptr, p.mask, p.dstside, nullptr, nullptr};
    hipLaunchKernelGGL(pull_merge_dst_kernel, dim3((unsigned)((hubs->d_nh * (ld / 4) + 255) / 256)), dim3(256), 0, st, m);
    BGNN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL((agg_heads_bwd_src_kernel<HEADS, EPV, UV>), dim3((unsigned)grid), dim3(256), 0, st, p);
  BGNN_LAUNCH_CHECK();
  if (hubs && hubs->s_nh > 0) {
    PullMergeParams m{hubs->s_rows, hubs->s_seg_ptr, hubs->s_nh, ld, p.s_vpartS, p.s_vpartT, p.mask, p.dstside, p.dh_t2s, p.dh_s2t};
    hipLaunchKernelGGL(pull_merge_src_kernel, dim3((unsigned)((hubs->s_nh * (ld / 4) + 255) / 256)), dim3(256), 0, st, m);
    BGNN_LAUNCH_CHECK();
  }
  return 0;
}

template <int LF>
int launch_pull(const PullParams& p, hipStream_t st, const PullHubs* hubs = nullptr) {
  constexpr int RPB = 4 * (64 / LF);
  static const int cap = [] {
    int a = 0, b = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2048;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, agg_bwd_dst_kernel<LF>, 256, 0) != hipSuccess || a < 1) return 2048;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, agg_bwd_src_kernel<LF>, 256, 0) != hipSuccess || b < 1) return 2048;
    int per_cu = a < b ? a : b;
    if (per_cu > 8) per_cu = 8;
    return per_cu * prop.multiProcessorCount / 8 * 8;
  }();
  const int64_t nmax = p.N + (p.d_nv > p.s_nv ? p.d_nv : p.s_nv);
  const int64_t ntiles = (nmax + RPB - 1) / RPB;
  int64_t grid = ntiles < cap ? (ntiles + 7) / 8 * 8 : cap;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL((agg_bwd_dst_kernel<LF>), dim3((unsigned)grid), dim3(256), 0, st, p);
  BGNN_LAUNCH_CHECK();
  if (hubs && hubs->d_nh > 0) {            // the hub destinations' dstside rows, before pass B reads them
    PullMergeParams m{hubs->d_rows, hubs->d_seg_ptr, hubs->d_nh, p.ldh, p.d_vpart, nullptr, p.mask, p.dstside, nullptr, nullptr};
    hipLaunchKernelGGL(pull_merge_dst_kernel, dim3((unsigned)((hubs->d_nh * (p.ldh / 4) + 255) / 256)), dim3(256), 0, st, m);
    BGNN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL((agg_bwd_src_kernel<LF>), dim3((unsigned)grid), dim3(256), 0, st, p);
  BGNN_LAUNCH_CHECK();
  if (hubs && hubs->s_nh > 0) {
    PullMergeParams m{hubs->s_rows, hubs->s_seg_ptr, hubs->s_nh, p.ldh, p.s_vpartS, p.s_vpartT, p.mask, p.dstside, p.dh_t2s, p.dh_s2t};
    hipLaunchKernelGGL(pull_merge_src_kernel, dim3((unsigned)((hubs->s_nh * (p.ldh / 4) + 255) / 256)), dim3(256), 0, st, m);
    BGNN_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

extern "C" int bgnn_adaptedconv_aggregate_bwd_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
                                                  const float* a_t2s, const float* a_s2t,
                                                  const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                                  int64_t row_begin, int64_t row_end, int32_t D, float negative_slope,
                                                  const float* out, int64_t ldo, const float* alpha,
                                                  const float* grad_out, int64_t ldg,
                                                  float* dh_t2s, float* dh_s2t, float* da_t2s, float* da_s2t,
                                                  void* stream) {
  if (!h_t2s || !h_s2t || !a_t2s || !a_s2t || !rowptr || !col || !mask || !out || !alpha || !grad_out || !dh_t2s ||
      !dh_s2t || !da_t2s || !da_s2t)
    return BGNN_E_NULL;
  if (row_begin < 0 || row_end < row_begin || D <= 0 || D > 256 || ldh < D || ldo < D || ldg < D) return BGNN_E_SHAPE;
  if (row_end == row_begin) return 0;
  BwdParams p{h_t2s, h_s2t, ldh, a_t2s, a_s2t, rowptr, col, mask, row_begin, row_end, D, negative_slope,
              out, ldo, alpha, grad_out, ldg, dh_t2s, dh_s2t, da_t2s, da_s2t};
  hipStream_t st = (hipStream_t)stream;
  const int nv = (D + 3) / 4;
  if (nv <= 1) return launch_bwd<1, 2>(p, st);
  if (nv <= 2) return launch_bwd<2, 2>(p, st);
  if (nv <= 4) return launch_bwd<4, 2>(p, st);
  if (nv <= 8) return launch_bwd<8, 2>(p, st);
  if (nv <= 16) return launch_bwd<16, 2>(p, st);
  if (nv <= 32) return launch_bwd<32, 2>(p, st);
  return launch_bwd<64, 2>(p, st);
}

extern "C" size_t bgnn_aggregate_bwd_pull_workspace_bytes(int64_t N, int64_t E, int64_t ldh) {
  return (size_t)32 * (size_t)(E > 0 ? E : 0) + sizeof(float) * (size_t)(N > 0 ? N : 0) * (size_t)(ldh > 0 ? ldh : 0) + 1024;
}

static int pull_impl(const float* h_t2s, const float* h_s2t, int64_t ldh, const float* a_t2s, const float* a_s2t,
                     const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                     const int32_t* t_rowptr, const int32_t* t_eid, const int32_t* t_dst,
                     int64_t N, int64_t E, int32_t D, float negative_slope, const float* out, int64_t ldo, const float* alpha,
                     const float* grad_out, int64_t ldg, float* dh_t2s, float* dh_s2t, float* da_t2s, float* da_s2t,
                     int32_t hub_threshold, const PullHubs* hubs, const int32_t* d_vbounds, const int32_t* d_vnode, int64_t d_nv,
                     const int32_t* s_vbounds, const int32_t* s_vnode, int64_t s_nv, void* ws, size_t ws_bytes, void* stream) {
  if (!h_t2s || !h_s2t || !a_t2s || !a_s2t || !rowptr || !col || !mask || !t_rowptr || !t_eid || !t_dst || !out || !alpha ||
      !grad_out || !dh_t2s || !dh_s2t || !da_t2s || !da_s2t || !ws)
    return BGNN_E_NULL;
  const bool narrow = D >= 1 && D <= 4 && ldh == 4 && ldo == 4 && ldg == 4;
  if (N < 0 || E < 0 || D < 1 || D > 128 || ldh < D || ldo < D || ldg < D || (ldh & 3) || (ldo & 3) || (ldg & 3)) return BGNN_E_SHAPE;
  if (hubs && (narrow || hub_threshold < 2 || d_nv < 0 || s_nv < 0)) return BGNN_E_SHAPE;
  if (!bgnn_aligned16(h_t2s) || !bgnn_aligned16(h_s2t) || !bgnn_aligned16(out) || !bgnn_aligned16(grad_out) ||
      !bgnn_aligned16(dh_t2s) || !bgnn_aligned16(dh_s2t) || !bgnn_aligned16(ws))
    return BGNN_E_ALIGN;
  const size_t base_bytes = bgnn_aggregate_bwd_pull_workspace_bytes(N, E, ldh);
  const size_t seg_bytes = hubs ? bgnn_align_up(sizeof(float) * (size_t)ldh * (size_t)(d_nv + 2 * s_nv), 256) + 256 : 0;
  if (ws_bytes < base_bytes + seg_bytes) return BGNN_E_WORKSPACE;
  if (N == 0) return 0;
  uint4* rec = (uint4*)ws;
  float* dstside = (float*)((char*)ws + bgnn_align_up((size_t)32 * (size_t)E, 256));
  unsigned int* queue = (unsigned int*)((char*)dstside + bgnn_align_up(sizeof(float) * (size_t)N * (size_t)ldh, 256));
  PullParams p{h_t2s, h_s2t, ldh, a_t2s, a_s2t, rowptr, col, mask, N, D, negative_slope, out, ldo, alpha, grad_out, ldg,
               t_rowptr, t_eid, t_dst, rec, queue, dstside, dh_t2s, dh_s2t, da_t2s, da_s2t};
  if (hubs) {
    float* seg = (float*)((char*)ws + bgnn_align_up(base_bytes, 256));
    p.hub_threshold = hub_threshold;
    p.d_vnode = d_vnode; p.d_vbounds = d_vbounds; p.d_nv = d_nv; p.d_vpart = seg;
    p.s_vnode = s_vnode; p.s_vbounds = s_vbounds; p.s_nv = s_nv; p.s_vpartS = seg + (size_t)d_nv * ldh; p.s_vpartT = seg + (size_t)(d_nv + s_nv) * ldh;
  }
  hipStream_t st = (hipStream_t)stream;
  if (bgnn_zero_async(queue, 16 * sizeof(unsigned int), st) != hipSuccess) return (int)hipErrorInvalidValue;
  if (narrow) return launch_pull_narrow(p, st);
  const int nv = (D + 3) / 4;
  p.E = E;
  if (!hubs && nv > 16 && bgnn_bwd::pull_fast_plan(p)) return bgnn_bwd::pull_fast_launch(p, st);
  if (nv <= 2) return launch_pull<2>(p, st, hubs);
  if (nv <= 4) return launch_pull<4>(p, st, hubs);
  if (nv <= 8) return launch_pull<8>(p, st, hubs);
  return nv <= 16 ? launch_pull<16>(p, st, hubs) : launch_pull<32>(p, st, hubs);
}

extern "C" int bgnn_adaptedconv_aggregate_bwd_pull_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
                                                       const float* a_t2s, const float* a_s2t,
                                                       const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                                       const int32_t* t_rowptr, const int32_t* t_eid, const int32_t* t_dst,
                                                       int64_t N, int64_t E, int32_t D, float negative_slope,
                                                       const float* out, int64_t ldo, const float* alpha,
                                                       const float* grad_out, int64_t ldg,
                                                       float* dh_t2s, float* dh_s2t, float* da_t2s, float* da_s2t,
                                                       void* ws, size_t ws_bytes, void* stream) {
  return pull_impl(h_t2s, h_s2t, ldh, a_t2s, a_s2t, rowptr, col, mask, t_rowptr, t_eid, t_dst, N, E, D, negative_slope, out, ldo,
                   alpha, grad_out, ldg, dh_t2s, dh_s2t, da_t2s, da_s2t, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0,
                   ws, ws_bytes, stream);
}

extern "C" size_t bgnn_aggregate_bwd_pull_hub_workspace_bytes(int64_t N, int64_t E, int64_t ldh, int64_t d_segments, int64_t s_segments) {
  const size_t nseg = (size_t)(d_segments > 0 ? d_segments : 0) + 2 * (size_t)(s_segments > 0 ? s_segments : 0);
  return bgnn_align_up(bgnn_aggregate_bwd_pull_workspace_bytes(N, E, ldh), 256) +
         bgnn_align_up(sizeof(float) * (size_t)(ldh > 0 ? ldh : 0) * nseg, 256) + 512;
}

extern "C" int bgnn_adaptedconv_aggregate_bwd_pull_hub_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
                                                           const float* a_t2s, const float* a_s2t,
                                                           const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                                           const int32_t* t_rowptr, const int32_t* t_eid, const int32_t* t_dst,
                                                           int64_t N, int64_t E, int32_t D, float negative_slope,
                                                           const float* out, int64_t ldo, const float* alpha,
                                                           const float* grad_out, int64_t ldg,
                                                           float* dh_t2s, float* dh_s2t, float* da_t2s, float* da_s2t,
                                                           int32_t hub_threshold,
                                                           const int32_t* d_hub_rows, int64_t d_n_hubs, const int32_t* d_hub_seg_ptr,
                                                           const int32_t* d_seg_bounds, const int32_t* d_seg_node, int64_t d_n_segments,
                                                           const int32_t* s_hub_rows, int64_t s_n_hubs, const int32_t* s_hub_seg_ptr,
                                                           const int32_t* s_seg_bounds, const int32_t* s_seg_node, int64_t s_n_segments,
                                                           void* ws, size_t ws_bytes, void* stream) {
  if (d_n_hubs < 0 || s_n_hubs < 0 || d_n_segments < d_n_hubs || s_n_segments < s_n_hubs) return BGNN_E_SHAPE;
  if ((d_n_hubs > 0 && (!d_hub_rows || !d_hub_seg_ptr || !d_seg_bounds || !d_seg_node)) ||
      (s_n_hubs > 0 && (!s_hub_rows || !s_hub_seg_ptr || !s_seg_bounds || !s_seg_node)))
    return BGNN_E_NULL;
  PullHubs hubs{d_hub_rows, d_hub_seg_ptr, d_n_hubs, s_hub_rows, s_hub_seg_ptr, s_n_hubs};
  return pull_impl(h_t2s, h_s2t, ldh, a_t2s, a_s2t, rowptr, col, mask, t_rowptr, t_eid, t_dst, N, E, D, negative_slope, out, ldo,
                   alpha, grad_out, ldg, dh_t2s, dh_s2t, da_t2s, da_s2t, hub_threshold, &hubs,
                   d_seg_bounds, d_seg_node, d_n_hubs > 0 ? d_n_segments : 0, s_seg_bounds, s_seg_node, s_n_hubs > 0 ? s_n_segments : 0,
                   ws, ws_bytes, stream);
}

extern "C" size_t bgnn_aggregate_heads_bwd_workspace_bytes(int64_t N, int64_t E, int32_t heads) {
  (void)E;                                       // nothing per edge is kept
  const size_t n = (size_t)(N > 0 ? N : 0), h = (size_t)(heads > 0 ? heads : 0);
  return bgnn_align_up(sizeof(float4) * n * 3 * h, 256) + bgnn_align_up(sizeof(float) * n * h * 4, 256) + 256;
}

static int heads_bwd_impl(const float* h_t2s, const float* h_s2t, const float* a_t2s, const float* a_s2t,
                          const int32_t* rowptr, const int32_t* col, const uint8_t* mask, const int32_t* t_rowptr, const int32_t* t_dst,
                          int64_t N, int64_t E, int32_t D, int32_t heads, float negative_slope, const float* out, const float* state_ms,
                          const float* grad_out, int log_softmax, float* dh_t2s, float* dh_s2t, float* da_t2s, float* da_s2t,
                          int32_t hub_threshold, const PullHubs* hubs, const int32_t* d_vbounds, const int32_t* d_vnode, int64_t d_nv,
                          const int32_t* s_vbounds, const int32_t* s_vnode, int64_t s_nv, void* ws, size_t ws_bytes, void* stream) {
  if (!h_t2s || !h_s2t || !a_t2s || !a_s2t || !rowptr || !col || !mask || !t_rowptr || !t_dst || !out || !state_ms ||
      !grad_out || !dh_t2s || !dh_s2t || !da_t2s || !da_s2t || !ws)
    return BGNN_E_NULL;
  if (N < 0 || E < 0 || D < 1 || D > 4 || (heads != 2 && heads != 3)) return BGNN_E_SHAPE;
  if (hubs && (hub_threshold < 2 || d_nv < 0 || s_nv < 0)) return BGNN_E_SHAPE;
  if (!bgnn_aligned16(h_t2s) || !bgnn_aligned16(h_s2t) || !bgnn_aligned16(out) || !bgnn_aligned16(grad_out) ||
      !bgnn_aligned16(dh_t2s) || !bgnn_aligned16(dh_s2t) || !bgnn_aligned16(ws))
    return BGNN_E_ALIGN;
  const size_t base_bytes = bgnn_aggregate_heads_bwd_workspace_bytes(N, E, heads);
  const size_t seg_bytes = hubs ? bgnn_align_up(sizeof(float) * (size_t)heads * 4 * (size_t)(d_nv + 2 * s_nv), 256) + 256 : 0;
  if (ws_bytes < base_bytes + seg_bytes) return BGNN_E_WORKSPACE;
  if (N == 0) return 0;
  float4* node = (float4*)ws;
  float* dstside = (float*)((char*)ws + bgnn_align_up(sizeof(float4) * (size_t)N * 3 * (size_t)heads, 256));
  HeadsBwdParams p{h_t2s, h_s2t, a_t2s, a_s2t, rowptr, col, mask, N, D, negative_slope, out, state_ms, grad_out, log_softmax,
                   t_rowptr, t_dst, node, dstside, dh_t2s, dh_s2t, da_t2s, da_s2t};
  if (hubs) {
    float* seg = (float*)((char*)ws + bgnn_align_up(base_bytes, 256));
    const size_t ld = (size_t)heads * 4;
    p.hub_threshold = hub_threshold;
    p.d_vnode = d_vnode; p.d_vbounds = d_vbounds; p.d_nv = d_nv; p.d_vpart = seg;
    p.s_vnode = s_vnode; p.s_vbounds = s_vbounds; p.s_nv = s_nv; p.s_vpartS = seg + (size_t)d_nv * ld; p.s_vpartT = seg + (size_t)(d_nv + s_nv) * ld;
  }
  hipStream_t st = (hipStream_t)stream;
  return heads == 3 ? launch_heads_bwd<3>(p, st, hubs) : launch_heads_bwd<2>(p, st, hubs);
}

extern "C" int bgnn_adaptedconv_aggregate_heads_bwd_f32(const float* h_t2s, const float* h_s2t, const float* a_t2s, const float* a_s2t,
                                                        const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                                        const int32_t* t_rowptr, const int32_t* t_eid, const int32_t* t_dst,
                                                        int64_t N, int64_t E, int32_t D, int32_t heads, float negative_slope,
                                                        const float* out, const float* state_ms, const float* grad_out,
                                                        int log_softmax, float* dh_t2s, float* dh_s2t, float* da_t2s,
                                                        float* da_s2t, void* ws, size_t ws_bytes, void* stream) {
  (void)t_eid;                                   // (kept in the signature: the by-source view is passed as one triple everywhere)
  return heads_bwd_impl(h_t2s, h_s2t, a_t2s, a_s2t, rowptr, col, mask, t_rowptr, t_dst, N, E, D, heads, negative_slope, out, state_ms,
                        grad_out, log_softmax, dh_t2s, dh_s2t, da_t2s, da_s2t, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0,
                        ws, ws_bytes, stream);
}

extern "C" size_t bgnn_aggregate_heads_bwd_hub_workspace_bytes(int64_t N, int64_t E, int32_t heads, int64_t d_segments, int64_t s_segments) {
  const size_t nseg = (size_t)(d_segments > 0 ? d_segments : 0) + 2 * (size_t)(s_segments > 0 ? s_segments : 0);
  return bgnn_align_up(bgnn_aggregate_heads_bwd_workspace_bytes(N, E, heads), 256) +
         bgnn_align_up(sizeof(float) * (size_t)(heads > 0 ? heads : 0) * 4 * nseg, 256) + 512;
}

extern "C" int bgnn_adaptedconv_aggregate_heads_bwd_hub_f32(const float* h_t2s, const float* h_s2t, const float* a_t2s, const float* a_s2t,
                                                            const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                                            const int32_t* t_rowptr, const int32_t* t_dst,
                                                            int64_t N, int64_t E, int32_t D, int32_t heads, float negative_slope,
                                                            const float* out, const float* state_ms, const float* grad_out,
                                                            int log_softmax, float* dh_t2s, float* dh_s2t, float* da_t2s, float* da_s2t,
                                                            int32_t hub_threshold,
                                                            const int32_t* d_hub_rows, int64_t d_n_hubs, const int32_t* d_hub_seg_ptr,
                                                            const int32_t* d_seg_bounds, const int32_t* d_seg_node, int64_t d_n_segments,
                                                            const int32_t* s_hub_rows, int64_t s_n_hubs, const int32_t* s_hub_seg_ptr,
                                                            const int32_t* s_seg_bounds, const int32_t* s_seg_node, int64_t s_n_segments,
                                                            void* ws, size_t ws_bytes, void* stream) {
  if (d_n_hubs < 0 || s_n_hubs < 0 || d_n_segments < d_n_hubs || s_n_segments < s_n_hubs) return BGNN_E_SHAPE;
  if ((d_n_hubs > 0 && (!d_hub_rows || !d_hub_seg_ptr || !d_seg_bounds || !d_seg_node)) ||
      (s_n_hubs > 0 && (!s_hub_rows || !s_hub_seg_ptr || !s_seg_bounds || !s_seg_node)))
    return BGNN_E_NULL;
  PullHubs hubs{d_hub_rows, d_hub_seg_ptr, d_n_hubs, s_hub_rows, s_hub_seg_ptr, s_n_hubs};
  return heads_bwd_impl(h_t2s, h_s2t, a_t2s, a_s2t, rowptr, col, mask, t_rowptr, t_dst, N, E, D, heads, negative_slope, out, state_ms,
                        grad_out, log_softmax, dh_t2s, dh_s2t, da_t2s, da_s2t, hub_threshold, &hubs,
                        d_seg_bounds, d_seg_node, d_n_hubs > 0 ? d_n_segments : 0, s_seg_bounds, s_seg_node, s_n_hubs > 0 ? s_n_segments : 0,
                        ws, ws_bytes, stream);
}
