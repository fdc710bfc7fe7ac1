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
  bgnn::XcdRange tr = bgnn::xcd_tile_range(ntiles);
  for (int64_t tile = tr.begin; tile < tr.end; tile += tr.step) {
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
