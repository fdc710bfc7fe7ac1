// AdaptedConv dense part for gfx950: domain means, domain-shift gates and the two Linear
// transforms (reference Bridged-GNN/models/KTGNN.py:275-284; Linear = PyG nn.dense.linear, :240-246).
//
//   delta   = mean_{i in S} x_i - mean_{i in T} x_i                                        (:275)
//   gate_s  = tanh([x_i || delta] . g_s2t) ,  gate_t = tanh([x_i || delta] . g_t2s)         (:277-278)
//   h_s2t_i = W_t (x_i - gate_s delta [i in S]) + b_t                                       (:279,:283)
//   h_t2s_i = W_s (x_i + gate_t delta [i in T]) + b_s                                       (:280,:284)
//
// The shifted inputs differ from x by a per-row scalar times delta, so by linearity
//   h_s2t_i = W_t x_i + b_t - [i in S] gate_s (W_t delta),  h_t2s_i = W_s x_i + b_s + [i in T] gate_t (W_s delta):
// ONE pass over x feeds a GEMM against the packed rows [W_t ; W_s] of up to two convs that share the
// input (clf_base / clf_target), the two gate GEMVs ride in the A-staging loads, and the rank-1 shift
// is applied in the epilogue.  x is read from HBM once, the [N,Din] shifted copies and the gate
// vector the reference materialises never exist.  The contraction is the only MFMA work on path B:
// v_mfma_f32_32x32x2_f32, exact-fp32 fmaf chains.
#include <cstdlib>
#include <type_traits>
#include "bgnn_common.h"
#include "bgnn_transform_params.h"

using bgnn_tf::GemmParams;
using bgnn_tf::MAXH;
using bgnn_tf::f32x16;
using bgnn_tf::tanh_fast;

namespace {

// ------------------------------------------------------------------ per-domain column sums (fp64)
// thread = (column float4 lane, row lane); 4 rows in flight per thread; block partials via LDS, one
// fp64 atomic per (block, column, domain).  Device-scope fp64 atomics on the same 2*Din addresses retire at only
// ~3 G/s (measured: +90 ns per block), so the launch is ONE 1024-thread block per CU rather than many small ones.
constexpr int DS_NT = 1024;
// `part` != nullptr: two-stage form -- block b leaves its [2*Din+2] partial sums in part[b] (plain stores) and
// domain_sums_reduce_kernel adds the blocks up; nullptr: the blocks add into `sums` with fp64 atomics (256 blocks x 258
// same-address atomics are a ~23 us tail, which is why that form wants few, long blocks)
__global__ __launch_bounds__(DS_NT) void domain_sums_kernel(const float* __restrict__ x, int64_t N, int32_t Din,
                                                          int64_t ldx, const uint8_t* __restrict__ mask,
                                                          double* __restrict__ sums, double* __restrict__ part) {
  if (part != nullptr) sums = part + (int64_t)blockIdx.x * (2 * Din + 2);
  const int nc4 = Din >> 2;                       // Din % 4 == 0
  const int cw = nc4 < 256 ? nc4 : 256;           // column lanes
  const int rl = DS_NT / cw;                       // row lanes
  const int tid = threadIdx.x;
  const int cl = tid % cw, r0 = tid / cw;
  const bool active = r0 < rl;
  const int64_t rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int64_t rb = blockIdx.x * rows_per_block;
  const int64_t re = min(rb + rows_per_block, N);
  extern __shared__ double sh[];                  // [rl][2][cw][4]
  for (int cb = 0; cb < nc4; cb += cw) {          // block-uniform trip count (barriers inside)
    const int c4 = cb + cl;
    const bool cok = active && c4 < nc4;
    double as[4] = {0, 0, 0, 0}, at[4] = {0, 0, 0, 0};
    if (cok) {
      int64_t r = rb + r0;
      for (; r + 3 * rl < re; r += 4 * rl) {
        float4 v[4];
        bool m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[u] = *reinterpret_cast<const float4*>(x + (r + u * rl) * ldx + c4 * 4);
          m[u] = mask[r + u * rl] != 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double a = v[u].x, b = v[u].y, c = v[u].z, d = v[u].w;
          if (m[u]) { as[0] += a; as[1] += b; as[2] += c; as[3] += d; }
          else      { at[0] += a; at[1] += b; at[2] += c; at[3] += d; }
        }
      }
      for (; r < re; r += rl) {
        const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c4 * 4);
        if (mask[r]) { as[0] += v.x; as[1] += v.y; as[2] += v.z; as[3] += v.w; }
        else         { at[0] += v.x; at[1] += v.y; at[2] += v.z; at[3] += v.w; }
      }
    }
    if (active) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { sh[((r0 * 2 + 0) * cw + cl) * 4 + e] = as[e]; sh[((r0 * 2 + 1) * cw + cl) * 4 + e] = at[e]; }
    }
    __syncthreads();
    if (cok && r0 == 0) {
      for (int q = 1; q < rl; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) { as[e] += sh[((q * 2 + 0) * cw + cl) * 4 + e]; at[e] += sh[((q * 2 + 1) * cw + cl) * 4 + e]; }
      if (part != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { sums[c4 * 4 + e] = as[e]; sums[Din + c4 * 4 + e] = at[e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { unsafeAtomicAdd(&sums[c4 * 4 + e], as[e]); unsafeAtomicAdd(&sums[Din + c4 * 4 + e], at[e]); }  // hardware fp64 atomic add
      }
    }
    __syncthreads();
  }
  float cs = 0.f, ct = 0.f;                         // node counts (exact in fp32 up to 2^24 per thread)
  for (int64_t r = rb + tid; r < re; r += DS_NT) { if (mask[r]) cs += 1.f; else ct += 1.f; }
  cs = bgnn::group_sum<64>(cs);
  ct = bgnn::group_sum<64>(ct);
  // one pair of atomics per BLOCK: atomics on one address serialise at the memory side (~20 ns each), and
  // 4096 waves x 2 of them used to cost more than streaming x
  if ((tid & 63) == 0) { sh[2 * (tid >> 6)] = (double)cs; sh[2 * (tid >> 6) + 1] = (double)ct; }
  __syncthreads();
  if (tid < 2) {
    double t = 0.0;
    for (int w = 0; w < DS_NT / 64; ++w) t += sh[2 * w + tid];
    if (part != nullptr) sums[2 * Din + tid] = t; else unsafeAtomicAdd(&sums[2 * Din + tid], t);
  }
}

__global__ __launch_bounds__(256) void domain_sums_reduce_kernel(const double* __restrict__ part, int nblocks, int ncol,
                                                                 double* __restrict__ sums) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncol) return;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;      // four independent load chains; fixed order: run-to-run identical
  int b = 0;
  for (; b + 3 < nblocks; b += 4) {
    a0 += part[(int64_t)b * ncol + c]; a1 += part[(int64_t)(b + 1) * ncol + c];
    a2 += part[(int64_t)(b + 2) * ncol + c]; a3 += part[(int64_t)(b + 3) * ncol + c];
  }
  for (; b < nblocks; ++b) a0 += part[(int64_t)b * ncol + c];
  sums[c] += (a0 + a1) + (a2 + a3);
}

__global__ void domain_delta_kernel(const double* __restrict__ sums, int32_t Din, float* __restrict__ delta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Din) return;
  const double ns = sums[2 * Din], nt = sums[2 * Din + 1];
  // mean over an empty domain is NaN in the reference (x[mask].mean(0) of 0 rows); keep that visible
  const float ms = (float)(sums[c] / ns), mt = (float)(sums[Din + c] / nt);
  delta[c] = ms - mt;
}

// ------------------------------------------------------------------ W.delta and gate constants
// wd[j] = Wp[j] . delta (j < NC) ; gc[h*2+t] = g[h][t][Din:] . delta
// delta comes either as a vector or, with `sums` (the [2*Din+2] domain sums), is formed on the fly with the arithmetic
// of domain_delta_kernel (one launch less per conv; bit-identical)
__global__ __launch_bounds__(256) void wd_kernel(const float* __restrict__ Wp, int32_t NC, int32_t Din,
                                                 const float* __restrict__ delta, const double* __restrict__ sums,
                                                 const float* __restrict__ g, const float* __restrict__ gate_const,
                                                 int32_t n_heads, float* __restrict__ wd, float* __restrict__ gc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nrows = NC + 2 * n_heads;
  double ns = 1.0, nt = 1.0;
  if (sums) { ns = sums[2 * Din]; nt = sums[2 * Din + 1]; }
  for (int j = blockIdx.x * 4 + wave; j < nrows; j += gridDim.x * 4) {
    const float* row = j < NC ? Wp + (int64_t)j * Din : g + (int64_t)(j - NC) * 2 * Din + Din;
    float acc = 0.f;
    for (int c = lane; c < Din; c += 64) {
      const float d = sums ? (float)(sums[c] / ns) - (float)(sums[Din + c] / nt) : delta[c];
      acc = fmaf(row[c], d, acc);
    }
    acc = bgnn::group_sum<64>(acc);
    if (lane == 0) { if (j < NC) wd[j] = acc; else gc[j - NC] = acc + (gate_const ? gate_const[j - NC] : 0.f); }
  }
}

// ------------------------------------------------------------------ fused MFMA fp32 GEMM
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BK = 32, LDS_LD = BK + 4;   // +4 floats: conflict-free ds_read_b128 (16-lane groups)

// BN output columns per block; waves arranged WM x WN, each computing TM x TN tiles of 32x32
template <int BN, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void transform_gemm_kernel(GemmParams p) {
  static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == BN, "tile layout");
  // one LDS arena: [As0 | As1 | Bs0 | Bs1] during the K loop, re-used as the C staging tile afterwards
  constexpr int A_SZ = BM * LDS_LD, B_SZ = BN * LDS_LD, C_LD = BN + 4;
  static_assert(BM * C_LD <= 2 * A_SZ + 2 * B_SZ, "C staging tile must fit in the A/B buffers");
  __shared__ __attribute__((aligned(16))) float arena[2 * A_SZ + 2 * B_SZ];
  __shared__ float coef[BM][MAXH][2];
  float* const As0 = arena;
  float* const Bs0 = arena + 2 * A_SZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch) -> give them the SAME
  // row tile and adjacent column tiles so the second read of the x tile hits that XCD's L2.
  const int nct = (p.NC + BN - 1) / BN;
  const int64_t b = blockIdx.x;
  const int64_t grp = b / (8 * nct), within = b % (8 * nct);
  const int64_t rt = grp * 8 + (within & 7);
  const int ct = (int)(within >> 3);
  const int64_t row0 = rt * BM;
  const int col0 = ct * BN;
  if (row0 >= p.N) return;                       // block-uniform

  // staging assignment: 8 threads cover one 32-float row segment, 32 rows per pass
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  constexpr int APASS = BM / 32, BPASS = BN / 32;
  float4 ra[APASS], rb[BPASS];
  float gd[APASS][MAXH][2];
#pragma unroll
  for (int j = 0; j < APASS; ++j)
#pragma unroll
    for (int h = 0; h < MAXH; ++h) gd[j][h][0] = gd[j][h][1] = 0.f;

  auto gload = [&](int k0) {
    const int k = k0 + sc;
    const bool kok = k < p.Din;   // Din % 4 == 0 -> a float4 is entirely in or out
    float4 g4[MAXH][2];
#pragma unroll
    for (int h = 0; h < MAXH; ++h)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        g4[h][t] = (kok && h < p.n_heads) ? *reinterpret_cast<const float4*>(p.g + ((int64_t)(h * 2 + t) * 2) * p.Din + k)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const int64_t r = row0 + sr + 32 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kok && r < p.N) v = *reinterpret_cast<const float4*>(p.x + r * p.ldx + k);
      ra[j] = v;
#pragma unroll
      for (int h = 0; h < MAXH; ++h)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float a = gd[j][h][t];
          a = fmaf(v.x, g4[h][t].x, a); a = fmaf(v.y, g4[h][t].y, a);
          a = fmaf(v.z, g4[h][t].z, a); a = fmaf(v.w, g4[h][t].w, a);
          gd[j][h][t] = a;
        }
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const int n = col0 + sr + 32 * j;
      rb[j] = (kok && n < p.NC) ? *reinterpret_cast<const float4*>(p.Wp + (int64_t)n * p.Din + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < APASS; ++j) *reinterpret_cast<float4*>(&As0[buf * A_SZ + (sr + 32 * j) * LDS_LD + sc]) = ra[j];
#pragma unroll
    for (int j = 0; j < BPASS; ++j) *reinterpret_cast<float4*>(&Bs0[buf * B_SZ + (sr + 32 * j) * LDS_LD + sc]) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int c = 0; c < TN; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  const int nk = (p.Din + BK - 1) / BK;
  gload(0);
  sstore(0);
  if (nk > 1) gload(BK);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) sstore(cur ^ 1);            // tile kt+1 (in registers since the previous compute phase)
    if (kt + 2 < nk) gload((kt + 2) * BK);       // tile kt+2 flies during this compute phase
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        af[a] = *reinterpret_cast<const float4*>(&As0[cur * A_SZ + ((wm * TM + a) * 32 + fr) * LDS_LD + kb * 8 + fh * 4]);
#pragma unroll
      for (int c = 0; c < TN; ++c)
        bf[c] = *reinterpret_cast<const float4*>(&Bs0[cur * B_SZ + ((wn * TN + c) * 32 + fr) * LDS_LD + kb * 8 + fh * 4]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int c = 0; c < TN; ++c) {
          // lanes 0-31 carry k = 8kb+s, lanes 32-63 carry k = 8kb+4+s (same permutation for A and B)
          acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[c].x, acc[a][c], 0, 0, 0);
          acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[c].y, acc[a][c], 0, 0, 0);
          acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[c].z, acc[a][c], 0, 0, 0);
          acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[c].w, acc[a][c], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // gates -> per-row rank-1 coefficients (KTGNN.py:277-280): coef[row][h] = (-gate_s [i in S], +gate_t [i in T])
#pragma unroll
  for (int j = 0; j < APASS; ++j) {
    const int64_t r = row0 + sr + 32 * j;
    const bool s = r < p.N ? (p.mask[r] != 0) : false;
#pragma unroll
    for (int h = 0; h < MAXH; ++h) {
      const float ds = bgnn::group_sum<8>(gd[j][h][0]);
      const float dt = bgnn::group_sum<8>(gd[j][h][1]);
      if ((tid & 7) == 0 && h < p.n_heads) {
        coef[sr + 32 * j][h][0] = s ? -tanhf(ds + p.gc[h * 2 + 0]) : 0.f;
        coef[sr + 32 * j][h][1] = s ? 0.f : tanhf(dt + p.gc[h * 2 + 1]);
      }
    }
  }
  // epilogue, stage 1: accumulators -> LDS tile (the K loop ended with a barrier, the A/B buffers are free).
  // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float* const Cs = arena;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int c = 0; c < TN; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        Cs[lr * C_LD + (wn * TN + c) * 32 + fr] = acc[a][c][r];
      }
  __syncthreads();
  // stage 2: whole rows leave as 16-byte stores (a dword-per-lane epilogue is store-issue bound),
  // with bias + rank-1 shift applied on the way out
  constexpr int F4R = BN / 4;                   // float4 per tile row
  constexpr int RPP = 256 / F4R;                // rows per pass
  const int c4 = tid % F4R, rr = tid / F4R;
  const int col = col0 + c4 * 4;
  if (col < p.NC) {                             // NC % 4 == 0: a float4 is entirely in or out
    const int ld2 = 2 * (int)p.ldh;
    const int h = col / ld2, rem = col % ld2;
    const int t = rem >= p.ldh ? 1 : 0;
    const int cc = rem - t * (int)p.ldh;
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + col);
    const float4 wv = *reinterpret_cast<const float4*>(p.wd + col);
    float* __restrict__ out = p.out[h][t];
#pragma unroll 4
    for (int lr = rr; lr < BM; lr += RPP) {
      const int64_t row = row0 + lr;
      if (row < p.N) {
        const float4 v = *reinterpret_cast<const float4*>(&Cs[lr * C_LD + c4 * 4]);
        const float cf = coef[lr][h][t];
        float4 o;
        o.x = fmaf(cf, wv.x, v.x + bv.x); o.y = fmaf(cf, wv.y, v.y + bv.y);
        o.z = fmaf(cf, wv.z, v.z + bv.z); o.w = fmaf(cf, wv.w, v.w + bv.w);
        *reinterpret_cast<float4*>(out + row * p.row_stride + cc) = o;
      }
    }
  }
}


// ------------------------------------------------------------------ W-stationary variant (NC > 32, Din <= 128)
// transform_gemm_kernel re-stages the [BN, 32] slab of W into LDS for every 128-row tile and pays a block-wide
// prologue / epilogue per tile; measured 0.87 ms on the [1M,128]x[128,256] hidden transform, of which 0.42 ms is fp32
// MFMA issue.  Here the weights never move: wave w keeps "its" 32 output columns of W in registers for the whole
// kernel (the MFMA A operand: Din/2 VGPRs), the block is persistent (one per CU) and walks 32-row tiles of x that are
// double-buffered in LDS (the B operand, shared by the 8 waves) -- one barrier per tile, the next tile's global loads
// fly during the MFMA phase, and with x as the B operand a lane ends up holding 4 CONSECUTIVE output columns of one row,
// so results leave as 16-byte stores straight from the accumulators (no C staging pass).
//   DK  : Din rounded up to 64/128 (zero-filled),  NCT : 32-column tiles per block (2/4/8); 8/NCT row sub-tiles.


template <int DK, int NCT, int NW, bool BF3, int MODE>
__global__ __launch_bounds__(64 * NW) void transform_wreg_kernel(GemmParams p) {
  constexpr int RS = NW / NCT;               // row sub-tiles (LDS: 2 x 32*RS x (DK+4) floats)
  constexpr int RPP = 4 * NW;                // staging: rows per pass (16 lanes per row)
  constexpr int BMW = 32 * RS, LD = DK + 4;
  constexpr int KB = DK / 8;                 // 16-byte k chunks per lane half
  constexpr int CPT = DK / 64;               // staging: 16 lanes cover a row, CPT float4 each (256-B runs)
  constexpr int NP = BMW / RPP;              // staging passes
  static_assert(NW % NCT == 0 && BMW % RPP == 0, "wave layout");
  constexpr int PRE_LD = MAXH * 2 + 1;       // gate pre-activations of a row + its domain flag
  // BF3: every fp32 value travels as three bf16 pieces (hi + mid + lo = the 24-bit significand) and a product is the six
  // bf16 MFMAs whose pieces are >= 2^-24 relative; fp32 MFMA shares the VALU datapath on CDNA (same 157 TFLOP/s peak,
  // VALU work ADDS to it -- measured), the bf16 matrix cores are 16x faster and run beside the VALU.
  constexpr int LDB = DK + 8;                // bf16 elements per piece row (272-B rows: conflict-free b128 reads)
  constexpr int XS_FLOATS = BF3 ? (3 * BMW * LDB) / 2 : BMW * LD;
  __shared__ __attribute__((aligned(16))) float xs[2][XS_FLOATS];
  __shared__ float pre[2][BMW][PRE_LD];
  constexpr int CT_LD = 36;                  // wave-private 32x32 output tile (+4 floats: conflict-free b128 writes)
  __shared__ __attribute__((aligned(16))) float ctile[NW][32 * CT_LD];
  // MODE 2: per-tile reduction of the second stage's 10 outputs per row over the NCT column waves
  constexpr int R2_LD = 12;
  __shared__ __attribute__((aligned(16))) float red2[MODE == 2 ? NCT : 1][MODE == 2 ? BMW * R2_LD : 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar branches, clean waitcnt placement
  const int fr = lane & 31, fh = lane >> 5;
  const int ct = wave % NCT, rs = wave / NCT;
  const int col_base = p.col_off + blockIdx.y * (32 * NCT) + ct * 32;
  // MODE 2 second stage: out2[o][row] = sum_c w2[o][c] * a[row][c] over this wave's 32 columns as 16 fp32 MFMAs whose B
  // operands are the activation registers as they are (register 4q+j of lane (fr, fh) = column 8q+4fh+j of row fr, i.e.
  // k-pair {8q+j, 8q+4+j}); the stationary A operand pairs the same way: w2r[4q+j] = w2[o = fr][col_base + 8q+4fh+j]
  // (BF3: the second stage runs on the bf16 matrix cores too -- the 16 activation registers of a lane ARE two k-blocks of a
  //  32x32x16 B operand once split into bf16 pieces (k-slot = register index, the same permutation on the stationary side);
  //  six products per k-block: 12 x 32 cycles instead of 16 fp32 MFMAs x 64 cycles on the vector ALU)
  float w2r[MODE == 2 && !BF3 ? 16 : 1];
  bf16x8 w2p[MODE == 2 && BF3 ? 3 : 1][2];   // [piece][k-block]
  if constexpr (MODE == 2) {                 // host: gridDim.y == 1, NC == 32 * NCT
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = col_base + 8 * q + 4 * fh + j;
        float v = 0.f;
        if (fr < 8) v = p.w2[(int64_t)fr * p.NC + c];
        else if (fr < 10) v = p.g2[(int64_t)(fr - 8) * 2 * p.NC + c];
        if constexpr (BF3) {
          const __bf16 h = (__bf16)v;
          const float r1 = v - (float)h;
          const __bf16 m = (__bf16)r1;
          const int r = 4 * q + j;
          w2p[0][r >> 3][r & 7] = h; w2p[1][r >> 3][r & 7] = m; w2p[2][r >> 3][r & 7] = (__bf16)(r1 - (float)m);
        } else {
          w2r[4 * q + j] = v;
        }
      }
    for (int t = tid; t < NCT * BMW * R2_LD; t += 64 * NW) (&red2[0][0])[t] = 0.f;   // (columns 10, 11 stay zero)
    // (the first __syncthreads of the tile loop orders these writes before their first use)
  }
  // (host guarantees (NC - col_off) % (32 * NCT) == 0: every wave owns a full column tile -- no per-wave branches in the tile loop,
  //  which keeps the compiler's s_waitcnt placement exact)

  // ---- stationary operands -------------------------------------------------------------------------------
  constexpr int KB16 = DK / 16;               // BF3: 16-wide k blocks; lane half fh holds k = 16kb + 8fh .. +7
  float wreg[BF3 ? 1 : KB * 4];               // fp32: W[col_base + fr][8kb + 4fh + j]  (the same k permutation as the x reads)
  bf16x8 wpc[BF3 ? 3 : 1][BF3 ? KB16 : 1];    // BF3: [piece][kb]
  {
    const int n = col_base + fr;
    if constexpr (!BF3) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int k = 8 * kb + 4 * fh;
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < p.NC && k < p.Din) w = *reinterpret_cast<const float4*>(p.Wp + (int64_t)n * p.Din + k);
        wreg[4 * kb] = w.x; wreg[4 * kb + 1] = w.y; wreg[4 * kb + 2] = w.z; wreg[4 * kb + 3] = w.w;
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < KB16; ++kb) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int k = 16 * kb + 8 * fh + 4 * hf;
          float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
          if (n < p.NC && k < p.Din) w = *reinterpret_cast<const float4*>(p.Wp + (int64_t)n * p.Din + k);
          const float wf[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const __bf16 h = (__bf16)wf[e];
            const float r1 = wf[e] - (float)h;
            const __bf16 m = (__bf16)r1;
            const __bf16 l = (__bf16)(r1 - (float)m);
            wpc[0][kb][4 * hf + e] = h; wpc[1][kb][4 * hf + e] = m; wpc[2][kb][4 * hf + e] = l;
          }
        }
      }
    }
  }
  // epilogue constants: accumulator registers 4q..4q+3 of a lane are columns col_base + 8q + 4fh + (0..3) of row fr
  float4 bv[4], wv[4];
  int hsel[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = col_base + 8 * q + 4 * fh;
    bv[q] = wv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    hsel[q] = 0;
    if (c < p.NC) {
      const int ld2 = 2 * (int)p.ldh;
      const int h = c / ld2, rem = c % ld2, t = rem >= p.ldh ? 1 : 0;
      bv[q] = *reinterpret_cast<const float4*>(p.bias + c);
      if constexpr (MODE == 0) wv[q] = *reinterpret_cast<const float4*>(p.wd + c);
      hsel[q] = h * 2 + t;
    }
  }
  // store side (after the transpose): lane owns columns col_base + 4*(lane & 7) .. +3 of rows (lane >> 3) + 8i
  float* ocol = nullptr;
  {
    const int c = col_base + 4 * (lane & 7);
    if (c < p.NC) {
      const int ld2 = 2 * (int)p.ldh;
      const int h = c / ld2, rem = c % ld2, t = rem >= p.ldh ? 1 : 0;
      float* base = h == 0 ? (t == 0 ? p.out[0][0] : p.out[0][1]) : (t == 0 ? p.out[1][0] : p.out[1][1]);
      ocol = base + (rem - t * (int)p.ldh);
    }
  }
  // staging: 16 consecutive lanes own one row; lane l16 holds float4 chunks l16 + 16c (c < CPT) -> its gate slices are fixed
  const int l16 = tid & 15, srow = tid >> 4;
  float4 g4[MAXH][2][CPT];
#pragma unroll
  for (int h = 0; h < MAXH; ++h)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int k = (l16 + 16 * c) * 4;
        g4[h][t][c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (MODE == 0)
          if (k < p.Din && h < p.n_heads) g4[h][t][c] = *reinterpret_cast<const float4*>(p.g + ((int64_t)(h * 2 + t) * 2) * p.Din + k);
      }
  float gcs[MAXH * 2];
#pragma unroll
  for (int h = 0; h < MAXH * 2; ++h) gcs[h] = (MODE == 0 && h < 2 * p.n_heads) ? p.gc[h] : 0.f;

  float4 ra[NP][CPT];
  uint8_t rm[NP];
  // branch-free loads (tail rows re-read row N-1, chunks past Din re-read chunk 0 and are zeroed): with every load
  // and its count unconditional the compiler can place exact s_waitcnt vmcnt(n) instead of draining the queue
  auto gload = [&](int64_t tl) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      int64_t r = tl * BMW + srow + RPP * j;
      r = r < p.N ? r : p.N - 1;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int k = (l16 + 16 * c) * 4;
        const bool kv = k < p.Din;
        float4 v = *reinterpret_cast<const float4*>(p.x + r * p.ldx + (kv ? k : 0));
        if (!kv) v = make_float4(0.f, 0.f, 0.f, 0.f);
        ra[j][c] = v;
      }
      rm[j] = (MODE == 0 || p.mask != nullptr) ? p.mask[r] : (uint8_t)0;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int lr = srow + RPP * j;
      float d[MAXH * 2];
#pragma unroll
      for (int h = 0; h < MAXH * 2; ++h) d[h] = 0.f;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const float4 v = ra[j][c];
        if constexpr (!BF3) {
          *reinterpret_cast<float4*>(&xs[buf][lr * LD + (l16 + 16 * c) * 4]) = v;
        } else {
          // split 4 floats into (hi, mid, lo) bf16 quadruples: v_cvt_pk_bf16_f32 + exact fp32 residuals
          const float vf[4] = {v.x, v.y, v.z, v.w};
          bf16x4 ph, pm, pl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const __bf16 h = (__bf16)vf[e];
            const float r1 = vf[e] - (float)h;
            const __bf16 m = (__bf16)r1;
            ph[e] = h; pm[e] = m; pl[e] = (__bf16)(r1 - (float)m);
          }
          __bf16* xb16 = reinterpret_cast<__bf16*>(xs[buf]);
          const int k = (l16 + 16 * c) * 4;
          *reinterpret_cast<bf16x4*>(&xb16[(0 * BMW + lr) * LDB + k]) = ph;
          *reinterpret_cast<bf16x4*>(&xb16[(1 * BMW + lr) * LDB + k]) = pm;
          *reinterpret_cast<bf16x4*>(&xb16[(2 * BMW + lr) * LDB + k]) = pl;
        }
        if constexpr (MODE == 0) {
#pragma unroll
          for (int h = 0; h < MAXH; ++h)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              float a = d[h * 2 + t];
              a = fmaf(v.x, g4[h][t][c].x, a); a = fmaf(v.y, g4[h][t][c].y, a);
              a = fmaf(v.z, g4[h][t][c].z, a); a = fmaf(v.w, g4[h][t][c].w, a);
              d[h * 2 + t] = a;
            }
        }
      }
      if constexpr (MODE == 0) {
#pragma unroll
        for (int h = 0; h < MAXH * 2; ++h) d[h] = bgnn::group_sum<16>(d[h]) + gcs[h];
      }
      // lanes 0..4 of the 16 publish the row's four pre-activations and its domain flag (tanh is taken by the consumer)
      float val = rm[j] != 0 ? 1.f : 0.f;
#pragma unroll
      for (int h = 0; h < MAXH * 2; ++h) val = l16 == h ? d[h] : val;
      if (l16 < PRE_LD) pre[buf][lr][l16] = val;
    }
  };

  float4 cs_s = make_float4(0.f, 0.f, 0.f, 0.f), cs_t = cs_s;   // MODE 1: column sums of the rows this lane stored
  float n_s = 0.f, n_t = 0.f;
  // MODE 2: a finished tile's 12 floats per row leave through red2 (summed over the NCT column waves there)
  auto flush2 = [&](int64_t tl) {
    if constexpr (MODE == 2) {
      if (tid < BMW * 3) {
        const int row = tid / 3, ch = tid % 3;
        float4 v = *reinterpret_cast<const float4*>(&red2[0][row * R2_LD + 4 * ch]);
#pragma unroll
        for (int w = 1; w < NCT; ++w) {
          const float4 u = *reinterpret_cast<const float4*>(&red2[w][row * R2_LD + 4 * ch]);
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const int64_t r = tl * BMW + row;
        if (r < p.N) *reinterpret_cast<float4*>(p.raw + r * 12 + 4 * ch) = v;
      }
    }
  };
  const int64_t ntiles = (p.N + BMW - 1) / BMW, last = ntiles - 1;     // gridDim.x <= ntiles (host)
  int64_t tile = blockIdx.x;
  gload(tile);
  sstore(0);
  gload(min(tile + (int64_t)gridDim.x, last));
  __builtin_amdgcn_s_waitcnt(0x0F70);         // enter the loop with no load pending (see the comment before the epilogue)
  int it = 0;
  for (; tile < ntiles; ++it, tile += gridDim.x) {   // block-uniform trip count
    const int cur = it & 1;
    __syncthreads();                          // buffer `cur` is complete; nobody still reads buffer cur^1
    // tile it+1 goes registers -> LDS, tile it+2 starts flying; past the end the last tile is staged again (never read)
    // (staggering the staging of the two waves that share a SIMD, or two 4-wave blocks per CU, measured no gain)
    sstore(cur ^ 1);                                          // tile it+1: registers -> LDS
    gload(min(tile + 2 * (int64_t)gridDim.x, last));          // tile it+2 flies during the MFMA phase
    __builtin_amdgcn_sched_barrier(0);        // keep the loads ABOVE the MFMA chain (the scheduler sinks them to their use)
    if constexpr (MODE == 0) {
      // single-table tail rows (GemmParams): a whole tile inside one tail group is skipped by the other table's waves
      const int64_t tb = tile * BMW + rs * 32, te = tb + 32;
      const int my_table = (col_base / (int)p.ldh) & 1;
      const bool tails = p.tail_s2t_begin > 0;                  // (zero-initialised params: no tail)
      const bool skip = tails && ((my_table == 0 && tb >= p.tail_t2s_begin && te <= p.tail_s2t_begin) || (my_table == 1 && tb >= p.tail_s2t_begin));
      if (skip) {                               // wave-uniform; the staging above and the barrier below are still shared
        __builtin_amdgcn_s_waitcnt(0x0F70);
        continue;
      }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
    if constexpr (!BF3) {
    const float* xb = &xs[cur][(rs * 32 + fr) * LD + 4 * fh];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const float4 b = *reinterpret_cast<const float4*>(xb + 8 * kb);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * kb], b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * kb + 1], b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * kb + 2], b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * kb + 3], b.w, acc, 0, 0, 0);
    }
    } else {
    const __bf16* xb16 = reinterpret_cast<const __bf16*>(xs[cur]) + (rs * 32 + fr) * LDB + 8 * fh;
#pragma unroll
    for (int kb = 0; kb < KB16; ++kb) {
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(xb16 + 0 * BMW * LDB + 16 * kb);
      const bf16x8 bm = *reinterpret_cast<const bf16x8*>(xb16 + 1 * BMW * LDB + 16 * kb);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(xb16 + 2 * BMW * LDB + 16 * kb);
      // smallest terms first; dropped: mid*lo, lo*mid, lo*lo (< 2^-24 relative)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wpc[2][kb], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wpc[0][kb], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wpc[1][kb], bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wpc[1][kb], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wpc[0][kb], bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wpc[0][kb], bh, acc, 0, 0, 0);
    }
    // (two independent accumulator chains -- a lone dependent bf16 chain issues at half rate -- measured no gain here:
    //  the two waves of a SIMD already fill the matrix pipe)
    }
    }
    // The loads issued above landed during the MFMA phase; retiring them HERE (vmcnt(0), free) lets the next
    // iteration's staging start without waiting for the stores below (vmcnt counts loads and stores in order).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    {
    // epilogue: bias + rank-1 shift in the accumulator layout (lane = row, 4 consecutive columns per q) ...
    const float* pr = pre[cur][rs * 32 + fr];
    const bool sdom = pr[MAXH * 2] != 0.f;
    float* cw = ctile[wave];
    float cf = 0.f;
    float a2[MODE == 2 ? 16 : 1];            // the activation values of this lane (row fr, 16 columns)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // rank-1 coefficient (KTGNN.py:277-280): -gate_s2t on source rows (table 0), +gate_t2s on target rows (table 1)
      if constexpr (MODE == 0) {
        if (q == 0 || hsel[q] != hsel[q - 1]) {
          const bool t1 = hsel[q] & 1;
          cf = (sdom != t1) ? tanh_fast(pr[hsel[q]]) : 0.f;
          cf = t1 ? cf : -cf;
        }
      }
      float4 o;
      o.x = fmaf(cf, wv[q].x, acc[4 * q] + bv[q].x);     o.y = fmaf(cf, wv[q].y, acc[4 * q + 1] + bv[q].y);
      o.z = fmaf(cf, wv[q].z, acc[4 * q + 2] + bv[q].z); o.w = fmaf(cf, wv[q].w, acc[4 * q + 3] + bv[q].w);
      if constexpr (MODE >= 1) {
        if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      }
      if constexpr (MODE == 2) { a2[4 * q] = o.x; a2[4 * q + 1] = o.y; a2[4 * q + 2] = o.z; a2[4 * q + 3] = o.w; }
      *reinterpret_cast<float4*>(&cw[fr * CT_LD + 8 * q + 4 * fh]) = o;     // MODE 2: only the column sums read it back
    }
    if constexpr (MODE == 2) {
      f32x16 acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
      if constexpr (BF3) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          bf16x8 ah, am, al;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = a2[8 * kb + e];
            const __bf16 h = (__bf16)v;
            const float r1 = v - (float)h;
            const __bf16 m = (__bf16)r1;
            ah[e] = h; am[e] = m; al[e] = (__bf16)(r1 - (float)m);
          }
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[2][kb], ah, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[0][kb], al, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[1][kb], am, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[1][kb], ah, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[0][kb], am, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[0][kb], ah, acc2, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2r[r], a2[r], acc2, 0, 0, 0);
      }
      // accumulator register i of lane (fr, fh) is output 8*(i/4) + 4*fh + i%4 of row fr: lane half 0 holds outputs
      // 0..3 and 8..11 (8, 9 are the gate products), lane half 1 outputs 4..7
      // every column wave leaves its partial sums in its own slot (plain stores: ds_add_f32 from the NCT waves into one
      // slot cost 0.16 ms per 1M rows); the flush adds the slots
      float* r2 = &red2[ct][(rs * 32 + fr) * R2_LD];
      if (fh == 0) {
        *reinterpret_cast<float4*>(r2) = make_float4(acc2[0], acc2[1], acc2[2], acc2[3]);
        *reinterpret_cast<float2*>(r2 + 8) = make_float2(acc2[4], acc2[5]);
      } else {
        *reinterpret_cast<float4*>(r2 + 4) = make_float4(acc2[0], acc2[1], acc2[2], acc2[3]);
      }
    }
    // ... then a wave-private LDS transpose so every store instruction writes 8 rows x 128 contiguous bytes (whole
    // cache lines); lane-per-row 16-byte stores scatter one instruction over 32 lines and cost 0.14 ms here
    if (MODE == 2 || ocol != nullptr) {
      const int64_t row0 = tile * BMW + rs * 32 + (lane >> 3);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(&cw[((lane >> 3) + 8 * i) * CT_LD + (lane & 7) * 4]);
        const int64_t row = row0 + 8 * i;
        if (row < p.N) {
          if constexpr (MODE != 2) *reinterpret_cast<float4*>(ocol + row * p.row_stride) = v;
          if constexpr (MODE >= 1) {
            if (p.colsum != nullptr) {            // per-lane fp32 partials (4 fixed columns, ~500 rows per lane and launch)
              if (pre[cur][rs * 32 + (lane >> 3) + 8 * i][MAXH * 2] != 0.f) { cs_s.x += v.x; cs_s.y += v.y; cs_s.z += v.z; cs_s.w += v.w; n_s += 1.f; }
              else { cs_t.x += v.x; cs_t.y += v.y; cs_t.z += v.z; cs_t.w += v.w; n_t += 1.f; }
            }
          }
        }
      }
    }
    }
    if constexpr (MODE == 2) {
      // the tile's rows leave through red2 in the same iteration (a second barrier per tile; deferring the flush to the
      // next iteration's top put a branch with stores in front of the staging and the compiler drained vmcnt there: 2x slower)
      __syncthreads();
      flush2(tile);
    }
  }
  if constexpr (MODE >= 1) {
    if (p.colsum != nullptr) {
      // lanes -> block (ds_add_f32) -> one hardware fp64 atomic per (block, column, domain)
      __shared__ float red[2][32 * NCT + 1];
      for (int t = tid; t < 2 * (32 * NCT + 1); t += 64 * NW) (&red[0][0])[t] = 0.f;
      __syncthreads();
      const int lc = ct * 32 + 4 * (lane & 7);
      unsafeAtomicAdd(&red[0][lc], cs_s.x); unsafeAtomicAdd(&red[0][lc + 1], cs_s.y);
      unsafeAtomicAdd(&red[0][lc + 2], cs_s.z); unsafeAtomicAdd(&red[0][lc + 3], cs_s.w);
      unsafeAtomicAdd(&red[1][lc], cs_t.x); unsafeAtomicAdd(&red[1][lc + 1], cs_t.y);
      unsafeAtomicAdd(&red[1][lc + 2], cs_t.z); unsafeAtomicAdd(&red[1][lc + 3], cs_t.w);
      if (ct == 0 && (lane & 7) == 0 && blockIdx.y == 0) { unsafeAtomicAdd(&red[0][32 * NCT], n_s); unsafeAtomicAdd(&red[1][32 * NCT], n_t); }
      __syncthreads();
      for (int t = tid; t < 2 * (32 * NCT + 1); t += 64 * NW) {
        const int d = t / (32 * NCT + 1), c = t % (32 * NCT + 1);
        const double sum = (double)red[d][c];
        const int gc0 = blockIdx.y * (32 * NCT) + c;
        if (c == 32 * NCT) { if (blockIdx.y == 0) unsafeAtomicAdd(&p.colsum[2 * p.NC + d], sum); }
        else if (gc0 < p.NC) unsafeAtomicAdd(&p.colsum[d * p.NC + gc0], sum);
      }
    }
  }
}


// ------------------------------------------------------------------ skinny variant (NC <= 24 packed columns, Din <= 128)
// KT-GNN's classifier convs have D = n_classes (2..10): the transform is a pure stream over x (HBM-bound, 0.09 ms for
// 512 MB) and the 128-row block tiles of transform_gemm_kernel<32,...> run it at 2.7 TB/s.  Here every WAVE is
// autonomous -- its own 32-row tile in a private LDS region, no block barrier, the next tile's loads in flight in
// registers during the MFMA phase -- so the 8 waves of a CU drift apart and keep HBM busy.  The 32-column MFMA tile
// has room for the gate vectors as well: rows 24..27 (and their copy 28..31, so both lane halves see them) of the
// stationary operand hold g[h][t][:Din], which turns the four gate GEMVs into accumulator registers 12..15.
template <int DK>
__global__ __launch_bounds__(256) void transform_skinny_kernel(GemmParams p) {
  constexpr int LD = DK + 4, KB = DK / 8, F4_ROW = DK / 4, NLD = 32 * F4_ROW / 64, RPL = 64 / F4_ROW;  // RPL rows per load instr
  __shared__ __attribute__((aligned(16))) float xs[4][32 * LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  float* const xw = xs[wave];

  float wreg[KB * 4];
  {
    const float* src = nullptr;               // row fr of the stationary operand
    if (fr < p.NC) src = p.Wp + (int64_t)fr * p.Din;
    else if (fr >= 24 && ((fr - 24) & 3) < 2 * p.n_heads) src = p.g + (int64_t)((fr - 24) & 3) * 2 * p.Din;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int k = 8 * kb + 4 * fh;
      float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
      if (src != nullptr && k < p.Din) w = *reinterpret_cast<const float4*>(src + k);
      wreg[4 * kb] = w.x; wreg[4 * kb + 1] = w.y; wreg[4 * kb + 2] = w.z; wreg[4 * kb + 3] = w.w;
    }
  }
  // epilogue constants: accumulator registers 4q..4q+3 of a lane are columns 8q + 4fh + (0..3) of row fr (q < 3)
  float4 bv[3], wv[3];
  float* optr[3];
  int hsel[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int c = 8 * q + 4 * fh;
    bv[q] = wv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    optr[q] = nullptr; hsel[q] = 0;
    if (c < p.NC) {
      const int ld2 = 2 * (int)p.ldh;
      const int h = c / ld2, rem = c % ld2, t = rem >= p.ldh ? 1 : 0;
      bv[q] = *reinterpret_cast<const float4*>(p.bias + c);
      wv[q] = *reinterpret_cast<const float4*>(p.wd + c);
      float* base = h == 0 ? (t == 0 ? p.out[0][0] : p.out[0][1]) : (t == 0 ? p.out[1][0] : p.out[1][1]);
      optr[q] = base + (rem - t * (int)p.ldh);
      hsel[q] = h * 2 + t;
    }
  }
  float gcs[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) gcs[h] = h < 2 * p.n_heads ? p.gc[h] : 0.f;

  const int c4 = lane % F4_ROW, lrow = lane / F4_ROW;
  const bool kv = c4 * 4 < p.Din;
  float4 ra[NLD];
  uint8_t rm = 0;
  auto gload = [&](int64_t tl) {              // branch-free (exact s_waitcnt counts): tail rows re-read row N-1
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      int64_t r = tl * 32 + lrow + RPL * j;
      r = r < p.N ? r : p.N - 1;
      float4 v = *reinterpret_cast<const float4*>(p.x + r * p.ldx + (kv ? c4 * 4 : 0));
      if (!kv) v = make_float4(0.f, 0.f, 0.f, 0.f);
      ra[j] = v;
    }
    int64_t r = tl * 32 + fr;
    rm = p.mask[r < p.N ? r : p.N - 1];
  };

  const int64_t ntiles = (p.N + 31) / 32, last = ntiles - 1;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile >= ntiles) return;                 // wave-uniform; the kernel has no block barrier
  gload(tile);
  for (; tile < ntiles; tile += stride) {
    // registers -> this wave's LDS tile (the previous tile's reads completed before its MFMAs issued)
#pragma unroll
    for (int j = 0; j < NLD; ++j) *reinterpret_cast<float4*>(&xw[(lrow + RPL * j) * LD + c4 * 4]) = ra[j];
    const bool sdom = rm != 0;
    gload(min(tile + stride, last));          // next tile flies during the MFMA phase (issuing the loads per tile
                                              // instead -- 200 VGPRs, two waves per SIMD -- measured the same or slower)
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* xb = &xw[fr * LD + 4 * fh];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const float4 b = *reinterpret_cast<const float4*>(xb + 8 * kb);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * kb], b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * kb + 1], b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * kb + 2], b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * kb + 3], b.w, acc, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);       // retire the prefetch before the stores (see transform_wreg_kernel)
    const int64_t row = tile * 32 + fr;
    if (row < p.N) {
      float cf = 0.f;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (optr[q] != nullptr) {
          if (q == 0 || hsel[q] != hsel[q - 1]) {
            const bool t1 = hsel[q] & 1;
            const float pre = hsel[q] == 0 ? acc[12] : hsel[q] == 1 ? acc[13] : hsel[q] == 2 ? acc[14] : acc[15];
            const float gcv = hsel[q] == 0 ? gcs[0] : hsel[q] == 1 ? gcs[1] : hsel[q] == 2 ? gcs[2] : gcs[3];
            cf = (sdom != t1) ? tanh_fast(pre + gcv) : 0.f;
            cf = t1 ? cf : -cf;
          }
          float4 o;
          o.x = fmaf(cf, wv[q].x, acc[4 * q] + bv[q].x);     o.y = fmaf(cf, wv[q].y, acc[4 * q + 1] + bv[q].y);
          o.z = fmaf(cf, wv[q].z, acc[4 * q + 2] + bv[q].z); o.w = fmaf(cf, wv[q].w, acc[4 * q + 3] + bv[q].w);
          *reinterpret_cast<float4*>(optr[q] + row * p.row_stride) = o;
        }
      }
    }
  }
}

}  // namespace

static int domain_sums_impl(const float* x, int64_t N, int32_t Din, int64_t ldx, const uint8_t* mask,
                            double* sums_io, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !mask || !sums_io) return BGNN_E_NULL;
  if (N < 0 || Din <= 0 || ldx < Din || (Din & 3) || (ldx & 3)) return BGNN_E_SHAPE;
  if (!bgnn_aligned16(x)) return BGNN_E_ALIGN;
  if (N == 0) return 0;
  const int nc4 = Din / 4, cw = nc4 < 256 ? nc4 : 256, rl = DS_NT / cw;
  const int ncol = 2 * Din + 2;
  hipStream_t st = (hipStream_t)stream;
  if (ws != nullptr) {
    // two-stage: every CU streams (>= 256 rows per block), partials in ws, one small launch adds them up
    if (ws_bytes < sizeof(double) * 256 * (size_t)ncol) return BGNN_E_WORKSPACE;
    int64_t grid = (N + 255) / 256;
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(domain_sums_kernel, dim3((unsigned)grid), dim3(DS_NT), sizeof(double) * rl * 2 * cw * 4, st,
                       x, N, Din, ldx, mask, sums_io, (double*)ws);
    BGNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(domain_sums_reduce_kernel, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, st,
                       (const double*)ws, (int)grid, ncol, sums_io);
    BGNN_LAUNCH_CHECK();
    return 0;
  }
  // 2*Din device-scope fp64 atomics per block at ~3 G/s: a block must stream >= 2048 rows to amortise them (a rank's
  // share of a partitioned graph is small), and never more blocks than CUs
  static const int64_t rpb = [] { const char* e = getenv("BGNN_DS_ROWS"); return e ? atoll(e) : 2048ll; }();
  int64_t grid = (N + rpb - 1) / rpb;
  static const int64_t gcap = [] { const char* e = getenv("BGNN_DS_GRID"); return e ? atoll(e) : 256ll; }();
  if (grid > gcap) grid = gcap;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(domain_sums_kernel, dim3((unsigned)grid), dim3(DS_NT), sizeof(double) * rl * 2 * cw * 4, st,
                     x, N, Din, ldx, mask, sums_io, (double*)nullptr);
  BGNN_LAUNCH_CHECK();
  return 0;
}

extern "C" int bgnn_domain_sums_f64(const float* x, int64_t N, int32_t Din, int64_t ldx, const uint8_t* mask,
                                    double* sums_io, void* stream) {
  return domain_sums_impl(x, N, Din, ldx, mask, sums_io, nullptr, 0, stream);
}

extern "C" size_t bgnn_domain_sums_workspace_bytes(int32_t Din) {
  return Din > 0 ? sizeof(double) * 256 * (size_t)(2 * Din + 2) : 0;
}

extern "C" int bgnn_domain_sums_ws_f64(const float* x, int64_t N, int32_t Din, int64_t ldx, const uint8_t* mask,
                                       double* sums_io, void* ws, size_t ws_bytes, void* stream) {
  if (!ws) return BGNN_E_NULL;
  return domain_sums_impl(x, N, Din, ldx, mask, sums_io, ws, ws_bytes, stream);
}

extern "C" int bgnn_domain_delta_f32(const double* sums, int32_t Din, float* delta, void* stream) {
  if (!sums || !delta) return BGNN_E_NULL;
  if (Din <= 0) return BGNN_E_SHAPE;
  hipLaunchKernelGGL(domain_delta_kernel, dim3((Din + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, Din, delta);
  BGNN_LAUNCH_CHECK();
  return 0;
}

static int transform_impl(const float* x, int64_t N, int32_t Din, int64_t ldx,
                          const uint8_t* mask, const float* delta, const double* sums,
                          int32_t n_heads, int32_t D, const float* Wp, const float* bias_p,
                          const float* gates, const float* gate_const_opt,
                          float* h_s2t_0, float* h_t2s_0, float* h_s2t_1, float* h_t2s_1,
                          int64_t ldh, int64_t row_stride, float* small_ws, void* stream,
                          int64_t n_tail_t2s = 0, int64_t n_tail_s2t = 0, const int32_t* tile_need = nullptr) {
  if (n_tail_t2s < 0 || n_tail_s2t < 0 || n_tail_t2s + n_tail_s2t > N) return BGNN_E_SHAPE;
  if (!x || !mask || (!delta && !sums) || !Wp || !bias_p || !gates || !h_s2t_0 || !h_t2s_0 || !small_ws) return BGNN_E_NULL;
  if (n_heads < 1 || n_heads > MAXH || (n_heads == 2 && (!h_s2t_1 || !h_t2s_1))) return BGNN_E_NULL;
  if (N < 0 || Din <= 0 || D <= 0 || ldx < Din || ldh < D) return BGNN_E_SHAPE;
  if ((Din & 3) || (ldx & 3) || (ldh & 3) || (row_stride & 3) || row_stride < ldh) return BGNN_E_SHAPE;
  if (!bgnn_aligned16(x) || !bgnn_aligned16(Wp) || !bgnn_aligned16(gates)) return BGNN_E_ALIGN;
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int NC = n_heads * 2 * (int)ldh;
  float* wd = small_ws;            // [NC]
  float* gc = small_ws + NC;       // [n_heads*2]
  hipLaunchKernelGGL(wd_kernel, dim3((unsigned)((NC + 2 * n_heads + 3) / 4)), dim3(256), 0, st, Wp, NC, Din, delta, sums, gates,
                     gate_const_opt, n_heads, wd, gc);
  BGNN_LAUNCH_CHECK();
  GemmParams p{};
  p.x = x; p.ldx = ldx; p.N = N; p.Din = Din; p.mask = mask; p.Wp = Wp; p.bias = bias_p; p.wd = wd; p.g = gates; p.gc = gc;
  p.out[0][0] = h_s2t_0; p.out[0][1] = h_t2s_0; p.out[1][0] = h_s2t_1; p.out[1][1] = h_t2s_1;
  p.ldh = ldh; p.row_stride = row_stride; p.NC = NC; p.n_heads = n_heads; p.relu = 0; p.colsum = nullptr; p.col_off = 0;
  const int64_t nrt = (N + BM - 1) / BM;
  const int64_t nrt8 = (nrt + 7) / 8 * 8;        // row tiles rounded up to the XCD group size
  static const int n_cu = [] {
    int dev = 0; hipDeviceProp_t prop;
    return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }();
  static const bool use_wreg = [] { const char* e = getenv("BGNN_GEMM_WREG"); return !e || atoi(e) != 0; }();
  // Tail rows that need ONE table (the resident input halo of a partitioned graph: a halo row feeds destinations of one
  // domain): the W-stationary kernel lets the other table's waves sit those row tiles out (GemmParams::tail_*).  Outside
  // its envelope the tail rows simply get both tables like every other row.
  p.tail_t2s_begin = 0; p.tail_s2t_begin = 0;
  if ((n_tail_t2s || n_tail_s2t) && use_wreg && n_heads == 1 && NC % 64 == 0 && Din <= 128) {
    p.tail_t2s_begin = N - n_tail_t2s - n_tail_s2t;
    p.tail_s2t_begin = N - n_tail_s2t;
    if (p.tail_s2t_begin == 0) p.tail_s2t_begin = 1, p.tail_t2s_begin = p.tail_t2s_begin > 1 ? 1 : p.tail_t2s_begin;   // (rows from 0: keep the "tails on" encoding; a tile never ends at row 0)
  }
  // barrier-free producer / consumer pipeline (bgnn_transform_stream.hip): one head, 128 or 256 packed columns, 64 < Din <= 128
  static const bool use_stream = [] { const char* e = getenv("BGNN_GEMM_STREAM"); return !e || atoi(e) != 0; }();
  p.tile_need = tile_need;                       // (honoured by the stream kernel only; the others write both tables everywhere)
  if (use_stream && bgnn_tf_stream_supported(p, 0)) return bgnn_tf_stream_launch(p, 0, st, n_cu);
  if (use_wreg && NC % 64 == 0 && Din <= 128) {   // (Din = 256 needs 128 weight registers per lane and spills)
    // W-stationary persistent kernel: one 512-thread block per CU, column groups of 32*NCT in grid.y
    // NC <= 64: 4-wave blocks (2 column tiles x 2 row sub-tiles), two per CU; wider: 8-wave blocks, one per CU
    const int nct = NC % 256 == 0 ? 8 : NC % 128 == 0 ? 4 : 2;   // full column groups only
    const int nw = nct == 2 ? 4 : 8;
    const int bmw = 32 * (nw / nct);
    const int64_t ntiles = (N + bmw - 1) / bmw;
    const int ncg = (NC + 32 * nct - 1) / (32 * nct);
    const int64_t gx = (int64_t)n_cu * (nw == 4 ? 2 : 1);
    const dim3 grid((unsigned)(ntiles < gx ? ntiles : gx), (unsigned)ncg);
    static const bool bf3 = [] { const char* e = getenv("BGNN_GEMM_BF3"); return !e || atoi(e) != 0; }();
#define BGNN_WREG(DK, NCT, NW) do { if (bf3 && NCT >= 4) hipLaunchKernelGGL((transform_wreg_kernel<DK, NCT, NW, true, 0>), grid, dim3(64 * NW), 0, st, p); \
                                     else hipLaunchKernelGGL((transform_wreg_kernel<DK, NCT, NW, false, 0>), grid, dim3(64 * NW), 0, st, p); } while (0)
#define BGNN_WREG_DK(NCT, NW) do { if (Din <= 64) BGNN_WREG(64, NCT, NW); else BGNN_WREG(128, NCT, NW); } while (0)
    if (nct == 2) BGNN_WREG_DK(2, 4); else if (nct == 4) BGNN_WREG_DK(4, 8); else BGNN_WREG_DK(8, 8);
#undef BGNN_WREG_DK
#undef BGNN_WREG
  } else if (use_wreg && NC <= 24 && Din <= 128) {
    // wave-autonomous stream: two 4-wave blocks per CU, each wave strides over 32-row tiles
    const int64_t nt = (N + 31) / 32, nb = (nt + 3) / 4;
    const unsigned grid = (unsigned)(nb < 2 * n_cu ? nb : 2 * n_cu);
    if (Din <= 64) hipLaunchKernelGGL((transform_skinny_kernel<64>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((transform_skinny_kernel<128>), dim3(grid), dim3(256), 0, st, p);
  } else if (NC <= 32) {
    hipLaunchKernelGGL((transform_gemm_kernel<32, 4, 1, 1, 1>), dim3((unsigned)nrt8), dim3(256), 0, st, p);
  } else if (NC <= 64) {
    hipLaunchKernelGGL((transform_gemm_kernel<64, 2, 2, 2, 1>), dim3((unsigned)nrt8), dim3(256), 0, st, p);
  } else {
    const int nct = (NC + 127) / 128;
    hipLaunchKernelGGL((transform_gemm_kernel<128, 2, 2, 2, 2>), dim3((unsigned)(nrt8 * nct)), dim3(256), 0, st, p);
  }
  BGNN_LAUNCH_CHECK();
  return 0;
}

extern "C" int bgnn_adaptedconv_transform_f32(const float* x, int64_t N, int32_t Din, int64_t ldx,
                                              const uint8_t* mask, const float* delta,
                                              int32_t n_heads, int32_t D, const float* Wp, const float* bias_p,
                                              const float* gates, const float* gate_const_opt,
                                              float* h_s2t_0, float* h_t2s_0, float* h_s2t_1, float* h_t2s_1,
                                              int64_t ldh, int64_t row_stride, float* small_ws, void* stream) {
  if (!delta) return BGNN_E_NULL;
  if (!bgnn_aligned16(delta)) return BGNN_E_ALIGN;
  return transform_impl(x, N, Din, ldx, mask, delta, nullptr, n_heads, D, Wp, bias_p, gates, gate_const_opt,
                        h_s2t_0, h_t2s_0, h_s2t_1, h_t2s_1, ldh, row_stride, small_ws, stream);
}

extern "C" int bgnn_adaptedconv_transform_sums_f32(const float* x, int64_t N, int32_t Din, int64_t ldx,
                                                   const uint8_t* mask, const double* sums,
                                                   int32_t n_heads, int32_t D, const float* Wp, const float* bias_p,
                                                   const float* gates, const float* gate_const_opt,
                                                   float* h_s2t_0, float* h_t2s_0, float* h_s2t_1, float* h_t2s_1,
                                                   int64_t ldh, int64_t row_stride, int64_t n_tail_t2s, int64_t n_tail_s2t,
                                                   float* small_ws, void* stream) {
  if (!sums) return BGNN_E_NULL;
  return transform_impl(x, N, Din, ldx, mask, nullptr, sums, n_heads, D, Wp, bias_p, gates, gate_const_opt,
                        h_s2t_0, h_t2s_0, h_s2t_1, h_t2s_1, ldh, row_stride, small_ws, stream, n_tail_t2s, n_tail_s2t);
}

extern "C" int bgnn_adaptedconv_transform_need_f32(const float* x, int64_t N, int32_t Din, int64_t ldx,
                                                   const uint8_t* mask, const double* sums,
                                                   int32_t n_heads, int32_t D, const float* Wp, const float* bias_p,
                                                   const float* gates, const float* gate_const_opt,
                                                   float* h_s2t_0, float* h_t2s_0, float* h_s2t_1, float* h_t2s_1,
                                                   int64_t ldh, int64_t row_stride, const int32_t* tile_need_opt,
                                                   float* small_ws, void* stream) {
  if (!sums) return BGNN_E_NULL;
  return transform_impl(x, N, Din, ldx, mask, nullptr, sums, n_heads, D, Wp, bias_p, gates, gate_const_opt,
                        h_s2t_0, h_t2s_0, h_s2t_1, h_t2s_1, ldh, row_stride, small_ws, stream, 0, 0, tile_need_opt);
}

extern "C" int bgnn_linear_f32(const float* x, int64_t N, int32_t Din, int64_t ldx, const float* W, const float* bias,
                               int32_t Dout, int relu, const uint8_t* mask_opt, double* colsum_opt,
                               float* out, int64_t ldo, void* stream) {
  if (!x || !W || !bias || !out || (colsum_opt && !mask_opt)) return BGNN_E_NULL;
  // the W-stationary kernel's envelope: full 64-column groups, weights of a column tile in <= 96 registers
  if (N < 0 || Din <= 0 || Din > 128 || (Din & 3) || (ldx & 3) || ldx < Din || Dout <= 0 || (Dout & 63) || ldo < Dout || (ldo & 3))
    return BGNN_E_SHAPE;
  if (!bgnn_aligned16(x) || !bgnn_aligned16(W) || !bgnn_aligned16(bias) || !bgnn_aligned16(out)) return BGNN_E_ALIGN;
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  GemmParams p{};
  p.x = x; p.ldx = ldx; p.N = N; p.Din = Din; p.mask = mask_opt; p.Wp = W; p.bias = bias; p.wd = nullptr; p.g = nullptr; p.gc = nullptr;
  p.out[0][0] = out; p.out[0][1] = out; p.out[1][0] = out; p.out[1][1] = out;
  p.ldh = Dout;            // one "table" spanning all columns: column c -> (h, t) = (0, 0), offset c
  p.row_stride = ldo; p.NC = Dout; p.n_heads = 1; p.relu = relu; p.colsum = colsum_opt;
  static const int n_cu = [] {
    int dev = 0; hipDeviceProp_t prop;
    return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }();
  const int nct = Dout % 256 == 0 ? 8 : Dout % 128 == 0 ? 4 : 2;
  const int nw = nct == 2 ? 4 : 8;
  const int bmw = 32 * (nw / nct);
  const int64_t ntiles = (N + bmw - 1) / bmw;
  const int64_t gx = (int64_t)n_cu * (nw == 4 ? 2 : 1);
  const dim3 grid((unsigned)(ntiles < gx ? ntiles : gx), (unsigned)(Dout / (32 * nct)));
#define BGNN_LIN(DK, NCT, NW, BF) hipLaunchKernelGGL((transform_wreg_kernel<DK, NCT, NW, BF, 1>), grid, dim3(64 * NW), 0, st, p)
#define BGNN_LIN_DK(NCT, NW, BF) do { if (Din <= 64) BGNN_LIN(64, NCT, NW, BF); else BGNN_LIN(128, NCT, NW, BF); } while (0)
  if (nct == 2) BGNN_LIN_DK(2, 4, false); else if (nct == 4) BGNN_LIN_DK(4, 8, true); else BGNN_LIN_DK(8, 8, true);
#undef BGNN_LIN_DK
#undef BGNN_LIN
  BGNN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ linear -> narrow AdaptedConv transform, fused
// KTGNN_no_complement.forward :433 evaluates clf_target on clf_transformer(h).  With the transformer's last Linear
// folded into the conv's weights (ktgnn.py: _composed_target_pack) the only thing between h and that conv's narrow
// tables is a1 = relu(BN(Linear0(h))) -- 512 MB written and read again at C4 size.  Stage A keeps a1 in the MFMA
// accumulators and leaves 12 floats per row: (W_t a1, W_s a1, a1.g_s2t, a1.g_t2s) plus the per-domain column sums of
// a1; stage B (after the sums are complete / all-reduced) applies bias and the rank-1 domain shift exactly like the
// transform kernels' epilogue (KTGNN.py:277-284 by linearity, see the file header).
namespace {
__global__ __launch_bounds__(256) void narrow_finish_kernel(const float* __restrict__ raw, int64_t N,
                                                            const uint8_t* __restrict__ mask,
                                                            const float* __restrict__ bias, const float* __restrict__ wd,
                                                            const float* __restrict__ gc, float* __restrict__ out_s2t,
                                                            float* __restrict__ out_t2s, int64_t row_stride) {
  const float4 b0 = *reinterpret_cast<const float4*>(bias), b1 = *reinterpret_cast<const float4*>(bias + 4);
  const float4 w0 = *reinterpret_cast<const float4*>(wd), w1 = *reinterpret_cast<const float4*>(wd + 4);
  const float g0 = gc[0], g1 = gc[1];
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(raw + r * 12);
    const float4 b = *reinterpret_cast<const float4*>(raw + r * 12 + 4);
    const float2 pg = *reinterpret_cast<const float2*>(raw + r * 12 + 8);
    const bool sdom = mask[r] != 0;
    const float c0 = sdom ? -tanh_fast(pg.x + g0) : 0.f;      // -gate_s2t on source rows (table 0 = h_s2t)
    const float c1 = sdom ? 0.f : tanh_fast(pg.y + g1);       // +gate_t2s on target rows (table 1 = h_t2s)
    float4 o0, o1;
    o0.x = fmaf(c0, w0.x, a.x + b0.x); o0.y = fmaf(c0, w0.y, a.y + b0.y);
    o0.z = fmaf(c0, w0.z, a.z + b0.z); o0.w = fmaf(c0, w0.w, a.w + b0.w);
    o1.x = fmaf(c1, w1.x, b.x + b1.x); o1.y = fmaf(c1, w1.y, b.y + b1.y);
    o1.z = fmaf(c1, w1.z, b.z + b1.z); o1.w = fmaf(c1, w1.w, b.w + b1.w);
    *reinterpret_cast<float4*>(out_s2t + r * row_stride) = o0;
    *reinterpret_cast<float4*>(out_t2s + r * row_stride) = o1;
  }
}
}  // namespace

extern "C" int bgnn_linear_narrow_transform_f32(const float* x, int64_t N, int32_t Din, int64_t ldx, const float* W,
                                                const float* bias, int32_t Dout, int relu, const uint8_t* mask,
                                                double* colsum, const float* Wp2, const float* gates2,
                                                float* raw, void* stream) {
  if (!x || !W || !bias || !mask || !colsum || !Wp2 || !gates2 || !raw) return BGNN_E_NULL;
  // one column group of the W-stationary kernel must hold the whole activation row
  if (N < 0 || Din <= 0 || Din > 128 || (Din & 3) || (ldx & 3) || ldx < Din || (Dout != 64 && Dout != 128 && Dout != 256))
    return BGNN_E_SHAPE;
  if (!bgnn_aligned16(x) || !bgnn_aligned16(W) || !bgnn_aligned16(bias) || !bgnn_aligned16(raw)) return BGNN_E_ALIGN;
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  GemmParams p{};
  p.x = x; p.ldx = ldx; p.N = N; p.Din = Din; p.mask = mask; p.Wp = W; p.bias = bias;
  p.ldh = Dout; p.row_stride = Dout; p.NC = Dout; p.n_heads = 1; p.relu = relu ? 1 : 0; p.colsum = colsum;
  p.w2 = Wp2; p.g2 = gates2; p.raw = raw;
  static const int n_cu = [] {
    int dev = 0; hipDeviceProp_t prop;
    return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }();
  // The barrier-free pipeline's form of this launch (transform_stream2_kernel) is OPT-IN (BGNN_GEMM_STREAM2=1): measured
  // 0.270 ms against this kernel's 0.254-0.276 ms on C4 -- its waves spend the time the block kernel spends at barriers waiting
  // at the ring's counters instead (profiles/r03/README.md); same results, kept for the next round's work on the tail behind
  // the first-stage MFMAs.
  static const bool use_stream2 = [] { const char* e = getenv("BGNN_GEMM_STREAM2"); return e && atoi(e) != 0; }();
  if (use_stream2 && bgnn_tf_stream_supported(p, 2)) return bgnn_tf_stream_launch(p, 2, st, n_cu);
  const int nct = Dout == 256 ? 8 : Dout == 128 ? 4 : 2;
  const int nw = nct == 2 ? 4 : 8;
  const int bmw = 32 * (nw / nct);
  const int64_t ntiles = (N + bmw - 1) / bmw;
  const int64_t gx = (int64_t)n_cu * (nw == 4 ? 2 : 1);
  const dim3 grid((unsigned)(ntiles < gx ? ntiles : gx), 1u);
#define BGNN_LIN2(DK, NCT, NW, BF) hipLaunchKernelGGL((transform_wreg_kernel<DK, NCT, NW, BF, 2>), grid, dim3(64 * NW), 0, st, p)
#define BGNN_LIN2_DK(NCT, NW, BF) do { if (Din <= 64) BGNN_LIN2(64, NCT, NW, BF); else BGNN_LIN2(128, NCT, NW, BF); } while (0)
  if (nct == 2) BGNN_LIN2_DK(2, 4, false); else if (nct == 4) BGNN_LIN2_DK(4, 8, true); else BGNN_LIN2_DK(8, 8, true);
#undef BGNN_LIN2_DK
#undef BGNN_LIN2
  BGNN_LAUNCH_CHECK();
  return 0;
}

extern "C" int bgnn_classifier_stage_f32(const float* x, int64_t N, int32_t Din, int64_t ldx, const uint8_t* mask,
                                         const double* sums_x, int32_t sk_heads, int32_t sk_D, const float* sk_Wp,
                                         const float* sk_bias, const float* sk_gates, const float* sk_gate_const_opt,
                                         float* h_s2t_0, float* h_t2s_0, float* h_s2t_1, float* h_t2s_1, int64_t sk_ldh,
                                         int64_t sk_row_stride, const float* W, const float* bias, int32_t Dout, int relu,
                                         double* colsum, const float* Wp2, const float* gates2, float* raw, float* small_ws,
                                         void* stream) {
  if (!x || !mask || !sums_x || !sk_Wp || !sk_bias || !sk_gates || !h_s2t_0 || !h_t2s_0 || !W || !bias || !colsum || !Wp2 || !gates2 ||
      !raw || !small_ws) return BGNN_E_NULL;
  if (sk_heads < 1 || sk_heads > MAXH || (sk_heads == 2 && (!h_s2t_1 || !h_t2s_1))) return BGNN_E_NULL;
  if (N < 0 || Din <= 0 || (Din & 3) || (ldx & 3) || ldx < Din || sk_D <= 0 || sk_ldh < sk_D || (sk_ldh & 3) || (sk_row_stride & 3) ||
      sk_row_stride < sk_ldh) return BGNN_E_SHAPE;
  if (!bgnn_aligned16(x) || !bgnn_aligned16(sk_Wp) || !bgnn_aligned16(sk_bias) || !bgnn_aligned16(raw) || !bgnn_aligned16(small_ws))
    return BGNN_E_ALIGN;
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int sk_NC = sk_heads * 2 * (int)sk_ldh;
  float* wd = small_ws;            // [sk_NC]
  float* gc = small_ws + sk_NC;    // [sk_heads * 2]
  GemmParams p{};
  p.x = x; p.ldx = ldx; p.N = N; p.Din = Din; p.mask = mask; p.Wp = W; p.bias = bias;
  p.ldh = Dout; p.row_stride = Dout; p.NC = Dout; p.n_heads = 1; p.relu = relu ? 1 : 0; p.colsum = colsum;
  p.w2 = Wp2; p.g2 = gates2; p.raw = raw;
  p.sk_Wp = sk_Wp; p.sk_bias = sk_bias; p.sk_wd = wd; p.sk_g = sk_gates; p.sk_gc = gc;
  p.sk_out[0][0] = h_s2t_0; p.sk_out[0][1] = h_t2s_0; p.sk_out[1][0] = h_s2t_1; p.sk_out[1][1] = h_t2s_1;
  p.sk_ldh = sk_ldh; p.sk_row_stride = sk_row_stride; p.sk_NC = sk_NC; p.sk_heads = sk_heads;
  if (!bgnn_tf_cls_supported(p)) return BGNN_E_SHAPE;
  hipLaunchKernelGGL(wd_kernel, dim3((unsigned)((sk_NC + 2 * sk_heads + 3) / 4)), dim3(256), 0, st, sk_Wp, sk_NC, Din, (const float*)nullptr,
                     sums_x, sk_gates, sk_gate_const_opt, sk_heads, wd, gc);
  BGNN_LAUNCH_CHECK();
  static const int n_cu = [] {
    int dev = 0; hipDeviceProp_t prop;
    return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }();
  return bgnn_tf_cls_launch(p, st, n_cu);
}

extern "C" int bgnn_narrow_transform_finish_f32(const float* raw, int64_t N, const uint8_t* mask, const double* sums,
                                                int32_t Din, const float* Wp2, const float* bias2, const float* gates2,
                                                const float* gate_const_opt, float* h_s2t, float* h_t2s,
                                                int64_t row_stride, float* small_ws, void* stream) {
  if (!raw || !mask || !sums || !Wp2 || !bias2 || !gates2 || !h_s2t || !h_t2s || !small_ws) return BGNN_E_NULL;
  if (N < 0 || Din <= 0 || (Din & 3) || row_stride < 4 || (row_stride & 3)) return BGNN_E_SHAPE;
  if (!bgnn_aligned16(raw) || !bgnn_aligned16(h_s2t) || !bgnn_aligned16(h_t2s) || !bgnn_aligned16(bias2) ||
      !bgnn_aligned16(small_ws)) return BGNN_E_ALIGN;
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  float* wd = small_ws;        // [8]
  float* gc = small_ws + 8;    // [2]
  hipLaunchKernelGGL(wd_kernel, dim3(3), dim3(256), 0, st, Wp2, 8, Din, (const float*)nullptr, sums, gates2, gate_const_opt, 1, wd, gc);
  BGNN_LAUNCH_CHECK();
  const int64_t nb = (N + 255) / 256;
  hipLaunchKernelGGL(narrow_finish_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, st, raw, N, mask, bias2, wd, gc,
                     h_s2t, h_t2s, row_stride);
  BGNN_LAUNCH_CHECK();
  return 0;
}

