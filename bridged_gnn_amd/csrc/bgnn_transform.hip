// AdaptedConv dense part for gfx950: domain means, domain-shift gates and the two Linear
// transforms (reference Bridged-GNN/models/KTGNN.py:275-284; Linear = PyG nn.dense.linear, :240-246).
//
//   delta   = mean_{i in S} x_i - mean_{i in T} x_i                                        (:275)
//   gate_s  = tanh([x_i || delta] . g_s2t) ,  gate_t = tanh([x_i || delta] . g_t2s)         (:277-278)
//   h_s2t_i = W_t (x_i - gate_s delta [i in S]) + b_t                                       (:279,:283)
//   h_t2s_i = W_s (x_i + gate_t delta [i in T]) + b_s                                       (:280,:284)
//
// The shifted inputs differ from x by a per-row scalar times delta, so the GEMM applies the shift as
// an A-operand transform while staging x through LDS (x is read from HBM once per output matrix,
// the [N,Din] shifted copies the reference materialises never exist).  The contraction itself is
// the only MFMA work on path B: v_mfma_f32_32x32x2_f32, exact-fp32 fmaf chains.
#include "bgnn_common.h"

namespace {

// ------------------------------------------------------------------ per-domain column sums (fp64)
__global__ __launch_bounds__(256) void domain_sums_kernel(const float* __restrict__ x, int64_t N, int32_t Din,
                                                          int64_t ldx, const uint8_t* __restrict__ mask,
                                                          double* __restrict__ sums) {
  // thread layout: cw column lanes x rl row lanes
  const int cw = Din < 256 ? Din : 256;
  const int rl = 256 / cw;
  const int tid = threadIdx.x;
  const int cl = tid % cw, r0 = tid / cw;
  const bool active = r0 < rl;
  const int64_t rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int64_t rb = blockIdx.x * rows_per_block;
  const int64_t re = min(rb + rows_per_block, N);
  extern __shared__ double sh[];   // [rl][2][cw]
  for (int cb = 0; cb < Din; cb += cw) {          // block-uniform trip count (barriers inside)
    const int c = cb + cl;
    const bool cok = active && c < Din;
    double as = 0.0, at = 0.0;
    if (cok)
      for (int64_t r = rb + r0; r < re; r += rl) {
        const double v = (double)x[r * ldx + c];
        if (mask[r]) as += v; else at += v;
      }
    if (active) { sh[(r0 * 2 + 0) * cw + cl] = as; sh[(r0 * 2 + 1) * cw + cl] = at; }
    __syncthreads();
    if (cok && r0 == 0) {
      for (int q = 1; q < rl; ++q) { as += sh[(q * 2 + 0) * cw + cl]; at += sh[(q * 2 + 1) * cw + cl]; }
      atomicAdd(&sums[c], as);
      atomicAdd(&sums[Din + c], at);
    }
    __syncthreads();
  }
  float cs = 0.f, ct = 0.f;                         // node counts (exact in fp32 up to 2^24 per thread)
  for (int64_t r = rb + tid; r < re; r += 256) { if (mask[r]) cs += 1.f; else ct += 1.f; }
  cs = bgnn::group_sum<64>(cs);
  ct = bgnn::group_sum<64>(ct);
  if ((tid & 63) == 0) {
    atomicAdd(&sums[2 * Din], (double)cs);
    atomicAdd(&sums[2 * Din + 1], (double)ct);
  }
}

__global__ void domain_delta_kernel(const double* __restrict__ sums, int32_t Din, float* __restrict__ delta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Din) return;
  const double ns = sums[2 * Din], nt = sums[2 * Din + 1];
  // mean over an empty domain is NaN in the reference (x[mask].mean(0) of 0 rows); keep that visible
  const float ms = (float)(sums[c] / ns), mt = (float)(sums[Din + c] / nt);
  delta[c] = ms - mt;
}

// ------------------------------------------------------------------ gates -> per-row shift coefficients
// coef[i] = (-gate_s * [i in S], +gate_t * [i in T])
template <int LF>
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ x, int64_t N, int32_t Din, int64_t ldx,
                                                   const uint8_t* __restrict__ mask, const float* __restrict__ delta,
                                                   const float* __restrict__ g_s2t, const float* __restrict__ g_t2s,
                                                   float* __restrict__ coef) {
  constexpr int RPW = 64 / LF;
  __shared__ float cst[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave == 0) {   // constant half of the GEMV: delta . g[Din:]
    float cs = 0.f, ct = 0.f;
    for (int c = lane; c < Din; c += 64) { cs = fmaf(delta[c], g_s2t[Din + c], cs); ct = fmaf(delta[c], g_t2s[Din + c], ct); }
    cs = bgnn::group_sum<64>(cs);
    ct = bgnn::group_sum<64>(ct);
    if (lane == 0) { cst[0] = cs; cst[1] = ct; }
  }
  __syncthreads();
  const float cs = cst[0], ct = cst[1];
  const int g = lane / LF, l = lane % LF;
  const int64_t nrt = (N + 4 * RPW - 1) / (4 * RPW);
  for (int64_t t = blockIdx.x; t < nrt; t += gridDim.x) {
    const int64_t i = t * (4 * RPW) + wave * RPW + g;
    const bool ok = i < N;
    float ds = 0.f, dt = 0.f;
    if (ok)
      for (int c = l * 4; c < Din; c += LF * 4) {
        const float4 v = *reinterpret_cast<const float4*>(x + i * ldx + c);
        const float4 a = *reinterpret_cast<const float4*>(g_s2t + c);
        const float4 b = *reinterpret_cast<const float4*>(g_t2s + c);
        ds = fmaf(v.x, a.x, ds); ds = fmaf(v.y, a.y, ds); ds = fmaf(v.z, a.z, ds); ds = fmaf(v.w, a.w, ds);
        dt = fmaf(v.x, b.x, dt); dt = fmaf(v.y, b.y, dt); dt = fmaf(v.z, b.z, dt); dt = fmaf(v.w, b.w, dt);
      }
    ds = bgnn::group_sum<LF>(ds);
    dt = bgnn::group_sum<LF>(dt);
    if (ok && l == 0) {
      const bool s = mask[i] != 0;
      float2 o;
      o.x = s ? -tanhf(ds + cs) : 0.f;
      o.y = s ? 0.f : tanhf(dt + ct);
      *reinterpret_cast<float2*>(coef + i * 2) = o;
    }
  }
}

// ------------------------------------------------------------------ MFMA fp32 GEMM with shift prologue
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BK = 32, LDS_LD = BK + 4;   // +4 floats: conflict-free ds_read_b128 (16-lane groups)

struct GemmParams {
  const float* x; int64_t ldx; int64_t N; int32_t Din;
  const float* coef; const float* delta;
  const float* W[2]; const float* b[2];   // [0] = lin_t -> h_s2t, [1] = lin_s -> h_t2s
  float* out[2]; int64_t ldh; int32_t D;
};

// BN output columns per block; waves arranged WM x WN, each computing TM x TN tiles of 32x32
template <int BN, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void transform_gemm_kernel(GemmParams p) {
  static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == BN, "tile layout");
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];
  const int which = blockIdx.z;
  const float* __restrict__ W = p.W[which];
  const float* __restrict__ bias = p.b[which];
  float* __restrict__ out = p.out[which];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int col0 = blockIdx.y * BN;

  // staging assignment: 8 threads cover one 32-float row segment, 32 rows per pass
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  constexpr int APASS = BM / 32, BPASS = BN / 32;
  float cf[APASS];
#pragma unroll
  for (int j = 0; j < APASS; ++j) {
    const int64_t r = row0 + sr + 32 * j;
    cf[j] = r < p.N ? p.coef[r * 2 + which] : 0.f;
  }
  float4 ra[APASS], rb[BPASS];
  auto gload = [&](int k0) {
    const int k = k0 + sc;
    const bool kok = k < p.Din;   // Din % 4 == 0 -> a float4 is entirely in or out
    float4 d4 = kok ? *reinterpret_cast<const float4*>(p.delta + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const int64_t r = row0 + sr + 32 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kok && r < p.N) {
        v = *reinterpret_cast<const float4*>(p.x + r * p.ldx + k);
        // x -/+ gate * delta * [domain]  (KTGNN.py:279-280); same op order as the reference: (gate*delta) then add
        v.x += cf[j] * d4.x; v.y += cf[j] * d4.y; v.z += cf[j] * d4.z; v.w += cf[j] * d4.w;
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const int n = col0 + sr + 32 * j;
      rb[j] = (kok && n < p.D) ? *reinterpret_cast<const float4*>(W + (int64_t)n * p.Din + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < APASS; ++j) *reinterpret_cast<float4*>(&As[(sr + 32 * j) * LDS_LD + sc]) = ra[j];
#pragma unroll
    for (int j = 0; j < BPASS; ++j) *reinterpret_cast<float4*>(&Bs[(sr + 32 * j) * LDS_LD + sc]) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (p.Din + BK - 1) / BK;
  gload(0);
  sstore();
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        af[a] = *reinterpret_cast<const float4*>(&As[((wm * TM + a) * 32 + fr) * LDS_LD + kb * 8 + fh * 4]);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        bf[b] = *reinterpret_cast<const float4*>(&Bs[((wn * TN + b) * 32 + fr) * LDS_LD + kb * 8 + fh * 4]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          // lanes 0-31 carry k = 8kb+s, lanes 32-63 carry k = 8kb+4+s (same permutation for A and B)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
        }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      sstore();
      __syncthreads();
    }
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int c = col0 + (wn * TN + b) * 32 + fr;
    const float bv = (bias != nullptr && c < p.D) ? bias[c] : 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row < p.N && c < p.ldh) out[row * p.ldh + c] = acc[a][b][r] + bv;
      }
  }
}

}  // namespace

extern "C" int bgnn_domain_sums_f64(const float* x, int64_t N, int32_t Din, int64_t ldx, const uint8_t* mask,
                                    double* sums_io, void* stream) {
  if (!x || !mask || !sums_io) return BGNN_E_NULL;
  if (N < 0 || Din <= 0 || ldx < Din) return BGNN_E_SHAPE;
  if (N == 0) return 0;
  const int cw = Din < 256 ? Din : 256, rl = 256 / cw;
  int64_t grid = (N + 255) / 256;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(domain_sums_kernel, dim3((unsigned)grid), dim3(256), sizeof(double) * rl * 2 * cw,
                     (hipStream_t)stream, x, N, Din, ldx, mask, sums_io);
  BGNN_LAUNCH_CHECK();
  return 0;
}

extern "C" int bgnn_domain_delta_f32(const double* sums, int32_t Din, float* delta, void* stream) {
  if (!sums || !delta) return BGNN_E_NULL;
  if (Din <= 0) return BGNN_E_SHAPE;
  hipLaunchKernelGGL(domain_delta_kernel, dim3((Din + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, Din, delta);
  BGNN_LAUNCH_CHECK();
  return 0;
}

extern "C" int bgnn_adaptedconv_transform_f32(const float* x, int64_t N, int32_t Din, int64_t ldx,
                                              const uint8_t* mask, const float* delta,
                                              const float* W_s, const float* b_s, const float* W_t, const float* b_t,
                                              const float* g_s2t, const float* g_t2s, int32_t D,
                                              float* h_t2s, float* h_s2t, int64_t ldh,
                                              float* coef_ws, void* stream) {
  if (!x || !mask || !delta || !W_s || !W_t || !g_s2t || !g_t2s || !h_t2s || !h_s2t || !coef_ws) return BGNN_E_NULL;
  if (N < 0 || Din <= 0 || D <= 0 || ldx < Din || ldh < D) return BGNN_E_SHAPE;
  if ((Din & 3) || (ldx & 3) || (ldh & 3)) return BGNN_E_SHAPE;
  if (!bgnn_aligned16(x) || !bgnn_aligned16(W_s) || !bgnn_aligned16(W_t) || !bgnn_aligned16(delta) ||
      !bgnn_aligned16(g_s2t) || !bgnn_aligned16(g_t2s) || !bgnn_aligned16(coef_ws))
    return BGNN_E_ALIGN;
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  {
    const int nv = Din / 4;
    int64_t grid = 2048;
#define GATE(LF)                                                                                          \
  {                                                                                                       \
    int64_t nrt = (N + 4 * (64 / LF) - 1) / (4 * (64 / LF));                                              \
    if (nrt < grid) grid = nrt;                                                                           \
    hipLaunchKernelGGL((gate_kernel<LF>), dim3((unsigned)grid), dim3(256), 0, st, x, N, Din, ldx, mask,   \
                       delta, g_s2t, g_t2s, coef_ws);                                                     \
  }
    if (nv <= 8) GATE(8) else if (nv <= 16) GATE(16) else if (nv <= 32) GATE(32) else GATE(64)
#undef GATE
    BGNN_LAUNCH_CHECK();
  }
  GemmParams p;
  p.x = x; p.ldx = ldx; p.N = N; p.Din = Din; p.coef = coef_ws; p.delta = delta;
  p.W[0] = W_t; p.b[0] = b_t; p.out[0] = h_s2t;
  p.W[1] = W_s; p.b[1] = b_s; p.out[1] = h_t2s;
  p.ldh = ldh; p.D = D;
  const unsigned gx = (unsigned)((N + BM - 1) / BM);
  // ldh columns are produced (pad columns come out as exact zeros: zero weight rows, no bias)
  if (ldh <= 32) {
    hipLaunchKernelGGL((transform_gemm_kernel<32, 4, 1, 1, 1>), dim3(gx, 1, 2), dim3(256), 0, st, p);
  } else if (ldh <= 64) {
    hipLaunchKernelGGL((transform_gemm_kernel<64, 2, 2, 2, 1>), dim3(gx, 1, 2), dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL((transform_gemm_kernel<128, 2, 2, 2, 2>), dim3(gx, (unsigned)((ldh + 127) / 128), 2), dim3(256), 0, st, p);
  }
  BGNN_LAUNCH_CHECK();
  return 0;
}
