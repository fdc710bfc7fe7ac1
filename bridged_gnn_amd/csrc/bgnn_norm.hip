// Training-mode BatchNorm1d -> ReLU -> dropout over node rows (models/KTGNN.py:420-430: `bns[ind](x)`, `F.relu`,
// `F.dropout`; :364-367 clf_transformer's BatchNorm1d + ReLU) as three streaming launches instead of torch's eight:
//   forward : colstats (fp64 column sums of x and x^2)  ->  apply (normalise + affine + ReLU + dropout mask)
//   backward: reduce (sum g', sum g'.xhat)              ->  apply (dx)
// Nothing but x is kept for the backward: the ReLU state is re-derived from x and the dropout mask from the
// counter-based hash of (seed, element index).  HBM-bound: 4 B/element read per reduction, 8-12 B/element per apply.
#include "bgnn_common.h"

namespace {

constexpr int NT = 256;
// the column sums are accumulated into NREP replicas (block b -> replica b % NREP) and added up by the readers: 2048 blocks x one
// fp64 atomic per column on ONE address each retire at ~25 ns apiece -- 50 us of tail under a 10 MB input
constexpr int NREP = 8;

// dropout: 16 random bits per element, keep <=> bits >= thr (thr = round(p * 65536)); two 32-bit words per float4
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ void drop_words(uint64_t q, uint64_t seed, uint32_t& w0, uint32_t& w1) {
  const uint32_t a = fmix32((uint32_t)q * 0x9E3779B1u + (uint32_t)seed);
  const uint32_t b = fmix32((uint32_t)(q >> 32) * 0x7FEB352Du + (uint32_t)(seed >> 32) + a);
  w0 = fmix32(a ^ b ^ 0x2C1B3C6Du);
  w1 = fmix32(w0 + b + 0x297A2D39u);
}

struct NormParams {
  const float* x; int64_t N; int32_t D; int64_t ldx;
  const double* stats;          // [2 * D]: sum x | sum x^2
  const float* gamma; const float* beta; float eps;
  int relu; uint32_t thr; float keep_scale; uint64_t seed;
  const uint64_t* seed_dev;     // optional device word added to `seed` (a captured HIP graph bakes `seed` in; the word moves per replay)
  float* y; int64_t ldy;
  float momentum; float* running_mean; float* running_var;   // updated by block 0 when non-null
  const float* gy; int64_t ldg; double* gsum;                 // backward: [2 * D]: sum g' | sum g'.xhat
  float* gx; int64_t ldgx;
};

// threads of a block: LPR = D/4 lanes per row (one float4 each), NT / LPR rows per pass
struct Lay {
  int lpr, rpp, cg, rof;
  bool on;
};
__device__ __forceinline__ Lay lay_of(int D) {
  Lay l;
  l.lpr = D >> 2;
  l.rpp = NT / l.lpr;
  l.cg = threadIdx.x % l.lpr;
  l.rof = threadIdx.x / l.lpr;
  l.on = l.rof < l.rpp;
  return l;
}

// accumulator layout (doubles): NREP x [2*D] replicas (block b adds into replica b % NREP); a reader block adds them up once,
// cooperatively (block_totals).  (A "last block finalises" ticket on one address cost 2048 x 90 ns -- worse than what it saved.)
__host__ __device__ inline size_t acc_doubles(int D) { return (size_t)NREP * 2 * D; }
constexpr int MAXD = 1024;

__device__ __forceinline__ void block_sum8_to_global(double v[8], const Lay& l, int D, double* out /*acc_doubles(D)*/) {
  // rows of the block -> one value per column through LDS, then one fp64 atomic per (block, column, quantity) into the block's replica
  __shared__ double red[NT * 8];
#pragma unroll
  for (int c = 0; c < 8; ++c) red[c * NT + threadIdx.x] = l.on ? v[c] : 0.0;
  __syncthreads();
  double* rep = out + (size_t)(blockIdx.x % NREP) * 2 * D;
  for (int t = threadIdx.x; t < 8 * l.lpr; t += NT) {
    const int c = t / l.lpr, cg = t % l.lpr;
    double s = 0.0;
    for (int r = 0; r < l.rpp; ++r) s += red[c * NT + r * l.lpr + cg];
    const int q = c >> 2, col = 4 * cg + (c & 3);
    unsafeAtomicAdd(&rep[q * D + col], s);
  }
}

// totals of a replicated accumulator into LDS, by the whole block (one entry per thread and pass), then a barrier
__device__ __forceinline__ void block_totals(const double* acc, int D, double* tot /*LDS [2*D]*/) {
  for (int t = threadIdx.x; t < 2 * D; t += NT) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < NREP; ++r) s += acc[(size_t)r * 2 * D + t];
    tot[t] = s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(NT) void colstats_kernel(NormParams p) {
  const Lay l = lay_of(p.D);
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (l.on) {
    for (int64_t r = (int64_t)blockIdx.x * l.rpp + l.rof; r < p.N; r += (int64_t)gridDim.x * l.rpp) {
      const float4 a = *reinterpret_cast<const float4*>(p.x + r * p.ldx + 4 * l.cg);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
      v[4] = fma((double)a.x, (double)a.x, v[4]); v[5] = fma((double)a.y, (double)a.y, v[5]);
      v[6] = fma((double)a.z, (double)a.z, v[6]); v[7] = fma((double)a.w, (double)a.w, v[7]);
    }
  }
  block_sum8_to_global(v, l, p.D, const_cast<double*>(p.stats));
}

// batch mean / biased variance of this lane's 4 columns (fp64 from the sums), as fp32 mean and 1/sqrt(var + eps)
__device__ __forceinline__ void col_consts(const NormParams& p, const double* st /*totals [2*D]*/, int cg, float mean[4], float rstd[4], float ga[4], float be[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * cg + e;
    const double m = st[c] / (double)p.N;
    double var = st[p.D + c] / (double)p.N - m * m;
    var = var > 0.0 ? var : 0.0;
    mean[e] = (float)m;
    rstd[e] = (float)(1.0 / sqrt(var + (double)p.eps));
    ga[e] = p.gamma ? p.gamma[c] : 1.f;
    be[e] = p.beta ? p.beta[c] : 0.f;
  }
}

__global__ __launch_bounds__(NT) void bn_apply_kernel(NormParams p) {
  if (p.seed_dev != nullptr) p.seed += *p.seed_dev;
  const Lay l = lay_of(p.D);
  __shared__ double st[2 * MAXD];
  block_totals(p.stats, p.D, st);
  if (blockIdx.x == 0 && p.running_mean != nullptr) {      // torch.nn.BatchNorm1d: unbiased variance into the running buffer
    for (int c = threadIdx.x; c < p.D; c += NT) {
      const double m = st[c] / (double)p.N;
      double var = st[p.D + c] / (double)p.N - m * m;
      var = var > 0.0 ? var : 0.0;
      const double unb = p.N > 1 ? var * (double)p.N / (double)(p.N - 1) : var;
      p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * (float)m;
      p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unb;
    }
  }
  if (!l.on) return;
  float mean[4], rstd[4], ga[4], be[4];
  col_consts(p, st, l.cg, mean, rstd, ga, be);
  for (int64_t r = (int64_t)blockIdx.x * l.rpp + l.rof; r < p.N; r += (int64_t)gridDim.x * l.rpp) {
    const float4 a = *reinterpret_cast<const float4*>(p.x + r * p.ldx + 4 * l.cg);
    float o[4] = {a.x, a.y, a.z, a.w};
    uint32_t w0 = 0xFFFFFFFFu, w1 = 0xFFFFFFFFu;
    if (p.thr != 0u) drop_words((uint64_t)r * (uint64_t)l.lpr + (uint64_t)l.cg, p.seed, w0, w1);
    const uint32_t bits[4] = {w0 & 0xFFFFu, w0 >> 16, w1 & 0xFFFFu, w1 >> 16};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = fmaf((o[e] - mean[e]) * rstd[e], ga[e], be[e]);
      if (p.relu) t = fmaxf(t, 0.f);
      o[e] = bits[e] >= p.thr ? t * p.keep_scale : 0.f;
    }
    *reinterpret_cast<float4*>(p.y + r * p.ldy + 4 * l.cg) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// g' = dL/d(bn output) = gy * keep_scale where kept and the ReLU is open, else 0
__device__ __forceinline__ void grad_prime(const NormParams& p, const Lay& l, int64_t r, const float mean[4], const float rstd[4],
                                           const float ga[4], const float be[4], float xh[4], float gp[4]) {
  const float4 a = *reinterpret_cast<const float4*>(p.x + r * p.ldx + 4 * l.cg);
  const float4 g = *reinterpret_cast<const float4*>(p.gy + r * p.ldg + 4 * l.cg);
  const float xa[4] = {a.x, a.y, a.z, a.w}, gv[4] = {g.x, g.y, g.z, g.w};
  uint32_t w0 = 0xFFFFFFFFu, w1 = 0xFFFFFFFFu;
  if (p.thr != 0u) drop_words((uint64_t)r * (uint64_t)l.lpr + (uint64_t)l.cg, p.seed, w0, w1);
  const uint32_t bits[4] = {w0 & 0xFFFFu, w0 >> 16, w1 & 0xFFFFu, w1 >> 16};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    xh[e] = (xa[e] - mean[e]) * rstd[e];
    const bool open = !p.relu || fmaf(xh[e], ga[e], be[e]) > 0.f;
    gp[e] = (open && bits[e] >= p.thr) ? gv[e] * p.keep_scale : 0.f;
  }
}

__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(NormParams p) {
  if (p.seed_dev != nullptr) p.seed += *p.seed_dev;
  const Lay l = lay_of(p.D);
  __shared__ double st[2 * MAXD];
  block_totals(p.stats, p.D, st);
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (l.on) {
    float mean[4], rstd[4], ga[4], be[4];
    col_consts(p, st, l.cg, mean, rstd, ga, be);
    for (int64_t r = (int64_t)blockIdx.x * l.rpp + l.rof; r < p.N; r += (int64_t)gridDim.x * l.rpp) {
      float xh[4], gp[4];
      grad_prime(p, l, r, mean, rstd, ga, be, xh, gp);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += (double)gp[e];
        v[4 + e] = fma((double)gp[e], (double)xh[e], v[4 + e]);
      }
    }
  }
  block_sum8_to_global(v, l, p.D, p.gsum);
}

__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(NormParams p) {
  if (p.seed_dev != nullptr) p.seed += *p.seed_dev;
  const Lay l = lay_of(p.D);
  __shared__ double st[2 * MAXD], gs[2 * MAXD];
  block_totals(p.stats, p.D, st);
  block_totals(p.gsum, p.D, gs);
  if (!l.on) return;
  float mean[4], rstd[4], ga[4], be[4], c1[4], c2[4];
  col_consts(p, st, l.cg, mean, rstd, ga, be);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    c1[e] = (float)(gs[4 * l.cg + e] / (double)p.N);
    c2[e] = (float)(gs[p.D + 4 * l.cg + e] / (double)p.N);
  }
  for (int64_t r = (int64_t)blockIdx.x * l.rpp + l.rof; r < p.N; r += (int64_t)gridDim.x * l.rpp) {
    float xh[4], gp[4], o[4];
    grad_prime(p, l, r, mean, rstd, ga, be, xh, gp);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = ga[e] * rstd[e] * (gp[e] - c1[e] - xh[e] * c2[e]);
    *reinterpret_cast<float4*>(p.gx + r * p.ldgx + 4 * l.cg) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

int grid_for(int64_t N, int D) {
  const int rpp = NT / (D / 4);
  int64_t g = (N + rpp - 1) / rpp;
  if (g > 2048) g = 2048;          // 8 blocks per CU: enough loads in flight for a plain stream
  return g < 1 ? 1 : (int)g;
}

bool shape_ok(int64_t N, int32_t D, int64_t ld) { return N >= 0 && D >= 4 && D <= 4 * NT && (D & 3) == 0 && ld >= D && (ld & 3) == 0; }

void drop_consts(float p_drop, uint32_t& thr, float& scale) {
  thr = 0u; scale = 1.f;
  if (p_drop > 0.f) {
    double t = (double)p_drop * 65536.0 + 0.5;
    thr = t >= 65535.0 ? 65535u : (uint32_t)t;
    scale = (float)(65536.0 / (65536.0 - (double)thr));
  }
}

}  // namespace

extern "C" int64_t bgnn_bn_acc_doubles(int32_t D) { return (int64_t)acc_doubles(D > 0 ? D : 0); }

extern "C" int bgnn_bn_relu_dropout_f32(const float* x, int64_t N, int32_t D, int64_t ldx, const float* gamma_opt,
                                        const float* beta_opt, float eps, int relu, float p_drop, uint64_t seed, const uint64_t* seed_dev_opt,
                                        float momentum, float* running_mean_opt, float* running_var_opt,
                                        float* y, int64_t ldy, double* stats, void* stream) {
  if (!x || !y || !stats) return BGNN_E_NULL;
  if ((running_mean_opt == nullptr) != (running_var_opt == nullptr)) return BGNN_E_NULL;
  if (!shape_ok(N, D, ldx) || !shape_ok(N, D, ldy) || !(p_drop >= 0.f && p_drop < 1.f)) return BGNN_E_SHAPE;
  if (!bgnn_aligned16(x) || !bgnn_aligned16(y)) return BGNN_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (bgnn_zero_async(stats, sizeof(double) * acc_doubles(D), st) != hipSuccess) return (int)hipErrorInvalidValue;
  if (N == 0) return 0;
  NormParams p{};
  p.x = x; p.N = N; p.D = D; p.ldx = ldx; p.stats = stats; p.gamma = gamma_opt; p.beta = beta_opt; p.eps = eps;
  p.relu = relu; p.seed = seed; p.seed_dev = seed_dev_opt; p.y = y; p.ldy = ldy;
  p.momentum = momentum; p.running_mean = running_mean_opt; p.running_var = running_var_opt;
  drop_consts(p_drop, p.thr, p.keep_scale);
  const int grid = grid_for(N, D);
  hipLaunchKernelGGL(colstats_kernel, dim3(grid), dim3(NT), 0, st, p);
  BGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(NT), 0, st, p);
  BGNN_LAUNCH_CHECK();
  return 0;
}

extern "C" int bgnn_bn_relu_dropout_bwd_f32(const float* x, const float* grad_y, int64_t N, int32_t D, int64_t ldx, int64_t ldg,
                                            const double* stats, const float* gamma_opt, const float* beta_opt, float eps,
                                            int relu, float p_drop, uint64_t seed, const uint64_t* seed_dev_opt, float* grad_x, int64_t ldgx,
                                            double* gsum, void* stream) {
  if (!x || !grad_y || !stats || !grad_x || !gsum) return BGNN_E_NULL;
  if (!shape_ok(N, D, ldx) || !shape_ok(N, D, ldg) || !shape_ok(N, D, ldgx) || !(p_drop >= 0.f && p_drop < 1.f)) return BGNN_E_SHAPE;
  if (!bgnn_aligned16(x) || !bgnn_aligned16(grad_y) || !bgnn_aligned16(grad_x)) return BGNN_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (bgnn_zero_async(gsum, sizeof(double) * acc_doubles(D), st) != hipSuccess) return (int)hipErrorInvalidValue;
  if (N == 0) return 0;
  NormParams p{};
  p.x = x; p.N = N; p.D = D; p.ldx = ldx; p.stats = stats; p.gamma = gamma_opt; p.beta = beta_opt; p.eps = eps;
  p.relu = relu; p.seed = seed; p.seed_dev = seed_dev_opt; p.gy = grad_y; p.ldg = ldg; p.gsum = gsum; p.gx = grad_x; p.ldgx = ldgx;
  drop_consts(p_drop, p.thr, p.keep_scale);
  const int grid = grid_for(N, D);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid), dim3(NT), 0, st, p);
  BGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(NT), 0, st, p);
  BGNN_LAUNCH_CHECK();
  return 0;
}
