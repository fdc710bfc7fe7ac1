/* ORACLE -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * Plain-C CPU restatement of the reference's hot-path arithmetic, for (i) parity tests at sizes
 * numpy is too slow for and (ii) bench.py's `cpu_baseline` leg ("kind":"port").  Only tests/,
 * __graft_entry__.smoke() and bench.py may load the shared object built from this file; nothing
 * under bridged_gnn_amd/ links or calls it.  Pinning: checked against oracle_np.py (which is
 * pinned against the reference's golden vectors) in tests/test_oracle_c.py.
 *
 * file:line citations are relative to /root/reference/Bridged-GNN/.
 * Build: make -C oracle   (gcc -O2 -fopenmp -fno-fast-math; fp contraction OFF so that the
 * canonical fp64 sums below are plain IEEE add/mul in index order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* AdaptedConv dense transform -- models/KTGNN.py:275-284.
 * h_s2t = lin_t(x - tanh([x||d] g_s2t) d [i in S]);  h_t2s = lin_s(x + tanh([x||d] g_t2s) d [i in T]) */
void orc_adaptedconv_transform_f32(const float* x, int64_t N, int32_t Din, const uint8_t* mask,
                                   const float* Ws, const float* bs, const float* Wt, const float* bt,
                                   const float* g_s2t, const float* g_t2s, int32_t D,
                                   float* h_s2t, float* h_t2s) {
  float* diff = (float*)calloc(Din, sizeof(float));
  double* ms = (double*)calloc(Din, sizeof(double));
  double* mt = (double*)calloc(Din, sizeof(double));
  int64_t ns = 0, nt = 0;
  for (int64_t i = 0; i < N; ++i) {                       /* :275 masked means */
    const float* xi = x + i * Din;
    if (mask[i]) { ns++; for (int c = 0; c < Din; ++c) ms[c] += xi[c]; }
    else         { nt++; for (int c = 0; c < Din; ++c) mt[c] += xi[c]; }
  }
  for (int c = 0; c < Din; ++c)
    diff[c] = (float)(ms[c] / (double)(ns ? ns : 1)) - (float)(mt[c] / (double)(nt ? nt : 1));
  float cs = 0.f, ct = 0.f;                               /* the [.. || diff] half of the GEMV */
  for (int c = 0; c < Din; ++c) { cs += diff[c] * g_s2t[Din + c]; ct += diff[c] * g_t2s[Din + c]; }
#pragma omp parallel
  {
    float* xs = (float*)malloc(sizeof(float) * Din);
    float* xt = (float*)malloc(sizeof(float) * Din);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
      const float* xi = x + i * Din;
      float ds = 0.f, dt = 0.f;
      for (int c = 0; c < Din; ++c) { ds += xi[c] * g_s2t[c]; dt += xi[c] * g_t2s[c]; }
      float gs = tanhf(ds + cs), gt = tanhf(dt + ct);     /* :277-278 */
      float fs = mask[i] ? 1.f : 0.f, ft = mask[i] ? 0.f : 1.f;
      for (int c = 0; c < Din; ++c) {                     /* :279-280 */
        xs[c] = xi[c] - gs * diff[c] * fs;
        xt[c] = xi[c] + gt * diff[c] * ft;
      }
      for (int o = 0; o < D; ++o) {                       /* :283-284 */
        const float* wt = Wt + (int64_t)o * Din;
        const float* ws = Ws + (int64_t)o * Din;
        float a = 0.f, b = 0.f;
        for (int c = 0; c < Din; ++c) { a += xs[c] * wt[c]; b += xt[c] * ws[c]; }
        h_s2t[i * D + o] = a + (bt ? bt[o] : 0.f);
        h_t2s[i * D + o] = b + (bs ? bs[o] : 0.f);
      }
    }
    free(xs); free(xt);
  }
  free(diff); free(ms); free(mt);
}

/* AdaptedConv attention + aggregation over a by-destination CSR -- models/KTGNN.py:292-305,
 * :317-319, PyG softmax (scatter-max / exp / scatter-add / div(+1e-16)) and propagate(aggr=add).
 * Destination i in S uses (h_t2s, a_t2s), i in T uses (h_s2t, a_s2t).  alpha (CSR order) optional. */
void orc_adaptedconv_aggregate_f32(const float* h_t2s, const float* h_s2t, const float* a_t2s,
                                   const float* a_s2t, const int32_t* rowptr, const int32_t* col,
                                   const uint8_t* mask, int64_t N, int32_t D, int64_t ldh,
                                   float slope, float* out, int64_t ldo, float* alpha) {
#pragma omp parallel
  {
    int cap = 1024;
    float* e = (float*)malloc(sizeof(float) * cap);
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = 0; i < N; ++i) {
      const float* H = mask[i] ? h_t2s : h_s2t;
      const float* a = mask[i] ? a_t2s : a_s2t;
      const float* hi = H + i * ldh;
      int32_t b = rowptr[i], en = rowptr[i + 1], deg = en - b;
      if (deg > cap) { cap = deg * 2; e = (float*)realloc(e, sizeof(float) * cap); }
      float m = -INFINITY;
      for (int32_t t = 0; t < deg; ++t) {                 /* :292-295 */
        const float* hj = H + (int64_t)col[b + t] * ldh;
        float s = 0.f;
        for (int c = 0; c < D; ++c) {
          float v = hj[c] + hi[c];
          v = v > 0.f ? v : slope * v;
          s += v * a[c];
        }
        e[t] = s;
        if (s > m) m = s;
      }
      float sum = 0.f;                                    /* :299 */
      for (int32_t t = 0; t < deg; ++t) { e[t] = expf(e[t] - m); sum += e[t]; }
      float* o = out + i * ldo;
      for (int c = 0; c < D; ++c) o[c] = 0.f;
      for (int32_t t = 0; t < deg; ++t) {                 /* :303-305 */
        float al = e[t] / (sum + 1e-16f);
        if (alpha) alpha[b + t] = al;
        const float* hj = H + (int64_t)col[b + t] * ldh;
        for (int c = 0; c < D; ++c) o[c] += hj[c] * al;
      }
    }
    free(e);
  }
}

/* The same aggregation with every intermediate in fp64 (inputs are the fp32 tables): the real-arithmetic value of the
 * reference's formula on those inputs, to ~1e-15.  Used to split the parity budget: |GPU - truth| and |fp32 oracle -
 * truth| are measured separately (tests/test_gpu_ktgnn.py), so a tolerance never has to absorb the CHECKER's own
 * fp32 rounding. */
void orc_adaptedconv_aggregate_f64(const float* h_t2s, const float* h_s2t, const float* a_t2s,
                                   const float* a_s2t, const int32_t* rowptr, const int32_t* col,
                                   const uint8_t* mask, int64_t N, int32_t D, int64_t ldh,
                                   float slope, double* out, int64_t ldo) {
#pragma omp parallel
  {
    int cap = 1024;
    double* e = (double*)malloc(sizeof(double) * cap);
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = 0; i < N; ++i) {
      const float* H = mask[i] ? h_t2s : h_s2t;
      const float* a = mask[i] ? a_t2s : a_s2t;
      const float* hi = H + i * ldh;
      int32_t b = rowptr[i], en = rowptr[i + 1], deg = en - b;
      if (deg > cap) { cap = deg * 2; e = (double*)realloc(e, sizeof(double) * cap); }
      double m = -INFINITY;
      for (int32_t t = 0; t < deg; ++t) {
        const float* hj = H + (int64_t)col[b + t] * ldh;
        double s = 0.0;
        for (int c = 0; c < D; ++c) {
          double v = (double)hj[c] + (double)hi[c];
          v = v > 0.0 ? v : (double)slope * v;
          s += v * (double)a[c];
        }
        e[t] = s;
        if (s > m) m = s;
      }
      double sum = 0.0;
      for (int32_t t = 0; t < deg; ++t) { e[t] = exp(e[t] - m); sum += e[t]; }
      double* o = out + i * ldo;
      for (int c = 0; c < D; ++c) o[c] = 0.0;
      for (int32_t t = 0; t < deg; ++t) {
        double al = e[t] / (sum + 1e-16);
        const float* hj = H + (int64_t)col[b + t] * ldh;
        for (int c = 0; c < D; ++c) o[c] += (double)hj[c] * al;
      }
    }
    free(e);
  }
}

/* CANONICAL row normalisation (shared bit-for-bit with the HIP kernel): fp64 sum of squares in
 * index order, fp64 sqrt, round to fp32, clamp at eps, one IEEE fp32 divide per element.
 * Real-arithmetic meaning: CosineSimilarity(dim=1, eps=1e-8), models/models.py:127. */
void orc_l2_normalize_rows_f32(const float* q, int64_t n, int32_t d, float eps, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int c = 0; c < d; ++c) { double v = (double)q[i * d + c]; s = s + v * v; }
    float nr = (float)sqrt(s);
    if (!(nr > eps)) nr = eps;
    for (int c = 0; c < d; ++c) out[i * d + c] = q[i * d + c] / nr;
  }
}

/* keep the k best (score desc, index asc) in sorted order by insertion */
static inline void topk_insert(double* bs, int64_t* bi, int k, int* cnt, double s, int64_t j) {
  int n = *cnt;
  if (n == k && !(s > bs[k - 1])) return;                 /* equal score, larger index loses */
  int p = n < k ? n : k - 1;
  while (p > 0 && (bs[p - 1] < s)) { bs[p] = bs[p - 1]; bi[p] = bi[p - 1]; --p; }
  bs[p] = s; bi[p] = j;
  if (n < k) *cnt = n + 1;
}

/* CANONICAL cosine top-k (main_bridged_graph.py:59-60 + models/models.py:127-129): score =
 * fp64 accumulation in feature-index order of exact fp32 products of NORMALISED embeddings;
 * selection = k largest by (score desc, index asc).  vals = fp64 scores (pre-sigmoid). */
void orc_cosine_topk(const float* qq, const float* qc, int64_t Nq, int64_t Nc, int32_t d, int32_t k,
                     int64_t* idx_out, double* val_out) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t i = 0; i < Nq; ++i) {
    double bs[256]; int64_t bi[256]; int cnt = 0;
    const float* a = qq + i * d;
    for (int64_t j = 0; j < Nc; ++j) {
      const float* b = qc + j * d;
      double s = 0.0;
      for (int c = 0; c < d; ++c) s = s + (double)a[c] * (double)b[c];
      topk_insert(bs, bi, k, &cnt, s, j);
    }
    for (int t = 0; t < k; ++t) { idx_out[i * k + t] = t < cnt ? bi[t] : -1; val_out[i * k + t] = t < cnt ? bs[t] : -INFINITY; }
  }
}

/* CANONICAL mlp pair logit (Similar_v2 'mlp', models/models.py:918-925,:949-951 in separable
 * eval form): sum_h w2[h]*relu(scale[h]*(A[c,h]+B[q,h])+shift[h]) + b2 in fp64, index order. */
void orc_mlp_topk(const float* A, const float* B, const float* scale, const float* shift,
                  const float* w2, float b2, int64_t Nq, int64_t Nc, int32_t H, int32_t k,
                  int64_t* idx_out, double* val_out) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t i = 0; i < Nq; ++i) {
    double bs[256]; int64_t bi[256]; int cnt = 0;
    const float* b = B + i * H;
    for (int64_t j = 0; j < Nc; ++j) {
      const float* a = A + j * H;
      double s = 0.0;
      for (int h = 0; h < H; ++h) {
        double t = (double)scale[h] * ((double)b[h] + (double)a[h]) + (double)shift[h];
        if (t < 0.0) t = 0.0;
        s = s + (double)w2[h] * t;
      }
      s = s + (double)b2;
      topk_insert(bs, bi, k, &cnt, s, j);
    }
    for (int t = 0; t < k; ++t) { idx_out[i * k + t] = t < cnt ? bi[t] : -1; val_out[i * k + t] = t < cnt ? bs[t] : -INFINITY; }
  }
}

/* Reference-SHAPED cosine scoring for the cpu_baseline leg: fp32 normalise-then-dot per pair
 * (models/models.py:127 as torch>=2.0 evaluates it), fp32 sigmoid (:129), per-row top-k. */
void orc_cosine_topk_f32_refshape(const float* qq, const float* qc, int64_t Nq, int64_t Nc, int32_t d,
                                  int32_t k, int64_t* idx_out, float* val_out) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t i = 0; i < Nq; ++i) {
    double bs[256]; int64_t bi[256]; int cnt = 0;
    const float* a = qq + i * d;
    for (int64_t j = 0; j < Nc; ++j) {
      const float* b = qc + j * d;
      float s = 0.f;
      for (int c = 0; c < d; ++c) s += a[c] * b[c];
      float p = 1.f / (1.f + expf(-s));
      topk_insert(bs, bi, k, &cnt, (double)p, j);
    }
    for (int t = 0; t < k; ++t) { idx_out[i * k + t] = t < cnt ? bi[t] : -1; val_out[i * k + t] = t < cnt ? (float)bs[t] : -INFINITY; }
  }
}
