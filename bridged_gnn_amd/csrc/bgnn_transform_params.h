// Parameter block and small helpers shared by the dense-transform kernels (bgnn_transform.hip: tiled fp32 MFMA GEMM,
// W-stationary block kernel, skinny stream; bgnn_transform_stream.hip: the barrier-free producer / consumer pipeline).
#pragma once
#include "bgnn_common.h"

namespace bgnn_tf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MAXH = 2;

struct GemmParams {
  const float* x; int64_t ldx; int64_t N; int32_t Din;
  const uint8_t* mask;
  const float* Wp;      // [NC, Din] packed: per head, ldh rows of W_t (zero padded) then ldh rows of W_s
  const float* bias;    // [NC]
  const float* wd;      // [NC]   Wp . delta
  const float* g;       // [n_heads][2][2*Din]  gate vectors (s2t, t2s), x-half first
  const float* gc;      // [n_heads][2]         delta-half constants
  float* out[MAXH][2];  // [head][table]  table 0 = h_s2t (W_t), 1 = h_t2s (W_s)
  int64_t ldh; int64_t row_stride; int32_t NC; int32_t n_heads;   // ldh = padded width of a head's row, row_stride >= ldh
  int32_t relu;         // plain-linear mode (MODE = 1): out = relu?(x W^T + b)
  double* colsum;       // plain-linear mode, optional [2*NC + 2]: per-domain column sums (+ node counts) of the output
  int32_t col_off;      // transform_wreg_kernel: the launch covers the packed columns [col_off, NC) (one table of a head)
  // MODE 2 (linear -> narrow transform, the activation never reaches HBM): second-stage operand and raw output
  const float* w2;      // [8][NC]: packed rows of the consumer conv (4 of W_t, 4 of W_s), zero padded
  const float* g2;      // [2][2*NC]: its gate vectors (s2t, t2s), x-half first
  float* raw;           // [N][12]: W_t.a (4) | W_s.a (4) | a.g_s2t | a.g_t2s | 0 | 0   for the activation row a
  // MODE 0, one head: rows [tail_t2s_begin, tail_s2t_begin) need table 1 (h_t2s) only, rows [tail_s2t_begin, N) table 0 only (the
  // resident input halo of a partitioned graph: a halo row feeds destinations of one domain).  The waves of the other
  // table sit a whole tile of such rows out (no MFMAs, no stores): half the matrix work and half the writes for them in the
  // SAME launch (three launches -- both / t2s-only / s2t-only -- measured slower than doing both tables everywhere).
  // Both = N: no tail.
  int64_t tail_t2s_begin, tail_s2t_begin;
  // stream kernel, MODE 0, optional: one int32 per 32-row tile, bit t set = some row of the tile needs table t (0 = h_s2t, 1 = h_t2s).
  // A node's h_t2s row is read only by source-domain destinations (and as its own row if it is one), its h_s2t row only by target-
  // domain destinations: in a bridged graph with s -> t bridge edges no target node ever feeds a source destination, so half of
  // the h_t2s table is dead (C4: 256 MB of writes and a quarter of the matrix work per forward).  nullptr: every tile needs both.
  const int32_t* tile_need;
  // stream kernel, fused classifier stage: a second packed operand on the SAME input rows (the narrow tables of two convs that read
  // x directly -- clf_base / clf_target on h) evaluated by one more consumer wave of the launch; nullptr: absent
  const float* sk_Wp; const float* sk_bias; const float* sk_wd; const float* sk_g; const float* sk_gc;
  float* sk_out[MAXH][2]; int64_t sk_ldh; int64_t sk_row_stride; int32_t sk_NC; int32_t sk_heads;
};

// tanh through v_exp_f32 + v_rcp_f32 (abs. error < 5e-7; the coefficient scales an O(1) rank-1 term)
__device__ __forceinline__ float tanh_fast(float z) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * z) + 1.f); }

}  // namespace bgnn_tf

// bgnn_transform_stream.hip.  mode 0: AdaptedConv transform (one head, NC = 2 * ldh in {128, 256}, ldh % 32 == 0, Din <= 128);
// mode 1: out = relu?(x W^T + b) (+ per-domain column sums); mode 2: linear -> narrow transform (+ the optional `sk_*` operand).
// Returns 0 when launched, BGNN_E_SHAPE when the shape is outside the kernel's envelope (the caller then takes another kernel).
int bgnn_tf_stream_launch(const bgnn_tf::GemmParams& p, int mode, hipStream_t st, int n_cu);
bool bgnn_tf_stream_supported(const bgnn_tf::GemmParams& p, int mode);

// bgnn_transform_cls.hip: the classifier stage's dense work in one pass over h (skinny pair + fused Linear -> narrow stage A)
int bgnn_tf_cls_launch(const bgnn_tf::GemmParams& p, hipStream_t st, int n_cu);
bool bgnn_tf_cls_supported(const bgnn_tf::GemmParams& p);
