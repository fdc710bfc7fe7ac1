/* bgnn.h -- C ABI of libbgnn_hip.so: the MI355X (gfx950) drop-in for Bridged-GNN's hot path.
 *
 * The reference (wendongbi/Bridged-GNN) is pure Python and has NO FFI layer; its "operator
 * surface" is Python call signatures.  Each entry point below replaces the
 * third-party/ATen kernels launched by one reference call site; the reference interface it
 * stands in for is cited as file:line relative to Bridged-GNN/.  bridged_gnn_amd/_lib.py is the
 * ctypes binding a maintainer would add; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in `_host`;
 *   - nothing is allocated inside; scratch is passed as (ws, ws_bytes) sized by *_workspace_bytes;
 *   - calls are asynchronous on `stream` (a hipStream_t passed as void*), reentrant, no globals;
 *   - return value: 0 = success, <0 = argument error (BGNN_E_*), >0 = hipError_t of a failed launch;
 *   - row-major, fp32 features, int32 CSR, int64 edge_index ([2,E] contiguous, row 0 = source /
 *     "from", row 1 = destination / "to" as in torch_geometric).
 */
#ifndef BGNN_H_
#define BGNN_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI revision.  110 (round 3) is NOT call-compatible with 100: `bgnn_transform_bwd_prep_f32` takes 22 arguments (was 17)
 * and the `n_fallback_opt` of `bgnn_cosine_topk_f32` / `bgnn_mlp_pair_topk_f32` is int32[2] (was int32[1]) -- a caller built
 * against the old header must be recompiled; compare bgnn_version() with the BGNN_VERSION it was built with at load time.
 * 111 adds bgnn_adaptedconv_transform_need_f32, 112 bgnn_classifier_stage_f32, 113 bgnn_adaptedconv_aggregate_bounded_f32 (all
 * call-compatible with 110). */
#define BGNN_VERSION 113
#define BGNN_E_NULL (-1)        /* required pointer is NULL                     */
#define BGNN_E_SHAPE (-2)       /* unsupported / inconsistent shape             */
#define BGNN_E_WORKSPACE (-3)   /* ws_bytes smaller than *_workspace_bytes()    */
#define BGNN_E_ALIGN (-4)       /* pointer or leading dimension not 16-B aligned */
#define BGNN_E_RANGE (-5)       /* k / index range not supported                */

int bgnn_version(void);
const char* bgnn_error_string(int code);
/* digest of the sources this library was built from (csrc/Makefile: HASHED); the loader compares it with the
 * tree next to the library so that a stale build cannot load silently.                         */
const char* bgnn_source_hash(void);

/* ------------------------------------------------------------------------------------------
 * (a10) graph_partition -> by-destination CSR.         models/KTGNN.py:385-398, cached :409-412
 * Drops self loops and appends one per node (rewrite_self_loops=1, the reference behaviour:
 * PyG remove_self_loops + add_self_loops), then groups edges by DESTINATION.  Inside a row the
 * order is input order with the self loop last (stable), so results are run-to-run identical.
 * The (edge_index1, edge_index2) split of the reference is the per-row domain flag mask[i].
 * col / eperm need capacity E+N.  *E_out_dev (device int64) receives E' = rowptr[N].
 * eperm_opt[t] = position of CSR slot t in the rewritten edge list (kept edges in input order,
 * then the N self loops) -- lets a caller map `alpha` back to the reference's edge order.      */
size_t bgnn_csr_workspace_bytes(int64_t N, int64_t E);
int bgnn_build_dst_csr(const int64_t* edge_index, int64_t E, int64_t N, int rewrite_self_loops,
                       int32_t* rowptr, int32_t* col, int32_t* eperm_opt, int64_t* E_out_dev,
                       void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a11) AdaptedConv dense part.                                       models/KTGNN.py:275-284
 * bgnn_domain_sums_f64: per-domain column sums of x ([2*Din] doubles: S then T) and node counts
 *   ([2] doubles) accumulated into sums_io (caller zeroes it; multi-GPU callers all-reduce it).
 * bgnn_domain_delta_f32: delta = sums_S/n_S - sums_T/n_T                        (:275)
 * bgnn_adaptedconv_transform_f32:                                               (:277-284)
 *   gate_s = tanh(x.g_s2t[:Din] + delta.g_s2t[Din:]),  gate_t likewise with g_t2s
 *   h_s2t = lin_t(x - gate_s*delta*[i in S]) ; h_t2s = lin_s(x + gate_t*delta*[i in T])
 *   evaluated by linearity as  W x + b -/+ gate * (W delta)  in ONE pass over x (the gate GEMVs ride in the staging
 *   loads, the rank-1 shift in the epilogue).  Products: for >= 128 packed columns and Din <= 128 every fp32 operand is
 *   split exactly into three bf16 pieces and a product is the six piece MFMAs >= 2^-24 relative
 *   (v_mfma_f32_32x32x16_bf16, fp32 accumulate; ~2e-7 relative to fp32 arithmetic); other shapes use
 *   v_mfma_f32_32x32x2_f32.  tanh of the gates: exp/rcp form, absolute error < 5e-7.  Up to n_heads = 2 convs that share the input x (clf_base / clf_target,
 *   KTGNN.py:432,:434) are evaluated together.
 *   Wp     [n_heads*2*ldh, Din] packed weights: per head ldh rows of lin_t.weight (rows >= D zero)
 *          followed by ldh rows of lin_s.weight (torch Linear.weight layout [out, in]);
 *   bias_p [n_heads*2*ldh]      matching packed biases (zeros where absent / padded);
 *   gates  [n_heads][2][2*Din]  a_g_s2t.weight then a_g_t2s.weight per head ([x || delta] order);
 *   gate_const_opt [n_heads][2] constants added to the gates' pre-activations (NULL = 0).  Together with composed
 *          weights this evaluates a conv on x' = x.M^T + c WITHOUT materialising x' (KTGNN.py:433: clf_target on
 *          clf_transformer's last Linear): W x' + b = (W M) x + (W c + b), [x'||delta'].g = x.(M^T g_x) + c.g_x + delta.(M^T g_d);
 *   each output row holds ldh >= D floats (ldh % 4 == 0; columns D..ldh-1 come out as 0) and rows are
 *   row_stride >= ldh floats apart (row_stride % 4 == 0), so several convs' tables can be interleaved in one
 *   allocation (multi-GPU: one halo exchange then carries the rows of all of them).
 * small_ws: n_heads*(2*ldh+2) floats of scratch (W.delta and the gates' delta halves).       */
int bgnn_domain_sums_f64(const float* x, int64_t N, int32_t Din, int64_t ldx, const uint8_t* mask,
                         double* sums_io /*[2*Din+2]*/, void* stream);
/* Two-stage form of bgnn_domain_sums_f64 (same result up to fp64 summation order, and run-to-run identical): every CU
 * streams its share and leaves a partial row in ws (bgnn_domain_sums_workspace_bytes(Din) bytes, no initialisation
 * needed), a second small launch adds the rows into sums_io -- no atomics.  Same speed as the one-launch form on MI355X
 * (1M x 128: 0.11 ms either way); its point is determinism. */
size_t bgnn_domain_sums_workspace_bytes(int32_t Din);
int bgnn_domain_sums_ws_f64(const float* x, int64_t N, int32_t Din, int64_t ldx, const uint8_t* mask,
                            double* sums_io /*[2*Din+2]*/, void* ws, size_t ws_bytes, void* stream);
int bgnn_domain_delta_f32(const double* sums /*[2*Din+2]*/, int32_t Din, float* delta, void* stream);
/* bgnn_linear_f32: out = relu?(x W^T + bias) for the first Linear (+ folded eval BatchNorm + ReLU) of
 *   KTGNN_no_complement.clf_transformer (models/KTGNN.py:407-411, applied at :433), on the same W-stationary MFMA kernel
 *   as the transform; colsum_opt ([2*Dout+2], zeroed by the caller, needs mask_opt) additionally receives the per-domain
 *   column sums + node counts of the OUTPUT, i.e. the domain sums (:275) of the conv that consumes it -- no extra pass.
 *   W [Dout, Din] is torch Linear.weight layout.  Envelope: Din <= 128, Din % 4 == 0, Dout % 64 == 0 (else
 *   BGNN_E_SHAPE; callers use a library GEMM outside it). */
int bgnn_linear_f32(const float* x, int64_t N, int32_t Din, int64_t ldx, const float* W, const float* bias,
                    int32_t Dout, int relu, const uint8_t* mask_opt, double* colsum_opt,
                    float* out, int64_t ldo, void* stream);
/* Fused pair for KTGNN_no_complement.forward :433 (clf_target on clf_transformer(h), eval): the activation
 *   a = relu?(x W^T + bias) of the transformer's first Linear (+ folded BatchNorm) never reaches HBM.
 *   bgnn_linear_narrow_transform_f32 (stage A) leaves raw[N][12] = (W_t a [4] | W_s a [4] | a.g_s2t | a.g_t2s | 0 | 0)
 *   for the consumer conv's packed operands Wp2 [8][Dout] (4 rows of W_t, 4 of W_s, zero padded: D <= 4) and gates2
 *   [2][2*Dout], and accumulates the per-domain column sums + counts of a into colsum [2*Dout+2] (caller zero-fills;
 *   multi-GPU callers all-reduce it).  bgnn_narrow_transform_finish_f32 (stage B) turns raw into the conv's h_s2t /
 *   h_t2s rows (4 floats each, row_stride apart) with bias and the rank-1 domain shift of :277-284.
 *   Envelope of stage A: Din <= 128, Din % 4 == 0, Dout in {64, 128, 256} (else BGNN_E_SHAPE: use bgnn_linear_f32 +
 *   the transform).  small_ws: 16 floats. */
int bgnn_linear_narrow_transform_f32(const float* x, int64_t N, int32_t Din, int64_t ldx, const float* W,
                                     const float* bias, int32_t Dout, int relu, const uint8_t* mask,
                                     double* colsum, const float* Wp2, const float* gates2, float* raw, void* stream);
/* The classifier stage's dense work in ONE pass over the hidden activation x = h (KTGNN.py:432-434; ABI 112): the narrow tables of
 * the `sk_heads` (1 or 2) convs that read h itself -- the transform of bgnn_adaptedconv_transform_sums_f32 with `sums_x` for the packed
 * operands sk_Wp / sk_bias / sk_gates (sk_heads * 2 * sk_ldh <= 24 packed columns) -- AND stage A of the fused pair above (W, bias,
 * Dout = 128, colsum, Wp2, gates2, raw).  Same results as the two separate launches; Din in (64, 128], Din % 4 == 0, else
 * BGNN_E_SHAPE.  small_ws: sk_heads * (2 * sk_ldh + 2) + 8 floats. */
int bgnn_classifier_stage_f32(const float* x, int64_t N, int32_t Din, int64_t ldx, const uint8_t* mask,
                              const double* sums_x, int32_t sk_heads, int32_t sk_D, const float* sk_Wp,
                              const float* sk_bias, const float* sk_gates, const float* sk_gate_const_opt,
                              float* h_s2t_0, float* h_t2s_0, float* h_s2t_1, float* h_t2s_1, int64_t sk_ldh,
                              int64_t sk_row_stride, const float* W, const float* bias, int32_t Dout, int relu,
                              double* colsum, const float* Wp2, const float* gates2, float* raw, float* small_ws,
                              void* stream);
int bgnn_narrow_transform_finish_f32(const float* raw, int64_t N, const uint8_t* mask, const double* sums,
                                     int32_t Din, const float* Wp2, const float* bias2, const float* gates2,
                                     const float* gate_const_opt, float* h_s2t, float* h_t2s, int64_t row_stride,
                                     float* small_ws, void* stream);
/* bgnn_gram_f32: out[a][b] = sum_i A[i][a] * B[i][b] for tall-skinny A [N,p], B [N,q] (p <= 288, q <= 128, both % 4 == 0).
 *   The training path's weight / gate gradients of the dense transform (KTGNN.py:275-284 under autograd) are
 *   [G_s2t | G_t2s | dgate]^T . x with the node count as the reduction dimension; one streaming pass, deterministic
 *   two-stage sum.  ws: bgnn_gram_workspace_bytes(p, q). */
size_t bgnn_gram_workspace_bytes(int32_t p, int32_t q);
int bgnn_gram_f32(const float* A, int64_t lda, int32_t p, const float* B, int64_t ldb, int32_t q, int64_t N,
                  float* out /*[p][q]*/, void* ws, size_t ws_bytes, void* stream);
/* Training-mode BatchNorm1d -> ReLU -> dropout over node rows (models/KTGNN.py:420-430: `self.bns[ind](x)`, `F.relu`,
 *   `F.dropout(x, p=self.dropout, training=self.training)`; :364-367 clf_transformer's BatchNorm1d + ReLU with p = 0) in two
 *   streaming launches.  x [N, D] (D % 4 == 0, D <= 1024), batch statistics over the N rows in fp64 (`stats`, bgnn_bn_acc_doubles(D) doubles = R x [2*D] partials: column sums
 *   of x | x^2, written here and kept for the backward); y = keep(seed, element) ? max(gamma*(x-mean)/sqrt(var+eps)+beta, 0) / (1-p) : 0.
 *   The dropout mask is a counter-based hash of (seed, element index) with 16 bits per element (p is rounded to 1/65536) -- the
 *   same Bernoulli(1-p) law as torch's Philox stream, not the same bits.  running_mean / running_var (both or neither) get torch's
 *   momentum update with the unbiased variance.  `seed_dev_opt` (device, may be NULL): a 64-bit word added to `seed` by the kernels --
 *   a captured HIP graph bakes the host value of `seed` in, the device word (advanced by the caller once per step) keeps the masks of
 *   successive replays different; forward and backward of one step must see the same word.
 * bgnn_bn_relu_dropout_bwd_f32: dL/dx from dL/dy; the ReLU state is re-derived from x and the mask from (seed, index); `gsum`
 *   (bgnn_bn_acc_doubles(D) doubles = R x [2*D] partials, written here; summed over R) returns sum g' = dL/dbeta and sum g'.xhat = dL/dgamma (g' = dL/d(BN output)). */
int64_t bgnn_bn_acc_doubles(int32_t D);   /* size in doubles of `stats` / `gsum` below: R partial accumulators of [2*D]; their sum over R is the total */
int bgnn_bn_relu_dropout_f32(const float* x, int64_t N, int32_t D, int64_t ldx, const float* gamma_opt,
                             const float* beta_opt, float eps, int relu, float p_drop, uint64_t seed, const uint64_t* seed_dev_opt,
                             float momentum, float* running_mean_opt, float* running_var_opt,
                             float* y, int64_t ldy, double* stats, void* stream);
int bgnn_bn_relu_dropout_bwd_f32(const float* x, const float* grad_y, int64_t N, int32_t D, int64_t ldx, int64_t ldg,
                                 const double* stats, const float* gamma_opt, const float* beta_opt, float eps,
                                 int relu, float p_drop, uint64_t seed, const uint64_t* seed_dev_opt, float* grad_x, int64_t ldgx,
                                 double* gsum, void* stream);
/* bgnn_transform_bwd_prep_f32: row-local part of the transform's hand-derived backward (KTGNN.py:275-284 under autograd) in
 *   one stream over x and the two incoming gradient tables: gate values (tanh of x.gx[g] + gconst[g]), the gates'
 *   adjoints G.(W delta) (wd [2][2D]: row 0 = -(W_t delta) in columns 0..D-1, row 1 = W_s delta in columns D..2D-1), and
 *   Gall[N][p] = [G_s2t[:, :D] | G_t2s[:, :D] | dpre_0 dpre_1 | +1/n_S or -1/n_T | 0] (p = pad4(2D+3); counts = the
 *   two node counts at the end of the domain sums), side[N][4] = (c1, c2, 1, 0): the operands of the Gram / linear
 *   launches that follow (column 2D+2 carries the gradient through the domain means as one more rank).  din % 4 == 0.
 *   ld_gall >= p / ld_side >= 4 are the row strides: two convs on the same x write side by side into one pair of buffers.
 *   ex_opt (D <= 128; ws: bgnn_transform_bwd_prep_workspace_bytes(N, p)): the entries of ex = Gall^T side that the backward uses
 *   (ex[c][0] = sum_i c1_i G_s2t[i][c], ex[D+c][1] = sum_i c2_i G_t2s[i][c], ex[.][2] = column sums of G_s2t | G_t2s | dpre; all
 *   other entries 0) from the same pass -- per-block partials summed in a fixed order, deterministic; side_opt may then be NULL. */
size_t bgnn_transform_bwd_prep_workspace_bytes(int64_t N, int32_t p);
int bgnn_transform_bwd_prep_f32(const float* x, int64_t ldx, int64_t N, int32_t din, const float* G_s2t, const float* G_t2s,
                                int64_t ldg, int32_t D, const uint8_t* mask, const float* gx /*[2][din]*/,
                                const float* gconst /*[2]*/, const float* wd /*[2][2D]*/, const double* counts /*[2]*/,
                                float* Gall, int32_t p, int64_t ld_gall, float* side_opt, int64_t ld_side,
                                float* ex_opt /*[p][4]*/, void* ws_opt, size_t ws_bytes, void* stream);
/* The O(D x Din) algebra around the streaming launches of the transform backward (KTGNN.py:275-284 under autograd), one launch
 *   each.  consts: gx [2][din] = the x-halves of the gate vectors g1 = a_g_s2t, g2 = a_g_t2s ([x || delta] order, 2*din each),
 *   gconst [2] = delta . (their delta-halves), wd [2][2D] as bgnn_transform_bwd_prep_f32 takes it.  finish: from dWall = Gall^T x
 *   [p][din] and ex [p][4]: dW_t = dWall[:D] - u1 (x) delta, dW_s = dWall[D:2D] + u2 (x) delta, dg_g = [dWall[2D+g] | sp_g delta],
 *   db_t / db_s = ex[:D,2] / ex[D:2D,2], and wcat_t [din][ld_wcat] = the TRANSPOSED operand (W_t | W_s | g1_x | g2_x | ddl | 0) of
 *   the input-gradient launch dX = Gall . Wcat (ddl = sp_0 g1_d + sp_1 g2_d - W_t^T u1 + W_s^T u2: through the domain means).
 *   W_s, W_t are [D][din] contiguous. */
int bgnn_transform_bwd_consts_f32(const float* W_s, const float* W_t, const float* g1, const float* g2, const float* delta,
                                  int32_t D, int32_t din, float* gx, float* gconst, float* wd, void* stream);
int bgnn_transform_bwd_finish_f32(const float* dWall, const float* ex, const float* W_s, const float* W_t, const float* g1,
                                  const float* g2, const float* delta, int32_t D, int32_t din, int32_t p, float* dW_s,
                                  float* dW_t, float* dg1, float* dg2, float* db_s_opt, float* db_t_opt, float* wcat_t,
                                  int64_t ld_wcat, void* stream);
/* bgnn_rowdot_f32: out[i][j] = X[i,:d] . V[j,:d], j < nv <= 4, d <= 256 (gate pre-activations and gate adjoints of the
 *   training path): one stream over X for all vectors. */
int bgnn_rowdot_f32(const float* X, int64_t ldx, int64_t N, int32_t d, const float* V, int64_t ldv, int32_t nv,
                    float* out /*[N][nv]*/, void* stream);
int bgnn_adaptedconv_transform_f32(const float* x, int64_t N, int32_t Din, int64_t ldx,
                                   const uint8_t* mask, const float* delta,
                                   int32_t n_heads, int32_t D, const float* Wp, const float* bias_p,
                                   const float* gates, const float* gate_const_opt,
                                   float* h_s2t_0, float* h_t2s_0, float* h_s2t_1, float* h_t2s_1,
                                   int64_t ldh, int64_t row_stride, float* small_ws, void* stream);
/* Same transform taking the domain sums ([2*Din+2] doubles, after any all-reduce) instead of delta: the tiny
 * W.delta kernel forms delta with the arithmetic of bgnn_domain_delta_f32 (bit-identical), one launch less per conv.
 * n_tail_t2s / n_tail_s2t: the LAST n_tail_t2s + n_tail_s2t of the N rows (in that order) are rows whose consumer reads
 * only h_t2s / only h_s2t -- the resident input halo of a partitioned graph (dist.py): for them the other table MAY be
 * left unwritten (single-table launches inside the W-stationary kernel's envelope, both tables outside it). */
int bgnn_adaptedconv_transform_sums_f32(const float* x, int64_t N, int32_t Din, int64_t ldx,
                                        const uint8_t* mask, const double* sums,
                                        int32_t n_heads, int32_t D, const float* Wp, const float* bias_p,
                                        const float* gates, const float* gate_const_opt,
                                        float* h_s2t_0, float* h_t2s_0, float* h_s2t_1, float* h_t2s_1,
                                        int64_t ldh, int64_t row_stride, int64_t n_tail_t2s, int64_t n_tail_s2t,
                                        float* small_ws, void* stream);
/* The sums form with a per-tile need mask (ABI 111).  tile_need_opt: NULL, or one int32 per 32-row tile of x, bit 0 set = some row
 * of the tile has its h_s2t row read by the aggregation, bit 1 = its h_t2s row (a node's h_t2s row is gathered only by source-
 * domain destinations and as the node's own row if it is one, KTGNN.py:292-295; with s -> t bridge edges no target node feeds a
 * source destination and half of the h_t2s table is dead).  Rows of a table that no tile needs MAY be left unwritten.  Only the
 * stream kernel (one head, 128 / 256 packed columns, 64 < Din <= 128) honours the mask; other shapes write both tables. */
int bgnn_adaptedconv_transform_need_f32(const float* x, int64_t N, int32_t Din, int64_t ldx,
                                        const uint8_t* mask, const double* sums,
                                        int32_t n_heads, int32_t D, const float* Wp, const float* bias_p,
                                        const float* gates, const float* gate_const_opt,
                                        float* h_s2t_0, float* h_t2s_0, float* h_s2t_1, float* h_t2s_1,
                                        int64_t ldh, int64_t row_stride, const int32_t* tile_need_opt,
                                        float* small_ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a11-a13) fused GATv2 logits + per-destination softmax + weighted neighbour sum.
 *                      models/KTGNN.py:292-305, message :317-319, PyG softmax (call site :299),
 *                      MessagePassing.propagate(aggr='add') (call sites :303-304)
 * For destination row i (row_begin <= i < row_end; rowptr/mask/out/H are indexed by the absolute
 * row so a caller can aggregate interior rows while a halo exchange for the boundary rows is in
 * flight): H = mask[i] ? h_t2s : h_s2t, a = mask[i] ? a_t2s : a_s2t,
 *   e_j = a . leaky_relu(H[col_j] + H[i], slope);  alpha = softmax_j(e_j) (+1e-16 in the
 *   denominator);  out[i] = sum_j alpha_j H[col_j].
 * One pass over the in-neighbours with an online softmax: every H row is read once per edge.
 * Feature tables may have more rows than row_end (multi-GPU: local rows then halo rows).
 * Optional fused node-wise epilogue of KTGNN_no_complement.forward (:425-430, eval mode):
 *   out = relu?(out * ep_scale[c] + ep_shift[c])  (BatchNorm1d eval affine; NULL = identity); ep_relu: 0 none, 1 ReLU,
 *   2 = log_softmax over the D classes of every head (:435; interleaved narrow heads only: heads in {2,3}, D <= 4,
 *   ldh == ldo == 4; like ReLU it applies when a row is finished, i.e. not in part = 1).
 * alpha_opt ([E'] in CSR order) is optional (tests / backward).
 * Two-part rows (multi-GPU overlap): part = 1 visits a row's first edge list and parks the online-softmax state
 * ((max, sum) in state_ms_opt [rows][2], the raw accumulator in out); part = 2 resumes from it over a second
 * edge list (another rowptr/col pair) and finishes the row.  part = 0 is the ordinary single launch; part = 3 (narrow
 * interleaved heads only) is a single launch that also leaves the finished rows' (max, sum) in state_ms_opt [rows][heads][2]
 * for bgnn_adaptedconv_aggregate_heads_bwd_f32.  In part = 1 the
 * rows [row_begin, park_begin) have no second part and are finished right away (epilogue, colsum), rows
 * [park_begin, row_end) are parked: one launch serves a rank's interior and boundary rows (park_begin = row_begin
 * parks every row).
 * heads > 1 evaluates several convs that share the graph in ONE pass (KTGNN.py:432-434: clf_base / clf_target /
 * clf_target-hat): their tables are interleaved row-wise ([rows, heads*ldh], head h of node r at (r*heads+h)*ldh),
 * a_t2s / a_s2t are [heads][D], out is [rows, heads*ldo] likewise; the in-neighbour ids are read once for all heads.
 * colsum_opt ([2*ldo+2] doubles, accumulated: caller zero-fills) receives the per-domain column sums and node
 * counts of the finished rows -- the `bgnn_domain_sums_f64` of the NEXT conv's input for free (heads == 1 only).
 * tile_queue_opt (8 x uint32 scratch, zeroed here on `stream`) switches the persistent blocks from static tile
 * striding to per-XCD dynamic tile counters (keeps the rows in flight inside the XCD's L2).
 * D <= 256; ldh % 4 == 0, ldo % 4 == 0, 16-B aligned tables; pad columns must be zero.        */
int bgnn_adaptedconv_aggregate_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
                                   const float* a_t2s, const float* a_s2t,
                                   const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                   int64_t row_begin, int64_t row_end, int32_t D, float negative_slope,
                                   float* out, int64_t ldo, float* alpha_opt,
                                   const float* ep_scale_opt, const float* ep_shift_opt, int ep_relu,
                                   float* state_ms_opt, int part, int64_t park_begin, int32_t heads, double* colsum_opt,
                                   uint32_t* tile_queue_opt, void* stream);

/* bgnn_adaptedconv_aggregate_f32 with one more promise from the caller (ABI 113): both tables have `table_rows` rows and every id in
 * `col` is below it (table_rows >= row_end, else BGNN_E_SHAPE).  Results are bit-identical; the promise lets the plain wide launch
 * (heads = 1, part = 0, no alpha, D > 32) address neighbour rows by 32-bit offsets inside one window that holds both tables, when
 * that window is below 4 GB and table_rows <= 2^24 (agg_wide_fast_kernel: C4 hidden conv 1.06 -> see DESIGN 4.1); every other
 * shape runs exactly what bgnn_adaptedconv_aggregate_f32 runs.  Ids >= table_rows read outside the tables (undefined), as they
 * do there.  gather_hint: 0 = unknown, 1 = neighbouring destination rows share neighbours (the gathers live off the L2s), 2 = no
 * neighbour reuse (HBM-bound gathers): only chooses how many blocks of the fast kernel stay resident per CU (results unchanged;
 * DstCSR.gather_hint() measures it once per graph).  BGNN_AGG_FAST=0 in the environment keeps the general kernel. */
int bgnn_adaptedconv_aggregate_bounded_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
                                           const float* a_t2s, const float* a_s2t,
                                           const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                           int64_t row_begin, int64_t row_end, int32_t D, float negative_slope,
                                           float* out, int64_t ldo, float* alpha_opt,
                                           const float* ep_scale_opt, const float* ep_shift_opt, int ep_relu,
                                           float* state_ms_opt, int part, int64_t park_begin, int32_t heads, double* colsum_opt,
                                           uint32_t* tile_queue_opt, int64_t table_rows, int32_t gather_hint, void* stream);

/* The same aggregation for graphs with HUB rows.  A destination row is walked by one lane group, so a row with hundreds of
 * in-edges (the 581 source nodes of the Twitter_Graph stand-in have ~750) is a chain of dependent gather steps that outlives
 * all other rows.  Rows with >= hub_threshold in-edges (hub_rows [n_hubs], ascending) are skipped by the main launch, walked as
 * segments -- seg_bounds [2 n_segments] = (begin, end) offsets into `col`, seg_node [n_segments] = the segment's row,
 * hub_seg_ptr [n_hubs + 1] = the segments of each hub -- whose online-softmax states are parked like two-part rows, and finished
 * by a merge launch (normalise, epilogue, colsum; part-3 state for the heads backward when state_ms_opt is given; alpha_opt [E'],
 * heads = 1: the attention coefficients in CSR order for the backward).  All N rows,
 * wide rows (D > 32, heads = 1) or interleaved narrow heads (heads = 2 | 3, ldh = ldo = 4); other shapes: BGNN_E_SHAPE (use
 * bgnn_adaptedconv_aggregate_f32).  n_hubs = 0 is the plain launch.  ws: bgnn_aggregate_hub_workspace_bytes(n_segments, heads, ldo). */
size_t bgnn_aggregate_hub_workspace_bytes(int64_t n_segments, int32_t heads, int64_t ldo);
int bgnn_adaptedconv_aggregate_hub_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
                                       const float* a_t2s, const float* a_s2t,
                                       const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                       int64_t N, int32_t D, float negative_slope, float* out, int64_t ldo,
                                       const float* ep_scale_opt, const float* ep_shift_opt, int ep_relu,
                                       float* state_ms_opt, int32_t heads, double* colsum_opt,
                                       uint32_t* tile_queue_opt, int32_t hub_threshold, const int32_t* hub_rows,
                                       int64_t n_hubs, const int32_t* hub_seg_ptr, const int32_t* seg_bounds,
                                       const int32_t* seg_node, int64_t n_segments, float* alpha_opt,
                                       void* ws, size_t ws_bytes, void* stream);

/*  backward of the aggregation above -- what autograd computes through
 * models/KTGNN.py:292-305 when main_graph_knowledge_transfer.py:39-68 calls loss.backward().
 * Inputs: the forward's tables, `out`, `alpha` (CSR order) and grad_out = dL/dout.  Outputs are
 * ACCUMULATED into (caller zero-fills): dh_t2s / dh_s2t [rows of the tables, ldh] and da_t2s / da_s2t [D].
 * Source-side sums use hardware fp32 atomics (order-dependent in the last bits).               */
int bgnn_adaptedconv_aggregate_bwd_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
                                       const float* a_t2s, const float* a_s2t,
                                       const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                       int64_t row_begin, int64_t row_end, int32_t D, float negative_slope,
                                       const float* out, int64_t ldo, const float* alpha,
                                       const float* grad_out, int64_t ldg,
                                       float* dh_t2s, float* dh_s2t, float* da_t2s, float* da_s2t,
                                       void* stream);

/* Atomic-free ("pull") form of the same backward for D <= 128: the source-side sums are gathered over a by-source
 * view of the edges (t_rowptr [N+1]; t_eid [E'] = position of the edge in the by-destination order; t_dst [E'] = its
 * destination) instead of scattered with float atomics; every dH row is written exactly once (no zero-fill needed,
 * deterministic).  All N rows are visited; ws: bgnn_aggregate_bwd_pull_workspace_bytes(N, E', ldh). */
size_t bgnn_aggregate_bwd_pull_workspace_bytes(int64_t N, int64_t E, int64_t ldh);
int bgnn_adaptedconv_aggregate_bwd_pull_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
                                            const float* a_t2s, const float* a_s2t,
                                            const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                            const int32_t* t_rowptr, const int32_t* t_eid, const int32_t* t_dst,
                                            int64_t N, int64_t E, int32_t D, float negative_slope,
                                            const float* out, int64_t ldo, const float* alpha,
                                            const float* grad_out, int64_t ldg,
                                            float* dh_t2s, float* dh_s2t, float* da_t2s, float* da_s2t,
                                            void* ws, size_t ws_bytes, void* stream);

/* The same pull-form backward for graphs with hub rows (wide rows only: not the ldh = 4 narrow form).  Pass A walks destinations,
 * pass B sources: a destination with >= hub_threshold in-edges / a source with that many out-edges is skipped as a row and walked
 * as segments -- d_* tables over `rowptr` / `col`, s_* tables over the by-source arrays, both in the layout of
 * bgnn_adaptedconv_aggregate_hub_f32 (hub_rows, hub_seg_ptr, seg_bounds = (begin, end) pairs, seg_node) -- that ride behind the real
 * rows of the same launch and leave partial row sums, merged in a fixed order (deterministic like the plain form).
 * ws: bgnn_aggregate_bwd_pull_hub_workspace_bytes(N, E', ldh, d_n_segments, s_n_segments). */
size_t bgnn_aggregate_bwd_pull_hub_workspace_bytes(int64_t N, int64_t E, int64_t ldh, int64_t d_segments, int64_t s_segments);
int bgnn_adaptedconv_aggregate_bwd_pull_hub_f32(const float* h_t2s, const float* h_s2t, int64_t ldh,
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
                                                void* ws, size_t ws_bytes, void* stream);

/* Pull-form backward for `heads` (2 or 3) interleaved narrow convs evaluated together (KT-GNN's classifier stage under
 * autograd: clf_base(x), clf_target(x), clf_target(T(x)), KTGNN.py:432-435, share the graph): tables / out / grad_out / dH are
 * [N][heads][4], a_* and da_* [heads][D] (da accumulated: caller zero-fills), D <= 4.  `state_ms` [N][heads][2] is the finished
 * rows' softmax state (max, sum) that bgnn_adaptedconv_aggregate_f32 leaves with part = 3; alpha is rebuilt from it, no per-edge
 * array is kept by the forward.  log_softmax != 0: `out` holds the fused log-probabilities (ep_relu = 2) and grad_out is
 * dL/dlogp -- the row-local adjoint is applied first.  Every dH row is written exactly once (deterministic).
 * ws: bgnn_aggregate_heads_bwd_workspace_bytes(N, E', heads). */
size_t bgnn_aggregate_heads_bwd_workspace_bytes(int64_t N, int64_t E, int32_t heads);
int bgnn_adaptedconv_aggregate_heads_bwd_f32(const float* h_t2s, const float* h_s2t, const float* a_t2s, const float* a_s2t,
                                             const int32_t* rowptr, const int32_t* col, const uint8_t* mask,
                                             const int32_t* t_rowptr, const int32_t* t_eid, const int32_t* t_dst,
                                             int64_t N, int64_t E, int32_t D, int32_t heads, float negative_slope,
                                             const float* out, const float* state_ms, const float* grad_out,
                                             int log_softmax, float* dh_t2s, float* dh_s2t, float* da_t2s,
                                             float* da_s2t, void* ws, size_t ws_bytes, void* stream);
/* ... and for graphs with hub rows (tables as in bgnn_adaptedconv_aggregate_bwd_pull_hub_f32; partial rows of heads * 4 floats). */
size_t bgnn_aggregate_heads_bwd_hub_workspace_bytes(int64_t N, int64_t E, int32_t heads, int64_t d_segments, int64_t s_segments);
int bgnn_adaptedconv_aggregate_heads_bwd_hub_f32(const float* h_t2s, const float* h_s2t, const float* a_t2s, const float* a_s2t,
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
                                                 void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a2,a3,a5,a6,a7) kNN bridge: pair scoring + per-query top-k.
 *     main_bridged_graph.py:45-67 / :90-111 (batched loop), models/models.py:124-130,:944-954
 *     (scorers), :265-282 (pair_enumeration -- never materialised here), Tensor.topk :60,:104.
 * Selection rule (declared; replaces torch's unspecified tie order): the k candidates with the
 * largest CANONICAL score, ties -> lower candidate index; rows come out sorted by that rule.
 * CANONICAL score = fp64 accumulation in feature-index order of the exact fp32 products
 * (oracle/oracle_c.c orc_cosine_topk / orc_mlp_topk).  Cosine: a cascade of filters with a proof per stage --
 * (0) every fp32 embedding is split exactly into bf16 pieces and the residual norms give a rigorous bound eps on
 * |approximate - exact| score; (1) a FAST pass streams all candidates on the bf16 matrix cores with one piece per
 * candidate, keeping per query the candidates within 2 eps of the running k-th best; (2) the survivors are re-scored
 * in canonical arithmetic and the row is proven (kth_exact > best excluded approximate score + eps) or queued;
 * (3) queued rows go through a PRECISE pass (three piece products, eps ~ 5e-5) and the same proof; (4) rows that are
 * still unproven (exact ties across the boundary) are re-done exhaustively.  mlp: fp32 VALU pass + stages (2), (4).
 * n_fallback_opt (optional, int32[2] on the device): [0] = rows re-done exhaustively, [1] = rows sent to the precise pass.
 * val_out = sigmoid(score) as fp32 if apply_sigmoid (models.py:129,:953) else the fp32 score.
 * k <= 56.  q* must already be L2-normalised by bgnn_l2_normalize_rows_f32 (cosine).          */
int bgnn_l2_normalize_rows_f32(const float* q, int64_t n, int32_t d, float eps, float* out, void* stream);
size_t bgnn_topk_workspace_bytes(int64_t Nq, int64_t Nc, int32_t k);
int bgnn_cosine_topk_f32(const float* qn_query, const float* qn_cand, int64_t Nq, int64_t Nc,
                         int32_t d, int32_t k, int apply_sigmoid,
                         int64_t* idx_out /*[Nq,k]*/, float* val_out /*[Nq,k]*/,
                         int32_t* n_fallback_opt, void* ws, size_t ws_bytes, void* stream);
int bgnn_mlp_pair_topk_f32(const float* A_cand /*[Nc,H]*/, const float* B_query /*[Nq,H]*/,
                           const float* bn_scale, const float* bn_shift, const float* w2, float b2,
                           int64_t Nq, int64_t Nc, int32_t H, int32_t k, int apply_sigmoid,
                           int64_t* idx_out, float* val_out, int32_t* n_fallback_opt,
                           void* ws, size_t ws_bytes, void* stream);

/* (a2/a3 tail) edge list from the top-k table: edge (from = idx[q,t] + cand_base, to = q + query_base),
 * main_bridged_graph.py:61-63,:105-107.  edge_index_out is [2, Nq*k].                          */
int bgnn_topk_edges_i64(const int64_t* idx, int64_t Nq, int32_t k, int64_t cand_base, int64_t query_base,
                        int64_t* edge_index_out, void* stream);
/* The same table as a COALESCED edge list (what main_bridged_graph.py:75 / :113 pass on: coalesce(edge_index_added)) in one
 * call: the k candidates of a query are distinct (0 <= idx < Nc, k <= Nc), so the list has no duplicates and coalescing is a
 * stable sort by candidate id of the query-major pairs -- 32-bit pairs over ceil(log2 Nc) bits, exactly Nq * k edges out, no
 * device-to-host read.  Equal to bgnn_coalesce_i64 applied to bgnn_topk_edges_i64's output.  ws:
 * bgnn_topk_edges_coalesced_workspace_bytes(Nq, k). */
size_t bgnn_topk_edges_coalesced_workspace_bytes(int64_t Nq, int32_t k);
int bgnn_topk_edges_coalesced_i64(const int64_t* idx, int64_t Nq, int32_t k, int64_t Nc, int64_t cand_base, int64_t query_base,
                                  int64_t* edge_index_out, void* ws, size_t ws_bytes, void* stream);

/* Packs table rows for a halo send list (bridged_gnn_amd/dist.py; the reference is single-device):
 * dst[r, :row_floats] = src[idx[r], :row_floats].  row_floats % 4 == 0, 16-B aligned tables with ld % 4 == 0; indices
 * are clamped into [0, src_rows).  Narrow rows (the classifier stage's 48-byte rows) are what the library gather is slow at. */
int bgnn_gather_rows_f32(const float* src, int64_t src_rows, int64_t ld_src, const int64_t* idx, int64_t n,
                         int32_t row_floats, float* dst, int64_t ld_dst, void* stream);

/* (a8) torch_geometric.utils.coalesce -- call sites main_bridged_graph.py:75,:113,:193.
 * key = row*num_nodes + col, ascending, duplicates dropped.  In place on [2,E] (row stride E);
 * the first *E_out_dev columns of each row are valid afterwards.                               */
size_t bgnn_coalesce_workspace_bytes(int64_t E);
int bgnn_coalesce_i64(int64_t* edge_index, int64_t E, int64_t num_nodes, int64_t* E_out_dev,
                      void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BGNN_H_ */
