// kNN bridge construction on gfx950: pair scoring + per-query top-k, never materialising the
// pair list (reference: Bridged-GNN/main_bridged_graph.py:45-67 / :90-111 batched loops,
// pair_enumeration models/models.py:265-282, scorers :124-130 (cosine) and :944-954 (mlp),
// Tensor.topk call sites main_bridged_graph.py:60,:104).
//
// A cascade of filters with a proof at every stage (see include/bgnn.h for the contract):
//   0. split: every fp32 embedding is split exactly into bf16 pieces hi + mid (+ rest); the largest L2 norms of the
//      residuals are measured, which gives a RIGOROUS bound eps on |approximate score - exact score| for each product set;
//   1. FAST pass: stream every candidate against a block of queries on the bf16 matrix cores with ONE piece per candidate
//      (cand_hi . (query_hi [+ query_mid])), keep per query the candidates whose approximate score lies within 2 eps of the
//      running k-th best (a sorted per-query buffer in LDS, threshold register per lane);
//   2. refine: re-score the survivors in CANONICAL arithmetic (fp64, feature-index order), rank by (score desc, index asc)
//      and prove  kth_exact > (best score any excluded candidate can have) + eps ; unproven rows are queued;
//   3. PRECISE pass on the queued rows only: the same kernel with three piece products (eps ~ 5e-5) + refine;
//   4. rows still unproven (exact ties across the boundary) are re-done exhaustively in canonical arithmetic.
// Index results are therefore bit-identical to oracle/oracle_c.c orc_cosine_topk / orc_mlp_topk.
#include <cstdlib>
#include "bgnn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int QPW = 32;            // queries per wave (one 32-wide MFMA column block)
constexpr int MT = 32;             // candidates per MFMA tile (pass 1 stages TPI of them per barrier)
constexpr int MAX_SLOTS = 8;       // shortlists per query (blocks whose tile range touches one query block)
constexpr int LDS_BYTES = 160 * 1024;
#ifndef KNN_SCHED_NUM
#define KNN_SCHED_NUM 3            // scheduled compactions at stream positions growing by (1 + NUM/8)
#endif
#ifndef KNN_SCHED_FREE
#define KNN_SCHED_FREE 26          // ... of the buffers with fewer free slots than this
#endif

// ---- orderable score bits: larger uint = larger float ------------------------------------------
__device__ __forceinline__ uint32_t ord_f32(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
constexpr uint32_t ORD_EMPTY = 0u;      // below every real score (ord(-inf) = 0x007FFFFF)

// ---- error bounds ------------------------------------------------------------------------------
// Residual maxima written by the split kernels (ordered-uint atomicMax): [0] = max_c |c - c_hi|_2, [1] = max_c |c - c_hi - c_mid|_2,
// [2] / [3] the same for the queries.  For unit vectors a, b (|a|, |b| <= 1 + 2^-20 after the fp32 normalisation):
//   1 product  (ah.bh)                : a.b - ah.bh         = ra.b + ah.rb                 -> |.| <= Ra (1+g) + (1+g+Ra) Rb
//   2 products (ah.bh + ah.bm)        : a.b - ah.(bh+bm)    = ra.b + ah.rrb                -> |.| <= Ra (1+g) + (1+g+Ra) Rbb
//   3 products (+ am.bh)              : a.b - (..)          = am.bm + rra.b + (ah+am).rrb  -> |.| <= Ra Rb (1+u)^2.. + Raa (1+g) + (1+g+Raa) Rbb
// (Cauchy-Schwarz on each term) plus the fp32 accumulation inside the MFMAs, counted as truncating: terms * 2^-23 * 1.01.
struct EpsSrc { const uint32_t* mx; int d; };
__device__ __forceinline__ float knn_eps(const EpsSrc& e, int nprod) {
  const float g = 1e-6f, infl = 1.002f;            // norm slack; the residual norms were summed in fp32
  const float Ra = unord_f32(e.mx[0]) * infl, Raa = unord_f32(e.mx[1]) * infl;
  const float Rb = unord_f32(e.mx[2]) * infl, Rbb = unord_f32(e.mx[3]) * infl;
  float eps;
  if (nprod == 1) eps = Ra * (1.f + g) + (1.f + g + Ra) * Rb;
  else if (nprod == 2) eps = Ra * (1.f + g) + (1.f + g + Ra) * Rbb;
  else eps = Ra * Rb * 1.02f + Raa * (1.f + g) + (1.f + g + Raa) * Rbb;
  eps += (float)(e.d * nprod) * 1.2e-7f * 1.01f;
  return eps * 1.001f;
}

extern __shared__ __attribute__((aligned(16))) unsigned char knn_smem[];     // the dynamic LDS of every kernel of this file

// wave-wide maximum of an unsigned value through DPP / swizzle moves (no LDS bpermute round trips)
__device__ __forceinline__ uint32_t bgnn_wave_max_u32(uint32_t x) {
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true));     // lane ^ 1
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true));     // lane ^ 2
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true));    // row_half_mirror
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true));    // row_mirror
  x = mx(x, (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x401F));                      // lane ^ 16
  return mx(x, (uint32_t)__shfl_xor((int)x, 32));
}

// ---- per-wave shortlist state in LDS -----------------------------------------------------------
// Per query: a buffer of CAP (ordered score, candidate) entries, a fill count and the admission threshold tau.  A score
// that beats the lane's threshold register is inserted DIRECTLY: one returning LDS atomic on the query's count hands out
// the slot (the two lanes that hold a query may insert at once), two stores fill it.  A full buffer is COMPACTED: the
// k-th best score is bracketed by a bisection on the ordered score bits (ballot + popcount per step: scalar work, no
// sorting network, no LDS traffic), the threshold becomes  max(old, k-th best - margin)  -- everything that can still
// matter for a proof with error bound eps = margin / 2 -- or a KP-th-best bracket when more than KP entries lie above it,
// and the kept entries are packed to the front by ballot prefix (their order is irrelevant).
// Invariant: every candidate that was ever refused or dropped has an approximate score <= tau (tau never decreases).
template <int CAPV, int KPV>
struct WaveTopK {
  static constexpr int CAP = CAPV, KP = KPV, EPL = (CAPV + 63) / 64, HC = CAPV / 2;
  static_assert(CAPV <= 128 && CAPV % 2 == 0, "one or two buffer entries per lane");
  static_assert(KPV < CAPV && (KPV + 1) / 2 < CAPV / 2, "slack between two compactions in both halves");
  static constexpr size_t BYTES = 2 * sizeof(uint32_t) * QPW * CAPV + sizeof(int) * 2 * QPW + sizeof(float) * QPW;
  // the state is addressed as an OFFSET into the kernel's dynamic LDS (not as generic pointers): the accessors below
  // keep the address space visible to the compiler across the noinline helpers (ds_* instead of flat_* accesses)
  unsigned base;                         // byte offset of this wave's state in knn_smem
  int k;                                 // wanted neighbours
  float margin_abs, margin_rel;          // margin(a) = margin_abs + margin_rel * |a|
  __device__ __forceinline__ uint32_t* skey() const { return reinterpret_cast<uint32_t*>(knn_smem + base); }            // [QPW][2][HC] score bits
  __device__ __forceinline__ uint32_t* sidx() const { return skey() + QPW * CAP; }                                      // [QPW][2][HC] candidate
  __device__ __forceinline__ int* cnt() const { return reinterpret_cast<int*>(sidx() + QPW * CAP); }                    // [2][QPW]: mailbox, index = lane
  __device__ __forceinline__ float* tau() const { return reinterpret_cast<float*>(cnt() + 2 * QPW); }                   // [QPW]

  __device__ __forceinline__ void carve(unsigned byte_offset) { base = byte_offset; }
  __device__ __forceinline__ void init(int lane, float tau0) {          // tau0: the lane's query (lane & 31)
    cnt()[lane] = 0;
    if (lane < QPW) tau()[lane] = tau0;
  }
  // number of buffer entries of this wave's lanes (EPL per lane) whose ordered score exceeds the scalar pivot
  __device__ __forceinline__ static int count_above(const uint32_t (&s)[EPL], uint32_t pivot) {
    int c = 0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) c += __popcll(__ballot(s[e] > pivot));
    return c;
  }
  // Compaction of query q's buffer (whole wave, uniform q); the halves' fill counts are in the mailbox cnt[h*32 + q].
  __device__ __forceinline__ void compact(int q, int lane) {
    const int cA = min(cnt()[q], HC), cB = min(cnt()[QPW + q], HC);
    const int n = cA + cB;
    const float tau_old = tau()[q];
    uint32_t s[EPL], x[EPL];
    uint32_t smax = ORD_EMPTY;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      // element (e, lane) -> (half, slot): one entry per lane walks half 0 then half 1; two entries per lane = the two halves
      const int half = EPL == 1 ? (lane >= HC ? 1 : 0) : e;
      const int slot = EPL == 1 ? lane - half * HC : lane;
      const bool valid = slot < (half ? cB : cA) && (EPL == 2 || lane < CAP);
      const int src = q * CAP + half * HC + slot;
      s[e] = valid ? ord_f32(__uint_as_float(skey()[src])) : ORD_EMPTY;              // empty slots carry 0 < ord(-inf)
      x[e] = valid ? sidx()[src] : 0u;
      smax = s[e] > smax ? s[e] : smax;
    }
    // bracket the `want`-th best: count(s > lo) >= want > count(s > hi)
    auto bracket = [&](int want, uint32_t lo, uint32_t hi) {
      for (int it = 0; it < 32 && hi - lo > 4096u; ++it) {     // 2^12 ordered steps of an fp32: < 2^-11 relative, far inside the margin
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const bool ge = count_above(s, mid) >= want;
        lo = ge ? mid : lo;
        hi = ge ? hi : mid;
      }
      return lo;                                               // the want-th best lies in (lo, hi]
    };
    const uint32_t hi0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)bgnn_wave_max_u32(smax));
    const uint32_t lo0 = ord_f32(tau_old);                     // every buffered entry was admitted above tau_old
    float tm = -INFINITY;
    if (n >= k) {
      const float a = unord_f32(bracket(k, lo0, hi0));         // a lower bound of the k-th best, tight to 2^-11
      tm = a - (margin_abs + margin_rel * fabsf(a));
    }
    tm = fmaxf(tm, tau_old);
    uint32_t tmo = ord_f32(tm);
    int cm = count_above(s, tmo);
    if (cm > KP) {                                  // wave-uniform, rare: more than KP candidates inside the margin
      // raise the threshold to the smallest bracket end with at most KP entries above it
      uint32_t lo = tmo, hi = hi0;                  // count(s > lo) > KP >= count(s > hi) = 0
      for (int it = 0; it < 40 && hi - lo > 1u; ++it) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const bool gt = count_above(s, mid) > KP;
        lo = gt ? mid : lo;
        hi = gt ? hi : mid;
      }
      tmo = hi;
      tm = unord_f32(tmo);
      cm = count_above(s, tmo);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);             // every read of the old layout is done before the packed stores
    int before = 0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const bool kp = s[e] > tmo;
      const unsigned long long b = __ballot(kp);
      if (kp) {                                     // packed position p -> half p & 1, slot p >> 1: both halves keep their slack
        const int pos = before + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
        const int dst = q * CAP + (pos & 1) * HC + (pos >> 1);
        skey()[dst] = __float_as_uint(unord_f32(s[e]));
        sidx()[dst] = x[e];
      }
      before += __popcll(b);
    }
    if (lane == 0) { cnt()[q] = (cm + 1) >> 1; cnt()[QPW + q] = cm >> 1; tau()[q] = tm; }
  }
};

// compaction of the queries in qmask; the caller has published every lane's fill count to the mailbox (cnt[lane]) and
// reloads count and threshold afterwards
template <class TK>
__device__ __noinline__ void compact_rows(TK tk, unsigned int qmask, int lane) {
  while (qmask) {
    const int qq = __ffs(qmask) - 1;
    qmask &= qmask - 1;
    tk.compact(qq, lane);
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the new counts / tau are visible to the wave's next LDS reads
}

// Lanes in `pend` hold an arrival (score bits so, candidate cand) that found their half of the query's buffer full:
// compact the buffers involved and insert what still beats the raised thresholds.  Rare path.  -> the lane's new count.
template <class TK>
__device__ __noinline__ int insert_overflow(TK tk, bool pend, uint32_t so, uint32_t cand, int mycnt, int lane) {
  constexpr int CAP = TK::CAP, HC = TK::HC;
  const int q = lane & 31, h = lane >> 5;
  tk.cnt()[lane] = mycnt;                           // publish the fill counts
  unsigned long long over = __ballot(pend);
  unsigned int qmask = 0;                           // distinct queries among the pending lanes
  while (over) {
    const int lead = __ffsll((long long)over) - 1;
    qmask |= 1u << (lead & 31);
    over &= over - 1;
  }
  compact_rows(tk, qmask, lane);
  mycnt = tk.cnt()[lane];
  if (pend && __uint_as_float(so) > tk.tau()[q]) {  // after a compaction every half holds at most (KP + 1) / 2 < HC entries
    tk.skey()[q * CAP + h * HC + mycnt] = so;
    tk.sidx()[q * CAP + h * HC + mycnt] = cand;
    ++mycnt;
  }
  return mycnt;
}

// Offer the 16 scores a lane holds (one query q = lane & 31, candidates cbase + cand(r, h)).  A register nobody beats the
// threshold with costs a compare and a scalar branch.  An arrival goes into the lane's OWN half of the query's buffer
// (lanes (q, 0) and (q, 1) each own HC slots and keep their fill count in a register): two LDS stores, no atomic and no
// wait for its return (the shared-count form spent ~1000 cycles per 64 candidates waiting for ds_add_rtn).
template <class TK>
__device__ __forceinline__ void offer_tile(TK& tk, f32x16 acc, int cbase, int64_t Nc, int cand_lo, int lane, float& tau, int& mycnt) {
  constexpr int CAP = TK::CAP, HC = TK::HC;
  const int q = lane & 31, h = lane >> 5;
  if (cbase + MT > Nc) {                            // last (partial) tile only: mask candidates >= Nc
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (cbase + (r & 3) + 8 * (r >> 2) + 4 * h >= Nc) acc[r] = -INFINITY;
  }
  float mx = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
  for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, acc[r]), acc[r + 1]);
  mx = fmaxf(mx, acc[15]);
#ifdef KNN_X_NOOFFER          /* ablation (tools/exp_libs): the screen only, nothing is ever inserted */
  if (mx > 1e30f) mycnt = 1;
  return;
#endif
  if (!__any(mx > tau)) return;
  uint32_t* const my_half = tk.skey() + q * CAP + h * HC;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool hit = acc[r] > tau;
    if (__any(hit)) {                               // wave-uniform
      const uint32_t so = __float_as_uint(acc[r]);    // raw bits; the ordered form is made where entries are compared
      const uint32_t cand = (uint32_t)(cand_lo + cbase + (r & 3) + 8 * (r >> 2) + 4 * h);
      const bool fits = hit && mycnt < HC;
      if (fits) { my_half[mycnt] = so; my_half[QPW * CAP + mycnt] = cand; ++mycnt; }
      if (__any(hit && !fits)) {                    // a half is full
        mycnt = insert_overflow(tk, hit && !fits, so, cand, mycnt, lane);
        tau = tk.tau()[q];
      }
    }
  }
}

// scheduled compaction of the fuller buffers (see the pass-1 kernel): publish the counts, compact, reload
template <class TK>
__device__ __forceinline__ void compact_fullish(TK& tk, int lane, float& tau, int& mycnt, int free_slots) {
  tk.cnt()[lane] = mycnt;
  const int other = __shfl_xor(mycnt, 32);
  const int hi = mycnt > other ? mycnt : other;       // a query overflows when ONE of its halves is full
  const unsigned long long fullish = __ballot(lane < QPW && (mycnt + other > TK::CAP - free_slots || hi > TK::HC - free_slots / 3));
  compact_rows(tk, (unsigned int)fullish, lane);
  mycnt = tk.cnt()[lane];
  tau = tk.tau()[lane & 31];
}

// final: every buffer trimmed to its threshold, then the (score, candidate) lists and the thresholds go to the workspace
template <class TK>
__device__ __forceinline__ void emit_shortlists(TK& tk, int lane, int mycnt, int64_t q0, int64_t nq, float* __restrict__ sl_score,
                                                int32_t* __restrict__ sl_idx, float* __restrict__ sl_tau, int slot, int nslots,
                                                uint32_t cand_lo = 0) {
  constexpr int CAP = TK::CAP, KP = TK::KP, EPL = TK::EPL, HC = TK::HC;
  tk.cnt()[lane] = mycnt;
  compact_rows(tk, 0xFFFFFFFFu, lane);
  for (int q = 0; q < QPW; ++q) {
    const int64_t gq = q0 + q;
    if (gq < nq) {
      const int n = tk.cnt()[q] + tk.cnt()[QPW + q];          // packed: entry p sits in half p & 1, slot p >> 1
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int pos = lane + 64 * e;
        if (pos < KP) {
          const int64_t o = (gq * nslots + slot) * KP + pos;
          const int src = q * CAP + (pos & 1) * HC + (pos >> 1);
          const bool own = pos < n && tk.sidx()[src] >= cand_lo;                  // seeds of the head pass stay in its slots
          sl_score[o] = own ? __uint_as_float(tk.skey()[src]) : -INFINITY;
          sl_idx[o] = own ? (int32_t)tk.sidx()[src] : -1;
        }
      }
      if (lane == 0) sl_tau[gq * nslots + slot] = tk.tau()[q];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// stage 0: exact bf16 split of fp32 rows (hi + mid) and the residual maxima for the error bounds
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, int64_t n, int d, __bf16* __restrict__ hi,
                                                         __bf16* __restrict__ mid, uint32_t* __restrict__ mx /*[2] of this operand*/) {
  // d in {32, 64, 128, 256}: a group of d/4 lanes owns a row (float4 per lane); blocks stride over the rows and keep the
  // running maxima in registers: ONE pair of atomics per block (one pair per row = 2 x 10^5 same-address atomics = 1.1 ms)
  __shared__ float red[2][4];
  const int lpr = d / 4;                                   // lanes per row: 8 .. 64 (a row never straddles a wave)
  const int rpb = 256 / lpr;                               // rows per block and step
  const int c4 = threadIdx.x % lpr;
  float M1 = 0.f, M2 = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * rpb + threadIdx.x / lpr; row < n; row += (int64_t)gridDim.x * rpb) {
    const float4 v = *reinterpret_cast<const float4*>(x + row * d + c4 * 4);
    const float vf[4] = {v.x, v.y, v.z, v.w};
    bf16x4 h, m;
    float r1 = 0.f, r2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 hh = (__bf16)vf[e];
      const float a = vf[e] - (float)hh;                   // exact
      const __bf16 mm = (__bf16)a;
      const float b = a - (float)mm;                       // exact
      h[e] = hh; m[e] = mm;
      r1 += a * a; r2 += b * b;
    }
    *reinterpret_cast<bf16x4*>(hi + row * d + c4 * 4) = h;
    *reinterpret_cast<bf16x4*>(mid + row * d + c4 * 4) = m;
    // sums over the lanes of the row (they share its trip count) through DPP moves, not LDS permutes; the order of these
    // fp32 additions is covered by knn_eps's inflation factor
    switch (lpr) {
      case 8: r1 = bgnn::group_sum<8>(r1); r2 = bgnn::group_sum<8>(r2); break;
      case 16: r1 = bgnn::group_sum<16>(r1); r2 = bgnn::group_sum<16>(r2); break;
      case 32: r1 = bgnn::group_sum<32>(r1); r2 = bgnn::group_sum<32>(r2); break;
      default: r1 = bgnn::group_sum<64>(r1); r2 = bgnn::group_sum<64>(r2); break;
    }
    M1 = fmaxf(M1, r1); M2 = fmaxf(M2, r2);
  }
  for (int o = 32; o > 0; o >>= 1) { M1 = fmaxf(M1, __shfl_xor(M1, o)); M2 = fmaxf(M2, __shfl_xor(M2, o)); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = M1; red[1][threadIdx.x >> 6] = M2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    const float b = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    atomicMax(&mx[0], ord_f32(sqrtf(a)));
    atomicMax(&mx[1], ord_f32(sqrtf(b)));
  }
}

// admission threshold for the main pass: the best threshold any slot of the head pass reached (each is a lower bound of
// (k-th best approximate score over ALL candidates) - margin, because it was derived from a subset)
__global__ void tau_from_slots_kernel(const float* __restrict__ sl_tau, int64_t nq, int nslots, int slot_lo, int slot_hi, float* __restrict__ tau_init) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  float t = -INFINITY;
  for (int sl = slot_lo; sl < slot_hi; ++sl) t = fmaxf(t, sl_tau[q * nslots + sl]);
  tau_init[q] = t;
}

__global__ void init_shortlists_kernel(int32_t* __restrict__ sl_idx, int64_t n_idx, float* __restrict__ sl_tau, int64_t n_tau) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_idx) sl_idx[i] = -1;
  if (i < n_tau) sl_tau[i] = -INFINITY;
}

// the same for the first *count rows only (the precise pass's lists are indexed by the position in the list of unproven rows:
// 179 MB of stores for C5 where that list is empty)
__global__ void init_shortlists_rows_kernel(int32_t* __restrict__ sl_idx, int64_t per_row_idx, float* __restrict__ sl_tau,
                                            int64_t per_row_tau, const int32_t* __restrict__ count) {
  const int64_t rows = *count;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    for (int64_t t = threadIdx.x; t < per_row_idx; t += blockDim.x) sl_idx[r * per_row_idx + t] = -1;
    for (int64_t t = threadIdx.x; t < per_row_tau; t += blockDim.x) sl_tau[r * per_row_tau + t] = -INFINITY;
  }
}

// ------------------------------------------------------------------------------------------------
// pass 1, cosine.  scores = cand tile (A operand, 32 x d, through LDS) . query block^T (B operand, d x 32, registers) on
// v_mfma_f32_32x32x16_bf16, so that after the MFMA chain EACH LANE HOLDS 16 SCORES OF ONE QUERY -> one threshold register
// per lane.  NPROD = 1: cand_hi.query_hi ; 2: + cand_hi.query_mid ; 3: + cand_mid.query_hi (the precise variant).
// Work = (query block, candidate tile) pairs, query-major; every persistent block takes one CONTIGUOUS range of `tpb`
// tiles (perfect balance, no tail), emitting one shortlist per query-block segment it touches.  NW waves = NW*32 queries
// per block, one block per CU; the candidate stage is double-buffered (one barrier per tile) and the loop is software
// pipelined: the MFMA chain of tile t is issued BEFORE the shortlist upkeep (VALU) of tile t-1, so a wave's matrix work
// runs beside its own vector work instead of in lock step with its SIMD partner's.
struct P1Params {
  const __bf16 *qh, *qm, *ch, *cm;
  int64_t Nq, Nc;              // queries; candidates OF THIS PASS (rows cand_lo .. cand_lo + Nc - 1 of ch / cm)
  const int32_t* qlist;        // optional: query rows to process (compact list), else rows 0..Nq-1
  const int32_t* nq_dev;       // optional: number of entries of qlist (device side); the plan is then made in the kernel
  int64_t tpb;
  int nslots;                  // shortlists per query in the workspace (all passes together)
  float* sl_score; int32_t* sl_idx; float* sl_tau;
  EpsSrc eps;
  int k;
  int64_t cand_lo;             // first candidate row of this pass
  int slot_base;               // this pass writes slots slot_base ...
  const float* tau_init;       // optional [Nq]: a valid admission threshold per query to start every segment from
  int carry_slots;             // with tau_init: slots 0 .. carry_slots-1 (the head pass) seed every segment's buffers
  int known_tiles;             // candidate tiles the seeds stand for (the head pass's), 0 without seeds
};

template <int D>
__device__ __forceinline__ int bf_swz(int row) {          // 16-byte chunk swizzle of the unpadded bf16 piece rows
  constexpr int NCH = D / 8;
  if constexpr (NCH >= 16) return row & 15;
  else if constexpr (NCH == 8) return (row >> 1) & 7;
  else return (row >> 2) & 3;
}

template <int DK, int NPROD, int CAPV, int KPV, int NW, int TPI>
__global__ __launch_bounds__(64 * NW) void cosine_pass1_kernel(const P1Params p) {
  typedef WaveTopK<CAPV, KPV> TK;
  constexpr int D = DK * 8, NCH = D / 8, NT = 64 * NW, QB = NW * QPW;
  constexpr int CT = MT * TPI;                                             // candidates staged (and scored) per barrier
  constexpr int CPIECES = NPROD == 3 ? 2 : 1, QPIECES = NPROD >= 2 ? 2 : 1;
  constexpr int PIECE_ELEMS = CT * D;                                      // bf16 elements of one staged piece
  constexpr int STAGE_ELEMS = CPIECES * PIECE_ELEMS;
  static_assert((size_t)2 * STAGE_ELEMS * 2 + (size_t)NW * TK::BYTES <= (size_t)LDS_BYTES, "LDS budget");
  __bf16* stage16 = reinterpret_cast<__bf16*>(knn_smem);                   // [2][CPIECES][CT][NCH ^ swizzle][8]  (swizzle by row & 15 ..)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t nq = p.nq_dev ? (int64_t)*p.nq_dev : p.Nq;
  if (nq <= 0) return;
  const int64_t ntiles = (p.Nc + CT - 1) / CT;
  const int64_t nqb = (nq + QB - 1) / QB;
  const int64_t T = nqb * ntiles;
  int64_t tpb = p.tpb;
  if (p.nq_dev) {                                     // plan made on the device: the row count is not known to the host
    tpb = (T + gridDim.x - 1) / gridDim.x;
    const int64_t tpb_min = (ntiles + MAX_SLOTS - 2) / (MAX_SLOTS - 1);
    if (tpb < tpb_min) tpb = tpb_min;
  }
  int64_t t = (int64_t)blockIdx.x * tpb;
  const int64_t t_end = min(T, t + tpb);
  if (t >= t_end) return;
  TK tk;
  tk.carve((unsigned)((size_t)2 * STAGE_ELEMS * 2 + (size_t)wave * TK::BYTES));
  tk.k = p.k;
  tk.margin_abs = 2.f * knn_eps(p.eps, NPROD) + 1e-7f;
  tk.margin_rel = 0.f;

  // staging: CT x NCH 16-byte chunks per piece over NT threads
  constexpr int CH_TILE = CT * NCH;
  constexpr int NLD = (CH_TILE + NT - 1) / NT;
  uint4 pre[CPIECES][NLD];
  auto gload = [&](int64_t ct) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = tid + NT * j;
      const int r = (f / NCH) % CT, c = f % NCH;      // (f >= CH_TILE only for tiny D: harmless duplicate)
      const int64_t gc = ct * CT + r;
      const bool ok = gc < p.Nc;
      pre[0][j] = ok ? *reinterpret_cast<const uint4*>(p.ch + (p.cand_lo + gc) * D + c * 8) : make_uint4(0, 0, 0, 0);
      if constexpr (CPIECES > 1) pre[CPIECES - 1][j] = ok ? *reinterpret_cast<const uint4*>(p.cm + (p.cand_lo + gc) * D + c * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int buf) {
    __bf16* st = stage16 + buf * STAGE_ELEMS;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = tid + NT * j;
      if (CH_TILE % NT == 0 || f < CH_TILE) {
        const int r = f / NCH, c = f % NCH;
        const int off = (r * NCH + (c ^ bf_swz<D>(r))) * 8;
        *reinterpret_cast<uint4*>(&st[off]) = pre[0][j];
        if constexpr (CPIECES > 1) *reinterpret_cast<uint4*>(&st[PIECE_ELEMS + off]) = pre[CPIECES - 1][j];
      }
    }
  };
  const int fr = lane & 31, fh = lane >> 5;
  const int sw = bf_swz<D>(fr);

  while (t < t_end) {                               // block-uniform: one segment per query block touched
    const int64_t qb = t / ntiles, ct0 = t % ntiles;
    const int64_t ct1 = min(ntiles, ct0 + (t_end - t));
    const int slot = p.slot_base + (int)(blockIdx.x - (qb * ntiles) / tpb);
    const int64_t q0 = qb * QB + wave * QPW;
    // B fragments: lane (j = lane & 31, h = lane >> 5) holds q[j][16 kb + 8 h + e], e = 0..7, per piece
    bf16x8 bq[QPIECES][D / 16];
    {
      const int64_t gq = q0 + fr;
      const bool ok = gq < nq;
      const int64_t qrow = ok ? (p.qlist ? (int64_t)p.qlist[gq] : gq) : 0;
#pragma unroll
      for (int kb = 0; kb < D / 16; ++kb) {          // rows beyond nq read row 0 and are zeroed (no pointer select: that became a flat load)
        uint4 v = *reinterpret_cast<const uint4*>(p.qh + qrow * D + kb * 16 + fh * 8);
        if (!ok) v = make_uint4(0, 0, 0, 0);
        bq[0][kb] = __builtin_bit_cast(bf16x8, v);
        if constexpr (QPIECES > 1) {
          uint4 m = *reinterpret_cast<const uint4*>(p.qm + qrow * D + kb * 16 + fh * 8);
          if (!ok) m = make_uint4(0, 0, 0, 0);
          bq[QPIECES - 1][kb] = __builtin_bit_cast(bf16x8, m);
        }
      }
    }
    float tau = -INFINITY;
    if (p.tau_init && q0 + fr < nq) tau = p.tau_init[q0 + fr];
    tk.init(lane, tau);
    int mycnt = 0;                                  // fill count of this lane's half of its query's buffer
    if (p.tau_init) {
      // Seed the buffers with the head pass's entries above the threshold: the segment then CONTINUES the head's stream
      // (its k-th best can only rise from there) instead of re-learning the threshold from its own candidates alone.
      // Seeds are not emitted again (emit_shortlists skips candidates below cand_lo); dropping seeds is always safe.
      for (int qq = 0; qq < QPW; ++qq) {
        const int64_t gq = q0 + qq;
        if (gq >= nq) break;
        const float t0 = p.tau_init[gq];
        int have = 0;
        const int ne = p.carry_slots * KPV;
        for (int e0 = 0; e0 < ne && have < KPV; e0 += 64) {
          const int e = e0 + lane;
          const int64_t o = gq * p.nslots * KPV + e;            // head slots are the first ones of the query's row
          const int32_t c = e < ne ? p.sl_idx[o] : -1;
          const float sc = e < ne ? p.sl_score[o] : -INFINITY;
          const bool kp = c >= 0 && sc > t0;
          const unsigned long long b = __ballot(kp);
          const int pos = have + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
          if (kp && pos < KPV) {                      // packed position p -> half p & 1, slot p >> 1
            const int dst = qq * CAPV + (pos & 1) * TK::HC + (pos >> 1);
            tk.skey()[dst] = __float_as_uint(sc);
            tk.sidx()[dst] = (uint32_t)c;
          }
          have += __popcll(b);
        }
        have = have < KPV ? have : KPV;
        if (lane == 0) { tk.cnt()[qq] = (have + 1) >> 1; tk.cnt()[QPW + qq] = have >> 1; }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      mycnt = tk.cnt()[lane];
    }
    __syncthreads();                                // the previous segment's last tile has been read by every wave
    gload(ct0);
    sstore(0);
    if (ct0 + 1 < ct1) gload(ct0 + 1);
    __syncthreads();
    auto score = [&](int cur, int sub) {
      // ONE accumulation chain: back-to-back 32x32x16 bf16 MFMAs on one accumulator issue at the pipe's full rate
      // (MI355X_MICROARCH.md, cycle constants), so no second chain and no merge adds
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const __bf16* st = stage16 + cur * STAGE_ELEMS;
      bf16x8 ahv[D / 16];                             // every A fragment of the tile is requested before the first MFMA waits
#pragma unroll
      for (int kb = 0; kb < D / 16; ++kb) ahv[kb] = *reinterpret_cast<const bf16x8*>(&st[((sub * MT + fr) * NCH + ((2 * kb + fh) ^ sw)) * 8]);
#pragma unroll
      for (int kb = 0; kb < D / 16; ++kb) {
        const int off = ((sub * MT + fr) * NCH + ((2 * kb + fh) ^ sw)) * 8;
        const bf16x8 ah = ahv[kb];
        f32x16& a = acc;
        // smallest terms first
        if constexpr (NPROD == 3) {
          const bf16x8 am = *reinterpret_cast<const bf16x8*>(&st[PIECE_ELEMS + off]);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bq[0][kb], a, 0, 0, 0);
        }
        if constexpr (NPROD >= 2) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[QPIECES - 1][kb], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[0][kb], a, 0, 0, 0);
      }
      return acc;
    };
    f32x16 accP[TPI];
#pragma unroll
    for (int u = 0; u < TPI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) accP[u][r] = -INFINITY;
    int cur = 0;
    const int nt = (int)(ct1 - ct0);                    // staged units of this segment (32-bit loop arithmetic)
    // SCHEDULED compactions.  A buffer would otherwise be compacted whenever it happens to fill: ~14 times per query at
    // unpredictable tiles, and every such call (~3000 cycles) holds all eight waves at the tile barrier (measured: 1860
    // of 4400 cycles per wave and tile were barrier wait).  The admission rate of a stream at position g is ~ K / g, so
    // compacting the fuller buffers when g crosses 2, 3, 4, 6, 8, 11, ... (x 1.375) keeps the expected arrivals between
    // two compactions at K ln 1.375 ~ 11 < the buffers' slack: the waves compact at the same tiles and stay in step.
    int next_c = p.known_tiles > 0 ? p.known_tiles + (p.known_tiles * KNN_SCHED_NUM >> 3) : 2;
    // The two waves of a SIMD (w and w + NW/2) run the halves of an iteration in OPPOSITE order: one issues its MFMA chain
    // while the other does its shortlist upkeep (VALU / LDS), then they swap -- in lock step behind the tile barrier both
    // would otherwise queue on the matrix pipe first and on the vector issue afterwards.
    const bool mfma_first = wave < NW / 2 || NW == 1;
    for (int i = 0; i < nt; ++i, cur ^= 1) {
      f32x16 accN[TPI];
      if (mfma_first) {                                 // unit ct0+i: LDS reads + MFMA chains
#pragma unroll
        for (int u = 0; u < TPI; ++u) accN[u] = score(cur, u);
      }
      if (i + 1 < nt) sstore(cur ^ 1);                  // unit i+1: registers -> the buffer unit i-1 was read from
      if (i + 2 < nt) gload(ct0 + i + 2);               // unit i+2 flies
      if (i > 0) {                                      // the upkeep of unit i-1
#pragma unroll
        for (int u = 0; u < TPI; ++u)
          offer_tile(tk, accP[u], (int)(ct0 + i - 1) * CT + u * MT, p.Nc, (int)p.cand_lo, lane, tau, mycnt);
        if (p.known_tiles + i >= next_c) {
          // only the buffers that could fill before the next scheduled compaction (the others keep their slack)
          compact_fullish(tk, lane, tau, mycnt, KNN_SCHED_FREE);
          next_c += (next_c * KNN_SCHED_NUM >> 3) > 0 ? (next_c * KNN_SCHED_NUM >> 3) : 1;
        }
      }
      if (!mfma_first) {
#pragma unroll
        for (int u = 0; u < TPI; ++u) accN[u] = score(cur, u);
      }
      __syncthreads();                                  // stage[cur^1] complete; nobody reads stage[cur] any more
#pragma unroll
      for (int u = 0; u < TPI; ++u) accP[u] = accN[u];
    }
#pragma unroll
    for (int u = 0; u < TPI; ++u)
      offer_tile(tk, accP[u], (int)(ct1 - 1) * CT + u * MT, p.Nc, (int)p.cand_lo, lane, tau, mycnt);
    emit_shortlists(tk, lane, mycnt, q0, nq, p.sl_score, p.sl_idx, p.sl_tau, slot, p.nslots, (uint32_t)p.cand_lo);
    t += ct1 - ct0;
  }
}

// ------------------------------------------------------------------------------------------------
// pass 1, mlp (Similar_v2 'mlp' in separable eval form, H = 128): fp32 VALU scoring, same shortlist.
constexpr int MLP_H = 128;
constexpr int MLP_WAVES = 4;
constexpr int MLP_CT = MT;          // candidates per staged tile of the mlp pass
template <int CAPV, int KPV>
__global__ __launch_bounds__(256, 1) void mlp_pass1_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ w2, float b2, int64_t Nq, int64_t Nc, int k,
                                                           float err_abs, float err_rel,
                                                           float* __restrict__ sl_score, int32_t* __restrict__ sl_idx,
                                                           float* __restrict__ sl_tau) {
  typedef WaveTopK<CAPV, KPV> TK;
  constexpr int H = MLP_H, LD = H + 4;
  float* stage = reinterpret_cast<float*>(knn_smem);                                // [MLP_CT][LD]
  float* coefs = stage + MLP_CT * LD;                                                   // scale|shift|w2 [3][H]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t q0 = (int64_t)blockIdx.x * (MLP_WAVES * QPW) + wave * QPW;
  TK tk;
  tk.carve((unsigned)(sizeof(float) * (MLP_CT * LD + 3 * H) + (size_t)wave * TK::BYTES));
  tk.k = k;
  tk.margin_abs = 2.f * err_abs + 1e-7f;
  tk.margin_rel = 2.f * err_rel;
  tk.init(lane, -INFINITY);
  for (int t = tid; t < H; t += 256) { coefs[t] = scale[t]; coefs[H + t] = shift[t]; coefs[2 * H + t] = w2[t]; }
  float4 bqv[H / 4];                 // this lane's query row B[q][:], statically indexed (registers)
  {
    const int64_t gq = q0 + (lane & 31);
#pragma unroll
    for (int h4 = 0; h4 < H / 4; ++h4)
      bqv[h4] = gq < Nq ? *reinterpret_cast<const float4*>(B + gq * H + h4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float tau = -INFINITY;
  int mycnt = 0;
  const int fh = lane >> 5;
  const float4* sc4 = reinterpret_cast<const float4*>(coefs);
  const float4* sh4 = reinterpret_cast<const float4*>(coefs + H);
  const float4* w4 = reinterpret_cast<const float4*>(coefs + 2 * H);
  for (int64_t cb = 0; cb < Nc; cb += MLP_CT) {
    __syncthreads();
    for (int f = tid; f < MLP_CT * (H / 4); f += 256) {
      const int r = f / (H / 4), c4 = f % (H / 4);
      const int64_t gc = cb + r;
      *reinterpret_cast<float4*>(&stage[r * LD + c4 * 4]) =
          gc < Nc ? *reinterpret_cast<const float4*>(A + gc * H + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b2;
#pragma unroll
    for (int h4 = 0; h4 < H / 4; ++h4) {
      const float4 sc = sc4[h4], sh = sh4[h4], w = w4[h4], bb = bqv[h4];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cr = (r & 3) + 8 * (r >> 2) + 4 * fh;
        const float4 a = *reinterpret_cast<const float4*>(&stage[cr * LD + h4 * 4]);
        acc[r] = fmaf(w.x, fmaxf(fmaf(sc.x, a.x + bb.x, sh.x), 0.f), acc[r]);
        acc[r] = fmaf(w.y, fmaxf(fmaf(sc.y, a.y + bb.y, sh.y), 0.f), acc[r]);
        acc[r] = fmaf(w.z, fmaxf(fmaf(sc.z, a.z + bb.z, sh.z), 0.f), acc[r]);
        acc[r] = fmaf(w.w, fmaxf(fmaf(sc.w, a.w + bb.w, sh.w), 0.f), acc[r]);
      }
    }
    offer_tile(tk, acc, (int)cb, Nc, 0, lane, tau, mycnt);
  }
  emit_shortlists(tk, lane, mycnt, q0, Nq, sl_score, sl_idx, sl_tau, 0, 1);
}

// ------------------------------------------------------------------------------------------------
// canonical scores (identical arithmetic to oracle/oracle_c.c)
// a double from lane `J` of every quad (DPP quad_perm broadcast, two 32-bit moves)
template <int J>
__device__ __forceinline__ double quad_bcast_f64(double v) {
  constexpr int CTRL = J | (J << 2) | (J << 4) | (J << 6);
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((unsigned long long)b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, (long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo));
}

struct CosineCanon {
  const float* qq; const float* qc; int d;
  static constexpr bool COOP4 = true;
  // The same sum, evaluated by a QUAD of lanes for one candidate: lane j holds floats 16i + 4j .. + 3 of both rows, so one load
  // instruction of the quad covers one 64-byte line (a lane walking "its" row alone touches a line per lane and instruction, and
  // the refine stage was bound by that request rate).  The running sum visits the elements in index order: within a line lane 0's
  // four terms, then lane 1's, ... -- every lane adds its own terms to the current sum and the quad takes lane jj's result, so
  // the sequence of fp64 additions is exactly operator()'s (bit-identical; d % 16 == 0).  Returns the sum in all four lanes.
  __device__ __forceinline__ double coop4(int64_t q, int64_t c, int j) const {
    const float* a = qq + q * d + 4 * j;
    const float* b = qc + c * d + 4 * j;
    double s = 0.0;
    for (int t = 0; t < d; t += 16) {
      const float4 x = *reinterpret_cast<const float4*>(a + t);
      const float4 y = *reinterpret_cast<const float4*>(b + t);
      const double p0 = (double)x.x * (double)y.x, p1 = (double)x.y * (double)y.y;      // exact
      const double p2 = (double)x.z * (double)y.z, p3 = (double)x.w * (double)y.w;
      double r;
      r = s + p0; r = r + p1; r = r + p2; r = r + p3; s = quad_bcast_f64<0>(r);
      r = s + p0; r = r + p1; r = r + p2; r = r + p3; s = quad_bcast_f64<1>(r);
      r = s + p0; r = r + p1; r = r + p2; r = r + p3; s = quad_bcast_f64<2>(r);
      r = s + p0; r = r + p1; r = r + p2; r = r + p3; s = quad_bcast_f64<3>(r);
    }
    return s;
  }
  __device__ __forceinline__ double operator()(int64_t q, int64_t c) const {
    const float* a = qq + q * d;
    const float* b = qc + c * d;
    double s = 0.0;
    for (int t = 0; t < d; t += 4) {
      const float4 x = *reinterpret_cast<const float4*>(a + t);
      const float4 y = *reinterpret_cast<const float4*>(b + t);
      s = s + (double)x.x * (double)y.x;     // products of fp32 are exact in fp64: fma == mul+add
      s = s + (double)x.y * (double)y.y;
      s = s + (double)x.z * (double)y.z;
      s = s + (double)x.w * (double)y.w;
    }
    return s;
  }
};
struct MlpCanon {
  const float* A; const float* B; const float* scale; const float* shift; const float* w2; float b2; int H;
  static constexpr bool COOP4 = false;
  __device__ __forceinline__ double coop4(int64_t, int64_t, int) const { return 0.0; }
  __device__ __forceinline__ double operator()(int64_t q, int64_t c) const {
    const float* a = A + c * H;
    const float* b = B + q * H;
    double s = 0.0;
    for (int h = 0; h < H; ++h) {
      // fp64 mul/add kept un-fused (non-exact products): matches gcc -ffp-contract=off
      double u = __dadd_rn((double)b[h], (double)a[h]);
      double t = __dadd_rn(__dmul_rn((double)scale[h], u), (double)shift[h]);
      if (t < 0.0) t = 0.0;
      s = __dadd_rn(s, __dmul_rn((double)w2[h], t));
    }
    return __dadd_rn(s, (double)b2);
  }
};

__device__ __forceinline__ float sigmoid_f32(float x) { return 1.f / (1.f + expf(-x)); }

// pass 2: one wave per query.  L = nslots*KP shortlist entries + nslots thresholds.
//   alast = max over the slots' thresholds = the best approximate score any NEVER-LISTED candidate can have;
//   a_k   = the k-th best approximate score among the listed candidates (bracketed by bisection on the ordered bits);
//   T     = max(alast, a_k - margin), margin = 2 eps (+ slack): a candidate with approximate score <= T is strictly
//           dominated by the k candidates at or above a_k (exact >= a_k - eps > T + eps >= its exact score);
//   survivors = listed candidates above T (~ k + a few): exact re-score + rank; proof  kth_exact > T + eps.
struct RefineParams {
  int64_t Nq; const int32_t* qlist; const int32_t* nq_dev;
  int k, L, KP, nslots;
  const float* sl_score; const int32_t* sl_idx; const float* sl_tau;
  EpsSrc eps; int nprod;                 // nprod > 0: bound from the residual maxima; else err_abs + err_rel |T|
  double err_abs, err_rel;
  int apply_sigmoid;
  int64_t* idx_out; float* val_out;
  int32_t* fail_list; int32_t* fail_count;
};
template <class Canon>
__global__ __launch_bounds__(256) void refine_kernel(Canon canon, const RefineParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.L;
  // per wave: exact scores [L] doubles, candidates [L] ints, ordered approximate scores [L] uints
  double* es = reinterpret_cast<double*>(knn_smem) + (size_t)wave * L;
  int32_t* ei = reinterpret_cast<int32_t*>(reinterpret_cast<double*>(knn_smem) + (size_t)4 * L) + (size_t)wave * L;
  uint32_t* ea = reinterpret_cast<uint32_t*>(reinterpret_cast<int32_t*>(reinterpret_cast<double*>(knn_smem) + (size_t)4 * L) + (size_t)4 * L) + (size_t)wave * L;
  const int64_t nq = p.nq_dev ? (int64_t)*p.nq_dev : p.Nq;
  const double eps = p.nprod > 0 ? (double)knn_eps(p.eps, p.nprod) : 0.0;
  for (int64_t qi = (int64_t)blockIdx.x * 4 + wave; qi < nq; qi += (int64_t)gridDim.x * 4) {
    const int64_t q = p.qlist ? (int64_t)p.qlist[qi] : qi;
    float alast = -INFINITY;
    for (int s = lane; s < p.nslots; s += 64) alast = fmaxf(alast, p.sl_tau[qi * p.nslots + s]);
    alast = bgnn::group_max<64>(alast);
    // (1) listed candidates above alast -> LDS (candidate, ordered approximate score)
    int n1 = 0;
    uint32_t amax = ORD_EMPTY;
    for (int e0 = 0; e0 < L; e0 += 64) {
      const int e = e0 + lane;
      const int32_t c = e < L ? p.sl_idx[qi * L + e] : -1;
      const float a = e < L ? p.sl_score[qi * L + e] : -INFINITY;
      const bool in = c >= 0 && a > alast;
      const unsigned long long b = __ballot(in);
      if (in) {
        const int pos = n1 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
        ei[pos] = c;
        ea[pos] = ord_f32(a);
        amax = ord_f32(a) > amax ? ord_f32(a) : amax;
      }
      n1 += __popcll(b);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // (2) bracket the k-th best approximate score: count(> lo) >= k > count(> hi)
    float T = alast;
    if (n1 >= p.k) {
      uint32_t lo = ord_f32(alast), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)bgnn_wave_max_u32(amax));
      auto count_above = [&](uint32_t piv) {
        int c = 0;
        for (int e0 = 0; e0 < n1; e0 += 64) c += __popcll(__ballot(e0 + lane < n1 && ea[e0 + lane] > piv));
        return c;
      };
      for (int it = 0; it < 32 && hi - lo > 64u; ++it) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const bool ge = count_above(mid) >= p.k;
        lo = ge ? mid : lo;
        hi = ge ? hi : mid;
      }
      const float ak = unord_f32(lo);                 // a lower bound of the k-th best approximate score
      const float margin = p.nprod > 0 ? (float)(2.0 * eps) + 1e-7f : (float)(2.0 * (p.err_abs + p.err_rel * fabs((double)ak))) + 1e-7f;
      T = fmaxf(alast, ak - margin);
    }
    // (3) survivors above T: exact canonical scores
    const uint32_t To = ord_f32(T);
    int ns = 0;
    for (int e0 = 0; e0 < n1; e0 += 64) {
      const int e = e0 + lane;
      const bool surv = e < n1 && ea[e] > To;
      const int32_t c = e < n1 ? ei[e] : -1;
      const unsigned long long b = __ballot(surv);
      __builtin_amdgcn_wave_barrier();                // every lane has read ei[e] before the packed writes below reuse the array
      if (surv) {
        const int pos = ns + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
        ei[pos] = c;                                  // pos <= e: packing in place is safe chunk by chunk
        if constexpr (!Canon::COOP4) es[pos] = canon(q, c);
      }
      ns += __popcll(b);
    }
    __builtin_amdgcn_s_waitcnt(0);   // LDS writes above complete before the cross-lane reads below
    __builtin_amdgcn_wave_barrier();
    if constexpr (Canon::COOP4) {
      // 16 survivors per step, a quad of lanes each (Canon::coop4); idle quads re-score survivor 0 (in-bounds, discarded)
      for (int e0 = 0; e0 < ns; e0 += 16) {
        const int e = e0 + (lane >> 2);
        const double sc = canon.coop4(q, (int64_t)ei[e < ns ? e : 0], lane & 3);
        if (e < ns && (lane & 3) == 0) es[e] = sc;
      }
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
    }
    double kth = -INFINITY;
    for (int e = lane; e < ns; e += 64) {
      const double s = es[e];
      const int32_t c = ei[e];
      int rank = 0;
      for (int j = 0; j < ns; ++j) {
        const double sj = es[j];
        const int32_t cj = ei[j];
        rank += (sj > s || (sj == s && cj < c));
      }
      if (rank < p.k) {
        p.idx_out[q * p.k + rank] = c;
        p.val_out[q * p.k + rank] = p.apply_sigmoid ? sigmoid_f32((float)s) : (float)s;
      }
      if (rank == p.k - 1) kth = s;
    }
    double kmax = kth;                 // exactly one lane holds it when >= k survivors exist
    for (int o = 32; o > 0; o >>= 1) {
      const double other = __shfl_xor(kmax, o);
      kmax = other > kmax ? other : kmax;
    }
    const bool have = ns >= p.k;
    const double bound = p.nprod > 0 ? eps : p.err_abs + p.err_rel * fabs((double)T);
    const bool proven = have && ((T == -INFINITY) || kmax > (double)T + bound);
    if (!proven && lane == 0) {
      const int slot = atomicAdd(p.fail_count, 1);
      p.fail_list[slot] = (int32_t)q;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// last stage: exhaustive canonical re-do of queued rows.  One block per queued row (grid-strided).
template <class Canon>
__global__ __launch_bounds__(256) void fallback_kernel(Canon canon, int64_t Nc, int k, int apply_sigmoid,
                                                       const int32_t* __restrict__ fb_list, const int32_t* __restrict__ fb_count,
                                                       double* __restrict__ scratch /*[gridDim.x][Nc]*/,
                                                       int64_t* __restrict__ idx_out, float* __restrict__ val_out) {
  __shared__ double red_s[256];
  __shared__ int32_t red_i[256];
  __shared__ double prev_s;
  __shared__ int32_t prev_i;
  const int n = *fb_count;
  double* sc = scratch + (size_t)blockIdx.x * Nc;
  for (int it = blockIdx.x; it < n; it += gridDim.x) {
    const int64_t q = fb_list[it];
    for (int64_t c = threadIdx.x; c < Nc; c += 256) sc[c] = canon(q, c);
    if (threadIdx.x == 0) { prev_s = INFINITY; prev_i = -1; }
    __syncthreads();
    for (int r = 0; r < k; ++r) {
      const double ps = prev_s;
      const int32_t pi = prev_i;
      double bs = -INFINITY;
      int32_t bi = 0x7FFFFFFF;
      for (int64_t c = threadIdx.x; c < Nc; c += 256) {
        const double s = sc[c];
        // strictly after (ps, pi) in (score desc, index asc) order
        const bool after = (s < ps) || (s == ps && (int32_t)c > pi);
        if (after && (s > bs || (s == bs && (int32_t)c < bi))) { bs = s; bi = (int32_t)c; }
      }
      red_s[threadIdx.x] = bs;
      red_i[threadIdx.x] = bi;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
          const double s2 = red_s[threadIdx.x + o];
          const int32_t i2 = red_i[threadIdx.x + o];
          if (s2 > red_s[threadIdx.x] || (s2 == red_s[threadIdx.x] && i2 < red_i[threadIdx.x])) {
            red_s[threadIdx.x] = s2;
            red_i[threadIdx.x] = i2;
          }
        }
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        const bool ok = red_i[0] != 0x7FFFFFFF;
        idx_out[q * k + r] = ok ? red_i[0] : -1;
        const float v = (float)red_s[0];
        val_out[q * k + r] = ok ? (apply_sigmoid ? sigmoid_f32(v) : v) : -INFINITY;
        prev_s = red_s[0];
        prev_i = ok ? red_i[0] : 0x7FFFFFFF;
      }
      __syncthreads();
    }
  }
}

// canonical: fp64 sum of squares in index order, fp64 sqrt, round to fp32, clamp, IEEE fp32 divide.  The per-row
// arithmetic stays one thread's sequential loop (its order is part of the canonical definition); the block moves its 64
// rows through LDS so that global loads and stores are coalesced (thread-per-row global access ran at 0.9 TB/s).
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ q, int64_t n, int d, float eps,
                                                             float* __restrict__ out, int NORM_ROWS) {
  extern __shared__ float nrm_tile[];               // [NORM_ROWS][d + 1]  (+1: conflict-free column walks)
  const int ld = d + 1, tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * NORM_ROWS;
  const int nrows = (int)(n - r0 < NORM_ROWS ? n - r0 : NORM_ROWS);
  const int total = nrows * d;
  const float* src = q + r0 * d;
  for (int t = tid; t < total; t += 256) nrm_tile[(t / d) * ld + (t % d)] = src[t];
  __syncthreads();
  __shared__ float nrm_len[64];                     // NORM_ROWS <= 64 (host)
  if (tid < nrows) {
    const float* r = nrm_tile + tid * ld;
    double s = 0.0;
    for (int c = 0; c < d; ++c) { const double v = (double)r[c]; s = s + v * v; }
    float nr = (float)sqrt(s);
    if (!(nr > eps)) nr = eps;
    nrm_len[tid] = nr;
  }
  __syncthreads();
  // the divisions are element-wise (no order to keep): all 256 threads, straight into the coalesced store
  float* dst = out + r0 * d;
  for (int t = tid; t < total; t += 256) dst[t] = __fdiv_rn(nrm_tile[(t / d) * ld + (t % d)], nrm_len[t / d]);
}

__global__ void normalize_rows_wide_kernel(const float* __restrict__ q, int64_t n, int d, float eps, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // rows too long for an LDS tile: thread per row
  if (i >= n) return;
  const float* r = q + i * d;
  double s = 0.0;
  for (int c = 0; c < d; ++c) { const double v = (double)r[c]; s = s + v * v; }
  float nr = (float)sqrt(s);
  if (!(nr > eps)) nr = eps;
  for (int c = 0; c < d; ++c) out[i * d + c] = __fdiv_rn(r[c], nr);
}

__global__ void copy_counts_kernel(const int32_t* __restrict__ exhaustive, const int32_t* __restrict__ precise, int32_t* __restrict__ dst) {
  dst[0] = *exhaustive;
  if (precise) dst[1] = *precise;
}

// ------------------------------------------------------------------------------------------------
// host side: geometry and workspace
struct Geom { int cap, kp, tpi; };                  // shortlist geometry by k; MFMA tiles staged per barrier
static Geom geom_fast(int k) { return k <= 20 ? Geom{62, 48, 2} : Geom{128, 112, 1}; }     // 62: leaves LDS for 2 x 64-row stages
static Geom geom_precise(int k) { return k <= 20 ? Geom{64, 56, 1} : Geom{128, 120, 1}; }   // wide windows: only rows the fast pass could not prove come here
static Geom geom_mlp(int k) { return k <= 24 ? Geom{64, 40, 1} : Geom{128, 112, 1}; }
static int max_kp(int k) { return geom_fast(k).kp; }

struct Pass1Plan { int64_t nblocks, tpb; int nslots; };
static Pass1Plan plan_pass1(int64_t Nq, int64_t Nc, int64_t resident_blocks, int qpb, int ct) {
  const int64_t ntiles = (Nc + ct - 1) / ct, nqb = (Nq + qpb - 1) / qpb, T = ntiles * nqb;
  Pass1Plan pl;
  pl.nblocks = T < resident_blocks ? T : resident_blocks;
  if (pl.nblocks < 1) pl.nblocks = 1;
  pl.tpb = (T + pl.nblocks - 1) / pl.nblocks;
  const int64_t tpb_min = (ntiles + MAX_SLOTS - 2) / (MAX_SLOTS - 1);   // keep <= MAX_SLOTS shortlists per query
  if (pl.tpb < tpb_min) pl.tpb = tpb_min;
  pl.nblocks = (T + pl.tpb - 1) / pl.tpb;
  pl.nslots = (int)((ntiles + pl.tpb - 1) / pl.tpb) + 1;     // blocks that can touch one query block
  if (pl.nslots > MAX_SLOTS) pl.nslots = MAX_SLOTS;
  return pl;
}

struct TopkWs {
  float* sl_score; int32_t* sl_idx; float* sl_tau;      // stage 1 shortlists  [Nq][<= 2 MAX_SLOTS][KP]
  float* tau_init;                                      // [Nq] thresholds handed from the head pass to the main pass
  float* sl2_score; int32_t* sl2_idx; float* sl2_tau;   // stage 3 (precise) shortlists, indexed by position in fail1
  int32_t* fail1; int32_t* fail2; int32_t* counts;      // counts[0] = precise rows, counts[1] = exhaustive rows
  uint32_t* mx;                                         // residual maxima [4]
  __bf16 *qh, *qm, *ch, *cm;
  double* scratch;
  int fb_blocks;
};
constexpr int FB_BLOCKS = 64;
static size_t topk_ws_layout(int64_t Nq, int64_t Nc, int k, int d, TopkWs* w, void* ws) {
  char* p = (char*)ws;
  size_t total = 0;
  auto take = [&](size_t bytes) { char* q = p ? p + total : nullptr; total += bgnn_align_up(bytes, 256); return q; };
  const size_t L1 = (size_t)2 * MAX_SLOTS * max_kp(k), L2 = (size_t)MAX_SLOTS * geom_precise(k).kp;      // head + main pass slots
  char* a0 = take(sizeof(float) * Nq * L1);
  char* a1 = take(sizeof(int32_t) * Nq * L1);
  char* a2 = take(sizeof(float) * Nq * 2 * MAX_SLOTS);
  char* a3 = take(sizeof(float) * Nq);
  char* b0 = take(sizeof(float) * Nq * L2);
  char* b1 = take(sizeof(int32_t) * Nq * L2);
  char* b2 = take(sizeof(float) * Nq * MAX_SLOTS);
  char* f1 = take(sizeof(int32_t) * (Nq + 1));
  char* f2 = take(sizeof(int32_t) * (Nq + 1));
  char* cn = take(256);
  char* mx = take(256);
  char* qh = take(sizeof(__bf16) * Nq * d);
  char* qm = take(sizeof(__bf16) * Nq * d);
  char* ch = take(sizeof(__bf16) * Nc * d);
  char* cm = take(sizeof(__bf16) * Nc * d);
  char* sc = take(sizeof(double) * FB_BLOCKS * Nc);
  if (w) {
    w->sl_score = (float*)a0; w->sl_idx = (int32_t*)a1; w->sl_tau = (float*)a2; w->tau_init = (float*)a3;
    w->sl2_score = (float*)b0; w->sl2_idx = (int32_t*)b1; w->sl2_tau = (float*)b2;
    w->fail1 = (int32_t*)f1; w->fail2 = (int32_t*)f2; w->counts = (int32_t*)cn; w->mx = (uint32_t*)mx;
    w->qh = (__bf16*)qh; w->qm = (__bf16*)qm; w->ch = (__bf16*)ch; w->cm = (__bf16*)cm;
    w->scratch = (double*)sc; w->fb_blocks = FB_BLOCKS;
  }
  return total + 256;
}

static int device_cus() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  return prop.multiProcessorCount;
}

// waves per block of a pass-1 instantiation: the most that fit the 160 KB of LDS next to the double-buffered stage
template <int DK, int NPROD, int CAPV, int TPI>
constexpr int pass1_waves() {
  constexpr size_t stage = (size_t)2 * (NPROD == 3 ? 2 : 1) * MT * TPI * DK * 8 * 2;
  constexpr size_t per_wave = sizeof(u64) * QPW * CAPV + 3 * 4 * QPW;
  return (stage + 8 * per_wave <= (size_t)LDS_BYTES) ? 8 : (stage + 4 * per_wave <= (size_t)LDS_BYTES) ? 4 : (stage + 2 * per_wave <= (size_t)LDS_BYTES) ? 2 : 1;
}

template <int DK, int NPROD, int CAPV, int KPV, int TPI>
static int launch_pass1(P1Params p, const Pass1Plan* plan /* nullptr: the kernel plans from the device-side row count */, hipStream_t st) {
  constexpr int NW = pass1_waves<DK, NPROD, CAPV, TPI>();
  constexpr size_t sh = (size_t)2 * (NPROD == 3 ? 2 : 1) * MT * TPI * DK * 8 * 2 + (size_t)NW * WaveTopK<CAPV, KPV>::BYTES;
  auto kern = cosine_pass1_kernel<DK, NPROD, CAPV, KPV, NW, TPI>;
  static int attr_done[BGNN_MAX_DEVICES];
  hipError_t e = bgnn_set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)sh, attr_done);
  if (e != hipSuccess) return (int)e;
  const int cus = device_cus();
  if (cus < 1) return (int)hipErrorInvalidDevice;
  p.tpb = plan ? plan->tpb : 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)(plan ? plan->nblocks : cus)), dim3(64 * NW), sh, st, p);
  BGNN_LAUNCH_CHECK();
  return 0;
}

// dispatch over (embedding width, product set, shortlist geometry); `qpb_out` != nullptr only asks for the query rows per block
template <int DK>
static int pass1_dk(int nprod, int k, const P1Params& p, const Pass1Plan* plan, int* qpb_out, hipStream_t st) {
#define BGNN_P1(NP, CAPV, KPV, TPI)                                                                    \
  do {                                                                                                 \
    if (qpb_out) { *qpb_out = pass1_waves<DK, NP, CAPV, TPI>() * QPW; return 0; }                      \
    return launch_pass1<DK, NP, CAPV, KPV, TPI>(p, plan, st);                                          \
  } while (0)
  if (nprod == 3) { if (geom_precise(k).cap == 64) BGNN_P1(3, 64, 56, 1); else BGNN_P1(3, 128, 120, 1); }
  if (nprod == 2) { if (geom_fast(k).cap == 62) BGNN_P1(2, 62, 48, 2); else BGNN_P1(2, 128, 112, 1); }
  if (geom_fast(k).cap == 62) BGNN_P1(1, 62, 48, 2); else BGNN_P1(1, 128, 112, 1);
#undef BGNN_P1
}
static int pass1_any(int d, int nprod, int k, const P1Params& p, const Pass1Plan* plan, int* qpb_out, hipStream_t st) {
  switch (d) {
    case 32: return pass1_dk<4>(nprod, k, p, plan, qpb_out, st);
    case 64: return pass1_dk<8>(nprod, k, p, plan, qpb_out, st);
    case 128: return pass1_dk<16>(nprod, k, p, plan, qpb_out, st);
    default: return pass1_dk<32>(nprod, k, p, plan, qpb_out, st);
  }
}

template <class Canon>
static int launch_refine(const Canon& canon, const RefineParams& rp, hipStream_t st) {
  int64_t grid = (rp.Nq + 3) / 4;
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  const size_t sh = (size_t)4 * rp.L * (sizeof(double) + sizeof(int32_t) + sizeof(uint32_t));
  hipLaunchKernelGGL((refine_kernel<Canon>), dim3((unsigned)grid), dim3(256), sh, st, canon, rp);
  BGNN_LAUNCH_CHECK();
  return 0;
}

static int init_shortlists(int32_t* sl_idx, int64_t n_idx, float* sl_tau, int64_t n_tau, hipStream_t st) {
  const int64_t n = n_idx > n_tau ? n_idx : n_tau;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(init_shortlists_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sl_idx, n_idx, sl_tau, n_tau);
  BGNN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int bgnn_l2_normalize_rows_f32(const float* q, int64_t n, int32_t d, float eps, float* out, void* stream) {
  if (!q || !out) return BGNN_E_NULL;
  if (n < 0 || d <= 0) return BGNN_E_SHAPE;
  if (n == 0) return 0;
  int rows = (int)((48 * 1024) / (sizeof(float) * (size_t)(d + 1)));       // LDS tile of <= 48 KB
  if (rows > 64) rows = 64;
  if (rows >= 4) {
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)((n + rows - 1) / rows)), dim3(256),
                       sizeof(float) * rows * (size_t)(d + 1), (hipStream_t)stream, q, n, d, eps, out, rows);
  } else {
    hipLaunchKernelGGL(normalize_rows_wide_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, n, d, eps, out);
  }
  BGNN_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t bgnn_topk_workspace_bytes(int64_t Nq, int64_t Nc, int32_t k) {
  return topk_ws_layout(Nq < 1 ? 1 : Nq, Nc < 1 ? 1 : Nc, k, 256, nullptr, nullptr);      // embedding width <= 256
}

extern "C" int bgnn_cosine_topk_f32(const float* qn_query, const float* qn_cand, int64_t Nq, int64_t Nc,
                                    int32_t d, int32_t k, int apply_sigmoid, int64_t* idx_out, float* val_out,
                                    int32_t* n_fallback_opt, void* ws, size_t ws_bytes, void* stream) {
  if (!qn_query || !qn_cand || !idx_out || !val_out || !ws) return BGNN_E_NULL;
  if (Nq < 0 || Nc <= 0 || Nc >= (int64_t)1 << 31 || d <= 0) return BGNN_E_SHAPE;
  if (d != 32 && d != 64 && d != 128 && d != 256) return BGNN_E_SHAPE;   // callers zero-pad d
  if (k <= 0 || k > 56 || k > Nc) return BGNN_E_RANGE;
  if (!bgnn_aligned16(qn_query) || !bgnn_aligned16(qn_cand)) return BGNN_E_ALIGN;
  if (ws_bytes < topk_ws_layout(Nq < 1 ? 1 : Nq, Nc, k, d, nullptr, nullptr)) return BGNN_E_WORKSPACE;
  if (Nq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  TopkWs w;
  topk_ws_layout(Nq, Nc, k, d, &w, ws);
  hipError_t e;
  if ((e = bgnn_zero_async(w.counts, 256, st)) != hipSuccess) return (int)e;
  if ((e = bgnn_zero_async(w.mx, 256, st)) != hipSuccess) return (int)e;
  // stage 0: bf16 pieces + residual maxima
  const int lpr = d / 4;
  auto split_grid = [&](int64_t n) { const int64_t g = (n * lpr + 255) / 256; return (unsigned)(g < 1024 ? (g < 1 ? 1 : g) : 1024); };
  hipLaunchKernelGGL(split_rows_kernel, dim3(split_grid(Nc)), dim3(256), 0, st, qn_cand, Nc, d, w.ch, w.cm, w.mx);
  BGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(split_rows_kernel, dim3(split_grid(Nq)), dim3(256), 0, st, qn_query, Nq, d, w.qh, w.qm, w.mx + 2);
  BGNN_LAUNCH_CHECK();
  static const int fast_nprod = [] { const char* s = getenv("BGNN_KNN_FAST_PRODUCTS"); const int v = s ? atoi(s) : 1; return v >= 1 && v <= 3 ? v : 1; }();
  const EpsSrc eps{w.mx, d};
  CosineCanon canon{qn_query, qn_cand, d};
  int rc;
  const int cus = device_cus();
  if (cus < 1) return (int)hipErrorInvalidDevice;
  // stage 1 + 2: fast pass over every row, refine.  Most admissions of a streaming top-k happen while the thresholds are
  // still low, i.e. at the start of every stream (K ln(n/K) in all, two thirds of them in the first sixteenth): a HEAD
  // pass over the first ~1/16 of the candidates pays that price once, and the MAIN pass over the rest starts every
  // segment from the head's per-query threshold (a valid lower bound of the final one) instead of from -inf.
  {
    const int kp = geom_fast(k).kp;
    int qpb = 0;
    P1Params p{w.qh, w.qm, w.ch, w.cm, Nq, Nc, nullptr, nullptr, 0, 0, w.sl_score, w.sl_idx, w.sl_tau, eps, k, 0, 0, nullptr, 0, 0};
    pass1_any(d, fast_nprod, k, p, nullptr, &qpb, st);
    const int ct = MT * geom_fast(k).tpi;                                       // candidates per staged unit
    const int64_t ntiles = (Nc + ct - 1) / ct;
    const int64_t head_tiles = ntiles >= 128 ? (ntiles + 15) / 16 : 0;          // small problems: one pass
    const int64_t NcA = head_tiles * ct, NcB = Nc - NcA;
    const Pass1Plan plA = head_tiles ? plan_pass1(Nq, NcA, cus, qpb, ct) : Pass1Plan{0, 0, 0};
    const Pass1Plan plB = plan_pass1(Nq, NcB, cus, qpb, ct);
    const int nslots = plA.nslots + plB.nslots;
    if ((rc = init_shortlists(w.sl_idx, Nq * nslots * (int64_t)kp, w.sl_tau, Nq * (int64_t)nslots, st))) return rc;
    p.nslots = nslots;
    if (head_tiles) {
      p.Nc = NcA; p.cand_lo = 0; p.slot_base = 0; p.tau_init = nullptr; p.carry_slots = 0; p.known_tiles = 0;
      if ((rc = pass1_any(d, fast_nprod, k, p, &plA, nullptr, st))) return rc;
      hipLaunchKernelGGL(tau_from_slots_kernel, dim3((unsigned)((Nq + 255) / 256)), dim3(256), 0, st, w.sl_tau, Nq, nslots, 0, plA.nslots, w.tau_init);
      BGNN_LAUNCH_CHECK();
    }
    p.Nc = NcB; p.cand_lo = NcA; p.slot_base = plA.nslots; p.tau_init = head_tiles ? w.tau_init : nullptr; p.carry_slots = plA.nslots; p.known_tiles = (int)head_tiles;
    if ((rc = pass1_any(d, fast_nprod, k, p, &plB, nullptr, st))) return rc;
    RefineParams rp{Nq, nullptr, nullptr, k, nslots * kp, kp, nslots, w.sl_score, w.sl_idx, w.sl_tau, eps, fast_nprod, 0.0, 0.0,
                    apply_sigmoid, idx_out, val_out, w.fail1, w.counts};
    if ((rc = launch_refine(canon, rp, st))) return rc;
  }
  // stage 3: precise pass on the rows the fast stage could not prove (plan made on the device from the row count)
  if (fast_nprod != 3) {
    const int kp2 = geom_precise(k).kp;
    P1Params p{w.qh, w.qm, w.ch, w.cm, Nq, Nc, w.fail1, w.counts, 0, MAX_SLOTS, w.sl2_score, w.sl2_idx, w.sl2_tau, eps, k, 0, 0, nullptr, 0, 0};
    hipLaunchKernelGGL(init_shortlists_rows_kernel, dim3(1024), dim3(256), 0, st, w.sl2_idx, (int64_t)MAX_SLOTS * kp2, w.sl2_tau,
                       (int64_t)MAX_SLOTS, w.counts);
    BGNN_LAUNCH_CHECK();
    if ((rc = pass1_any(d, 3, k, p, nullptr, nullptr, st))) return rc;
    RefineParams rp{Nq, w.fail1, w.counts, k, MAX_SLOTS * kp2, kp2, MAX_SLOTS, w.sl2_score, w.sl2_idx, w.sl2_tau, eps, 3, 0.0, 0.0,
                    apply_sigmoid, idx_out, val_out, w.fail2, w.counts + 1};
    if ((rc = launch_refine(canon, rp, st))) return rc;
  } else {
    hipLaunchKernelGGL(copy_counts_kernel, dim3(1), dim3(1), 0, st, w.counts, nullptr, w.counts + 1);
    BGNN_LAUNCH_CHECK();
    if ((e = hipMemcpyAsync(w.fail2, w.fail1, sizeof(int32_t) * (Nq + 1), hipMemcpyDeviceToDevice, st)) != hipSuccess) return (int)e;
  }
  // stage 4: exhaustive canonical pass for what is left (exact ties across the boundary)
  hipLaunchKernelGGL((fallback_kernel<CosineCanon>), dim3(w.fb_blocks), dim3(256), 0, st, canon, Nc, k, apply_sigmoid,
                     w.fail2, w.counts + 1, w.scratch, idx_out, val_out);
  BGNN_LAUNCH_CHECK();
  if (n_fallback_opt) {
    hipLaunchKernelGGL(copy_counts_kernel, dim3(1), dim3(1), 0, st, w.counts + 1, fast_nprod != 3 ? w.counts : nullptr, n_fallback_opt);
    BGNN_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int bgnn_mlp_pair_topk_f32(const float* A_cand, const float* B_query, const float* bn_scale,
                                      const float* bn_shift, const float* w2, float b2, int64_t Nq, int64_t Nc,
                                      int32_t H, int32_t k, int apply_sigmoid, int64_t* idx_out, float* val_out,
                                      int32_t* n_fallback_opt, void* ws, size_t ws_bytes, void* stream) {
  if (!A_cand || !B_query || !bn_scale || !bn_shift || !w2 || !idx_out || !val_out || !ws) return BGNN_E_NULL;
  if (Nq < 0 || Nc <= 0 || Nc >= (int64_t)1 << 31) return BGNN_E_SHAPE;
  if (H != MLP_H) return BGNN_E_SHAPE;   // Similar_v2 'mlp' hidden width is fixed at 128 (models.py:921)
  if (k <= 0 || k > 56 || k > Nc) return BGNN_E_RANGE;
  if (!bgnn_aligned16(A_cand) || !bgnn_aligned16(B_query)) return BGNN_E_ALIGN;
  if (ws_bytes < topk_ws_layout(Nq < 1 ? 1 : Nq, Nc, k, 32, nullptr, nullptr)) return BGNN_E_WORKSPACE;
  if (Nq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  TopkWs w;
  topk_ws_layout(Nq, Nc, k, 32, &w, ws);
  hipError_t e;
  if ((e = bgnn_zero_async(w.counts, 256, st)) != hipSuccess) return (int)e;
  // fp32 evaluation of a 128-term sum of O(1) terms: relative bound on the logit magnitude plus an absolute floor;
  // generous (a loose bound only widens the shortlist / costs a few more exhaustive rows)
  const float err_abs = 1e-4f, err_rel = 1e-4f;
  const Geom g = geom_mlp(k);
  int rc;
  if ((rc = init_shortlists(w.sl_idx, Nq * (int64_t)g.kp, w.sl_tau, Nq, st))) return rc;
  {
    const size_t stage = sizeof(float) * (MLP_CT * (MLP_H + 4) + 3 * MLP_H);
    const unsigned grid = (unsigned)((Nq + MLP_WAVES * QPW - 1) / (MLP_WAVES * QPW));
    if (g.cap == 64) {
      auto kern = mlp_pass1_kernel<64, 40>;
      const size_t sh = stage + MLP_WAVES * WaveTopK<64, 40>::BYTES;
      static int done[BGNN_MAX_DEVICES];
      if ((e = bgnn_set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)sh, done)) != hipSuccess) return (int)e;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), sh, st, A_cand, B_query, bn_scale, bn_shift, w2, b2, Nq, Nc, (int)k, err_abs, err_rel,
                         w.sl_score, w.sl_idx, w.sl_tau);
    } else {
      auto kern = mlp_pass1_kernel<128, 112>;
      const size_t sh = stage + MLP_WAVES * WaveTopK<128, 112>::BYTES;
      static int done[BGNN_MAX_DEVICES];
      if ((e = bgnn_set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)sh, done)) != hipSuccess) return (int)e;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), sh, st, A_cand, B_query, bn_scale, bn_shift, w2, b2, Nq, Nc, (int)k, err_abs, err_rel,
                         w.sl_score, w.sl_idx, w.sl_tau);
    }
    BGNN_LAUNCH_CHECK();
  }
  MlpCanon canon{A_cand, B_query, bn_scale, bn_shift, w2, b2, H};
  RefineParams rp{Nq, nullptr, nullptr, k, g.kp, g.kp, 1, w.sl_score, w.sl_idx, w.sl_tau, EpsSrc{nullptr, 0}, 0, (double)err_abs, (double)err_rel,
                  apply_sigmoid, idx_out, val_out, w.fail2, w.counts + 1};
  if ((rc = launch_refine(canon, rp, st))) return rc;
  hipLaunchKernelGGL((fallback_kernel<MlpCanon>), dim3(w.fb_blocks), dim3(256), 0, st, canon, Nc, k, apply_sigmoid,
                     w.fail2, w.counts + 1, w.scratch, idx_out, val_out);
  BGNN_LAUNCH_CHECK();
  if (n_fallback_opt) {
    hipLaunchKernelGGL(copy_counts_kernel, dim3(1), dim3(1), 0, st, w.counts + 1, nullptr, n_fallback_opt);
    BGNN_LAUNCH_CHECK();
  }
  return 0;
}

