// kNN bridge construction on gfx950: pair scoring + per-query top-k, never materialising the
// pair list (reference: Bridged-GNN/main_bridged_graph.py:45-67 / :90-111 batched loops,
// pair_enumeration models/models.py:265-282, scorers :124-130 (cosine) and :944-954 (mlp),
// Tensor.topk call sites main_bridged_graph.py:60,:104).
//
// Three passes (see include/bgnn.h for the contract):
//   1. stream every candidate against a block of queries, fp32 scores (cosine: v_mfma_f32_32x32x2_f32
//      tiles, candidates = A operand so that one lane holds 16 scores of ONE query -> a single
//      threshold register per lane; mlp: VALU), keep a KP-entry shortlist per query in LDS
//      (threshold filter + rare wave-wide bitonic compaction);
//   2. re-score the shortlist in CANONICAL arithmetic (fp64, feature-index order), rank by
//      (score desc, index asc), and prove by an error-bound margin that the exact top-k lies inside
//      the shortlist; unproven rows are queued;
//   3. queued rows are re-done exhaustively in canonical arithmetic.
// Index results are therefore bit-identical to oracle/oracle_c.c orc_cosine_topk / orc_mlp_topk.
#include <cstdio>

#include <cstdlib>
#include "bgnn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#ifndef KNN_CAND_PIECES
#define KNN_CAND_PIECES 2
#endif
constexpr int CPIECES = KNN_CAND_PIECES;   // bf16 pieces per candidate value (3 = full fp32 significand, 2 = 2^-18)
constexpr int QPW = 32;            // queries per wave (one 32-wide MFMA column block)
constexpr int WAVES = 4;
constexpr int QPB = QPW * WAVES;   // queries per block
constexpr int CT = 32;             // candidates per MFMA tile

// ---- sortable keys: larger key = better (higher score, then LOWER candidate index) --------------
__device__ __forceinline__ uint32_t ord_f32(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__device__ __forceinline__ u64 make_key(float s, int32_t idx) {
  return ((u64)ord_f32(s) << 32) | (u64)(0xFFFFFFFFu - (uint32_t)idx);
}
__device__ __forceinline__ int32_t key_idx(u64 k) { return (int32_t)(0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFu)); }
__device__ __forceinline__ float key_score(u64 k) { return unord_f32((uint32_t)(k >> 32)); }
constexpr u64 KEY_EMPTY = 0ull;    // below every real key (ord(-inf) = 0x007FFFFF > 0)

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_xor(lo, m);
  hi = __shfl_xor(hi, m);
  return ((u64)hi << 32) | lo;
}

// wave-wide bitonic sort, DESCENDING, of 64*EPL keys (element e of lane l has global index l + 64*e)
template <int EPL>
__device__ __forceinline__ void wave_sort_desc(u64 (&v)[EPL], int lane) {
  constexpr int NTOT = 64 * EPL;
#pragma unroll
  for (int k = 2; k <= NTOT; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {   // partner lives in the same lane (only EPL == 2, j == 64)
        const int i0 = lane, i1 = lane + 64;
        const bool desc = ((i0 & k) == 0);   // k == 128 here -> always true
        u64 a = v[0], b = v[EPL - 1];
        const bool sw = desc ? (a < b) : (a > b);
        if (sw) { v[0] = b; v[EPL - 1] = a; }
        (void)i1;
      } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          const int i = lane + 64 * e;
          const u64 o = shfl_xor_u64(v[e], j);
          const bool desc = ((i & k) == 0);
          const bool lower = ((i & j) == 0);          // I am the lower index of the pair
          // descending block: lower index keeps the max
          const bool keep_max = (desc == lower);
          v[e] = keep_max ? (v[e] > o ? v[e] : o) : (v[e] < o ? v[e] : o);
        }
      }
    }
  }
}

// ---- per-wave shortlist state in LDS -----------------------------------------------------------
// Per query: CAP buffer entries of which the best KP are "live"; the CAP-KP others absorb arrivals
// between two compactions.  Per wave: a small QUEUE of (key, query) pairs.  Scores that beat the lane's
// threshold register are only queued while tiles stream (ballot-prefix offsets, no atomics, bounded
// time -> the block barrier is never held up by one wave's bookkeeping); the queue is drained in
// batches of 64 (one LDS atomic round for the whole batch) every few tiles.
#if defined(KNN_EXP) && (KNN_EXP == 8 || KNN_EXP == 9 || KNN_EXP == 10)
__device__ unsigned long long g_knn_cnt[8];
__device__ unsigned long long g_knn_blk[1024 * 3];
#define KCOUNT(i, v) do { if (KNN_EXP == 8 && lane == 0) atomicAdd(&g_knn_cnt[i], (unsigned long long)(v)); } while (0)
#else
#define KCOUNT(i, v) do { } while (0)
#endif
#if defined(KNN_EXP) && KNN_EXP == 9
#define KSTAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define KACC(i, a, b) do { if (lane == 0) tk.dbg[i] += (b) - (a); } while (0)
#else
#define KSTAMP(var)
#define KACC(i, a, b) do { } while (0)
#endif
constexpr int QCAP = 128;                  // queue entries per wave
#ifdef DRAIN_EVERY_OVERRIDE
constexpr int DRAIN_EVERY = DRAIN_EVERY_OVERRIDE;
#else
constexpr int DRAIN_EVERY = 8;
#endif             // tiles between unconditional drains (all waves drain together)

template <int CAPV, int KPV>
struct WaveTopK {
  static constexpr int CAP = CAPV, KP = KPV, EPL = (CAPV + 63) / 64;
  static_assert(KPV < CAPV && CAPV - KPV >= 2 && CAPV <= 128, "shortlist geometry");
  static constexpr size_t BYTES = sizeof(u64) * QPW * CAPV + sizeof(int) * QPW + sizeof(float) * QPW +
                                  sizeof(u64) * QCAP + sizeof(int) * QCAP + 64;
  u64* keys;                             // [QPW][CAP]
  int* cnt;                              // [QPW]
  float* tau;                            // [QPW]  current admission threshold (score of the KP-th best)
  u64* qkey;                             // [QCAP] queued keys
  int* qqry;                             // [QCAP] their query (0..31)
  int qcount;                            // wave-uniform
  unsigned long long* dbg;               // [8] cycle counters (diagnostic builds only)

  __device__ __forceinline__ void carve(unsigned char* base) {
    keys = reinterpret_cast<u64*>(base);
    qkey = keys + QPW * CAP;
    cnt = reinterpret_cast<int*>(qkey + QCAP);
    tau = reinterpret_cast<float*>(cnt + QPW);
    qqry = reinterpret_cast<int*>(tau + QPW);
    dbg = reinterpret_cast<unsigned long long*>(qqry + QCAP);
  }
  __device__ __forceinline__ void kq_store(int q, int slot, u64 k) { keys[q * CAP + slot] = k; }
  __device__ __forceinline__ void init(int lane) {
    for (int t = lane; t < QPW * CAP; t += 64) keys[t] = KEY_EMPTY;
    if (lane < QPW) { cnt[lane] = 0; tau[lane] = -INFINITY; }
    qcount = 0;
  }
  // keep the best KP entries of query q (sorted descending), refresh tau.  Whole wave, uniform q.
  __device__ __forceinline__ void compact(int q, int lane) {
    const int n = min(cnt[q], CAP);
    if constexpr (EPL == 1) {
      // rank by counting: every lane streams the query's CAP keys from LDS (same address in all lanes ->
      // broadcast reads) and counts the larger ones; no shuffle network, ~3 VALU ops per key.
      // Valid keys are unique (distinct candidate index) and all exceed KEY_EMPTY, so ranks of valid keys
      // are a permutation of [0, n); slots >= n are never read.
      const u64* kq = keys + q * CAP;
      const u64 key = lane < n ? kq[lane] : KEY_EMPTY;
      int rank = 0;
      // batches of 16 keys (8 x ds_read_b128, independent, issued back to back), then 16 compares
      static_assert(CAP % 8 == 0, "CAP multiple of 8");
#pragma unroll
      for (int j0 = 0; j0 < CAP; j0 += 16) {
        constexpr int NB = 16;
        u64 kk[NB];
#pragma unroll
        for (int t = 0; t < NB; t += 2) {
          if (j0 + t < CAP) {
            const ulonglong2 two = *reinterpret_cast<const ulonglong2*>(kq + j0 + t);
            kk[t] = two.x; kk[t + 1] = two.y;
          } else { kk[t] = KEY_EMPTY; kk[t + 1] = KEY_EMPTY; }
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const u64 kj = (j0 + t < n) ? kk[t] : KEY_EMPTY;
          rank += (kj > key) ? 1 : 0;
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): all reads of the old order done before the scatter
      if (lane < n) kq_store(q, rank, key);
      if (lane < n && rank == KP - 1) tau[q] = unord_f32((uint32_t)(key >> 32));
      if (lane == 0) cnt[q] = n < KP ? n : KP;
    } else {
      u64 v[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int i = lane + 64 * e;
        v[e] = i < n ? keys[q * CAP + i] : KEY_EMPTY;
      }
      wave_sort_desc<EPL>(v, lane);
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int i = lane + 64 * e;
        if (i < CAP) keys[q * CAP + i] = v[e];
      }
      constexpr int TL = (KP - 1) % 64, TE = (KP - 1) / 64;
      const uint32_t kb = __shfl((uint32_t)(v[TE] >> 32), TL);
      if (lane == 0) {
        cnt[q] = n < KP ? n : KP;
        tau[q] = (n >= KP) ? unord_f32(kb) : -INFINITY;
      }
    }
  }
};

template <class TK>
__device__ __noinline__ void compact_rows(TK tk, unsigned int qmask, int lane) {
  KSTAMP(c0);
  while (qmask) {
    const int qq = __ffs(qmask) - 1;
    qmask &= qmask - 1;
    KCOUNT(4, 1);
    tk.compact(qq, lane);
  }
  KSTAMP(c1);
  KACC(5, c0, c1);
}

// Move the queued pairs into the per-query buffers, 64 at a time; full buffers are compacted and the
// affected pairs re-checked against the raised threshold.
template <class TK>
__device__ __noinline__ void drain_queue(TK tk, int n, int lane) {
  constexpr int CAP = TK::CAP;
#if defined(KNN_EXP) && KNN_EXP == 6
  return;                                           // timing experiment: queueing only
#endif
  KCOUNT(3, 1);
  KSTAMP(d0);
  for (int base = 0; base < n; base += 64) {
    const int e = base + lane;
    const bool have = e < n;
    const u64 key = have ? tk.qkey[e] : KEY_EMPTY;
    const int qi = have ? tk.qqry[e] : 0;
    bool pend = have && key_score(key) > tk.tau[qi];
    for (int guard = 0; guard < 64; ++guard) {
      if (pend) {
        const int slot = atomicAdd(&tk.cnt[qi], 1);
        if (slot < CAP) { tk.keys[qi * CAP + slot] = key; pend = false; }
      }
      unsigned long long over = __ballot(pend);
      if (!over) break;
      unsigned int qmask = 0;                       // distinct queries that overflowed
      while (over) {
        const int lead = __ffsll((long long)over) - 1;
        const int qsel = __builtin_amdgcn_readlane(qi, lead);
        qmask |= 1u << qsel;
        over &= ~__ballot(pend && qi == qsel);
      }
#if defined(KNN_EXP) && KNN_EXP == 7
      if (lane < 32 && ((qmask >> lane) & 1)) tk.cnt[lane] = TK::KP;   // timing experiment: no compaction work
#else
      compact_rows(tk, qmask, lane);
#endif
      pend = pend && key_score(key) > tk.tau[qi];
    }
  }
  KSTAMP(d1);
  KACC(4, d0, d1);
}

__device__ __forceinline__ float select16(const f32x16& acc, int r) {
  float v = acc[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) v = (r == i) ? acc[i] : v;
  return v;
}

// Offer the 16 scores a lane holds (one query q = lane&31, candidates cbase + cand(r)).
template <class TK>
__device__ __forceinline__ void offer_tile(TK& tk, f32x16 acc, int cbase, int64_t Nc, int lane, float& tau,
                                           bool force_drain) {
  const int q = lane & 31, h = lane >> 5;
  if (cbase + CT > Nc) {                            // last (partial) tile only: mask candidates >= Nc
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (cbase + (r & 3) + 8 * (r >> 2) + 4 * h >= Nc) acc[r] = -INFINITY;
  }
  float mx = acc[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[r]);
#if defined(KNN_EXP) && KNN_EXP == 4
  if (__any(mx > tau)) asm volatile("" ::"v"(acc[3]));
  return;                                         // timing experiment: fast-path test only
#endif
  if (__any(mx > tau)) {
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m |= (acc[r] > tau) ? (1u << r) : 0u;
    while (true) {                                  // one queued score per lane and round, lowest register first
      const bool pend = m != 0;
      const unsigned long long b = __ballot(pend);
      if (!b) break;
      if (pend) {
        const int r = __ffs(m) - 1;
        const int off = tk.qcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
        tk.qkey[off] = make_key(select16(acc, r), cbase + (r & 3) + 8 * (r >> 2) + 4 * h);
        tk.qqry[off] = q;
        m &= m - 1;
      }
      tk.qcount += __popcll(b);
      if (tk.qcount > QCAP - 64) {                  // wave-uniform: the next round might not fit
        drain_queue(tk, tk.qcount, lane);
        tk.qcount = 0;
        tau = tk.tau[q];
        uint32_t keep = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) keep |= (acc[r] > tau) ? (1u << r) : 0u;
        m &= keep;
      }
    }
  }
  if (force_drain && tk.qcount > 0) {
    drain_queue(tk, tk.qcount, lane);
    tk.qcount = 0;
    tau = tk.tau[q];
  }
}

// final: sort each query's buffer and emit the best KP (score, idx) pairs, descending
template <class TK>
__device__ __forceinline__ void emit_shortlists(TK& tk, int lane, int64_t q0, int64_t Nq,
                                                float* __restrict__ sl_score, int32_t* __restrict__ sl_idx,
                                                int split, int nsplit) {
  constexpr int CAP = TK::CAP, KP = TK::KP, EPL = TK::EPL;
  if (tk.qcount > 0) { drain_queue(tk, tk.qcount, lane); tk.qcount = 0; }
  compact_rows(tk, 0xFFFFFFFFu, lane);
  for (int q = 0; q < QPW; ++q) {
    const int64_t gq = q0 + q;
    if (gq < Nq) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int pos = lane + 64 * e;
        if (pos < KP) {
          const u64 k = tk.keys[q * CAP + pos];
          const int64_t o = (gq * nsplit + split) * KP + pos;
          sl_score[o] = (k == KEY_EMPTY) ? -INFINITY : key_score(k);
          sl_idx[o] = (k == KEY_EMPTY) ? -1 : key_idx(k);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// pass 1, cosine: scores = Qn_cand_tile (A, 32 x d) . Qn_query_block^T (B, d x 32) on fp32 MFMA.
// DK = d / 8.  Work = (query block, candidate tile) pairs, query-major; every persistent block takes one
// CONTIGUOUS range of `tpb` tiles (perfect balance, no tail), emitting one shortlist per query-block
// segment it touches (slot = block - first block touching that query block).
// BF3: the scores come from bf16 MFMAs on bf16 pieces of both operands (candidates hi + mid + lo = the fp32 significand,
// queries hi + mid; five piece products).  fp32 MFMA shares the VALU datapath on CDNA (its 157 TFLOP/s
// is the vector peak, and the shortlist upkeep's VALU work adds to it); the bf16 matrix cores are 16x faster and run
// beside the VALU, so pass 1 becomes bound by the upkeep alone.  Candidates are split while they are staged
// (fp32 in HBM/L2: no extra traffic), queries once per segment into registers.
template <int D>
__device__ __forceinline__ int bf_swz(int row) {          // 16-byte chunk swizzle of the unpadded bf16 piece rows
  constexpr int NCH = D / 8;
  if constexpr (NCH >= 16) return row & 15;
  else if constexpr (NCH == 8) return (row >> 1) & 7;
  else return (row >> 2) & 3;
}
__device__ __forceinline__ void bf_split4(const float4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
  const float vf[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 hh = (__bf16)vf[e];
    const float r1 = vf[e] - (float)hh;                   // exact
    const __bf16 mm = (__bf16)r1;
    h[e] = hh; m[e] = mm; l[e] = (__bf16)(r1 - (float)mm);
  }
}

template <int DK, int CAPV, int KPV, bool BF3>
__global__ __launch_bounds__(256) void cosine_pass1_kernel(const float* __restrict__ qq, const float* __restrict__ qc,
                                                           int64_t Nq, int64_t Nc, int64_t tpb, int nslots,
                                                           float* __restrict__ sl_score, int32_t* __restrict__ sl_idx) {
  typedef WaveTopK<CAPV, KPV> TK;
  constexpr int D = DK * 8, LD = D + 4;
  constexpr int NCH = D / 8;                                               // BF3: 16-byte chunks per piece row
  constexpr size_t STAGE_BYTES = BF3 ? (size_t)CPIECES * CT * D * 2 : sizeof(float) * CT * LD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* stage = reinterpret_cast<float*>(smem);                           // fp32: [CT][LD]
  __bf16* stage16 = reinterpret_cast<__bf16*>(smem);                       // BF3 : [3][CT][NCH ^ swizzle][8]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t ntiles = (Nc + CT - 1) / CT;
  const int64_t nqb = (Nq + QPB - 1) / QPB;
  const int64_t T = nqb * ntiles;
  int64_t t = (int64_t)blockIdx.x * tpb;
  const int64_t t_end = min(T, t + tpb);

  TK tk;
  tk.carve(smem + STAGE_BYTES + (size_t)wave * TK::BYTES);
#if defined(KNN_EXP) && KNN_EXP == 9
  if (lane < 8) tk.dbg[lane] = 0;
  const unsigned long long kt0 = __builtin_amdgcn_s_memtime(), kr0 = __builtin_amdgcn_s_memrealtime();
#endif

  // staging: CT x D floats over 256 threads
  constexpr int F4_PER_ROW = D / 4;
  constexpr int NLD = CT * F4_PER_ROW / 256;
  static_assert(CT * F4_PER_ROW % 256 == 0 && NLD >= 1, "staging split");
  float4 pre[NLD];
  auto gload = [&](int64_t ct) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = tid + 256 * j;
      const int r = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
      const int64_t gc = ct * CT + r;
      pre[j] = gc < Nc ? *reinterpret_cast<const float4*>(qc + gc * D + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = tid + 256 * j;
      const int r = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
      if constexpr (!BF3) {
        *reinterpret_cast<float4*>(&stage[r * LD + c4 * 4]) = pre[j];
      } else {
        bf16x4 h, m, l;
        bf_split4(pre[j], h, m, l);
        const int off = (r * NCH + ((c4 >> 1) ^ bf_swz<D>(r))) * 8 + (c4 & 1) * 4;
        *reinterpret_cast<bf16x4*>(&stage16[off]) = h;
        *reinterpret_cast<bf16x4*>(&stage16[CT * D + off]) = m;
        if constexpr (CPIECES > 2) *reinterpret_cast<bf16x4*>(&stage16[2 * CT * D + off]) = l;
      }
    }
  };
  const int fr = lane & 31, fh = lane >> 5;
#if defined(KNN_EXP) && KNN_EXP == 10
  const unsigned long long cen0 = __builtin_amdgcn_s_memrealtime();
#endif
#if defined(KNN_STAGGER)
  // de-phase the co-resident block pair of a CU (second wave of blocks lands on CUs that already host one):
  // without it both run their MFMA chains and their bookkeeping at the same time (speed only)
  if (blockIdx.x >= gridDim.x / 2) { for (int i = 0; i < KNN_STAGGER; ++i) __builtin_amdgcn_s_sleep(32); }
#endif

  while (t < t_end) {                               // block-uniform: one segment per query block touched
    const int64_t qb = t / ntiles, ct0 = t % ntiles;
    const int64_t ct1 = min(ntiles, ct0 + (t_end - t));
    const int slot = (int)(blockIdx.x - (qb * ntiles) / tpb);
    const int64_t q0 = qb * QPB + wave * QPW;
    tk.init(lane);
    // B fragments (this wave's 32 queries): lane (j = lane&31, h = lane>>5) holds q[j][8kb + 4h + s], s = 0..3
    float4 bq[BF3 ? 1 : DK];
    bf16x8 bqp[BF3 ? 2 : 1][BF3 ? D / 16 : 1];   // BF3: lane (j, h) holds the (hi, mid) pieces of q[j][16kb + 8h + e], e = 0..7
                                                 // (queries keep two pieces -- |q - hi - mid| <= 2^-18 |q| -- to stay within 256 VGPRs)
    {
      const int64_t gq = q0 + fr;
      if constexpr (!BF3) {
#pragma unroll
        for (int kb = 0; kb < DK; ++kb)
          bq[kb] = gq < Nq ? *reinterpret_cast<const float4*>(qq + gq * D + kb * 8 + fh * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
#pragma unroll
        for (int kb = 0; kb < D / 16; ++kb)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const float4 v = gq < Nq ? *reinterpret_cast<const float4*>(qq + gq * D + kb * 16 + fh * 8 + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            bf16x4 h, m, l;
            bf_split4(v, h, m, l);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bqp[0][kb][4 * hf + e] = h[e]; bqp[1][kb][4 * hf + e] = m[e]; }
          }
      }
    }
    float tau = -INFINITY;
    gload(ct0);
    for (int64_t ct = ct0; ct < ct1; ++ct) {
      KSTAMP(s0);
      sstore();
      __syncthreads();
      KSTAMP(s1);
      if (ct + 1 < ct1) gload(ct + 1);             // next tile flies while this one is scored
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#if defined(KNN_PRIO)
      __builtin_amdgcn_s_setprio(KNN_PRIO);
#endif
      if constexpr (!BF3) {
        const float* arow = &stage[fr * LD + fh * 4];
#pragma unroll
        for (int kb = 0; kb < DK; ++kb) {
          const float4 a = *reinterpret_cast<const float4*>(arow + kb * 8);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[kb].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[kb].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[kb].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[kb].w, acc, 0, 0, 0);
        }
      } else {
        const int sw = bf_swz<D>(fr);
#pragma unroll
        for (int kb = 0; kb < D / 16; ++kb) {
          const int off = (fr * NCH + ((2 * kb + fh) ^ sw)) * 8;
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&stage16[off]);
          const bf16x8 am = *reinterpret_cast<const bf16x8*>(&stage16[CT * D + off]);
          // smallest terms first; dropped pieces are in the error bound (2^-18 per operand with two pieces)
          if constexpr (CPIECES > 2) {
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(&stage16[2 * CT * D + off]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bqp[0][kb], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bqp[1][kb], acc, 0, 0, 0);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bqp[0][kb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bqp[1][kb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bqp[0][kb], acc, 0, 0, 0);
        }
      }
#if defined(KNN_PRIO)
      __builtin_amdgcn_s_setprio(0);
#endif
      KSTAMP(s2);
#if !defined(KNN_EXP) || KNN_EXP != 2
      __syncthreads();                             // the stage buffer may be overwritten from here on
#endif
      KSTAMP(s3);
#if defined(KNN_EXP) && (KNN_EXP == 1 || KNN_EXP == 3)
      asm volatile("" ::"v"(acc[0]), "v"(acc[5]), "v"(acc[15]));   // timing experiment: scores stay live, no shortlist
#else
      offer_tile(tk, acc, (int)(ct * CT), Nc, lane, tau, ((ct - ct0) % DRAIN_EVERY) == DRAIN_EVERY - 1);
#endif
      KSTAMP(s4);
      KACC(1, s0, s1); KACC(0, s1, s2); KACC(2, s2, s3); KACC(3, s3, s4);
    }
    emit_shortlists(tk, lane, q0, Nq, sl_score, sl_idx, slot, nslots);
    t += ct1 - ct0;
  }
#if defined(KNN_EXP) && KNN_EXP == 10
  if (tid == 0 && blockIdx.x < 1024) { g_knn_blk[blockIdx.x * 3] = cen0; g_knn_blk[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime();
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); g_knn_blk[blockIdx.x * 3 + 2] = ((unsigned long long)xcc << 32) | hwid; }
#endif
#if defined(KNN_EXP) && KNN_EXP == 9
  if (lane == 0) { tk.dbg[6] = __builtin_amdgcn_s_memtime() - kt0; tk.dbg[7] = __builtin_amdgcn_s_memrealtime() - kr0; }
  if (tid == 0 && blockIdx.x < 1024) { g_knn_blk[blockIdx.x * 3] = kr0; g_knn_blk[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime();
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); g_knn_blk[blockIdx.x * 3 + 2] = ((unsigned long long)xcc << 32) | hwid; }
  if (lane < 8) atomicAdd(&g_knn_cnt[lane], tk.dbg[lane]);
#endif
}

// pass 1, cosine, bf16 pieces, NW waves per block (NW * 32 queries) and a DOUBLE-BUFFERED candidate stage: one block
// barrier per tile instead of two, and with NW = 8 (one block per CU) the staging work (global loads, bf16 split, LDS
// writes) and the candidate traffic per query are half of the 4-wave form.  A wave that is busy with shortlist upkeep
// holds the others up only when it falls a whole tile behind.
template <int DK, int CAPV, int KPV, int NW>
__global__ __launch_bounds__(64 * NW) void cosine_pass1_db_kernel(const float* __restrict__ qq, const float* __restrict__ qc,
                                                                  int64_t Nq, int64_t Nc, int64_t tpb, int nslots,
                                                                  float* __restrict__ sl_score, int32_t* __restrict__ sl_idx) {
  typedef WaveTopK<CAPV, KPV> TK;
  constexpr int D = DK * 8, NCH = D / 8, NT = 64 * NW, QB = NW * QPW;
  constexpr int STAGE_ELEMS = CPIECES * CT * D;                             // bf16 elements per buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* stage16 = reinterpret_cast<__bf16*>(smem);                       // [2][CPIECES][CT][NCH ^ swizzle][8]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t ntiles = (Nc + CT - 1) / CT;
  const int64_t nqb = (Nq + QB - 1) / QB;
  const int64_t T = nqb * ntiles;
  int64_t t = (int64_t)blockIdx.x * tpb;
  const int64_t t_end = min(T, t + tpb);
  TK tk;
  tk.carve(smem + (size_t)2 * STAGE_ELEMS * 2 + (size_t)wave * TK::BYTES);
#if defined(KNN_EXP) && KNN_EXP == 9
  if (lane < 8) tk.dbg[lane] = 0;
  const unsigned long long kt0 = __builtin_amdgcn_s_memtime(), kr0 = __builtin_amdgcn_s_memrealtime();
#endif

  constexpr int F4_PER_ROW = D / 4, F4_TILE = CT * F4_PER_ROW;
  constexpr int NLD = (F4_TILE + NT - 1) / NT;
  float4 pre[NLD];
  auto gload = [&](int64_t ct) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = tid + NT * j;
      const int r = (f / F4_PER_ROW) % CT, c4 = f % F4_PER_ROW;   // (f >= F4_TILE only for tiny D: harmless duplicate)
      const int64_t gc = ct * CT + r;
      pre[j] = gc < Nc ? *reinterpret_cast<const float4*>(qc + gc * D + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&](int buf) {
    __bf16* st = stage16 + buf * STAGE_ELEMS;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = tid + NT * j;
      if (F4_TILE % NT == 0 || f < F4_TILE) {
        const int r = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
        bf16x4 h, m, l;
        bf_split4(pre[j], h, m, l);
        const int off = (r * NCH + ((c4 >> 1) ^ bf_swz<D>(r))) * 8 + (c4 & 1) * 4;
        *reinterpret_cast<bf16x4*>(&st[off]) = h;
        *reinterpret_cast<bf16x4*>(&st[CT * D + off]) = m;
        if constexpr (CPIECES > 2) *reinterpret_cast<bf16x4*>(&st[2 * CT * D + off]) = l;
      }
    }
  };
  const int fr = lane & 31, fh = lane >> 5;
  const int sw = bf_swz<D>(fr);

  while (t < t_end) {                               // block-uniform: one segment per query block touched
    const int64_t qb = t / ntiles, ct0 = t % ntiles;
    const int64_t ct1 = min(ntiles, ct0 + (t_end - t));
    const int slot = (int)(blockIdx.x - (qb * ntiles) / tpb);
    const int64_t q0 = qb * QB + wave * QPW;
    tk.init(lane);
    bf16x8 bqp[2][D / 16];                          // (hi, mid) pieces of q[j][16kb + 8h + e]
    {
      const int64_t gq = q0 + fr;
#pragma unroll
      for (int kb = 0; kb < D / 16; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const float4 v = gq < Nq ? *reinterpret_cast<const float4*>(qq + gq * D + kb * 16 + fh * 8 + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
          bf16x4 h, m, l;
          bf_split4(v, h, m, l);
#pragma unroll
          for (int e = 0; e < 4; ++e) { bqp[0][kb][4 * hf + e] = h[e]; bqp[1][kb][4 * hf + e] = m[e]; }
        }
    }
    float tau = -INFINITY;
    __syncthreads();                                // the previous segment's last tile has been read by every wave
    gload(ct0);
    sstore(0);
    if (ct0 + 1 < ct1) gload(ct0 + 1);
    // score one tile from stage[cur] on the matrix cores
    auto score = [&](int cur) {
      // two independent accumulator chains (even / odd k blocks): a single dependent chain of 32x32x16 bf16 MFMAs issues
      // one instruction per 64 cycles, half the matrix pipe's rate (tools/micro/mfma_valu_overlap.hip)
      f32x16 acc, acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
      const __bf16* st = stage16 + cur * STAGE_ELEMS;
#pragma unroll
      for (int kb = 0; kb < D / 16; ++kb) {
        const int off = (fr * NCH + ((2 * kb + fh) ^ sw)) * 8;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&st[off]);
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(&st[CT * D + off]);
        f32x16& a = (kb & 1) ? acc2 : acc;
        if constexpr (CPIECES > 2) {
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(&st[2 * CT * D + off]);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bqp[0][kb], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bqp[1][kb], a, 0, 0, 0);
        }
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bqp[0][kb], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bqp[1][kb], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bqp[0][kb], a, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
      return acc;
    };
    auto offer = [&](const f32x16& acc, int64_t ct) {
#if defined(KNN_EXP) && (KNN_EXP == 1 || KNN_EXP == 3)
      asm volatile("" ::"v"(acc[0]), "v"(acc[5]), "v"(acc[15]));   // timing experiment: scores stay live, no shortlist
#else
      offer_tile(tk, acc, (int)(ct * CT), Nc, lane, tau, ((ct - ct0) % DRAIN_EVERY) == DRAIN_EVERY - 1);
#endif
    };
    // Cycle stamps (-DKNN_EXP=9, tools/knn_exp.py) per wave and tile: scoring 1530 (two waves share a SIMD's matrix pipe:
    // 2 x 768), staging 660, barrier wait 1350, shortlist upkeep 1670 (queueing 980, drains 690 of which compaction 520).
    // Negative results: the skewed schedule below and parking single passes in per-lane registers until the next drain
    // (12.5 ms).
    // Tried and removed: a skewed schedule in which the second wave of every SIMD (waves w and w + NW/2
    // share one: tools/micro/wave_simd_map.hip) does the upkeep of tile t-1 before it scores tile t -- 13.2 vs 11.8 ms;
    // skewing only the staging was equally flat.
    {
    int cur = 0;
    for (int64_t ct = ct0; ct < ct1; ++ct, cur ^= 1) {
      KSTAMP(s0);
      __syncthreads();                              // stage[cur] complete; nobody reads stage[cur^1] (tile ct-1) any more
      KSTAMP(s1);
      if (ct + 1 < ct1) sstore(cur ^ 1);            // tile ct+1: registers -> the free buffer
      if (ct + 2 < ct1) gload(ct + 2);              // tile ct+2 flies while this one is scored
      KSTAMP(s2);
      const f32x16 acc = score(cur);
#if defined(KNN_EXP) && KNN_EXP == 9
      asm volatile("s_nop 0" ::"v"(acc[15]));
#endif
      KSTAMP(s3);
      offer(acc, ct);
      KSTAMP(s4);
      KACC(2, s0, s1); KACC(1, s1, s2); KACC(0, s2, s3); KACC(3, s3, s4);
    }
    }
    emit_shortlists(tk, lane, q0, Nq, sl_score, sl_idx, slot, nslots);
    t += ct1 - ct0;
  }
#if defined(KNN_EXP) && KNN_EXP == 9
  if (lane == 0) { tk.dbg[6] = __builtin_amdgcn_s_memtime() - kt0; tk.dbg[7] = __builtin_amdgcn_s_memrealtime() - kr0; }
  if (lane < 8) atomicAdd(&g_knn_cnt[lane], tk.dbg[lane]);
#endif
}


// ------------------------------------------------------------------------------------------------
// pass 1, mlp (Similar_v2 'mlp' in separable eval form, H = 128): fp32 VALU scoring, same shortlist.
constexpr int MLP_H = 128;
template <int EPL>
__global__ __launch_bounds__(256, 1) void mlp_pass1_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ w2, float b2, int64_t Nq, int64_t Nc,
                                                           float* __restrict__ sl_score, int32_t* __restrict__ sl_idx) {
  typedef WaveTopK<64 * EPL, 32 * EPL> TK;
  constexpr int H = MLP_H, LD = H + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* stage = reinterpret_cast<float*>(smem);                                    // [CT][LD]
  float* coefs = stage + CT * LD;                                                   // scale|shift|w2 [3][H]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t q0 = (int64_t)blockIdx.x * QPB + wave * QPW;
  TK tk;
  tk.carve(reinterpret_cast<unsigned char*>(coefs + 3 * H) + (size_t)wave * TK::BYTES);
  tk.init(lane);
  for (int t = tid; t < H; t += 256) { coefs[t] = scale[t]; coefs[H + t] = shift[t]; coefs[2 * H + t] = w2[t]; }
  float4 bqv[H / 4];                 // this lane's query row B[q][:], statically indexed (registers)
  {
    const int64_t gq = q0 + (lane & 31);
#pragma unroll
    for (int h4 = 0; h4 < H / 4; ++h4)
      bqv[h4] = gq < Nq ? *reinterpret_cast<const float4*>(B + gq * H + h4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float tau = -INFINITY;
  const int fh = lane >> 5;
  const float4* sc4 = reinterpret_cast<const float4*>(coefs);
  const float4* sh4 = reinterpret_cast<const float4*>(coefs + H);
  const float4* w4 = reinterpret_cast<const float4*>(coefs + 2 * H);
  for (int64_t cb = 0; cb < Nc; cb += CT) {
    __syncthreads();
    for (int f = tid; f < CT * (H / 4); f += 256) {
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
    offer_tile(tk, acc, (int)cb, Nc, lane, tau, true);
  }
  emit_shortlists(tk, lane, q0, Nq, sl_score, sl_idx, 0, 1);
}

// ------------------------------------------------------------------------------------------------
// canonical scores (identical arithmetic to oracle/oracle_c.c)
struct CosineCanon {
  const float* qq; const float* qc; int d;
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

// pass 2: one wave per query.  L = nsplit*KP shortlist entries.
template <class Canon>
__global__ __launch_bounds__(256) void refine_kernel(Canon canon, int64_t Nq, int k, int L, int KP, int nsplit,
                                                     const float* __restrict__ sl_score, const int32_t* __restrict__ sl_idx,
                                                     double err_abs, double err_rel, int apply_sigmoid,
                                                     int64_t* __restrict__ idx_out, float* __restrict__ val_out,
                                                     int32_t* __restrict__ fb_list, int32_t* __restrict__ fb_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* es = reinterpret_cast<double*>(smem) + (size_t)wave * L;                        // exact scores
  int32_t* ei = reinterpret_cast<int32_t*>(reinterpret_cast<double*>(smem) + (size_t)4 * L) + (size_t)wave * L;
  for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < Nq; q += (int64_t)gridDim.x * 4) {
    float alast = -INFINITY;   // best approximate score any excluded candidate can have
    for (int e = lane; e < L; e += 64) {
      const int32_t c = sl_idx[q * L + e];
      ei[e] = c;
      es[e] = c >= 0 ? canon(q, c) : -INFINITY;
      if ((e % KP) == KP - 1 && c >= 0) alast = fmaxf(alast, sl_score[q * L + e]);   // full split
    }
    alast = bgnn::group_max<64>(alast);
    __builtin_amdgcn_s_waitcnt(0);   // LDS writes above complete before the cross-lane reads below
    __builtin_amdgcn_wave_barrier();
    double kth = -INFINITY;
    for (int e = lane; e < L; e += 64) {
      const double s = es[e];
      const int32_t c = ei[e];
      int rank = 0;
      if (c >= 0) {
        for (int j = 0; j < L; ++j) {
          const double sj = es[j];
          const int32_t cj = ei[j];
          rank += (cj >= 0) && (sj > s || (sj == s && cj < c));
        }
        if (rank < k) {
          idx_out[q * k + rank] = c;
          val_out[q * k + rank] = apply_sigmoid ? sigmoid_f32((float)s) : (float)s;
        }
        if (rank == k - 1) kth = s;
      }
    }
    // broadcast kth (exactly one lane holds it when >= k valid entries exist)
    double kmax = kth;
    for (int o = 32; o > 0; o >>= 1) {
      const double other = __shfl_xor(kmax, o);
      kmax = other > kmax ? other : kmax;
    }
    const bool have = kmax > -INFINITY;
    // proof: every excluded candidate has exact score <= alast + bound < kth
    const double bound = err_abs + err_rel * fabs((double)alast);
    const bool proven = (alast == -INFINITY) || (have && kmax > (double)alast + bound);
    if (!proven && lane == 0) {
      const int slot = atomicAdd(fb_count, 1);
      fb_list[slot] = (int32_t)q;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// pass 3: exhaustive canonical re-do of queued rows.  One block per queued row (grid-strided).
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
  if (tid < nrows) {
    float* r = nrm_tile + tid * ld;
    double s = 0.0;
    for (int c = 0; c < d; ++c) { const double v = (double)r[c]; s = s + v * v; }
    float nr = (float)sqrt(s);
    if (!(nr > eps)) nr = eps;
    for (int c = 0; c < d; ++c) r[c] = __fdiv_rn(r[c], nr);
  }
  __syncthreads();
  float* dst = out + r0 * d;
  for (int t = tid; t < total; t += 256) dst[t] = nrm_tile[(t / d) * ld + (t % d)];
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

__global__ void copy_count_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst) { *dst = *src; }

struct TopkWs {
  float* sl_score; int32_t* sl_idx; int32_t* fb_list; int32_t* fb_count; double* scratch;
  int fb_blocks;
};
constexpr int FB_BLOCKS = 64;
#ifndef KNN_KP_SMALL
#define KNN_KP_SMALL 24      // live shortlist size for k <= KNN_KP_SMALL - 4
#define KNN_CAP_SMALL 48
#endif
static int pick_kp(int k) { return k <= KNN_KP_SMALL - 4 ? KNN_KP_SMALL : (k <= 24 ? 32 : 64); }
constexpr int MAX_SLOTS = 8;

// geometry of pass 1 for a problem: persistent blocks, tiles per block, shortlist slots per query
struct Pass1Plan {
  int64_t nblocks, tpb;
  int nslots;
};
static Pass1Plan plan_pass1(int64_t Nq, int64_t Nc, int64_t resident_blocks, int qpb = QPB) {
  const int64_t ntiles = (Nc + CT - 1) / CT, nqb = (Nq + qpb - 1) / qpb, T = ntiles * nqb;
  Pass1Plan pl;
  pl.nblocks = T < resident_blocks ? T : resident_blocks;
  if (pl.nblocks < 1) pl.nblocks = 1;
  pl.tpb = (T + pl.nblocks - 1) / pl.nblocks;
  const int64_t tpb_min = (ntiles + MAX_SLOTS - 2) / (MAX_SLOTS - 1);   // keep <= MAX_SLOTS shortlists per query
  if (pl.tpb < tpb_min) pl.tpb = tpb_min;
  pl.nblocks = (T + pl.tpb - 1) / pl.tpb;
  pl.nslots = (int)((ntiles + pl.tpb - 1) / pl.tpb) + 1;     // blocks that can touch one query block
  return pl;
}
// upper bound used for workspace sizing when the occupancy query is not yet known: 2 blocks per CU, 256 CUs
constexpr int64_t RESIDENT_MAX = 1024;
static int worst_slots(int64_t Nq, int64_t Nc) {
  int m = 2;
  for (int64_t rb = 1; rb <= RESIDENT_MAX; rb *= 2) {
    const int s1 = plan_pass1(Nq, Nc, rb).nslots, s2 = plan_pass1(Nq, Nc, rb, 2 * QPB).nslots;
    if (s1 > m) m = s1;
    if (s2 > m) m = s2;
  }
  return m;
}

static size_t topk_ws_bytes(int64_t Nq, int64_t Nc, int k) {
  const size_t L = (size_t)worst_slots(Nq, Nc) * pick_kp(k);
  size_t b = 0;
  b += bgnn_align_up(sizeof(float) * Nq * L, 256);
  b += bgnn_align_up(sizeof(int32_t) * Nq * L, 256);
  b += bgnn_align_up(sizeof(int32_t) * (Nq + 1), 256);
  b += 256;
  b += bgnn_align_up(sizeof(double) * FB_BLOCKS * Nc, 256);
  return b + 256;
}
static TopkWs topk_carve(void* ws, int64_t Nq, int64_t Nc, int k) {
  const size_t L = (size_t)worst_slots(Nq, Nc) * pick_kp(k);
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { char* q = p; p += bgnn_align_up(bytes, 256); return q; };
  TopkWs w;
  w.sl_score = (float*)take(sizeof(float) * Nq * L);
  w.sl_idx = (int32_t*)take(sizeof(int32_t) * Nq * L);
  w.fb_list = (int32_t*)take(sizeof(int32_t) * (Nq + 1));
  w.fb_count = (int32_t*)take(256);
  w.scratch = (double*)take(sizeof(double) * FB_BLOCKS * Nc);
  w.fb_blocks = FB_BLOCKS;
  return w;
}

template <class Canon>
static int run_refine(const Canon& canon, int64_t Nq, int64_t Nc, int k, int KP, int nsplit, const TopkWs& w,
                      double err_abs, double err_rel, int apply_sigmoid, int64_t* idx_out, float* val_out,
                      int32_t* n_fallback_opt, hipStream_t st) {
  hipError_t e;
  if ((e = hipMemsetAsync(w.fb_count, 0, sizeof(int32_t), st)) != hipSuccess) return (int)e;
  const int L = KP * nsplit;
  int64_t grid = (Nq + 3) / 4;
  if (grid > 2048) grid = 2048;
  const size_t sh = (size_t)4 * L * (sizeof(double) + sizeof(int32_t));
  hipLaunchKernelGGL((refine_kernel<Canon>), dim3((unsigned)grid), dim3(256), sh, st, canon, Nq, k, L, KP, nsplit,
                     w.sl_score, w.sl_idx, err_abs, err_rel, apply_sigmoid, idx_out, val_out, w.fb_list, w.fb_count);
  BGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL((fallback_kernel<Canon>), dim3(w.fb_blocks), dim3(256), 0, st, canon, Nc, k, apply_sigmoid,
                     w.fb_list, w.fb_count, w.scratch, idx_out, val_out);
  BGNN_LAUNCH_CHECK();
  if (n_fallback_opt) {
    hipLaunchKernelGGL(copy_count_kernel, dim3(1), dim3(1), 0, st, w.fb_count, n_fallback_opt);
    BGNN_LAUNCH_CHECK();
  }
  return 0;
}

template <int DK, int CAPV, int KPV, bool BF3>
static int launch_cosine_pass1(const float* qq, const float* qc, int64_t Nq, int64_t Nc, const TopkWs& w, int* nslots_out,
                               hipStream_t st) {
  constexpr int D = DK * 8, LD = D + 4;
  hipError_t e;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipErrorInvalidDevice;
  static const int db_nw = [] { const char* e = getenv("BGNN_KNN_DB"); return e ? atoi(e) : 8; }();   // 0 = two-barrier 4-wave form
  if constexpr (BF3) {
    constexpr size_t sh8 = (size_t)2 * CPIECES * CT * D * 2 + 8 * WaveTopK<CAPV, KPV>::BYTES;
    if (db_nw == 8 && sh8 <= 160 * 1024) {
      auto k8 = cosine_pass1_db_kernel<DK, CAPV, KPV, 8>;
      static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh8);
      if (attr != hipSuccess) return (int)attr;
      const Pass1Plan pl = plan_pass1(Nq, Nc, prop.multiProcessorCount, 2 * QPB);     // one 8-wave block per CU
      *nslots_out = pl.nslots;
      if ((e = hipMemsetAsync(w.sl_idx, 0xFF, sizeof(int32_t) * Nq * pl.nslots * KPV, st)) != hipSuccess) return (int)e;
      hipLaunchKernelGGL(k8, dim3((unsigned)pl.nblocks), dim3(512), sh8, st, qq, qc, Nq, Nc, pl.tpb, pl.nslots, w.sl_score, w.sl_idx);
      BGNN_LAUNCH_CHECK();
      return 0;
    }
  }
  const size_t sh = (BF3 ? (size_t)CPIECES * CT * D * 2 : sizeof(float) * CT * LD) + WAVES * WaveTopK<CAPV, KPV>::BYTES;
  auto kern = cosine_pass1_kernel<DK, CAPV, KPV, BF3>;
  // immutable per (instantiation, device): how many blocks are co-resident.  The occupancy API prices LDS
  // against 64 KB per CU on ROCm 7.2 and answers 1 here; gfx950 has 160 KB per CU, and this kernel's
  // <= 256 VGPRs allow two waves per SIMD, so the residency is computed from those two budgets.
  static const int resident = [&] {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) return -1;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)) != hipSuccess) return -1;
    int per_cu = (int)((160 * 1024) / sh);          // the kernel has no static LDS
    const int by_regs = fa.numRegs > 0 ? 512 / ((fa.numRegs + 7) / 8 * 8) : 1;     // waves per SIMD = blocks per CU (4 waves)
    if (per_cu > by_regs) per_cu = by_regs;
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    const int r = per_cu * prop.multiProcessorCount;
    return r > (int)RESIDENT_MAX ? (int)RESIDENT_MAX : r;
  }();
  if (resident < 1) return (int)hipErrorInvalidValue;
  const Pass1Plan pl = plan_pass1(Nq, Nc, resident);
  *nslots_out = pl.nslots;
  if ((e = hipMemsetAsync(w.sl_idx, 0xFF, sizeof(int32_t) * Nq * pl.nslots * KPV, st)) != hipSuccess) return (int)e;   // every slot starts empty (-1)
  hipLaunchKernelGGL(kern, dim3((unsigned)pl.nblocks), dim3(256), sh, st, qq, qc, Nq, Nc, pl.tpb, pl.nslots, w.sl_score, w.sl_idx);
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
  return topk_ws_bytes(Nq < 1 ? 1 : Nq, Nc < 1 ? 1 : Nc, k);
}

extern "C" int bgnn_cosine_topk_f32(const float* qn_query, const float* qn_cand, int64_t Nq, int64_t Nc,
                                    int32_t d, int32_t k, int apply_sigmoid, int64_t* idx_out, float* val_out,
                                    int32_t* n_fallback_opt, void* ws, size_t ws_bytes, void* stream) {
  if (!qn_query || !qn_cand || !idx_out || !val_out || !ws) return BGNN_E_NULL;
  if (Nq < 0 || Nc <= 0 || Nc >= (int64_t)1 << 31 || d <= 0) return BGNN_E_SHAPE;
  if (d != 32 && d != 64 && d != 128 && d != 256) return BGNN_E_SHAPE;   // callers zero-pad d
  if (k <= 0 || k > 56 || k > Nc) return BGNN_E_RANGE;
  if (d == 256 && k > 24) return BGNN_E_SHAPE;   // LDS budget (shortlists 128 KB + staging)
  if (!bgnn_aligned16(qn_query) || !bgnn_aligned16(qn_cand)) return BGNN_E_ALIGN;
  if (ws_bytes < topk_ws_bytes(Nq, Nc, k)) return BGNN_E_WORKSPACE;
  if (Nq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  TopkWs w = topk_carve(ws, Nq, Nc, k);
  const int KP = pick_kp(k);
  int nsplit = 1;
  int rc;
  static const bool bf3 = [] { const char* e = getenv("BGNN_KNN_BF3"); return !e || atoi(e) != 0; }();
#define COS(DKV, BF)                                                                                    \
  rc = KP == KNN_KP_SMALL ? launch_cosine_pass1<DKV, KNN_CAP_SMALL, KNN_KP_SMALL, BF>(qn_query, qn_cand, Nq, Nc, w, &nsplit, st) \
     : KP == 32 ? launch_cosine_pass1<DKV, 48, 32, BF>(qn_query, qn_cand, Nq, Nc, w, &nsplit, st)           \
                : launch_cosine_pass1<DKV, 128, 64, BF>(qn_query, qn_cand, Nq, Nc, w, &nsplit, st)
  if (d == 256) { COS(32, false); }            // 256-wide rows keep the fp32 MFMA path (LDS budget)
  else if (bf3) { if (d == 32) { COS(4, true); } else if (d == 64) { COS(8, true); } else { COS(16, true); } }
  else { if (d == 32) { COS(4, false); } else if (d == 64) { COS(8, false); } else { COS(16, false); } }
#undef COS
  if (rc) return rc;
  // |fp32 MFMA dot - exact| <= d * 2^-24 * sum|a_c b_c| <= d * 2^-24 for unit vectors (Cauchy-Schwarz);
  // x2 safety + the fp32 rounding of the stored shortlist score
  // bf16 path: an operand kept as two pieces is exact to 2^-18 relative, as three pieces to 2^-27; for unit vectors the
  // score error is <= (sum of the operands' piece errors + the dropped mid*mid term 2^-18) + d 2^-23 for the fp32
  // accumulation inside the MFMA (counted as truncating).  Two pieces on both sides: 3 * 2^-18 + d 2^-23 = 2.7e-5 at
  // d = 128; err_abs = 2^-16 + 4 (d + 2) 2^-24 = 4.6e-5 keeps a factor ~2 of safety like the fp32 bound does.
  const double err_abs = (bf3 && d != 256) ? (CPIECES > 2 ? 0.0 : 1.52587890625e-05) + 4.0 * (double)(d + 2) * 5.9604644775390625e-08
                                           : 2.0 * (double)(d + 2) * 5.9604644775390625e-08;
  CosineCanon canon{qn_query, qn_cand, d};
  return run_refine(canon, Nq, Nc, k, KP, nsplit, w, err_abs, 0.0, apply_sigmoid, idx_out, val_out, n_fallback_opt, st);
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
  if (ws_bytes < topk_ws_bytes(Nq, Nc, k)) return BGNN_E_WORKSPACE;
  if (Nq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  TopkWs w = topk_carve(ws, Nq, Nc, k);
  const int epl = k <= 24 ? 1 : 2;
  const int KP = 32 * epl;
  {
    const size_t sh = sizeof(float) * (CT * (MLP_H + 4) + 3 * MLP_H) +
                      WAVES * (epl == 1 ? WaveTopK<64, 32>::BYTES : WaveTopK<128, 64>::BYTES);
    const unsigned grid = (unsigned)((Nq + QPB - 1) / QPB);
    hipError_t e;
    if (epl == 1) {
      auto kern = mlp_pass1_kernel<1>;
      if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)) != hipSuccess) return (int)e;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), sh, st, A_cand, B_query, bn_scale, bn_shift, w2, b2, Nq, Nc, w.sl_score, w.sl_idx);
    } else {
      auto kern = mlp_pass1_kernel<2>;
      if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)) != hipSuccess) return (int)e;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), sh, st, A_cand, B_query, bn_scale, bn_shift, w2, b2, Nq, Nc, w.sl_score, w.sl_idx);
    }
    BGNN_LAUNCH_CHECK();
  }
  // fp32 evaluation of a 128-term sum of O(1) terms: relative bound on the logit magnitude plus an
  // absolute floor; generous (a loose bound only costs a few more exhaustive rows)
  MlpCanon canon{A_cand, B_query, bn_scale, bn_shift, w2, b2, H};
  return run_refine(canon, Nq, Nc, k, KP, 1, w, 1e-4, 1e-4, apply_sigmoid, idx_out, val_out, n_fallback_opt, st);
}

#if defined(KNN_EXP) && (KNN_EXP == 8 || KNN_EXP == 9 || KNN_EXP == 10)
extern "C" int bgnn_debug_knn_blocks(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_knn_blk), sizeof(unsigned long long) * 1024 * 3);
}
extern "C" int bgnn_debug_knn_counters(unsigned long long* host_out) {
  hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_knn_cnt), sizeof(unsigned long long) * 8);
  return (int)e;
}
#endif
