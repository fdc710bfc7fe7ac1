// Tall-skinny Gram product for the training path:  out[a][b] = sum_i A[i][a] * B[i][b]   (A [N,p], B [N,q], N ~ 1e6).
//
// Where it sits (reference Bridged-GNN/models/KTGNN.py:275-284 under autograd): the weight / gate gradients of the
// AdaptedConv dense transform are  [G_s2t | G_t2s | dgate]^T . x  -- a [<=260, N] x [N, <=128] product whose reduction
// dimension is the node count.  The library GEMM picks 16..32-row macro tiles for this shape (2 ms for 66 GFLOP); here
// every persistent block streams its slice of the rows ONCE through LDS (coalesced 16-B loads, double buffered), keeps
// the whole p x q result of its slice in MFMA accumulators (wave w owns output columns 32w..32w+31 and all <= 9 row
// blocks: 144 VGPRs) and writes one partial; a second tiny kernel sums the partials in a fixed order (deterministic, no
// atomics).  HBM-bound: (p + q) * 4 bytes per node, read once.
#include "bgnn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int GR_RT = 32;       // rows per LDS tile
constexpr int GR_PA = 9;        // max 32-row blocks of the output (p <= 288)

struct GramParams {
  const float* A; int64_t lda; int32_t p;
  const float* B; int64_t ldb; int32_t q;
  int64_t N;
  float* part;                  // [gridDim.x][p][q]
};

template <int PA>
__global__ __launch_bounds__(256) void gram_partial_kernel(GramParams g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];     // [2][GR_RT][ldA + ldB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int pa = (g.p + 31) / 32;
  const int ldA = pa * 32, ldB = 128 + ((pa & 1) ? 0 : 32);   // LDS rows, zero-filled past p / q; row stride = 32 (mod 64)
  const int ldT = ldA + ldB;                                  // floats: the two rows of one MFMA step use disjoint banks
  const int nA4 = g.p / 4, nB4 = g.q / 4;                // p, q % 4 == 0 (host)
  const int64_t rows_per_blk = (g.N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r1 = min(g.N, r0 + rows_per_blk);
  const int ntile = r1 > r0 ? (int)((r1 - r0 + GR_RT - 1) / GR_RT) : 0;

  // zero the padding columns once (they are never written by the staging loop)
  for (int t = tid; t < 2 * GR_RT * ldT; t += 256) sm[t] = 0.f;
  __syncthreads();

  // staging: float4 slots of a tile = GR_RT * (nA4 + nB4), strided over the block; <= 13 per thread for p=260, q=128.
  // slot -> (row, column) is tile-invariant: the integer divisions happen once, outside the row loop (fp32 MFMA shares
  // the VALU, so staging arithmetic competes with it directly)
  const int n4 = nA4 + nB4, nslot = GR_RT * n4;
  constexpr int MAXS = PA == 1 ? 5 : PA <= 4 ? 8 : 13;   // (32 * (p + q) / 4) slots over 256 threads
  int srow[MAXS], slds[MAXS], scol[MAXS];       // row inside the tile (-1: unused slot), LDS float offset, global column
  unsigned int isA = 0;                          // bit j: slot j belongs to A
#pragma unroll
  for (int j = 0; j < MAXS; ++j) {
    const int sidx = tid + 256 * j;
    srow[j] = -1; slds[j] = 0; scol[j] = 0;
    if (sidx < nslot) {
      const int r = sidx / n4, c = sidx % n4;
      srow[j] = r;
      if (c < nA4) { isA |= 1u << j; scol[j] = c * 4; slds[j] = r * ldT + c * 4; }
      else { scol[j] = (c - nA4) * 4; slds[j] = r * ldT + ldA + (c - nA4) * 4; }
    }
  }
  float4 st[MAXS];
  auto gload = [&](int t) {
#pragma unroll
    for (int j = 0; j < MAXS; ++j) {
      st[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (srow[j] >= 0) {
        const int64_t row = r0 + (int64_t)t * GR_RT + srow[j];
        const bool a = (isA >> j) & 1u;
        const float4 v = *reinterpret_cast<const float4*>((a ? g.A : g.B) + (row < r1 ? row : r1 - 1) * (a ? g.lda : g.ldb) + scol[j]);   // tail rows re-read r1-1
        if (row < r1) st[j] = v;
      }
    }
  };
  auto sstore = [&](int buf) {
    float* base = sm + buf * GR_RT * ldT;
#pragma unroll
    for (int j = 0; j < MAXS; ++j)
      if (srow[j] >= 0) *reinterpret_cast<float4*>(base + slds[j]) = st[j];
  };

  f32x16 acc[PA];
#pragma unroll
  for (int a = 0; a < PA; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const bool wave_on = wave * 32 < g.q;                  // wave-uniform

  if (ntile > 0) { gload(0); sstore(0); }
  if (ntile > 1) gload(1);
  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    __syncthreads();                                     // tile t complete in `cur`; nobody reads cur^1 any more
    if (t + 1 < ntile) sstore(cur ^ 1);
    if (t + 2 < ntile) gload(t + 2);
    if (wave_on) {
      const float* base = sm + cur * GR_RT * ldT;
#pragma unroll 4
      for (int k = 0; k < GR_RT; k += 2) {
        // D[i = A column][j = B column] += A[row k+fh][i] * B[row k+fh][j]
        const float* rowp = base + (k + fh) * ldT;
        const float b = rowp[ldA + wave * 32 + fr];
#pragma unroll
        for (int a = 0; a < PA; ++a)
          if (a < pa) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(rowp[a * 32 + fr], b, acc[a], 0, 0, 0);
      }
    }
  }
  // partial [p][q]: accumulator register r of lane (fr, fh) is D[i = (r&3) + 8(r>>2) + 4fh][j = fr]
  if (wave_on) {
    float* out = g.part + (int64_t)blockIdx.x * g.p * g.q;
    const int col = wave * 32 + fr;
#pragma unroll
    for (int a = 0; a < PA; ++a)
      if (a < pa) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
          if (i < g.p && col < g.q) out[(int64_t)i * g.q + col] = acc[a][r];
        }
      }
  }
}

// bf16 x 3 variant for p > 32 (the hidden conv's [260, N] x [N, 128] and the Linears' [128, N] x [N, 128] weight gradients).
// fp32 MFMA runs on the vector ALU (64 cycles per 32x32x2 step, and the staging arithmetic competes with it: 0.98 ms for
// p = 260); here every fp32 value is split exactly into three bf16 pieces while it is staged (hi + mid + lo = the 24-bit
// significand) and a product is the six bf16 MFMAs whose pieces are >= 2^-24 relative -- the matrix cores do 16 rows of
// the reduction in 6 x 32 cycles instead of 8 x 64.  The reduction index (the node) is the ROW of both operands in memory, but
// an MFMA operand wants 8 consecutive reduction indices per lane: the LDS image stays row-major [piece][node][column]
// (8-byte writes of 4 columns) and the operands are fetched with the transposing LDS read `ds_read_b64_tr_b16` (per 16 lanes
// a 4 x 16 block, delivered column-major).  Row pitch = 64 B (mod 256 B): the four rows of a block fall on disjoint banks.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int GB_RT = 16;       // nodes per LDS tile = one k-step of v_mfma_f32_32x32x16_bf16

__host__ __device__ inline int gram_bf16_pitch(int pa) {       // bf16 elements per row: 32 (mod 128)
  int P = pa * 32 + 128;
  while (P % 128 != 32) P += 32;
  return P;
}

__device__ __forceinline__ bf16x8 tr_read8(const __bf16* base /* this lane's address for rows 0..3 */, int row4_stride) {
  // two transposing reads: reduction rows 8fh + 0..3 and 8fh + 4..7 of this lane's column
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + row4_stride));
  const bf16x4 l = __builtin_bit_cast(bf16x4, lo), h = __builtin_bit_cast(bf16x4, hi);
  bf16x8 r;
  r[0] = l[0]; r[1] = l[1]; r[2] = l[2]; r[3] = l[3]; r[4] = h[0]; r[5] = h[1]; r[6] = h[2]; r[7] = h[3];
  return r;
}

// Eight waves with two roles (wave-uniform): waves 0..3 only multiply -- wave w owns output columns 32w..32w+31 and all PA row
// blocks (the MFMA chains of two row blocks interleaved: a lone dependent bf16 chain issues at half rate) -- and waves 4..7
// only stage (global loads two tiles ahead, the three-way split, LDS writes).  A SIMD then holds one wave of each kind and the
// conversion arithmetic runs beside the matrix work instead of in front of it (same-role waves between two barriers run the
// two halves one after the other: measured 0.65 ms for p = 260; opposite-order halves in one code path spilled).
template <int PA>
__global__ __launch_bounds__(512) void gram_bf16_kernel(GramParams g) {
  constexpr int NT = 256;                               // threads of one role
  extern __shared__ __attribute__((aligned(16))) float sm[];     // bf16 [2][3][GB_RT][P]
  __bf16* smb = reinterpret_cast<__bf16*>(sm);
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool stager = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) != 0;
  const int fr = lane & 31, fh = lane >> 5;
  const int pa = (g.p + 31) / 32;
  const int ldA = pa * 32;
  const int P = gram_bf16_pitch(pa);
  const int piece = GB_RT * P, bufsz = 3 * piece;       // bf16 elements
  const int nA4 = g.p / 4, nB4 = g.q / 4;
  const int64_t rows_per_blk = (g.N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r1 = min(g.N, r0 + rows_per_blk);
  const int ntile = r1 > r0 ? (int)((r1 - r0 + GB_RT - 1) / GB_RT) : 0;
  for (int t = threadIdx.x; t < bufsz; t += 512) sm[t] = 0.f;    // (2 buffers x bufsz bf16 = bufsz floats) padding columns stay zero
  __syncthreads();

  if (stager) {
    const int n4 = nA4 + nB4, nslot = GB_RT * n4;
    constexpr int MAXS = (GB_RT * (PA * 32 + 128) / 4 + NT - 1) / NT;
    int srow[MAXS], slds[MAXS], scol[MAXS];
    unsigned int isA = 0;
#pragma unroll
    for (int j = 0; j < MAXS; ++j) {
      // slots past the tile wrap around: such a thread stages an element a second time (the same value to the same place)
      // instead of branching around a load / store
      const int sidx = (tid + NT * j) % nslot;
      {
        const int r = sidx / n4, c = sidx % n4;
        srow[j] = r;
        if (c < nA4) { isA |= 1u << j; scol[j] = c * 4; slds[j] = r * P + c * 4; }
        else { scol[j] = (c - nA4) * 4; slds[j] = r * P + ldA + (c - nA4) * 4; }
      }
    }
    // two register sets: tiles u+2 and u+3 are in flight while tile u is multiplied.  The loads are branch-free (tail rows
    // re-read a valid row; their values are zeroed when they are stored): a load inside a lane-masked branch
    // whose result is assigned there makes the compiler wait for it on the spot, i.e. one load in flight at a time
    float4 stA[MAXS], stB[MAXS];
    unsigned int okA = 0, okB = 0;
    auto gload = [&](int t, float4 (&st)[MAXS], unsigned int& okm) {
      okm = 0;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) {
        const int64_t row = r0 + (int64_t)t * GB_RT + srow[j];
        const bool a = (isA >> j) & 1u;
        okm |= row < r1 ? 1u << j : 0u;
        st[j] = *reinterpret_cast<const float4*>((a ? g.A : g.B) + (row < r1 ? row : r1 - 1) * (a ? g.lda : g.ldb) + scol[j]);
      }
    };
    auto sstore = [&](int buf, const float4 (&st)[MAXS], unsigned int okm) {
      __bf16* base = smb + buf * bufsz;
#pragma unroll
      for (int j = 0; j < MAXS; ++j) {                   // branch-free (see gload)
        {
          const bool ok = (okm >> j) & 1u;
          const float vf[4] = {ok ? st[j].x : 0.f, ok ? st[j].y : 0.f, ok ? st[j].z : 0.f, ok ? st[j].w : 0.f};
          bf16x4 ph, pm, pl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const __bf16 h = (__bf16)vf[e];
            const float r1_ = vf[e] - (float)h;
            const __bf16 m = (__bf16)r1_;
            ph[e] = h; pm[e] = m; pl[e] = (__bf16)(r1_ - (float)m);
          }
          *reinterpret_cast<bf16x4*>(base + slds[j]) = ph;
          *reinterpret_cast<bf16x4*>(base + piece + slds[j]) = pm;
          *reinterpret_cast<bf16x4*>(base + 2 * piece + slds[j]) = pl;
        }
      }
    };
    if (ntile > 0) { gload(0, stA, okA); sstore(0, stA, okA); }
    if (ntile > 1) gload(1, stA, okA);
    if (ntile > 2) gload(2, stB, okB);
    int t = 0;
    for (; t + 4 < ntile; t += 2) {                      // steady state without conditions: exact s_waitcnt vmcnt counts
      __syncthreads();                                   // tile t complete in buffer 0; nobody reads buffer 1 any more
      sstore(1, stA, okA);
      gload(t + 3, stA, okA);
      __syncthreads();
      sstore(0, stB, okB);
      gload(t + 4, stB, okB);
    }
    for (; t < ntile; t += 2) {
      __syncthreads();
      if (t + 1 < ntile) sstore(1, stA, okA);
      if (t + 3 < ntile) gload(t + 3, stA, okA);
      if (t + 1 < ntile) {
        __syncthreads();
        if (t + 2 < ntile) sstore(0, stB, okB);
        if (t + 4 < ntile) gload(t + 4, stB, okB);
      }
    }
    return;
  }

  f32x16 acc[PA];
#pragma unroll
  for (int a = 0; a < PA; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const bool wave_on = wave * 32 < g.q;                  // wave-uniform
  // transposing-read address of this lane inside a 32-column operand block: 16-lane group gq = lane >> 4 covers columns
  // 16 (gq & 1) .. +15 and reduction rows 8 (gq >> 1) .. ; lane 4q + pp of the group points at row q, columns 4pp .. 4pp + 3
  const int gq = lane >> 4, i16 = lane & 15;
  const int toff = (8 * (gq >> 1) + (i16 >> 2)) * P + 16 * (gq & 1) + 4 * (i16 & 3);
  for (int t = 0; t < ntile; ++t) {
    __syncthreads();                                     // (the same barrier sequence as the staging waves: one per tile)
    // (every lane of the wave runs the reads: the transposing read needs EXEC all ones; the conditions are wave-uniform)
    if (wave_on) {
      const __bf16* base = smb + (t & 1) * bufsz + toff;
      const __bf16* bb = base + ldA + wave * 32;
      const bf16x8 bh = tr_read8(bb, 4 * P), bm = tr_read8(bb + piece, 4 * P), bl = tr_read8(bb + 2 * piece, 4 * P);
#pragma unroll
      for (int a = 0; a < PA; a += 2) {
        const bool on0 = a < pa, on1 = (a + 1 < PA) && (a + 1 < pa);
        if (on0 && on1) {
          const __bf16* ab = base + a * 32;
          const bf16x8 ah = tr_read8(ab, 4 * P), am = tr_read8(ab + piece, 4 * P), al = tr_read8(ab + 2 * piece, 4 * P);
          const bf16x8 ch = tr_read8(ab + 32, 4 * P), cm = tr_read8(ab + 32 + piece, 4 * P), cl = tr_read8(ab + 32 + 2 * piece, 4 * P);
          const int a1 = a + 1 < PA ? a + 1 : a;           // (unrolled: a compile-time register index)
          // smallest terms first; dropped: mid*lo, lo*mid, lo*lo (< 2^-24 relative)
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[a], 0, 0, 0);
          acc[a1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl, bh, acc[a1], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[a], 0, 0, 0);
          acc[a1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, bl, acc[a1], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[a], 0, 0, 0);
          acc[a1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cm, bm, acc[a1], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[a], 0, 0, 0);
          acc[a1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cm, bh, acc[a1], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[a], 0, 0, 0);
          acc[a1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, bm, acc[a1], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[a], 0, 0, 0);
          acc[a1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, bh, acc[a1], 0, 0, 0);
        } else if (on0) {
          const __bf16* ab = base + a * 32;
          const bf16x8 ah = tr_read8(ab, 4 * P), am = tr_read8(ab + piece, 4 * P), al = tr_read8(ab + 2 * piece, 4 * P);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[a], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[a], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[a], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[a], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[a], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[a], 0, 0, 0);
        }
      }
    }
  }
  if (wave_on) {
    float* out = g.part + (int64_t)blockIdx.x * g.p * g.q;
    const int col = wave * 32 + fr;
#pragma unroll
    for (int a = 0; a < PA; ++a)
      if (a < pa) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
          if (i < g.p && col < g.q) out[(int64_t)i * g.q + col] = acc[a][r];
        }
      }
  }
}

__global__ void gram_reduce_kernel(const float* __restrict__ part, int nblk, int64_t pq, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= pq) return;
  // fixed order (deterministic); 8 independent loads in flight per thread instead of a chain of nblk dependent ones
  double s = 0.0;
  int b = 0;
  for (; b + 8 <= nblk; b += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(b + u) * pq + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u];
  }
  for (; b < nblk; ++b) s += (double)part[(int64_t)b * pq + e];
  out[e] = (float)s;
}

// persistent blocks: as many as fit a CU's 160 KB of LDS (up to 3) x 256 CUs
static size_t gram_lds_bytes(int p) {
  const int pa = (p + 31) / 32;
  return sizeof(float) * 2 * GR_RT * (size_t)(pa * 32 + 128 + ((pa & 1) ? 0 : 32));
}
static size_t gram_bf16_lds_bytes(int p) {
  const int pa = (p + 31) / 32;
  return sizeof(__bf16) * 2 * 3 * GB_RT * (size_t)gram_bf16_pitch(pa);
}
static bool gram_use_bf16(int p) { return p > 32; }
static int gram_blocks(int p) {
  int per_cu = (int)((150 * 1024) / (gram_use_bf16(p) ? gram_bf16_lds_bytes(p) : gram_lds_bytes(p)));
  per_cu = per_cu < 1 ? 1 : per_cu > 3 ? 3 : per_cu;
  return 256 * per_cu;
}

// out[i][j] = X[i,:d] . V[j,:d] for up to 4 vectors at once (gate pre-activations x.a_g and the gates' adjoints
// G.(W delta) of the training path): one HBM stream over X instead of one library GEMV per vector (0.5 ms each).
// 16 lanes own a row (float4 per lane and 64-column chunk), 4 rows per wave and step, 4 steps in flight.
template <int NV>
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ X, int64_t ldx, int64_t N, int d,
                                                     const float* __restrict__ V, int64_t ldv, float* __restrict__ out) {
  constexpr int MAXC = 4;                         // d <= 256
  const int tid = threadIdx.x, l16 = tid & 15;
  const int nch = (d + 63) / 64;
  float4 v[NV][MAXC];
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int k = (c * 16 + l16) * 4;
      v[j][c] = (c < nch && k < d) ? *reinterpret_cast<const float4*>(V + j * ldv + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  const int64_t rows_per_pass = (int64_t)gridDim.x * 16;          // 16 rows per block and pass
  for (int64_t r = (int64_t)blockIdx.x * 16 + (tid >> 4); r < N; r += rows_per_pass) {
    float acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int k = (c * 16 + l16) * 4;
      if (c < nch && k < d) {
        const float4 xv = *reinterpret_cast<const float4*>(X + r * ldx + k);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          acc[j] = fmaf(xv.x, v[j][c].x, acc[j]); acc[j] = fmaf(xv.y, v[j][c].y, acc[j]);
          acc[j] = fmaf(xv.z, v[j][c].z, acc[j]); acc[j] = fmaf(xv.w, v[j][c].w, acc[j]);
        }
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) { const float t = bgnn::group_sum<16>(acc[j]); mine = l16 == j ? t : mine; }
    if (l16 < NV) out[r * NV + l16] = mine;
  }
}

// Backward of the dense transform, row-local part (ktgnn.py: _TransformFn.backward): one stream over x and over the two
// incoming gradient tables replaces ~25 torch element-wise / copy launches per conv.  For row i (S = source domain):
//   pre_g = x_i . gx[g] + gconst[g];  gam_g = tanh(pre_g);  c1 = S ? gam_0 : 0;  c2 = S ? 0 : gam_1
//   dc_0 = G_s2t[i,:D] . wd[0][:D];  dc_1 = G_t2s[i,:D] . wd[1][D:2D]          (adjoints of the gates)
//   dpre = (S ? dc_0 (1 - gam_0^2) : 0,  S ? 0 : dc_1 (1 - gam_1^2))
//   Gall[i] = [G_s2t[i,:D] | G_t2s[i,:D] | dpre_0 dpre_1 | +-1/n_dom | 0...]  (p = pad4(2D+3) columns);  side[i] = (c1, c2, 1, 0)
// 32 lanes own a row (two rows per wave); element-wise indexing over D so that any D works.
template <bool EX, int MAXK>
__global__ __launch_bounds__(256) void transform_bwd_prep_kernel(const float* __restrict__ x, int64_t ldx, int64_t N, int din,
                                                                 const float* __restrict__ G_s2t, const float* __restrict__ G_t2s,
                                                                 int64_t ldg, int D, const uint8_t* __restrict__ mask,
                                                                 const float* __restrict__ gx, const float* __restrict__ gconst,
                                                                 const float* __restrict__ wd, const double* __restrict__ counts,
                                                                 float* __restrict__ Gall, int p, int64_t ld_gall,
                                                                 float* __restrict__ side, int64_t ld_side, float* __restrict__ ex_part) {
  // EX (D <= 128): the N-reductions of Gall against side that the backward needs -- sum_i c1_i G1_i, sum_i c2_i G2_i, the
  // column sums of G1 / G2 (bias gradients) and sum_i dpre_i -- are accumulated here per lane (a lane owns columns
  // l32 + 32k of its rows), reduced over the block in a fixed order and written as one partial [p][4] per block: Gall is
  // not streamed a second time by a side Gram.  Layout of a partial = the entries of ex = Gall^T side that are used.
  extern __shared__ float exs[];                 // EX: [8 row groups][p * 4]
  // MAXK: 32-column chunks of a G row a lane owns (EX: D <= 32 * MAXK; the narrow convs take MAXK = 1 -- with 4 their launches issued three
  // predicated-off copies of every per-column instruction and were VALU-bound at 0.22 ms for 0.5 GB)
  const int tid = threadIdx.x, l32 = tid & 31;
  const float gc0 = gconst[0], gc1 = gconst[1];
  // column 2D+2 of Gall: d(delta)/d(x_i) = +1/n_S on source rows, -1/n_T on target rows -- the input gradient's term
  // through the domain means is then one more rank of the Gall . Wcat product instead of an [N,Din] multiply + add
  const float cS = (float)(1.0 / counts[0]), cT = (float)(-1.0 / counts[1]);
  float eu1[MAXK], eb1[MAXK], eu2[MAXK], eb2[MAXK], es0 = 0.f, es1 = 0.f;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) eu1[k] = eb1[k] = eu2[k] = eb2[k] = 0.f;
  const int64_t rows_per_pass = (int64_t)gridDim.x * 8;             // 8 rows per block and pass
  // A row in two halves so that the loads of R rows are in flight together: written as one body per row the compiler kept every row's
  // loads behind the previous row's stores (load -> 2 us -> arithmetic -> store, one row at a time: the narrow convs' launches ran at
  // 2 TB/s, 0.22 ms for the 0.5 GB of x they read).
  constexpr int R = 4, XK = 1;                                     // rows in flight per 32-lane group; 128-column chunks of x held in registers (wider inputs: loads in the finishing half)
  struct RowRegs { float4 xv[XK]; float g1v[MAXK], g2v[MAXK]; bool S; };
  const bool x_in_regs = din <= 128 * XK;
  auto load_row = [&](RowRegs& q, int64_t r) {
    if (x_in_regs) {
#pragma unroll
      for (int c = 0; c < XK; ++c) {
        const int k = l32 * 4 + 128 * c;
        q.xv[c] = k < din ? *reinterpret_cast<const float4*>(x + r * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if constexpr (EX) {
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int c = l32 + 32 * k;
        q.g1v[k] = c < D ? G_s2t[r * ldg + c] : 0.f;
        q.g2v[k] = c < D ? G_t2s[r * ldg + c] : 0.f;
      }
    }
    q.S = mask[r] != 0;
  };
  auto finish_row = [&](const RowRegs& q, int64_t r) {
    float a0 = 0.f, a1 = 0.f, d0 = 0.f, d1 = 0.f;
    if (x_in_regs) {
#pragma unroll
      for (int c = 0; c < XK; ++c) {
        const int k = l32 * 4 + 128 * c;
        if (k < din) {
          const float4 xv = q.xv[c];
          const float4 u = *reinterpret_cast<const float4*>(gx + k), v = *reinterpret_cast<const float4*>(gx + din + k);
          a0 = fmaf(xv.x, u.x, a0); a0 = fmaf(xv.y, u.y, a0); a0 = fmaf(xv.z, u.z, a0); a0 = fmaf(xv.w, u.w, a0);
          a1 = fmaf(xv.x, v.x, a1); a1 = fmaf(xv.y, v.y, a1); a1 = fmaf(xv.z, v.z, a1); a1 = fmaf(xv.w, v.w, a1);
        }
      }
    } else {
      for (int k = l32 * 4; k < din; k += 128) {                    // din % 4 == 0 (host)
        const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + k);
        const float4 u = *reinterpret_cast<const float4*>(gx + k), v = *reinterpret_cast<const float4*>(gx + din + k);
        a0 = fmaf(xv.x, u.x, a0); a0 = fmaf(xv.y, u.y, a0); a0 = fmaf(xv.z, u.z, a0); a0 = fmaf(xv.w, u.w, a0);
        a1 = fmaf(xv.x, v.x, a1); a1 = fmaf(xv.y, v.y, a1); a1 = fmaf(xv.z, v.z, a1); a1 = fmaf(xv.w, v.w, a1);
      }
    }
    float* go = Gall + r * ld_gall;
    if constexpr (EX) {
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int c = l32 + 32 * k;
        if (c < D) {
          d0 = fmaf(q.g1v[k], wd[c], d0);
          d1 = fmaf(q.g2v[k], wd[2 * D + D + c], d1);
          go[c] = q.g1v[k]; go[D + c] = q.g2v[k];
        }
      }
    } else {
      for (int c = l32; c < D; c += 32) {
        const float g1 = G_s2t[r * ldg + c], g2 = G_t2s[r * ldg + c];
        d0 = fmaf(g1, wd[c], d0);
        d1 = fmaf(g2, wd[2 * D + D + c], d1);
        go[c] = g1; go[D + c] = g2;
      }
    }
    {
      // the four row sums through ONE transposing butterfly (lane l ends with the total of value l & 3; quad broadcasts hand all four back
      // to every lane): 6 selects + 6 DPP adds + a swizzle + 4 broadcasts instead of 4 x (5 adds incl. a swizzle each)
      const bool b0 = l32 & 1, b1 = l32 & 2;
      const float k0 = b0 ? a1 : a0, s0 = b0 ? a0 : a1, k1 = b0 ? d1 : d0, s1 = b0 ? d0 : d1;
      const float u0 = k0 + bgnn::dpp_mov<0xB1>(s0), u1 = k1 + bgnn::dpp_mov<0xB1>(s1);     // lane parity: a0 | a1 and d0 | d1, summed over lane ^ 1
      const float kk = b1 ? u1 : u0, ss = b1 ? u0 : u1;
      float t = kk + bgnn::dpp_mov<0x4E>(ss);                                                   // lane & 3 = 0..3: a0, a1, d0, d1 over the quad
      t += bgnn::dpp_mov<0x124>(t);                                                             // row_ror:4 (keeps lane & 3)
      t += bgnn::dpp_mov<0x128>(t);                                                             // row_ror:8
      t += bgnn::swz_xor16(t);
      a0 = bgnn::dpp_mov<0x00>(t); a1 = bgnn::dpp_mov<0x55>(t); d0 = bgnn::dpp_mov<0xAA>(t); d1 = bgnn::dpp_mov<0xFF>(t);
    }
    const bool S = q.S;
    float g0 = 0.f, g1 = 0.f;
    // the gate values as the FORWARD kernels form them (v_exp_f32 + v_rcp_f32, abs. error < 5e-7; bgnn_transform_params.h: tanh_fast).
    // libm's tanhf on one lane per row was ~150 instructions per pair of rows: half of the launch on the narrow convs.
    g0 = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * (a0 + gc0)) + 1.f);
    g1 = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * (a1 + gc1)) + 1.f);
    if constexpr (EX) {                          // (group_sum leaves the sums in every lane of the row: so are the gate values)
      const float c1 = S ? g0 : 0.f, c2 = S ? 0.f : g1;
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        eu1[k] = fmaf(c1, q.g1v[k], eu1[k]); eb1[k] += q.g1v[k];
        eu2[k] = fmaf(c2, q.g2v[k], eu2[k]); eb2[k] += q.g2v[k];
      }
    }
    if (l32 == 0) {
      const float dp0 = S ? d0 * (1.f - g0 * g0) : 0.f, dp1 = S ? 0.f : d1 * (1.f - g1 * g1);
      go[2 * D] = dp0;
      go[2 * D + 1] = dp1;
      go[2 * D + 2] = S ? cS : cT;
      for (int c = 2 * D + 3; c < p; ++c) go[c] = 0.f;
      if (side != nullptr) *reinterpret_cast<float4*>(side + r * ld_side) = make_float4(S ? g0 : 0.f, S ? 0.f : g1, 1.f, 0.f);
      es0 += dp0; es1 += dp1;
    }
  };
  int64_t r = (int64_t)blockIdx.x * 8 + (tid >> 5);
  for (; r + (R - 1) * rows_per_pass < N; r += R * rows_per_pass) {
    RowRegs q[R];
#pragma unroll
    for (int j = 0; j < R; ++j) load_row(q[j], r + j * rows_per_pass);
#pragma unroll
    for (int j = 0; j < R; ++j) finish_row(q[j], r + j * rows_per_pass);
  }
  for (; r < N; r += rows_per_pass) {
    RowRegs q;
    load_row(q, r);
    finish_row(q, r);
  }
  if constexpr (EX) {
    const int P4 = p * 4, gi = tid >> 5;
    for (int t = tid; t < 8 * P4; t += 256) exs[t] = 0.f;
    __syncthreads();
    float* mine = exs + gi * P4;                 // every (column, slot) of a group's copy has exactly one owner lane
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int c = l32 + 32 * k;
      if (c < D) {
        mine[c * 4 + 0] = eu1[k]; mine[c * 4 + 2] = eb1[k];
        mine[(D + c) * 4 + 1] = eu2[k]; mine[(D + c) * 4 + 2] = eb2[k];
      }
    }
    if (l32 == 0) { mine[(2 * D) * 4 + 2] = es0; mine[(2 * D + 1) * 4 + 2] = es1; }
    __syncthreads();
    for (int t = tid; t < P4; t += 256) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += exs[q * P4 + t];
      ex_part[(int64_t)blockIdx.x * P4 + t] = sum;
    }
  }
}

// The O(D x Din) algebra around the streaming launches of the transform backward (ktgnn.py: _TransformFn.backward), as two
// one-launch kernels instead of ~40 torch element-wise / BLAS-1 launches per conv (each ~5 us on an idle GPU: 0.8 ms of a
// 15 ms training step for KT-GNN's four convs).
// consts: gx = [g1_x; g2_x], gconst = (delta . g1_d, delta . g2_d), wd = [-(W_t delta) | 0 ; 0 | W_s delta]
__global__ __launch_bounds__(256) void transform_bwd_consts_kernel(const float* __restrict__ W_s, const float* __restrict__ W_t,
                                                                   const float* __restrict__ g1, const float* __restrict__ g2,
                                                                   const float* __restrict__ delta, int D, int din,
                                                                   float* __restrict__ gx, float* __restrict__ gconst,
                                                                   float* __restrict__ wd) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < din; k += 256) { gx[k] = g1[k]; gx[din + k] = g2[k]; }
  for (int t = tid; t < 4 * D; t += 256) wd[t] = 0.f;
  __syncthreads();
  for (int row = wave; row < 2 * D + 2; row += 4) {   // one wave per dot product of length din
    const float* v = row < D ? W_t + (int64_t)row * din : row < 2 * D ? W_s + (int64_t)(row - D) * din
                     : row == 2 * D ? g1 + din : g2 + din;
    float a = 0.f;
    for (int k = lane; k < din; k += 64) a = fmaf(v[k], delta[k], a);
    a = bgnn::group_sum<64>(a);
    if (lane == 0) {
      if (row < D) wd[row] = -a;
      else if (row < 2 * D) wd[2 * D + row] = a;      // wd[1][D + c]
      else gconst[row - 2 * D] = a;
    }
  }
}

// finish: from dWall = Gall^T x [p][din] and ex [p][4] (u1 = ex[:D,0], u2 = ex[D:2D,1], sp = ex[2D:2D+2,2], bias sums ex[:,2]):
//   dW_t = dWall[:D] - u1 (x) delta,  dW_s = dWall[D:2D] + u2 (x) delta,  dg = [dWall[2D+g] | sp_g delta],  db_* = ex[.][2],
//   ddl = sp_0 g1_d + sp_1 g2_d - W_t^T u1 + W_s^T u2  (gradient through the domain means), and the operand of the
//   input-gradient launch, written transposed: wcat_t[k][r] = (W_t | W_s | g1_x | g2_x | ddl | 0)[r][k].
__global__ __launch_bounds__(128) void transform_bwd_finish_kernel(const float* __restrict__ dWall, const float* __restrict__ ex,
                                                                   const float* __restrict__ W_s, const float* __restrict__ W_t,
                                                                   const float* __restrict__ g1, const float* __restrict__ g2,
                                                                   const float* __restrict__ delta, int D, int din, int p,
                                                                   float* __restrict__ dW_s, float* __restrict__ dW_t,
                                                                   float* __restrict__ dg1, float* __restrict__ dg2,
                                                                   float* __restrict__ db_s, float* __restrict__ db_t,
                                                                   float* __restrict__ wcat_t, int64_t ld_wcat) {
  const int r = blockIdx.x;                       // row of Wcat / dWall
  const float sp0 = ex[(2 * D) * 4 + 2], sp1 = ex[(2 * D + 1) * 4 + 2];
  if (threadIdx.x == 0) {
    if (r < D) { if (db_t) db_t[r] = ex[r * 4 + 2]; }
    else if (r < 2 * D) { if (db_s) db_s[r - D] = ex[r * 4 + 2]; }
  }
  for (int k = threadIdx.x; k < din; k += 128) {
    float w;
    if (r < D) {
      dW_t[(int64_t)r * din + k] = dWall[(int64_t)r * din + k] - ex[r * 4 + 0] * delta[k];
      w = W_t[(int64_t)r * din + k];
    } else if (r < 2 * D) {
      const int c = r - D;
      dW_s[(int64_t)c * din + k] = dWall[(int64_t)r * din + k] + ex[r * 4 + 1] * delta[k];
      w = W_s[(int64_t)c * din + k];
    } else if (r == 2 * D) {
      dg1[k] = dWall[(int64_t)r * din + k]; dg1[din + k] = sp0 * delta[k];
      w = g1[k];
    } else if (r == 2 * D + 1) {
      dg2[k] = dWall[(int64_t)r * din + k]; dg2[din + k] = sp1 * delta[k];
      w = g2[k];
    } else if (r == 2 * D + 2) {
      float a = sp0 * g1[din + k] + sp1 * g2[din + k];
      for (int c = 0; c < D; ++c) {
        a = fmaf(-W_t[(int64_t)c * din + k], ex[c * 4 + 0], a);
        a = fmaf(W_s[(int64_t)c * din + k], ex[(D + c) * 4 + 1], a);
      }
      w = a;
    } else {
      w = 0.f;
    }
    wcat_t[(int64_t)k * ld_wcat + r] = w;
  }
}

}  // namespace

extern "C" int bgnn_rowdot_f32(const float* X, int64_t ldx, int64_t N, int32_t d, const float* V, int64_t ldv, int32_t nv,
                               float* out, void* stream) {
  if (!X || !V || !out) return BGNN_E_NULL;
  if (N < 0 || d <= 0 || d > 256 || (d & 3) || (ldx & 3) || (ldv & 3) || ldx < d || ldv < d || nv < 1 || nv > 4) return BGNN_E_SHAPE;
  if (!bgnn_aligned16(X) || !bgnn_aligned16(V)) return BGNN_E_ALIGN;
  if (N == 0) return 0;
  int64_t grid = (N + 15) / 16;
  if (grid > 4096) grid = 4096;
  hipStream_t st = (hipStream_t)stream;
  switch (nv) {
    case 1: hipLaunchKernelGGL(rowdot_kernel<1>, dim3((unsigned)grid), dim3(256), 0, st, X, ldx, N, d, V, ldv, out); break;
    case 2: hipLaunchKernelGGL(rowdot_kernel<2>, dim3((unsigned)grid), dim3(256), 0, st, X, ldx, N, d, V, ldv, out); break;
    case 3: hipLaunchKernelGGL(rowdot_kernel<3>, dim3((unsigned)grid), dim3(256), 0, st, X, ldx, N, d, V, ldv, out); break;
    default: hipLaunchKernelGGL(rowdot_kernel<4>, dim3((unsigned)grid), dim3(256), 0, st, X, ldx, N, d, V, ldv, out); break;
  }
  BGNN_LAUNCH_CHECK();
  return 0;
}

// out[e] = sum over the per-block partials part[b][e] in a fixed order (deterministic): 64 consecutive elements per block
// (coalesced 256-byte reads), 16 groups of threads take the partials round-robin, 8 loads in flight each
__global__ __launch_bounds__(1024) void partial_reduce_kernel(const float* __restrict__ part, int nblk, int64_t pq, float* __restrict__ out) {
  __shared__ double red[16][64];
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + el;
  double s = 0.0;
  if (e < pq) {
    int b = grp;
    for (; b + 7 * 16 < nblk; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(b + 16 * u) * pq + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; b < nblk; b += 16) s += (double)part[(int64_t)b * pq + e];
  }
  red[grp][el] = s;
  __syncthreads();
  if (grp == 0 && e < pq) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][el];
    out[e] = (float)t;
  }
}

static int prep_blocks(int64_t N, bool ex, int p = 1 << 20) {
  int64_t grid = (N + 7) / 8;
  const int64_t cap = (ex && p > 16) ? 2048 : 8192;          // EX: one [p][4] partial per block (narrow convs: 128-byte partials, the full grid)
  if (grid > cap) grid = cap;
  return grid < 1 ? 1 : (int)grid;
}

extern "C" size_t bgnn_transform_bwd_prep_workspace_bytes(int64_t N, int32_t p) {
  return sizeof(float) * (size_t)prep_blocks(N > 0 ? N : 0, true, p) * (size_t)(p > 0 ? p : 0) * 4 + 256;
}

extern "C" int bgnn_transform_bwd_prep_f32(const float* x, int64_t ldx, int64_t N, int32_t din, const float* G_s2t,
                                           const float* G_t2s, int64_t ldg, int32_t D, const uint8_t* mask,
                                           const float* gx, const float* gconst, const float* wd, const double* counts,
                                           float* Gall, int32_t p, int64_t ld_gall, float* side_opt, int64_t ld_side,
                                           float* ex_opt, void* ws_opt, size_t ws_bytes, void* stream) {
  if (!x || !G_s2t || !G_t2s || !mask || !gx || !gconst || !wd || !counts || !Gall || (!side_opt && !ex_opt)) return BGNN_E_NULL;
  if (N < 0 || din <= 0 || (din & 3) || (ldx & 3) || ldx < din || D <= 0 || ldg < D || p < 2 * D + 3 || (p & 3) || ld_gall < p ||
      (side_opt && (ld_side < 4 || (ld_side & 3))))
    return BGNN_E_SHAPE;
  if (ex_opt && (D > 128 || !ws_opt)) return ex_opt && D > 128 ? BGNN_E_SHAPE : BGNN_E_NULL;
  if (ex_opt && ws_bytes < bgnn_transform_bwd_prep_workspace_bytes(N, p)) return BGNN_E_WORKSPACE;
  if (!bgnn_aligned16(x) || !bgnn_aligned16(gx) || (side_opt && !bgnn_aligned16(side_opt))) return BGNN_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {
    if (ex_opt && bgnn_zero_async(ex_opt, sizeof(float) * 4 * (size_t)p, st) != hipSuccess) return (int)hipErrorInvalidValue;
    return 0;
  }
  const int grid = prep_blocks(N, ex_opt != nullptr, p);
  if (ex_opt) {
    if (D <= 32)
      hipLaunchKernelGGL((transform_bwd_prep_kernel<true, 1>), dim3((unsigned)grid), dim3(256), sizeof(float) * 8 * 4 * (size_t)p, st, x,
                         ldx, N, din, G_s2t, G_t2s, ldg, D, mask, gx, gconst, wd, counts, Gall, p, ld_gall, side_opt, ld_side,
                         (float*)ws_opt);
    else
      hipLaunchKernelGGL((transform_bwd_prep_kernel<true, 4>), dim3((unsigned)grid), dim3(256), sizeof(float) * 8 * 4 * (size_t)p, st, x,
                         ldx, N, din, G_s2t, G_t2s, ldg, D, mask, gx, gconst, wd, counts, Gall, p, ld_gall, side_opt, ld_side,
                         (float*)ws_opt);
    BGNN_LAUNCH_CHECK();
    const int64_t pq = (int64_t)p * 4;
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((unsigned)((pq + 63) / 64)), dim3(1024), 0, st, (const float*)ws_opt, grid, pq, ex_opt);
  } else {
    hipLaunchKernelGGL((transform_bwd_prep_kernel<false, 4>), dim3((unsigned)grid), dim3(256), 0, st, x, ldx, N, din, G_s2t,
                       G_t2s, ldg, D, mask, gx, gconst, wd, counts, Gall, p, ld_gall, side_opt, ld_side, (float*)nullptr);
  }
  BGNN_LAUNCH_CHECK();
  return 0;
}

extern "C" int bgnn_transform_bwd_consts_f32(const float* W_s, const float* W_t, const float* g1, const float* g2,
                                             const float* delta, int32_t D, int32_t din, float* gx, float* gconst, float* wd,
                                             void* stream) {
  if (!W_s || !W_t || !g1 || !g2 || !delta || !gx || !gconst || !wd) return BGNN_E_NULL;
  if (D <= 0 || din <= 0) return BGNN_E_SHAPE;
  hipLaunchKernelGGL(transform_bwd_consts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, W_s, W_t, g1, g2, delta, D, din,
                     gx, gconst, wd);
  BGNN_LAUNCH_CHECK();
  return 0;
}

extern "C" int bgnn_transform_bwd_finish_f32(const float* dWall, const float* ex, const float* W_s, const float* W_t,
                                             const float* g1, const float* g2, const float* delta, int32_t D, int32_t din,
                                             int32_t p, float* dW_s, float* dW_t, float* dg1, float* dg2, float* db_s_opt,
                                             float* db_t_opt, float* wcat_t, int64_t ld_wcat, void* stream) {
  if (!dWall || !ex || !W_s || !W_t || !g1 || !g2 || !delta || !dW_s || !dW_t || !dg1 || !dg2 || !wcat_t) return BGNN_E_NULL;
  if (D <= 0 || din <= 0 || p < 2 * D + 3 || ld_wcat < p) return BGNN_E_SHAPE;
  hipLaunchKernelGGL(transform_bwd_finish_kernel, dim3((unsigned)p), dim3(128), 0, (hipStream_t)stream, dWall, ex, W_s, W_t, g1,
                     g2, delta, D, din, p, dW_s, dW_t, dg1, dg2, db_s_opt, db_t_opt, wcat_t, ld_wcat);
  BGNN_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t bgnn_gram_workspace_bytes(int32_t p, int32_t q) {
  return sizeof(float) * (size_t)gram_blocks(p > 0 ? p : 1) * (size_t)(p > 0 ? p : 0) * (size_t)(q > 0 ? q : 0) + 256;
}

extern "C" int bgnn_gram_f32(const float* A, int64_t lda, int32_t p, const float* B, int64_t ldb, int32_t q, int64_t N,
                             float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!A || !B || !out || !ws) return BGNN_E_NULL;
  if (N < 0 || p <= 0 || q <= 0 || p > 32 * GR_PA || q > 128 || (p & 3) || (q & 3) || lda < p || ldb < q || (lda & 3) || (ldb & 3))
    return BGNN_E_SHAPE;
  if (!bgnn_aligned16(A) || !bgnn_aligned16(B)) return BGNN_E_ALIGN;
  if (ws_bytes < bgnn_gram_workspace_bytes(p, q)) return BGNN_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t pq = (int64_t)p * q;
  int nblk = (int)((N + GR_RT - 1) / GR_RT);          // (a block's slice is at least one fp32 tile = two bf16 tiles)
  if (nblk > gram_blocks(p)) nblk = gram_blocks(p);
  if (nblk < 1) nblk = 1;
  GramParams g{A, lda, p, B, ldb, q, N, (float*)ws};
  const size_t sh = gram_use_bf16(p) ? gram_bf16_lds_bytes(p) : gram_lds_bytes(p);
  const int pa = (p + 31) / 32;
#define BGNN_GRAM(PA)                                                                                                  \
  do {                                                                                                                 \
    static int attr_done[BGNN_MAX_DEVICES];                                                                            \
    const hipError_t attr = bgnn_set_max_dynamic_lds(reinterpret_cast<const void*>(gram_partial_kernel<PA>),           \
                                                     160 * 1024 - 1024, attr_done);                                    \
    if (attr != hipSuccess) return (int)attr;                                                                          \
    hipLaunchKernelGGL(gram_partial_kernel<PA>, dim3(nblk), dim3(256), sh, st, g);                                     \
  } while (0)
#define BGNN_GRAM_BF16(PA)                                                                                             \
  do {                                                                                                                 \
    static int attr_done[BGNN_MAX_DEVICES];                                                                            \
    const hipError_t attr = bgnn_set_max_dynamic_lds(reinterpret_cast<const void*>(gram_bf16_kernel<PA>),           \
                                                     160 * 1024 - 1024, attr_done);                                    \
    if (attr != hipSuccess) return (int)attr;                                                                          \
    hipLaunchKernelGGL((gram_bf16_kernel<PA>), dim3(nblk), dim3(512), sh, st, g);                                   \
  } while (0)
  if (pa <= 1) BGNN_GRAM(1); else if (pa <= 4) BGNN_GRAM_BF16(4); else BGNN_GRAM_BF16(GR_PA);
#undef BGNN_GRAM
#undef BGNN_GRAM_BF16
  BGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((pq + 255) / 256)), dim3(256), 0, st, (const float*)ws, nblk, pq, out);
  BGNN_LAUNCH_CHECK();
  return 0;
}
