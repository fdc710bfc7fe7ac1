// The classifier stage's dense work in ONE pass over the hidden activation h (KTGNN_no_complement.forward, reference
// Bridged-GNN/models/KTGNN.py:432-434): the narrow (h_t2s, h_s2t) tables of clf_base(h) and clf_target(h) (the transform of
// :275-284 for two convs that share the input) AND stage A of clf_target(clf_transformer(h)) (:407-411, :433): the hidden
// activation a1 = relu(BN(Linear0(h))) kept on chip, its 12 per-row products with the consumer conv's packed rows / gate vectors
// and its per-domain column sums.  Round 2 streamed h twice (transform_skinny_kernel 0.13 ms + transform_wreg_kernel<..,2> 0.25 ms
// on C4); the barrier-free ring of bgnn_transform_stream.hip brought the second no gain (its waves wait at the ring's counters).
//
// Here nobody synchronises at all.  A WAVE owns whole 32-row tiles: it converts its tile into fp16 pieces in a private LDS slot,
// holds the tile's fragments in registers (the MFMA B operand, 64 VGPRs) and runs them against FIVE stationary 32-column operands
// that live in LDS for the whole kernel (the four column tiles of W1' and the skinny tile: 28 rows of narrow weights and gate
// vectors; 80 KB as fp16 hi / lo pieces, XOR-swizzled like the tiles) -- so the 128 activation columns of a row meet in one wave
// and the second stage needs no reduction across waves.  One 4-wave block per CU (one wave per SIMD with the whole 512-register budget; 8 waves under 256 registers spilled 205); the next
// tile's 16 KB of loads are in flight in registers while the current one is worked on.  144 MFMAs per tile and wave
// (5 x 24 + 4 x 6) are the backbone; the vector work of a column tile's tail (activation, column sums, split, second stage) issues
// in the gaps of the next tile's chain.
//
// Split products and scales as in bgnn_transform_stream.hip: rows scaled per row, each stationary operand by ONE power of two
// (a lane holds 16 different columns of a row, so per-column scales do not factor out), hi = fp16(v), lo = fp16(v - hi), three
// MFMAs per k block.
#include <cstdio>
#include <cstdlib>
#include "bgnn_common.h"
#include "bgnn_transform_params.h"

using bgnn_tf::GemmParams;
using bgnn_tf::MAXH;
using bgnn_tf::f32x16;
using bgnn_tf::tanh_fast;

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int CLS_TARGET_EXP = 10;
constexpr int DK = 128, KB16 = DK / 16, NCT = 4;         // four activation column tiles + the skinny tile
constexpr int PIECE = 32 * DK * 2;                       // one fp16 piece of a 32 x 128 operand: 8 KB
constexpr int SLOT = 32 * 36 * 4 + 2 * 32 * 4;           // a wave's private LDS: 32 x 32 transpose scratch | column-sum weights of its rows (S, T)
constexpr int T_LD = 36;                                 // floats per row of the 32 x 32 transpose scratch (inside the slot, once its fragments are in registers)

__device__ __forceinline__ int chunk_off(int r, int c) { return r * 256 + ((c ^ (r & 15)) << 4); }
__device__ __forceinline__ void pow2_scales(float mx, float& sc, float& inv) {
  int E = (__builtin_bit_cast(int, mx) >> 23) & 0xff;
  E = E < 40 ? 40 : E;
  sc = __builtin_bit_cast(float, (127 + CLS_TARGET_EXP + 127 - E) << 23);
  inv = __builtin_bit_cast(float, (E - CLS_TARGET_EXP) << 23);
}

constexpr int CLS_NW = 4;                                // waves per block (two per SIMD: one converts / waits while the other feeds the matrix pipe)

__global__ __launch_bounds__(64 * CLS_NW) void cls_stage_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char wl[(NCT + 1) * 2 * PIECE];      // stationary operands [tile][piece][32][128]
  __shared__ __attribute__((aligned(16))) unsigned char slots[CLS_NW * SLOT];
  __shared__ __attribute__((aligned(16))) h8 w2lds[NCT][2][2][64];     // second-stage operand pieces [column tile][k block][hi, lo][lane]
  __shared__ __attribute__((aligned(16))) float skc[2][32];            // skinny epilogue constants: bias | W.delta of the packed columns
  __shared__ __attribute__((aligned(16))) float bias1[DK];
  __shared__ uint32_t wmax[3];
  __shared__ float red[2][DK + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int NC = p.NC;                        // = 128
  const bool din_full = p.Din == DK;
#ifdef CLS_STAMP
  uint32_t st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t st_last = 0;
#define CLS_MARK(k) do { const uint32_t t__ = (uint32_t)__builtin_amdgcn_s_memtime(); st_acc[k] += t__ - st_last; st_last = t__; } while (0)
#else
#define CLS_MARK(k) do { } while (0)
#endif

  // ------------------------------------------------------------------ stationary operands -> LDS (all 256 threads, once)
  if (tid < 3) wmax[tid] = 0u;
  for (int t = tid; t < 2 * (DK + 1); t += 64 * CLS_NW) (&red[0][0])[t] = 0.f;
  if (tid < 32) { skc[0][tid] = tid < p.sk_NC ? p.sk_bias[tid] : 0.f; skc[1][tid] = tid < p.sk_NC ? p.sk_wd[tid] : 0.f; }
  if (tid < DK) bias1[tid] = tid < NC ? p.bias[tid] : 0.f;
  __syncthreads();
  // row r of stationary tile ct: W1'[32 ct + r] (ct < 4); skinny tile: packed narrow rows, gate vectors at rows 24..27 and their
  // copy 28..31 (both lane halves of the accumulator see them), see transform_skinny_kernel
  auto src_row = [&](int ct, int r) -> const float* {
    if (ct < NCT) return (32 * ct + r) < NC ? p.Wp + (int64_t)(32 * ct + r) * p.Din : nullptr;
    if (r < p.sk_NC) return p.sk_Wp + (int64_t)r * p.Din;
    if (r >= 24 && ((r - 24) & 3) < 2 * p.sk_heads) return p.sk_g + (int64_t)((r - 24) & 3) * 2 * p.Din;
    return nullptr;
  };
  // item = (tile, row, 8-float chunk): 5 * 32 * 16 = 2560 items, 10 per thread
  float mx1 = 0.f, mxs = 0.f;
  for (int it = tid; it < (NCT + 1) * 32 * 16; it += 64 * CLS_NW) {
    const int ct = it / 512, r = (it >> 4) & 31, c = it & 15;
    const float* s = src_row(ct, r);
    float m = 0.f;
    if (s != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const int k = 8 * c + e; m = fmaxf(m, k < p.Din ? fabsf(s[k]) : 0.f); }
    }
    if (ct < NCT) mx1 = fmaxf(mx1, m); else mxs = fmaxf(mxs, m);
  }
  // second-stage operand (rows of w2 | g2): this wave's registers, one scale for the matrix
  float w2v[NCT][16];
  float mx2 = 0.f;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = 32 * ct + 8 * (r >> 2) + 4 * fh + (r & 3);
      float v = 0.f;
      if (c < NC) {
        if (fr < 8) v = p.w2[(int64_t)fr * NC + c];
        else if (fr < 10) v = p.g2[(int64_t)(fr - 8) * 2 * NC + c];
      }
      w2v[ct][r] = v;
      mx2 = fmaxf(mx2, fabsf(v));
    }
  mx1 = bgnn::group_max<64>(mx1); mxs = bgnn::group_max<64>(mxs); mx2 = bgnn::group_max<64>(mx2);
  if (lane == 0) {
    atomicMax(&wmax[0], __builtin_bit_cast(unsigned, mx1));
    atomicMax(&wmax[1], __builtin_bit_cast(unsigned, mxs));
    atomicMax(&wmax[2], __builtin_bit_cast(unsigned, mx2));
  }
  __syncthreads();
  float scW, cinvW, scS, cinvS, sc2w, cinv2;
  pow2_scales(__builtin_bit_cast(float, wmax[0]), scW, cinvW);
  pow2_scales(__builtin_bit_cast(float, wmax[1]), scS, cinvS);
  pow2_scales(__builtin_bit_cast(float, wmax[2]), sc2w, cinv2);
  for (int it = tid; it < (NCT + 1) * 32 * 16; it += 64 * CLS_NW) {
    const int ct = it / 512, r = (it >> 4) & 31, c = it & 15;
    const float* s = src_row(ct, r);
    const float sc = ct < NCT ? scW : scS;
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * c + e;
      const float v = (s != nullptr && k < p.Din) ? s[k] : 0.f;
      const _Float16 h = (_Float16)(v * sc);
      hi[e] = h;
      lo[e] = (_Float16)fmaf(v, sc, -(float)h);
    }
    unsigned char* base = wl + (ct * 2) * PIECE + chunk_off(r, c);
    *reinterpret_cast<h8*>(base) = hi;
    *reinterpret_cast<h8*>(base + PIECE) = lo;
  }
  if (wave == 0) {                            // (every wave computed the same values; one writes them)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      h8 hh[2], ll[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const _Float16 h = (_Float16)(w2v[ct][r] * sc2w);
        hh[r >> 3][r & 7] = h;
        ll[r >> 3][r & 7] = (_Float16)fmaf(w2v[ct][r], sc2w, -(float)h);
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) { w2lds[ct][kb][0][lane] = hh[kb]; w2lds[ct][kb][1][lane] = ll[kb]; }
    }
  }
  __syncthreads();                            // the last block-wide barrier: from here on every wave is on its own

  // ------------------------------------------------------------------ skinny epilogue constants (as transform_skinny_kernel):
  // accumulator registers 4q..4q+3 of a lane are packed columns 8q + 4fh + (0..3) of row fr (q < 3); 12..15 the gate products
  float* optr[3];
  int hsel[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int c = 8 * q + 4 * fh;
    optr[q] = nullptr; hsel[q] = 0;
    if (c < p.sk_NC) {
      const int ld2 = 2 * (int)p.sk_ldh;
      const int h = c / ld2, rem = c % ld2, t = rem >= p.sk_ldh ? 1 : 0;
      float* base = h == 0 ? (t == 0 ? p.sk_out[0][0] : p.sk_out[0][1]) : (t == 0 ? p.sk_out[1][0] : p.sk_out[1][1]);
      optr[q] = base + (rem - t * (int)p.sk_ldh);
      hsel[q] = h * 2 + t;
    }
  }
  float gcs[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) gcs[h] = h < 2 * p.sk_heads ? p.sk_gc[h] : 0.f;

  // ------------------------------------------------------------------ the wave's tiles
  unsigned char* const sb = slots + wave * SLOT;
  float* const tsc = reinterpret_cast<float*>(sb);                     // transpose scratch
  float* const cfs = reinterpret_cast<float*>(sb + 32 * T_LD * 4);     // [0..31] / [32..63]: 1.0 for an existing source- / target-domain row
  const int roff0 = chunk_off(fr, fh);
  auto roff = [&](int kb) { return roff0 ^ (kb << 5); };
  const int64_t ntiles = (p.N + 31) / 32;
  const int64_t stride = (int64_t)gridDim.x * CLS_NW;
  int64_t tile = (int64_t)blockIdx.x * CLS_NW + wave;
  // The tile is loaded straight in the MFMA operand layout: lane (fr, fh) takes elements 16kb + 8fh .. +7 (kb < 8) of row fr -- 32
  // bytes per k block, the two lane halves of a row together one 64-byte segment, so a row's 128-byte lines are fetched once
  // (the second half hits the L1).  No LDS round trip, no cross-lane reduction beyond one exchange between the two halves.
  float4 ra[KB16][2];
  uint8_t rmk = 0;
  auto gload = [&](int64_t tl) {              // branch-free: rows past N re-read row N-1 (never stored), chunks past Din chunk 0
    int64_t r = tl * 32 + fr;
    r = r < p.N ? r : p.N - 1;
    const float* src = p.x + r * p.ldx;
#pragma unroll
    for (int kb = 0; kb < KB16; ++kb)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int k = 16 * kb + 8 * fh + 4 * c;
        ra[kb][c] = *reinterpret_cast<const float4*>(src + (din_full || k < p.Din ? k : 0));
      }
    rmk = p.mask[r];
  };
  // column sums of the activation: lane (c = fr, fh) accumulates column 32ct + c over rows 16fh .. 16fh+15 of every tile (the
  // activation tile goes through a 32 x 32 transpose in the slot; per-lane sums of all 16 registers cost 128 VGPRs and spilled)
  float cs_s[NCT], cs_t[NCT], cnt_s = 0.f, cnt_t = 0.f;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) cs_s[ct] = cs_t[ct] = 0.f;

  if (tile < ntiles) gload(tile);
  for (; tile < ntiles; tile += stride) {
    // ---- (1) the tile becomes fp16 fragments in registers (row scale from the row's two lane halves, split); the next tile's loads
    //          take the place of the fp32 values
    CLS_MARK(0);
    const bool sdom = rmk != 0;
    const int64_t row = tile * 32 + fr;
    const bool valid = row < p.N;
    float mx = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB16; ++kb)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (!din_full && 16 * kb + 8 * fh + 4 * c >= p.Din) ra[kb][c] = make_float4(0.f, 0.f, 0.f, 0.f);
        mx = fmaxf(fmaxf(fabsf(ra[kb][c].x), fabsf(ra[kb][c].y)), mx);
        mx = fmaxf(fmaxf(fabsf(ra[kb][c].z), fabsf(ra[kb][c].w)), mx);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float scx, s_row;
    pow2_scales(mx, scx, s_row);
    h8 xh[KB16], xl[KB16];                    // B operand: row fr, k = 16kb + 8fh .. +7, for all five stationary operands
#pragma unroll
    for (int kb = 0; kb < KB16; ++kb)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float f[4] = {ra[kb][c].x, ra[kb][c].y, ra[kb][c].z, ra[kb][c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const _Float16 h = (_Float16)(f[e] * scx);
          xh[kb][4 * c + e] = h;
          xl[kb][4 * c + e] = (_Float16)fmaf(f[e], scx, -(float)h);
        }
      }
    const float dom = sdom ? 1.f : 0.f;
    if (fh == 0) {
      cfs[fr] = (valid && sdom) ? 1.f : 0.f;
      cfs[32 + fr] = (valid && !sdom) ? 1.f : 0.f;
      cnt_s += (valid && sdom) ? 1.f : 0.f;
      cnt_t += (valid && !sdom) ? 1.f : 0.f;
    }
    const int64_t tnext = tile + stride < ntiles ? tile + stride : tile;      // past the end: re-read (never used)
    gload(tnext);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");           // the wave's own LDS writes before its reads
    // The stationary operand's fragments are requested HALF A CHAIN ahead (two buffers of 4 k blocks = 2 x 32 registers; the
    // sequence of ten half chains per tile is skinny, tile 0 .. tile 3): left to the compiler every k block was read -> wait ->
    // 3 MFMAs through one register pair, i.e. 40 exposed LDS round trips per tile with nothing else on the SIMD to cover them.
    constexpr int HK = KB16 / 2;
    h8 wf[2][HK][2];
    auto fetch_w = [&](int hc, int buf) {     // half chain hc = 2 * operand + half; operand 0 = skinny tile (LDS tile NCT), 1.. = tiles 0..
      const int op = hc >> 1, half = hc & 1;
      const unsigned char* wb = wl + ((op == 0 ? NCT : op - 1) * 2) * PIECE;
#pragma unroll
      for (int j = 0; j < HK; ++j) {
        wf[buf][j][0] = *reinterpret_cast<const h8*>(wb + roff(HK * half + j));
        wf[buf][j][1] = *reinterpret_cast<const h8*>(wb + PIECE + roff(HK * half + j));
      }
    };
    // `between(kb)`: vector work dropped behind the three MFMAs of k block kb (the second stage of the PREVIOUS column tile: with one
    // wave per SIMD nothing else fills the chain's issue gaps)
    auto half_chain = [&](int hc, f32x16& acc, auto&& between) {          // prefetches half chain hc + 1, runs half chain hc (buffer hc & 1)
      if (hc + 1 < 2 * (NCT + 1)) fetch_w(hc + 1, (hc + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      const int half = hc & 1, buf = hc & 1;
#pragma unroll
      for (int j = 0; j < HK; ++j) {
        const int kb = HK * half + j;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[buf][j][0], xl[kb], acc, 0, 0, 0);      // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[buf][j][1], xh[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[buf][j][0], xh[kb], acc, 0, 0, 0);
        between(kb);
      }
    };
    auto nothing = [](int) {};
    CLS_MARK(1);                              // convert (+ wait for the tile's loads)
    fetch_w(0, 0);
    // ---- (3) the skinny tile (needs only the fragment registers; the slot becomes scratch afterwards): narrow tables of the two convs on h itself (KTGNN.py:277-284 by linearity, see bgnn_transform.hip)
    {                                         // (without a skinny operand its LDS tile is zero and nothing is stored)
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      half_chain(0, acc, nothing);
      half_chain(1, acc, nothing);
      CLS_MARK(2);                            // skinny chain
      const float ss = s_row * cinvS;
      if (valid) {
        float cf = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if (optr[q] != nullptr) {
            if (q == 0 || hsel[q] != hsel[q - 1]) {
              const bool t1 = hsel[q] & 1;
              const float pre = hsel[q] == 0 ? acc[12] : hsel[q] == 1 ? acc[13] : hsel[q] == 2 ? acc[14] : acc[15];
              const float gcv = hsel[q] == 0 ? gcs[0] : hsel[q] == 1 ? gcs[1] : hsel[q] == 2 ? gcs[2] : gcs[3];
              cf = ((dom != 0.f) != t1) ? tanh_fast(fmaf(pre, ss, gcv)) : 0.f;
              cf = t1 ? cf : -cf;
            }
            const float4 bq = *reinterpret_cast<const float4*>(&skc[0][8 * q + 4 * fh]), wq = *reinterpret_cast<const float4*>(&skc[1][8 * q + 4 * fh]);
            float4 o;
            o.x = fmaf(cf, wq.x, fmaf(acc[4 * q], ss, bq.x));     o.y = fmaf(cf, wq.y, fmaf(acc[4 * q + 1], ss, bq.y));
            o.z = fmaf(cf, wq.z, fmaf(acc[4 * q + 2], ss, bq.z)); o.w = fmaf(cf, wq.w, fmaf(acc[4 * q + 3], ss, bq.w));
            *reinterpret_cast<float4*>(optr[q] + row * p.sk_row_stride) = o;
          }
        }
      }
    }
    // ---- (4) activation column tiles -> second stage; accumulator register r of a lane = column 32ct + 8(r/4) + 4fh + r%4 of row fr
    float o2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o2[i] = 0.f;
    const float sc1 = s_row * cinvW;
    // second stage of a column tile, in eight steps (one per k block of the NEXT tile's chain): row scale of the activation ->
    // fp16 pieces -> 6 MFMAs against the tile's slice of (w2 | g2) -> the 12 per-row products; column sums from the transposed tile
    struct Stage2 { float a1[16]; float mx, sc2, inv2; h8 ah[2], al[2]; f32x16 acc2; h8 w2h[2], w2l[2]; };
    auto stage2_step = [&](Stage2& q, int ct, int j) {
      if (j == 0) {
        q.mx = fmaxf(q.mx, __shfl_xor(q.mx, 32));          // the row's two lane halves share the scale
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { q.w2h[kb] = w2lds[ct][kb][0][lane]; q.w2l[kb] = w2lds[ct][kb][1][lane]; }
      } else if (j == 1) {
        pow2_scales(q.mx, q.sc2, q.inv2);
#pragma unroll
        for (int r = 0; r < 16; ++r) q.acc2[r] = 0.f;
      } else if (j == 2 || j == 3) {
        const int kb = j - 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const _Float16 h = (_Float16)(q.a1[8 * kb + e] * q.sc2);
          q.ah[kb][e] = h;
          q.al[kb][e] = (_Float16)fmaf(q.a1[8 * kb + e], q.sc2, -(float)h);
        }
      } else if (j == 4 || j == 5) {
        const int kb = j - 4;
        q.acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q.w2h[kb], q.al[kb], q.acc2, 0, 0, 0);
        q.acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q.w2l[kb], q.ah[kb], q.acc2, 0, 0, 0);
        q.acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q.w2h[kb], q.ah[kb], q.acc2, 0, 0, 0);
      } else if (j == 6) {
        // column sums: this lane's column of the transposed tile, its 16 rows (written by this wave before the chain started)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = tsc[(16 * fh + r) * T_LD + fr];
          cs_s[ct] = fmaf(cfs[16 * fh + r], v, cs_s[ct]);  // (weights of the lane's 16 rows: LDS broadcasts, no registers held)
          cs_t[ct] = fmaf(cfs[32 + 16 * fh + r], v, cs_t[ct]);
        }
      } else {
        const float s2 = q.inv2 * cinv2;
#pragma unroll
        for (int i = 0; i < 8; ++i) o2[i] = fmaf(q.acc2[i], s2, o2[i]);
      }
    };
    Stage2 q2;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    CLS_MARK(3);                              // skinny epilogue
    half_chain(2, acc, nothing);
    half_chain(3, acc, nothing);
    CLS_MARK(4);                              // first activation chain (nothing behind it)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      // activation of column tile ct from its finished accumulators; the tile also goes (transposed) through the scratch
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");          // the previous tile's column-sum reads are done
      q2.mx = 0.f;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float4 b4 = *reinterpret_cast<const float4*>(&bias1[32 * ct + 8 * qq + 4 * fh]);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = fmaf(acc[4 * qq + e], sc1, bb[e]);
          if (p.relu) v = fmaxf(v, 0.f);
          q2.a1[4 * qq + e] = v;
          q2.mx = fmaxf(q2.mx, fabsf(v));
        }
        *reinterpret_cast<float4*>(&tsc[fr * T_LD + 8 * qq + 4 * fh]) =
            make_float4(q2.a1[4 * qq], q2.a1[4 * qq + 1], q2.a1[4 * qq + 2], q2.a1[4 * qq + 3]);
      }
      CLS_MARK(5);                            // activation + transpose writes
      if (ct + 1 < NCT) {
        // the next column tile's chain, the second stage of THIS tile behind its MFMAs
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        half_chain(2 * (ct + 2), acc, [&](int kb) { stage2_step(q2, ct, kb); });
        half_chain(2 * (ct + 2) + 1, acc, [&](int kb) { stage2_step(q2, ct, kb); });
        CLS_MARK(6);                          // chain with the second stage behind it
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) stage2_step(q2, ct, j);
        CLS_MARK(7);                          // last second stage (exposed)
      }
    }
    // accumulator register i of lane (fr, fh) is output 8*(i/4) + 4*fh + i%4 of row fr: lane half 0 holds outputs 0..3 and
    // 8..11 (8, 9 are the gate products), lane half 1 outputs 4..7
    if (valid) {
      float* rr = p.raw + row * 12;
      if (fh == 0) {
        *reinterpret_cast<float4*>(rr) = make_float4(o2[0], o2[1], o2[2], o2[3]);
        *reinterpret_cast<float4*>(rr + 8) = make_float4(o2[4], o2[5], 0.f, 0.f);
      } else {
        *reinterpret_cast<float4*>(rr + 4) = make_float4(o2[0], o2[1], o2[2], o2[3]);
      }
    }
  }
#ifdef CLS_STAMP
  if (lane == 0 && p.wd != nullptr) {          // (p.wd is not used by this kernel: the stamped build borrows it for its debug buffer)
    float* o = const_cast<float*>(p.wd) + ((int64_t)blockIdx.x * CLS_NW + wave) * 8;
    for (int k = 0; k < 8; ++k) o[k] = (float)st_acc[k];
  }
  return;
#endif
  // ------------------------------------------------------------------ per-domain column sums of the activation (+ node counts):
  // lanes -> wave (xor butterfly over the 32 rows of a lane half) -> block (LDS float adds) -> one fp64 atomic per (block, column, domain)
  if (p.colsum != nullptr) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float a = cs_s[ct] + __shfl_xor(cs_s[ct], 32), b = cs_t[ct] + __shfl_xor(cs_t[ct], 32);      // the two row halves
      if (fh == 0) {
        unsafeAtomicAdd(&red[0][32 * ct + fr], a);
        unsafeAtomicAdd(&red[1][32 * ct + fr], b);
      }
    }
    {
      float a = cnt_s, b = cnt_t;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
      if (lane == 0) { unsafeAtomicAdd(&red[0][DK], a); unsafeAtomicAdd(&red[1][DK], b); }
    }
    __syncthreads();
    for (int t = tid; t < 2 * (DK + 1); t += 64 * CLS_NW) {
      const int d = t / (DK + 1), c = t % (DK + 1);
      const double v = (double)red[d][c];
      if (c == DK) unsafeAtomicAdd(&p.colsum[2 * NC + d], v);
      else if (c < NC) unsafeAtomicAdd(&p.colsum[d * NC + c], v);
    }
  }
}

}  // namespace

bool bgnn_tf_cls_supported(const GemmParams& p) {
  if (p.NC != 128 || p.Din > 128 || p.Din <= 64 || (p.Din & 3) || !p.mask || !p.w2 || !p.g2 || !p.raw) return false;
  if (p.sk_NC < 0 || p.sk_NC > 24 || (p.sk_NC & 3) || p.sk_heads < 0 || p.sk_heads > 2) return false;
  if (p.sk_NC > 0 && (!p.sk_Wp || !p.sk_bias || !p.sk_wd || !p.sk_g || !p.sk_gc || (p.sk_ldh & 3) || p.sk_NC != p.sk_heads * 2 * p.sk_ldh)) return false;
  return true;
}

int bgnn_tf_cls_launch(const GemmParams& p, hipStream_t st, int n_cu) {
  if (!bgnn_tf_cls_supported(p)) return BGNN_E_SHAPE;
  const int64_t nt = (p.N + 31) / 32, nb = (nt + CLS_NW - 1) / CLS_NW;
#ifdef CLS_STAMP
  {
    GemmParams q = p;
    static float* dbuf = nullptr;
    if (!dbuf) (void)hipMalloc((void**)&dbuf, sizeof(float) * 256 * 8 * 8);
    q.wd = dbuf;
    const unsigned g = (unsigned)(nb < n_cu ? nb : n_cu);
    hipLaunchKernelGGL(cls_stage_kernel, dim3(g), dim3(64 * CLS_NW), 0, st, q);
    (void)hipDeviceSynchronize();
    static float host[256 * 8 * 8];
    (void)hipMemcpy(host, dbuf, sizeof(float) * g * CLS_NW * 8, hipMemcpyDeviceToHost);
    double acc[8] = {0};
    for (unsigned w = 0; w < g * CLS_NW; ++w) for (int k = 0; k < 8; ++k) acc[k] += host[w * 8 + k];
    const double d = (double)nt;
    fprintf(stderr, "[cls stamp] cycles per tile: loop %.0f convert %.0f skinny-chain %.0f skinny-epilogue %.0f first-chain %.0f activation(x4) %.0f chain+stage2(x3) %.0f last-stage2 %.0f\n",
            acc[0] / d, acc[1] / d, acc[2] / d, acc[3] / d, acc[4] / d, acc[5] / d, acc[6] / d, acc[7] / d);
    return 0;
  }
#endif
  hipLaunchKernelGGL(cls_stage_kernel, dim3((unsigned)(nb < n_cu ? nb : n_cu)), dim3(64 * CLS_NW), 0, st, p);
  BGNN_LAUNCH_CHECK();
  return 0;
}
