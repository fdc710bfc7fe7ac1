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
// and the second stage needs no reduction across waves.  One 4-wave block per CU (one wave per SIMD, 150 KB of LDS); the next
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
constexpr int SLOT = 2 * PIECE + 4 * 32 * 4;             // a wave's tile: hi, lo | inverse scales, domain flags, column-sum weights (S, T)
constexpr int T_LD = 36;                                 // floats per row of the 32 x 32 transpose scratch (inside the slot, once its fragments are in registers)

__device__ __forceinline__ int chunk_off(int r, int c) { return r * 256 + ((c ^ (r & 15)) << 4); }
__device__ __forceinline__ void pow2_scales(float mx, float& sc, float& inv) {
  int E = (__builtin_bit_cast(int, mx) >> 23) & 0xff;
  E = E < 40 ? 40 : E;
  sc = __builtin_bit_cast(float, (127 + CLS_TARGET_EXP + 127 - E) << 23);
  inv = __builtin_bit_cast(float, (E - CLS_TARGET_EXP) << 23);
}

__global__ __launch_bounds__(256) void cls_stage_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char wl[(NCT + 1) * 2 * PIECE];      // stationary operands [tile][piece][32][128]
  __shared__ __attribute__((aligned(16))) unsigned char slots[4 * SLOT];
  __shared__ __attribute__((aligned(16))) float bias1[DK];
  __shared__ uint32_t wmax[3];
  __shared__ float red[2][DK + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int NC = p.NC;                        // = 128
  const bool din_full = p.Din == DK;

  // ------------------------------------------------------------------ stationary operands -> LDS (all 256 threads, once)
  if (tid < 3) wmax[tid] = 0u;
  for (int t = tid; t < 2 * (DK + 1); t += 256) (&red[0][0])[t] = 0.f;
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
  for (int it = tid; it < (NCT + 1) * 32 * 16; it += 256) {
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
  for (int it = tid; it < (NCT + 1) * 32 * 16; it += 256) {
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
  h8 w2h[NCT][2], w2l[NCT][2];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const _Float16 h = (_Float16)(w2v[ct][r] * sc2w);
      w2h[ct][r >> 3][r & 7] = h;
      w2l[ct][r >> 3][r & 7] = (_Float16)fmaf(w2v[ct][r], sc2w, -(float)h);
    }
  __syncthreads();                            // the last block-wide barrier: from here on every wave is on its own

  // ------------------------------------------------------------------ skinny epilogue constants (as transform_skinny_kernel):
  // accumulator registers 4q..4q+3 of a lane are packed columns 8q + 4fh + (0..3) of row fr (q < 3); 12..15 the gate products
  float4 bv[3], wv[3];
  float* optr[3];
  int hsel[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int c = 8 * q + 4 * fh;
    bv[q] = wv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    optr[q] = nullptr; hsel[q] = 0;
    if (c < p.sk_NC) {
      const int ld2 = 2 * (int)p.sk_ldh;
      const int h = c / ld2, rem = c % ld2, t = rem >= p.sk_ldh ? 1 : 0;
      bv[q] = *reinterpret_cast<const float4*>(p.sk_bias + c);
      wv[q] = *reinterpret_cast<const float4*>(p.sk_wd + c);
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
  float* const cfs = reinterpret_cast<float*>(sb + 2 * PIECE);          // [0..31] inverse row scales, [32..63] domain flags, [64..95] / [96..127] 1.0 for
                                                                        // an existing source- / target-domain row (weights of the column sums)
  float* const tsc = reinterpret_cast<float*>(sb);                     // transpose scratch (the hi piece's space)
  const int l16 = lane & 15, rsub = lane >> 4;
  const int roff0 = chunk_off(fr, fh);
  auto roff = [&](int kb) { return roff0 ^ (kb << 5); };
  const int64_t ntiles = (p.N + 31) / 32;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  float4 ra[8][2];
  uint8_t rmk = 0;
  auto gload = [&](int64_t tl) {              // branch-free: rows past N re-read row N-1 (never stored), chunks past Din chunk 0
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      int64_t r = tl * 32 + 4 * g + rsub;
      r = r < p.N ? r : p.N - 1;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int k = (l16 + 16 * c) * 4;
        ra[g][c] = *reinterpret_cast<const float4*>(p.x + r * p.ldx + (din_full || k < p.Din ? k : 0));
      }
    }
    int64_t r = tl * 32 + fr;
    rmk = p.mask[r < p.N ? r : p.N - 1];
  };
  // column sums of the activation: lane (c = fr, fh) accumulates column 32ct + c over rows 16fh .. 16fh+15 of every tile (the
  // activation tile goes through a 32 x 32 transpose in the slot; per-lane sums of all 16 registers cost 128 VGPRs and spilled)
  float cs_s[NCT], cs_t[NCT], cnt_s = 0.f, cnt_t = 0.f;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) cs_s[ct] = cs_t[ct] = 0.f;

  if (tile < ntiles) gload(tile);
  for (; tile < ntiles; tile += stride) {
    // ---- (1) the tile leaves the registers as fp16 pieces (row scale, split), the next tile's loads take their place
    const bool sdom = rmk != 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float f[8];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float4 v = ra[g][c];
        if (!din_full && (l16 + 16 * c) * 4 >= p.Din) v = make_float4(0.f, 0.f, 0.f, 0.f);
        f[4 * c] = v.x; f[4 * c + 1] = v.y; f[4 * c + 2] = v.z; f[4 * c + 3] = v.w;
      }
      float mx = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e += 2) mx = fmaxf(fmaxf(fabsf(f[e]), fabsf(f[e + 1])), mx);
      unsigned b = __builtin_bit_cast(unsigned, mx);
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true));
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xF, 0xF, true));
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x141, 0xF, 0xF, true));
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x140, 0xF, 0xF, true));
      float sc, inv;
      pow2_scales(__builtin_bit_cast(float, b), sc, inv);
      const int row = 4 * g + rsub;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const _Float16 h = (_Float16)(f[4 * c + e] * sc);
          hi[e] = h;
          lo[e] = (_Float16)fmaf(f[4 * c + e], sc, -(float)h);
        }
        const int c4 = l16 + 16 * c;
        const int off = chunk_off(row, c4 >> 1) + ((c4 & 1) << 3);
        *reinterpret_cast<h4*>(sb + off) = hi;
        *reinterpret_cast<h4*>(sb + PIECE + off) = lo;
      }
      if (l16 == 0) cfs[row] = inv;
    }
    if (fh == 0) {
      const bool ex = tile * 32 + fr < p.N;
      cfs[32 + fr] = sdom ? 1.f : 0.f;
      cfs[64 + fr] = (ex && sdom) ? 1.f : 0.f;
      cfs[96 + fr] = (ex && !sdom) ? 1.f : 0.f;
    }
    const int64_t tnext = tile + stride < ntiles ? tile + stride : tile;      // past the end: re-read (never staged)
    gload(tnext);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");           // the wave's own LDS writes before its reads
    // ---- (2) the tile's fragments (B operand: row fr, k = 16kb + 8fh .. +7) stay in registers for all five operands
    h8 xh[KB16], xl[KB16];
#pragma unroll
    for (int kb = 0; kb < KB16; ++kb) {
      xh[kb] = *reinterpret_cast<const h8*>(sb + roff(kb));
      xl[kb] = *reinterpret_cast<const h8*>(sb + PIECE + roff(kb));
    }
    const float s_row = cfs[fr], dom = cfs[32 + fr];
    const int64_t row = tile * 32 + fr;
    const bool valid = row < p.N;
    if (fh == 0) { cnt_s += cfs[64 + fr]; cnt_t += cfs[96 + fr]; }
    float wsr[16], wtr[16];                   // the weights of this lane's 16 rows (16fh ..) for the column sums
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(&cfs[64 + 16 * fh + 4 * q]);
      const float4 b = *reinterpret_cast<const float4*>(&cfs[96 + 16 * fh + 4 * q]);
      wsr[4 * q] = a.x; wsr[4 * q + 1] = a.y; wsr[4 * q + 2] = a.z; wsr[4 * q + 3] = a.w;
      wtr[4 * q] = b.x; wtr[4 * q + 1] = b.y; wtr[4 * q + 2] = b.z; wtr[4 * q + 3] = b.w;
    }
    auto chain = [&](int ct, f32x16& acc) {
      const unsigned char* wb = wl + (ct * 2) * PIECE;
#pragma unroll
      for (int kb = 0; kb < KB16; ++kb) {
        const h8 wh = *reinterpret_cast<const h8*>(wb + roff(kb));
        const h8 wlo = *reinterpret_cast<const h8*>(wb + PIECE + roff(kb));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[kb], acc, 0, 0, 0);      // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xh[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[kb], acc, 0, 0, 0);
      }
    };
    // ---- (3) the skinny tile (needs only the fragment registers; the slot becomes scratch afterwards): narrow tables of the two convs on h itself (KTGNN.py:277-284 by linearity, see bgnn_transform.hip)
    if (p.sk_NC > 0) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      chain(NCT, acc);
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
            float4 o;
            o.x = fmaf(cf, wv[q].x, fmaf(acc[4 * q], ss, bv[q].x));     o.y = fmaf(cf, wv[q].y, fmaf(acc[4 * q + 1], ss, bv[q].y));
            o.z = fmaf(cf, wv[q].z, fmaf(acc[4 * q + 2], ss, bv[q].z)); o.w = fmaf(cf, wv[q].w, fmaf(acc[4 * q + 3], ss, bv[q].w));
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
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      chain(ct, acc);
      float a1[16];
      float mx = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(&bias1[32 * ct + 8 * q + 4 * fh]);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = fmaf(acc[4 * q + e], sc1, bb[e]);
          if (p.relu) v = fmaxf(v, 0.f);
          a1[4 * q + e] = v;
          mx = fmaxf(mx, fabsf(v));
        }
        *reinterpret_cast<float4*>(&tsc[fr * T_LD + 8 * q + 4 * fh]) = make_float4(a1[4 * q], a1[4 * q + 1], a1[4 * q + 2], a1[4 * q + 3]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));                  // the row's two lane halves share the scale
      float sc2, inv2;
      pow2_scales(mx, sc2, inv2);
      h8 ah[2], al[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const _Float16 h = (_Float16)(a1[r] * sc2);
        ah[r >> 3][r & 7] = h;
        al[r >> 3][r & 7] = (_Float16)fmaf(a1[r], sc2, -(float)h);
      }
      f32x16 acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[ct][kb], al[kb], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l[ct][kb], ah[kb], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[ct][kb], ah[kb], acc2, 0, 0, 0);
      }
      const float s2 = inv2 * cinv2;
#pragma unroll
      for (int i = 0; i < 8; ++i) o2[i] = fmaf(acc2[i], s2, o2[i]);
      // column sums: this lane's column of the transposed tile, its 16 rows (the wave's own writes: in-order LDS queue)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float v = tsc[(16 * fh + j) * T_LD + fr];
        cs_s[ct] = fmaf(wsr[j], v, cs_s[ct]);
        cs_t[ct] = fmaf(wtr[j], v, cs_t[ct]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");          // reads done before the next column tile overwrites the scratch
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
    for (int t = tid; t < 2 * (DK + 1); t += 256) {
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
  const int64_t nt = (p.N + 31) / 32, nb = (nt + 3) / 4;
  hipLaunchKernelGGL(cls_stage_kernel, dim3((unsigned)(nb < n_cu ? nb : n_cu)), dim3(256), 0, st, p);
  BGNN_LAUNCH_CHECK();
  return 0;
}
