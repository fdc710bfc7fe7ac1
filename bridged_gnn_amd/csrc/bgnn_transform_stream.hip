// Barrier-free pipeline for the dense transform of AdaptedConv (reference Bridged-GNN/models/KTGNN.py:275-284,
// Linear = PyG nn.dense.linear :240-246) on gfx950.
//
// Why another kernel: the W-stationary block kernel of bgnn_transform.hip walks 32-row tiles with ONE block-wide barrier per tile,
// so all eight waves of a CU sit in the same phase (stage -> barrier -> MFMA -> epilogue) and the phases add up: 0.38 ms for the
// 1.5 GB of the hidden transform (4 TB/s); six re-schedulings of that structure moved nothing.  Here nobody
// waits at a barrier.  One persistent block of NW = 8 waves per CU walks the tiles blockIdx.x + i * gridDim.x; in iteration i
// every wave
//   (a) STAGES its 32/NW rows of tile i + AHEAD: the rows were requested DEPTH tiles earlier (2 KB per wave and tile, in
//       registers), the wave forms the rows' gate dot products and tanh (KTGNN.py:277-278), scales every row by a power of two,
//       splits it into two fp16 pieces (below), writes the pieces into a slot of an LDS ring (XOR-swizzled 16-byte chunks) with
//       the row's rank-1 coefficient and inverse scale, and adds one to the slot's `ready` counter;
//   (b) CONSUMES tile i once `ready` shows all NW shares: the wave keeps "its" 32 output columns of W in registers for the whole
//       kernel (MFMA B operand, scaled per column and split the same way: 64 VGPRs), reads the tile as the A operand (16
//       ds_read_b128), runs 24 v_mfma_f32_32x32x16_f16, adds one to the slot's `done` counter (a slot is overwritten once it shows
//       all consumers of its previous tile) and writes out = acc * (row scale x column scale) + bias + coef * (W.delta) straight
//       from the accumulators: with x as the A operand a store instruction writes 2 rows x 128 contiguous bytes, no transpose.
// The only synchronisation is through those LDS counters (monotone, never reset); ordering is the in-order LDS queue of a wave
// plus s_waitcnt lgkmcnt(0) -- fences restricted to the "local" address space, so outstanding global stores are never waited for.
// Waves drift up to SLOTS - AHEAD tiles apart.
// Vector memory goes through buffer descriptors re-based per tile (rows past N, tiles past the end and waves without an output
// column read zeros / drop their stores by the descriptor's range check): EVERY path through an iteration issues the same
// loads and stores, so the compiler's s_waitcnt vmcnt(n) counts are exact and a wave really keeps DEPTH tiles of loads in flight
// behind its stores (loads, stores and their vmcnt slots are in issue order; one conditional store would make the count at the
// join conservative and drain the queue every tile).
// A first version with dedicated producer waves (4 stagers + 8 consumers) ran 0.36 ms: one wave executes a tile's ~1200 staging
// instructions at ~13 cycles each, which bounded the launch.
//
// fp16 x 2 split products.  A fp32 value v scaled into [2^10, 2^11) relative to its row (column) maximum is written as
// hi = fp16(v), lo = fp16(v - hi): |v - hi - lo| <= 2^-22 |v| (two round-to-nearest 11-bit significands).  A product is the three
// MFMAs hi*hi + hi*lo + lo*hi (fp32 accumulate); the dropped lo*lo term is <= 2^-22 relative.  Measured against the fp64 oracle
// this is as exact as the bf16 x 3 / six-product scheme of the block kernel (3e-7 of the output scale) at HALF the matrix work
// and two conversions instead of three per element.  The power-of-two row / column scales make the split independent of the
// magnitude of the inputs (fp16 has 5 exponent bits): elements below 2^-13 of their row's maximum lose relative, not absolute,
// accuracy -- their error stays below 2^-35 of the row maximum.
#include <cstdio>
#include <cstdlib>
#include "bgnn_common.h"
#include "bgnn_transform_params.h"

using bgnn_tf::GemmParams;
using bgnn_tf::f32x16;
using bgnn_tf::tanh_fast;

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef TS_SLOTS
#define TS_SLOTS 6        // ring slots (16.5 KB each)
#endif
#ifndef TS_AHEAD
#define TS_AHEAD 2        // staging runs this many tiles in front of consumption
#endif
#ifndef TS_DEPTH
#define TS_DEPTH 2        // loads are requested this many tiles in front of staging
#endif
constexpr int TS_TARGET_EXP = 10;   // a row's / column's largest magnitude is scaled into [2^10, 2^11)

// raw buffer descriptor (gfx9 dword 3 = 0x00020000: 32-bit data format, no swizzle); `bytes` <= 0: every access is out of range
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t bytes) {
  const int64_t cap = 0x7fffffff;
  const int nrec = (int)(bytes < 0 ? 0 : (bytes > cap ? cap : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, nrec, 0x00020000);
}

// ---- LDS ring ---------------------------------------------------------------------------------------------------------
// slot = [hi: 32 rows x DK fp16][lo: same][coef0[32] | coef1[32] | inv_scale[32] | domain[32]] ; DK = 128: 16.5 KB
template <int DK>
struct Slot {
  static constexpr int PIECE = 32 * DK * 2;
  static constexpr int COEF = 2 * PIECE;
  static constexpr int BYTES = COEF + 5 * 32 * 4;      // (+ one dump row for the lanes that have nothing to publish)
  // byte offset of the 16-byte chunk `c` (8 fp16) of row r inside a piece: rows are 256 B (DK = 128) or two rows share 256 B
  // (DK = 64); the chunk index is XOR-swizzled with the (super-)row so that the 16 lanes of a ds_read_b128 group, which
  // read the same k chunk of 16 different rows, hit 16 different bank quads (MI355X_MICROARCH: LDS table)
  __device__ static __forceinline__ int chunk_off(int r, int c) {
    if constexpr (DK == 128) return r * 256 + ((c ^ (r & 15)) << 4);
    else return (r >> 1) * 256 + (((((r & 1) << 3) | c) ^ ((r >> 1) & 15)) << 4);
  }
};

__device__ __forceinline__ unsigned lds_addr(const void* p) {       // LDS byte address of a __shared__ object
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// one lane adds 1 to an LDS counter (plain ds_add_u32: the compiler's atomic optimiser wraps a uniform add into mbcnt / bcnt logic)
__device__ __forceinline__ void lds_inc(uint32_t* ctr) {
  if ((threadIdx.x & 63) == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(lds_addr(ctr)), "v"(1u) : "memory");
}
__device__ __forceinline__ void lds_arrive(uint32_t* ctr) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");      // s_waitcnt lgkmcnt(0): this wave's LDS reads / writes are done
  lds_inc(ctr);
}
__device__ __forceinline__ uint32_t lds_peek(uint32_t* flag) {
  return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_wait_ge(uint32_t* flag, uint32_t v) {
  while (lds_peek(flag) < v) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// both counters of an iteration in ONE LDS round trip; `pa` / `pb`: values read earlier (the counters only grow, so an old
// value that already meets its bar settles the matter without touching LDS again)
__device__ __forceinline__ void lds_wait_ge2(uint32_t* fa, uint32_t va, uint32_t pa, uint32_t* fb, uint32_t vb, uint32_t pb) {
  while (pa < va || pb < vb) {
    __builtin_amdgcn_s_sleep(1);
    pa = lds_peek(fa); pb = lds_peek(fb);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// power-of-two scale that brings a magnitude with biased exponent E into [2^TARGET, 2^(TARGET+1)), and its inverse
__device__ __forceinline__ void pow2_scales(float mx, float& sc, float& inv) {
  int E = (__builtin_bit_cast(int, mx) >> 23) & 0xff;
  E = E < 40 ? 40 : E;                                   // zero / tiny rows: a fixed (large) scale, products stay tiny
  sc = __builtin_bit_cast(float, (127 + TS_TARGET_EXP + 127 - E) << 23);       // E = 255 (inf / nan): 2^-118, propagates
  inv = __builtin_bit_cast(float, (E - TS_TARGET_EXP) << 23);
}

// v * sc (exact: sc is a power of two) -> hi, lo fp16 quadruples
__device__ __forceinline__ void split4(const float4 v, const float sc, h4& hi, h4& lo) {
  const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const _Float16 h = (_Float16)(f[e] * sc);
    hi[e] = h;
    lo[e] = (_Float16)fmaf(f[e], sc, -(float)h);        // the residual is exact in fp32
  }
}

// MODE 0: AdaptedConv transform (one head: NC = 2 * ldh).
// NW waves stage; waves < NCW also own a 32-column tile (NC = 32 * NCW).  SLOTS ring slots, staging runs AHEAD tiles in front of
// consumption, loads DEPTH tiles in front of staging.
template <int DK, int NW, int NCW, int SLOTS, int AHEAD, int DEPTH, int MODE>
__global__ __launch_bounds__(64 * NW) void transform_stream_kernel(GemmParams p) {
  using SL = Slot<DK>;
  static_assert(AHEAD >= 1 && AHEAD < SLOTS && NCW <= NW && 32 % (4 * NW) == 0, "pipeline shape");
  constexpr int KB16 = DK / 16;              // 16-wide k blocks of the 32x32x16 MFMA
  constexpr int CPT = DK / 64;               // staging: 16 lanes cover a row, CPT float4 chunks each (256-byte runs)
  constexpr int GPW = 32 / (4 * NW);         // 4-row groups a wave stages per tile
  constexpr int RPW = 32 / NW;               // rows a wave stages per tile
  __shared__ __attribute__((aligned(16))) unsigned char ring[SLOTS * SL::BYTES];
  __shared__ uint32_t ready[SLOTS], done[SLOTS];
  __shared__ __attribute__((aligned(16))) float gtab[MODE == 0 ? 2 * DK : 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < SLOTS) { ready[tid] = 0u; done[tid] = 0u; }
  const int64_t ntiles = (p.N + 31) / 32;
  const int64_t nlocal = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;      // tiles blockIdx.x + i * gridDim.x, i < nlocal
  const bool consumer = wave < NCW;
#ifdef TS_STAMP
  // diagnostic build only (tools/build_variant.sh -DTS_STAMP): cycles per wave spent in each phase, written to p.raw at the end
  uint32_t st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t st_last = (uint32_t)__builtin_amdgcn_s_memtime();
#define TS_MARK(k) do { const uint32_t t__ = (uint32_t)__builtin_amdgcn_s_memtime(); st_acc[k] += t__ - st_last; st_last = t__; } while (0)
#else
#define TS_MARK(k) do { } while (0)
#endif

  // ---------------------------------------------------------------- staging state
  const int l16 = lane & 15, rsub = lane >> 4;
  // gate vectors as (g_s2t[k], g_t2s[k]) pairs in LDS: a lane reads the pairs of its fixed k chunks when it stages (16 VGPRs
  // if kept in registers -- the fused iteration has none to spare)
  float gcs[2] = {0.f, 0.f};
  if constexpr (MODE == 0) {
    for (int k = tid; k < DK; k += 64 * NW) {
      gtab[2 * k] = k < p.Din ? p.g[k] : 0.f;
      gtab[2 * k + 1] = k < p.Din ? p.g[2 * p.Din + k] : 0.f;
    }
  }
  if constexpr (MODE == 0) { gcs[0] = p.gc[0]; gcs[1] = p.gc[1]; }
  __syncthreads();                            // the only block-wide barrier (flags zeroed, gate table written)
  const bool din_full = p.Din == DK;          // (block-uniform) no zero-fill of chunks past Din needed
  // per-lane offsets that are the same in every tile: LDS bytes of the chunks this lane writes, byte offsets of its loads
  int woff[GPW][CPT], xoff[GPW][CPT], moff[GPW], coff[GPW];
#pragma unroll
  for (int g = 0; g < GPW; ++g) {
    const int row = RPW * wave + 4 * g + rsub;
    moff[g] = row;
    coff[g] = l16 < 4 ? 32 * l16 + row : 128 + row;   // word of the slot's coefficient block this lane publishes (row 4: dump)
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int c4 = l16 + 16 * c, k = c4 * 4;
      woff[g][c] = SL::chunk_off(row, c4 >> 1) + ((c4 & 1) << 3);
      xoff[g][c] = (int)((row * p.ldx + (din_full || k < p.Din ? k : 0)) * 4);
    }
  }
  u32x4 ra[DEPTH][GPW][CPT];
  uint8_t rm[DEPTH][GPW];
  // request this wave's rows of local tile j into register set `set`.  ALWAYS the same loads: a tile past the end gets an empty
  // descriptor (zeros, no traffic), rows past N are out of the descriptor's range (zeros).
  // (running state instead of 64-bit multiplies per tile: first row / rows left / base pointers of the tile to request next)
  const int64_t tstep = (int64_t)gridDim.x * 32;                 // rows between two tiles of this block
  int64_t rq_left = p.N - (int64_t)blockIdx.x * 32;              // N - first row of the tile requested next
  const float* rq_x = p.x + (int64_t)blockIdx.x * 32 * p.ldx;
  const uint8_t* rq_m = p.mask + (int64_t)blockIdx.x * 32;
  const bool has_mask = MODE == 0 || p.mask != nullptr;
  auto request = [&](int set) {
    const int rows_here = rq_left > 32 ? 32 : (rq_left < 0 ? 0 : (int)rq_left);     // tile past the end: 0 rows, an empty descriptor
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rq_x), (short)0, rows_here * (int)p.ldx * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(rq_m), (short)0, has_mask ? rows_here : 0, 0x00020000);
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) ra[set][g][c] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[g][c], 0, 0);
      rm[set][g] = __builtin_amdgcn_raw_buffer_load_b8(rmk, moff[g], 0, 0);
    }
    rq_left -= tstep; rq_x += tstep * p.ldx; rq_m += tstep;
  };
  // One row group's staging arithmetic, cut into steps so that the fused iteration can drop a step behind each MFMA of the wave's
  // own chain (the chain is dependency-paced: one MFMA per 32-64 cycles; the wave's vector instructions issue in between).
  struct StageState { float f[4 * CPT]; bool sdom; f2 d; float mx, sc, inv, t; h4 hi[CPT], lo[CPT]; };
  constexpr int STAGE_STEPS = CPT + 4 + 3 * CPT + 4;
  auto stage_step = [&](StageState& q, int m, unsigned char* sb, int g) {
    if (m < CPT) {                            // gate dot products, both gates per v_pk_fma_f32
      if constexpr (MODE == 0) {
        const float4 ga = *reinterpret_cast<const float4*>(&gtab[8 * (l16 + 16 * m)]);
        const float4 gb = *reinterpret_cast<const float4*>(&gtab[8 * (l16 + 16 * m) + 4]);
        q.d = __builtin_elementwise_fma(f2{q.f[4 * m], q.f[4 * m]}, f2{ga.x, ga.y}, q.d);
        q.d = __builtin_elementwise_fma(f2{q.f[4 * m + 1], q.f[4 * m + 1]}, f2{ga.z, ga.w}, q.d);
        q.d = __builtin_elementwise_fma(f2{q.f[4 * m + 2], q.f[4 * m + 2]}, f2{gb.x, gb.y}, q.d);
        q.d = __builtin_elementwise_fma(f2{q.f[4 * m + 3], q.f[4 * m + 3]}, f2{gb.z, gb.w}, q.d);
      }
    } else if (m == CPT) {                    // row maximum (this lane's share)
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        q.mx = fmaxf(fmaxf(fabsf(q.f[4 * c]), fabsf(q.f[4 * c + 1])), q.mx);
        q.mx = fmaxf(fmaxf(fabsf(q.f[4 * c + 2]), fabsf(q.f[4 * c + 3])), q.mx);
      }
    } else if (m == CPT + 1) {               // maximum over the row's 16 lanes on the BITS (magnitudes: unsigned order = float
      unsigned b = __builtin_bit_cast(unsigned, q.mx);   // order, NaN on top; v_max_u32 takes the DPP operand, fmaxf costs 3 per step)
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true));
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xF, 0xF, true));
      q.mx = __builtin_bit_cast(float, b);
    } else if (m == CPT + 2) {
      unsigned b = __builtin_bit_cast(unsigned, q.mx);
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x141, 0xF, 0xF, true));
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x140, 0xF, 0xF, true));
      q.mx = __builtin_bit_cast(float, b);
    } else if (m == CPT + 3) {
      pow2_scales(q.mx, q.sc, q.inv);
      if constexpr (MODE == 0) q.t = q.sdom ? q.d.x : q.d.y;     // a row needs ONE of its two gates: select before the sum and the tanh
    } else if (m < CPT + 4 + 3 * CPT) {       // per chunk: hi pieces, lo pieces, the two LDS writes
      const int c = (m - (CPT + 4)) / 3, part = (m - (CPT + 4)) % 3;
      if (part == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) q.hi[c][e] = (_Float16)(q.f[4 * c + e] * q.sc);
      } else if (part == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) q.lo[c][e] = (_Float16)fmaf(q.f[4 * c + e], q.sc, -(float)q.hi[c][e]);   // exact residual
      } else {
        *reinterpret_cast<h4*>(sb + woff[g][c]) = q.hi[c];
        *reinterpret_cast<h4*>(sb + SL::PIECE + woff[g][c]) = q.lo[c];
      }
    } else {
      const int r = m - (CPT + 4 + 3 * CPT);
      if (r == 0) { if constexpr (MODE == 0) { q.t += bgnn::dpp_mov<0xB1>(q.t); q.t += bgnn::dpp_mov<0x4E>(q.t); } }
      else if (r == 1) { if constexpr (MODE == 0) { q.t += bgnn::dpp_mov<0x141>(q.t); q.t += bgnn::dpp_mov<0x140>(q.t); q.t += q.sdom ? gcs[0] : gcs[1]; } }
      else if (r == 2) { if constexpr (MODE == 0) q.t = tanh_fast(q.t); }
      else {
        // rank-1 coefficients (KTGNN.py:277-280): -gate_s2t on source rows (table 0), +gate_t2s on target rows (table 1);
        // lanes 0..3 of the row's 16 publish coef0 / coef1 / inverse scale / domain flag
        // (branch-free: lanes 4..15 write the same word of a dump row -- an exec-masked store splits the block the MFMAs sit in)
        float c0 = 0.f, c1 = 0.f;
        if constexpr (MODE == 0) { c0 = q.sdom ? -q.t : 0.f; c1 = q.sdom ? 0.f : q.t; }
        float val = l16 == 0 ? c0 : c1;
        val = l16 == 2 ? q.inv : val;
        val = l16 >= 3 ? (q.sdom ? 1.f : 0.f) : val;
        reinterpret_cast<float*>(sb + SL::COEF)[coff[g]] = val;
      }
    }
  };
  static_assert(STAGE_STEPS == CPT + 4 + 3 * CPT + 4, "step count");
  auto stage_init = [&](StageState& q, const u32x4 (&rv)[CPT], uint8_t mk) {
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      float4 v = __builtin_bit_cast(float4, rv[c]);
      if (!din_full && (l16 + 16 * c) * 4 >= p.Din) v = make_float4(0.f, 0.f, 0.f, 0.f);
      q.f[4 * c] = v.x; q.f[4 * c + 1] = v.y; q.f[4 * c + 2] = v.z; q.f[4 * c + 3] = v.w;
    }
    q.sdom = mk != 0; q.d = f2{0.f, 0.f}; q.mx = 0.f; q.sc = q.inv = q.t = 0.f;
  };
  // stage this wave's rows of local tile j (taken out of their register set as `rv` / `mk`) into ring slot `slot`; `round` =
  // j / SLOTS.  No vector memory here.
  auto stage = [&](int64_t j, const u32x4 (&rv)[GPW][CPT], const uint8_t (&mk)[GPW], int slot, uint32_t round) {
    if (j >= nlocal) return;
    TS_MARK(0);
    lds_wait_ge(&done[slot], (uint32_t)NCW * round);            // every consumer has released the slot's previous tile
    TS_MARK(1);
    unsigned char* const sb = ring + slot * SL::BYTES;
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
      StageState q;
      stage_init(q, rv[g], mk[g]);
#pragma unroll
      for (int m = 0; m < STAGE_STEPS; ++m) stage_step(q, m, sb, g);
    }
    lds_arrive(&ready[slot]);
    TS_MARK(3);
  };

  // ---------------------------------------------------------------- consumer state: columns col_base .. col_base + 31
  const int fr = lane & 31, fh = lane >> 5;
  const int col_base = p.col_off + wave * 32;
  const int n = col_base + fr;
  // stationary B operand: W[n][16kb + 8fh .. +7] scaled by this column's power of two, split into fp16 pieces
  h8 whi[KB16], wlo[KB16];
  float cinv = 0.f;
  float bias_n = 0.f, wd_n = 0.f;
  const float* otab = nullptr;                // first element of this wave's output table (nullptr: nothing to store)
  int ooff = 0;                               // byte offset of (row 4*fh of a tile, this lane's column) from the tile's first row
  int my_table = 0;
  // per-lane LDS byte offset of the A fragment of k block 0 (row fr, chunk fh); k block kb: the chunk index is XOR-swizzled, so
  // its offset is roff0 ^ (kb << 5) (DK = 128; bits 5..7 of the offset = bits 1..3 of the chunk) -- one v_xad_u32 with the slot base
  const int roff0 = SL::chunk_off(fr, fh);
  static_assert(DK == 128, "fragment offsets: roff0 ^ (kb << 5) holds for 256-byte rows");
  auto roff = [&](int kb) { return roff0 ^ (kb << 5); };
  if (consumer) {
    float4 w[KB16][2];
    float mx = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB16; ++kb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int k = 16 * kb + 8 * fh + 4 * hf;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < p.NC && k < p.Din) t = *reinterpret_cast<const float4*>(p.Wp + (int64_t)n * p.Din + k);
        w[kb][hf] = t;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sc;
    pow2_scales(mx, sc, cinv);
#pragma unroll
    for (int kb = 0; kb < KB16; ++kb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        h4 hi, lo;
        split4(w[kb][hf], sc, hi, lo);
#pragma unroll
        for (int e = 0; e < 4; ++e) { whi[kb][4 * hf + e] = hi[e]; wlo[kb][4 * hf + e] = lo[e]; }
      }
    // epilogue constants of this lane's column; accumulator register v of a lane is row 8*(v/4) + 4*fh + v%4 of the tile.
    // (host: NC % 32 == 0 and ldh % 32 == 0, so a wave's 32 columns lie in ONE table of one head)
    const int ld2 = 2 * (int)p.ldh;
    const int h = col_base / ld2, rem = col_base % ld2, t = rem >= p.ldh ? 1 : 0;
    bias_n = p.bias[n];
    if constexpr (MODE == 0) wd_n = p.wd[n];
    otab = h == 0 ? (t == 0 ? p.out[0][0] : p.out[0][1]) : (t == 0 ? p.out[1][0] : p.out[1][1]);
    ooff = (int)((4 * fh * p.row_stride + (rem - t * (int)p.ldh) + fr) * 4);
    if constexpr (MODE == 0) my_table = __builtin_amdgcn_readfirstlane(t);
  }
  const int rs4 = (int)(p.row_stride * 4);    // bytes per output row
  // LDS part of a tile's consumption (no vector memory): -> res[16] for the stores; false: the wave sits the tile out
  auto consume = [&](int64_t i, int slot, uint32_t round, float (&res)[16]) -> bool {
    if (!consumer || i >= nlocal) return false;
    const int64_t tile = blockIdx.x + i * (int64_t)gridDim.x;
    const unsigned char* const sb = ring + slot * SL::BYTES;
    TS_MARK(0);
    lds_wait_ge(&ready[slot], (uint32_t)NW * (round + 1));
    TS_MARK(4);
    bool skip = false;
    if constexpr (MODE == 0) {
      // single-table tail rows (GemmParams): a tile inside one tail group is sat out by the other table's waves
      const int64_t tb = tile * 32, te = tb + 32;
      skip = p.tail_s2t_begin > 0 && ((my_table == 0 && tb >= p.tail_t2s_begin && te <= p.tail_s2t_begin) || (my_table == 1 && tb >= p.tail_s2t_begin));
      if (p.tile_need != nullptr) skip = skip || ((p.tile_need[tile] >> my_table) & 1) == 0;     // (scalar load: uniform address)
    }
    if (skip) {                               // wave-uniform
      lds_arrive(&done[slot]);
      return false;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // A fragments in batches of two k blocks (4 ds_read_b128), two batches in flight: the reads of batch b + 2 are issued behind
    // the six MFMAs of batch b (192 cycles of matrix pipe cover the LDS latency; left alone the compiler chains
    // read -> wait -> MFMA through ONE register quad and exposes that latency 16 times per tile, and all 16 reads up front
    // cost 64 VGPRs)
    constexpr int NB = KB16 / 2;
    h8 fa[2][2][2];                           // [buffer][k block of the batch][hi, lo]
    auto fetch = [&](int b, int buf) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[buf][t][0] = *reinterpret_cast<const h8*>(sb + roff(2 * b + t));
        fa[buf][t][1] = *reinterpret_cast<const h8*>(sb + SL::PIECE + roff(2 * b + t));
      }
    };
    fetch(0, 0);
    if constexpr (NB > 1) fetch(1, 1);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kb = 2 * b + t;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[b & 1][t][1], whi[kb], acc, 0, 0, 0);   // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[b & 1][t][0], wlo[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[b & 1][t][0], whi[kb], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (b + 2 < NB) fetch(b + 2, b & 1);
    }
    const float* cf = reinterpret_cast<const float*>(sb + SL::COEF);
    float4 cq[4], sq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sq[q] = *reinterpret_cast<const float4*>(cf + 64 + 8 * q + 4 * fh);
      if constexpr (MODE == 0) cq[q] = *reinterpret_cast<const float4*>(cf + 32 * my_table + 8 * q + 4 * fh);
    }
    lds_arrive(&done[slot]);                  // everything this wave needs from the slot is in registers
    TS_MARK(5);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float s4[4] = {sq[q].x, sq[q].y, sq[q].z, sq[q].w};
      float c4[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (MODE == 0) { c4[0] = cq[q].x; c4[1] = cq[q].y; c4[2] = cq[q].z; c4[3] = cq[q].w; }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        res[4 * q + e] = fmaf(acc[4 * q + e], s4[e] * cinv, MODE == 0 ? fmaf(c4[e], wd_n, bias_n) : bias_n);
    }
    return true;
  };
  // Steady-state iteration: the staging arithmetic of tile i + AHEAD and the epilogue constants of tile i are dropped, a step at a
  // time, behind the MFMAs of tile i -- the wave's own chain is dependency-paced (one MFMA per 32-64 cycles), its vector
  // instructions issue in the gaps.  As separate phases (stage, then MFMA, then epilogue; the slow path below) a wave spent
  // 850 + 1450 + 700 cycles per tile, the chain mostly idle issue slots.  Both flags are polled first; both arrivals come last.
  auto fused = [&](int64_t i, const u32x4 (&rv)[GPW][CPT], const uint8_t (&mk)[GPW], int sslot, uint32_t sround, int cslot,
                   uint32_t cround, uint32_t pf_done, uint32_t pf_ready, float (&res)[16]) {
    static_assert(GPW == 1, "fused iteration: one 4-row group per wave");
    TS_MARK(0);
    lds_wait_ge2(&done[sslot], (uint32_t)NCW * sround, pf_done, &ready[cslot], (uint32_t)NW * (cround + 1), pf_ready);
    TS_MARK(4);
    unsigned char* const wb = ring + sslot * SL::BYTES;
    const unsigned char* const sb = ring + cslot * SL::BYTES;
    StageState q;
    stage_init(q, rv[0], mk[0]);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int NB = KB16 / 2;
    h8 fa[2][2][2];
    auto fetch = [&](int b, int buf) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[buf][t][0] = *reinterpret_cast<const h8*>(sb + roff(2 * b + t));
        fa[buf][t][1] = *reinterpret_cast<const h8*>(sb + SL::PIECE + roff(2 * b + t));
      }
    };
    fetch(0, 0);
    if constexpr (NB > 1) fetch(1, 1);
    const float* cf = reinterpret_cast<const float*>(sb + SL::COEF);
    float4 cq[4], sq[4];                      // read late: registers are short while the staging state is alive
    float tt[16], ssc[16];
    int step = 0;                             // (compile-time after unrolling)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kb = 2 * b + t;
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
#ifndef TS_X_NOMFMA
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? fa[b & 1][t][1] : fa[b & 1][t][0], pr == 1 ? wlo[kb] : whi[kb], acc, 0, 0, 0);
#else
          if (pr == 0) acc[kb] += (float)fa[b & 1][t][0][0] + (float)fa[b & 1][t][1][1];     // ablation: keep the fragment reads alive
#endif
          if (step < STAGE_STEPS) {
            stage_step(q, step, wb, 0);
          } else if (step < STAGE_STEPS + 4) {
            const int qq = step - STAGE_STEPS;
            const float c4[4] = {cq[qq].x, cq[qq].y, cq[qq].z, cq[qq].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) tt[4 * qq + e] = MODE == 0 ? fmaf(c4[e], wd_n, bias_n) : bias_n;
          } else if (step < STAGE_STEPS + 8) {
            const int qq = step - STAGE_STEPS - 4;
            const float s4[4] = {sq[qq].x, sq[qq].y, sq[qq].z, sq[qq].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) ssc[4 * qq + e] = s4[e] * cinv;
          }
          if (step == STAGE_STEPS - 6) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
              if constexpr (MODE == 0) cq[qq] = *reinterpret_cast<const float4*>(cf + 32 * my_table + 8 * qq + 4 * fh);
          }
          if (step == STAGE_STEPS - 2) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) sq[qq] = *reinterpret_cast<const float4*>(cf + 64 + 8 * qq + 4 * fh);
          }
          ++step;
        }
      }
      if (b + 2 < NB) fetch(b + 2, b & 1);
    }
    static_assert(STAGE_STEPS + 8 <= 3 * KB16, "the staging steps fit behind the MFMAs of a tile");
    // both arrivals behind ONE wait for this wave's LDS traffic (the staged rows are written, the fragments and coefficients read)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    lds_inc(&ready[sslot]);
    lds_inc(&done[cslot]);
    TS_MARK(5);
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = fmaf(acc[r], ssc[r], tt[r]);
  };
  // the tile's 16 stores (2 rows x 128 contiguous bytes each); `live` false / rows past N: dropped by the descriptor's range check
  int64_t st_left = 0;                        // N - first row of the tile stored next (set where the loop starts)
  const float* st_o = otab;
  auto store16 = [&](bool live, const float (&res)[16]) {
#ifdef TS_X_NOSTORE
    const int rows_here = 0;                  // ablation: every store dropped by the range check
#else
    const int rows_here = (!live || st_left < 0) ? 0 : (st_left > 32 ? 32 : (int)st_left);
#endif
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st_o), (short)0, rows_here * rs4, 0x00020000);
    st_left -= tstep; st_o += tstep * p.row_stride;
#ifdef TS_X_STORE4
    // experiment (wrong layout, same bytes): four 16-byte stores per lane instead of sixteen 4-byte ones
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32x4 v = {__builtin_bit_cast(unsigned, res[4 * q]), __builtin_bit_cast(unsigned, res[4 * q + 1]),
                       __builtin_bit_cast(unsigned, res[4 * q + 2]), __builtin_bit_cast(unsigned, res[4 * q + 3])};
      __builtin_amdgcn_raw_buffer_store_b128(v, ro, (lane & 31) * 16 + ((lane >> 5) * 4 + (wave & 3)) * rs4 + q * 8 * rs4, 0, 0);
    }
    return;
#endif
    // (the row offset rides in the VECTOR offset: the scalar offset of a buffer instruction is excluded from the range check)
    if (rs4 == 512) {                         // block-uniform: 128-float rows -- the row offsets inside a group of 4 are immediates
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res[r]), ro, ooff + (r >> 2) * 4096 + (r & 3) * 512, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res[r]), ro, ooff + (8 * (r >> 2) + (r & 3)) * rs4, 0, 0);
    }
  };

  // ---------------------------------------------------------------- pipeline
  // The tile loop is unrolled DEPTH times so that the register-set index is a compile-time constant; ring slots and rounds are
  // run-time counters (compile-time slots made the compiler keep every slot's LDS addresses in registers: 60 VGPRs, spills).
  // There is no prologue: the loop starts AHEAD + DEPTH iterations early with the out-of-range parts switched off (guards for
  // the LDS parts, empty descriptors for the loads and stores).  Loads issued in front of the loop would still be pending at
  // its header, and the compiler merges that state with the back edge's conservatively: it then waits for all but ~1 tile of
  // operations in every iteration instead of DEPTH tiles.
  const bool steady_wave = consumer && otab != nullptr && !(MODE == 0 && p.tail_s2t_begin > 0) && GPW == 1;
  uint32_t pf_done = 0, pf_ready = 0;         // counters of the next iteration's slots as read at the end of the previous one
  int ss = 0, cs = 0;                         // ring slot of the next tile to stage / to consume
  uint32_t sr = 0, cr = 0;                    // ... and its round (tile index / SLOTS)
  auto bump = [](int& slot, uint32_t& round) { ++slot; if (slot == SLOTS) { slot = 0; ++round; } };
  int64_t i_first = -(AHEAD + DEPTH);
  asm volatile("" : "+s"(i_first));            // opaque start: keeps the compiler from peeling the switched-off iterations off the
                                              // loop (peeled copies leave loads pending at the loop header, see above)
  st_left = p.N - ((int64_t)blockIdx.x + i_first * (int64_t)gridDim.x) * 32;
  st_o = otab != nullptr ? otab + ((int64_t)blockIdx.x + i_first * (int64_t)gridDim.x) * 32 * p.row_stride : nullptr;
  for (int64_t i0 = i_first; i0 < nlocal; i0 += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int64_t i = i0 + u, s = i + AHEAD;        // (s + DEPTH) % DEPTH == u
      // the staged tile's rows leave their register set first, so that the set can be re-requested in front of the LDS work
      u32x4 rv[GPW][CPT];
      uint8_t mk[GPW];
      // (explicit v_mov copies: the loop-carried set is then dead before it is re-requested, so the new loads can land in the SAME
      //  registers.  Left to the compiler the copy is made at the bottom of the loop instead -- behind a wait for the loads just issued)
#pragma unroll
      for (int g = 0; g < GPW; ++g) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { unsigned t; asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(ra[u][g][c][e])); rv[g][c][e] = t; }
        }
        unsigned tm;
        asm volatile("v_mov_b32 %0, %1" : "=v"(tm) : "v"((unsigned)rm[u][g]));
        mk[g] = (uint8_t)tm;
      }
      TS_MARK(0);
      request(u);
      TS_MARK(2);                             // (stamped build: the request's issue time is booked under "wait_loads")
      float res[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) res[r] = 0.f;
      bool live = false;
      bool wanted = true;                     // does any row of tile i need this wave's table?  (GemmParams::tile_need)
      if constexpr (MODE == 0) {
        if (p.tile_need != nullptr && i >= 0 && i < nlocal)
          wanted = ((p.tile_need[blockIdx.x + i * (int64_t)gridDim.x] >> my_table) & 1) != 0;
      }
      if (steady_wave && wanted && i >= 0 && s < nlocal) {      // block-uniform per wave; no vector memory in either branch
        fused(i, rv, mk, ss, sr, cs, cr, pf_done, pf_ready, res);
        bump(ss, sr);
        bump(cs, cr);
        live = true;
      } else {
        if (s >= 0) {
          stage(s, rv, mk, ss, sr);
          bump(ss, sr);
        }
        if (i >= 0) {
          live = consume(i, cs, cr, res);
          bump(cs, cr);
        }
      }
      TS_MARK(6);
      store16(live, res);
      // the next iteration's two counters, read now: the round trip hides behind the stores and the next request
      pf_done = lds_peek(&done[ss]);
      pf_ready = lds_peek(&ready[cs]);
      TS_MARK(7);
    }
  }
#ifdef TS_STAMP
  if (lane == 0 && p.raw != nullptr) {
    float* o = p.raw + ((int64_t)blockIdx.x * NW + wave) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (float)st_acc[k];
  }
#endif
}


// ======================================================================================================================
// Linear -> ReLU -> narrow AdaptedConv transform in the same pipeline (KTGNN_no_complement.forward :433: clf_target on
// clf_transformer(h); stage A of bgnn_linear_narrow_transform_f32).  The activation a1 = relu(x W^T + b) of a tile never
// leaves the registers: NA = 4 waves own 32 of its 128 columns each, here with W as the MFMA's A operand, so that a lane ends
// up with 16 columns of ONE row -- which is the B operand of the second stage as it stands (k slot = register index).  The
// second stage (6 fp16 MFMAs per wave against the stationary 10 x 32 slice of the consumer conv's packed rows and gate vectors)
// leaves 12 partial sums per row in the tile's LDS slot of that wave; the wave whose arrival completes the slot's counter adds
// the four slices and stores the tile's 32 x 12 floats.  The per-domain column sums of a1 (the next delta) accumulate per lane
// (fp32, ~120 rows per lane) and are reduced once at the end.  The W operands get ONE power-of-two scale per matrix (a lane's
// registers hold 16 different columns, so per-column scales do not factor out); rows are scaled per row as in the transform.
// All waves issue the same vector-memory instructions per iteration (3 loads, 2 stores; empty descriptors where a wave has
// nothing to store), see the header comment.
template <int DK, int NW, int NA, int SLOTS, int AHEAD, int DEPTH>
__global__ __launch_bounds__(64 * NW) void transform_stream2_kernel(GemmParams p) {
  using SL = Slot<DK>;
  static_assert(DK == 128 && NW == 8 && AHEAD >= 1 && AHEAD < SLOTS, "pipeline shape");
  constexpr int KB16 = DK / 16, CPT = DK / 64, RPW = 32 / NW;
  constexpr int R2 = 12;                      // floats per row of a partial / of the raw output
  __shared__ __attribute__((aligned(16))) unsigned char ring[SLOTS * SL::BYTES];
  __shared__ __attribute__((aligned(16))) float red[SLOTS][NA][32 * R2];
  __shared__ uint32_t ready[SLOTS], done[SLOTS], redcnt[SLOTS], wmax[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < SLOTS) { ready[tid] = 0u; done[tid] = 0u; redcnt[tid] = 0u; }
  if (tid < 2) wmax[tid] = 0u;
  __syncthreads();
  const int64_t ntiles = (p.N + 31) / 32;
  const int64_t nlocal = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
  // two teams of NA waves (waves 0..3 / 4..7, one of each per SIMD) take the tiles in turn: the long tail behind a tile's first-stage
  // MFMAs (activation, column sums, split, second stage, reduction) then overlaps with the partner's next chain instead of
  // running beside an idle wave (one team of 4 consumers + 4 staging-only waves measured 0.27 ms, no better than the block kernel)
  static_assert(NW == 2 * NA, "two consumer teams");
  const bool consumer = true;
  const int team = wave / NA, cw = wave % NA;
  const int NC = p.NC;                        // = 32 * NA
#ifdef TS_STAMP
  uint32_t st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t st_last = (uint32_t)__builtin_amdgcn_s_memtime();
#endif

  // ---------------------------------------------------------------- staging (no gates here: the coefficient block carries the
  // row's inverse scale and its domain flag)
  const int l16 = lane & 15, rsub = lane >> 4;
  const bool din_full = p.Din == DK;
  int woff[CPT], xoff[CPT];
  const int srow = RPW * wave + rsub;
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int c4 = l16 + 16 * c, k = c4 * 4;
    woff[c] = SL::chunk_off(srow, c4 >> 1) + ((c4 & 1) << 3);
    xoff[c] = (int)((srow * p.ldx + (din_full || k < p.Din ? k : 0)) * 4);
  }
  const int coff = l16 < 4 ? 32 * l16 + srow : 128 + srow;
  u32x4 ra[DEPTH][CPT];
  uint8_t rm[DEPTH];
  const int64_t tstep = (int64_t)gridDim.x * 32;
  int64_t rq_left = p.N - (int64_t)blockIdx.x * 32;
  const float* rq_x = p.x + (int64_t)blockIdx.x * 32 * p.ldx;
  const uint8_t* rq_m = p.mask + (int64_t)blockIdx.x * 32;
  auto request = [&](int set) {
    const int rows_here = rq_left > 32 ? 32 : (rq_left < 0 ? 0 : (int)rq_left);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rq_x), (short)0, rows_here * (int)p.ldx * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(rq_m), (short)0, rows_here, 0x00020000);
#pragma unroll
    for (int c = 0; c < CPT; ++c) ra[set][c] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[c], 0, 0);
    rm[set] = __builtin_amdgcn_raw_buffer_load_b8(rmk, srow, 0, 0);
    rq_left -= tstep; rq_x += tstep * p.ldx; rq_m += tstep;
  };
  struct StageState { float f[4 * CPT]; bool sdom; float mx, sc, inv; h4 hi[CPT], lo[CPT]; };
  constexpr int STAGE_STEPS = 4 + 3 * CPT + 1;
  auto stage_step = [&](StageState& q, int m, unsigned char* sb) {
    if (m == 0) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        q.mx = fmaxf(fmaxf(fabsf(q.f[4 * c]), fabsf(q.f[4 * c + 1])), q.mx);
        q.mx = fmaxf(fmaxf(fabsf(q.f[4 * c + 2]), fabsf(q.f[4 * c + 3])), q.mx);
      }
    } else if (m == 1) {
      unsigned b = __builtin_bit_cast(unsigned, q.mx);
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true));
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xF, 0xF, true));
      q.mx = __builtin_bit_cast(float, b);
    } else if (m == 2) {
      unsigned b = __builtin_bit_cast(unsigned, q.mx);
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x141, 0xF, 0xF, true));
      b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x140, 0xF, 0xF, true));
      q.mx = __builtin_bit_cast(float, b);
    } else if (m == 3) {
      pow2_scales(q.mx, q.sc, q.inv);
    } else if (m < 4 + 3 * CPT) {
      const int c = (m - 4) / 3, part = (m - 4) % 3;
      if (part == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) q.hi[c][e] = (_Float16)(q.f[4 * c + e] * q.sc);
      } else if (part == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) q.lo[c][e] = (_Float16)fmaf(q.f[4 * c + e], q.sc, -(float)q.hi[c][e]);
      } else {
        *reinterpret_cast<h4*>(sb + woff[c]) = q.hi[c];
        *reinterpret_cast<h4*>(sb + SL::PIECE + woff[c]) = q.lo[c];
      }
    } else {
      float val = l16 == 2 ? q.inv : 0.f;
      val = l16 >= 3 ? (q.sdom ? 1.f : 0.f) : val;
      reinterpret_cast<float*>(sb + SL::COEF)[coff] = val;
    }
  };
  auto stage_init = [&](StageState& q, const u32x4 (&rv)[CPT], uint8_t mk) {
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      float4 v = __builtin_bit_cast(float4, rv[c]);
      if (!din_full && (l16 + 16 * c) * 4 >= p.Din) v = make_float4(0.f, 0.f, 0.f, 0.f);
      q.f[4 * c] = v.x; q.f[4 * c + 1] = v.y; q.f[4 * c + 2] = v.z; q.f[4 * c + 3] = v.w;
    }
    q.sdom = mk != 0; q.mx = 0.f; q.sc = q.inv = 0.f;
  };
  auto stage = [&](int64_t j, const u32x4 (&rv)[CPT], uint8_t mk, int slot, uint32_t round) {
    if (j >= nlocal) return;
    lds_wait_ge(&done[slot], (uint32_t)NA * round);
    unsigned char* const sb = ring + slot * SL::BYTES;
    StageState q;
    stage_init(q, rv, mk);
#pragma unroll
    for (int m = 0; m < STAGE_STEPS; ++m) stage_step(q, m, sb);
    lds_arrive(&ready[slot]);
  };

  // ---------------------------------------------------------------- consumer state
  const int fr = lane & 31, fh = lane >> 5;
  const int col_base = cw * 32;
  h8 whi[KB16], wlo[KB16];                    // stationary A operand: W[col_base + fr][16kb + 8fh .. +7]
  h8 w2h[2], w2l[2];                          // second stage, stationary A operand: row fr of (w2 | g2), k slot r = column col_base + 8(r/4) + 4fh + r%4
  float bias16[16];
  float cinvW = 0.f, cinvW2 = 0.f;
  float cs_s[16], cs_t[16], cnt_s = 0.f, cnt_t = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { cs_s[r] = cs_t[r] = 0.f; bias16[r] = 0.f; }
  {
    float4 w[KB16][2];
    float w2v[16];
    float mx = 0.f, mx2 = 0.f;
    if (consumer) {
#pragma unroll
      for (int kb = 0; kb < KB16; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int k = 16 * kb + 8 * fh + 4 * hf;
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (k < p.Din) t = *reinterpret_cast<const float4*>(p.Wp + (int64_t)(col_base + fr) * p.Din + k);
          w[kb][hf] = t;
          mx = fmaxf(fmaxf(mx, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = col_base + 8 * (r >> 2) + 4 * fh + (r & 3);
        float v = 0.f;
        if (fr < 8) v = p.w2[(int64_t)fr * NC + c];
        else if (fr < 10) v = p.g2[(int64_t)(fr - 8) * 2 * NC + c];
        w2v[r] = v;
        mx2 = fmaxf(mx2, fabsf(v));
        bias16[r] = p.bias[c];
      }
      mx = bgnn::group_max<64>(mx);
      mx2 = bgnn::group_max<64>(mx2);
      if (lane == 0) { atomicMax(&wmax[0], __builtin_bit_cast(unsigned, mx)); atomicMax(&wmax[1], __builtin_bit_cast(unsigned, mx2)); }
    }
    __syncthreads();                          // one scale per matrix: the largest magnitude over all NA waves' slices
    if (consumer) {
      float sc, sc2;
      pow2_scales(__builtin_bit_cast(float, wmax[0]), sc, cinvW);
      pow2_scales(__builtin_bit_cast(float, wmax[1]), sc2, cinvW2);
#pragma unroll
      for (int kb = 0; kb < KB16; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          h4 hi, lo;
          split4(w[kb][hf], sc, hi, lo);
#pragma unroll
          for (int e = 0; e < 4; ++e) { whi[kb][4 * hf + e] = hi[e]; wlo[kb][4 * hf + e] = lo[e]; }
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const _Float16 h = (_Float16)(w2v[r] * sc2);
        w2h[r >> 3][r & 7] = h;
        w2l[r >> 3][r & 7] = (_Float16)fmaf(w2v[r], sc2, -(float)h);
      }
    }
  }
  const int roff0 = SL::chunk_off(fr, fh);
  auto roff = [&](int kb) { return roff0 ^ (kb << 5); };

  // everything after the first-stage accumulators of a tile: activation, column sums, second stage, partial sums into the slot's
  // reduction block; -> true (with the tile's two float4 items of this lane in o0 / o1) for the wave that completes the block
  auto post = [&](const f32x16& acc, float s_row, float dom, int64_t tile, int slot, uint32_t round, float4& o0, float4& o1) -> bool {
    float a1[16];
    const float sc1 = s_row * cinvW;
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = fmaf(acc[r], sc1, bias16[r]);
      if (p.relu) v = fmaxf(v, 0.f);
      a1[r] = v;
      mx = fmaxf(mx, fabsf(v));
    }
    const bool valid = tile * 32 + fr < p.N;
    const float ws = (valid && dom != 0.f) ? 1.f : 0.f, wt = (valid && dom == 0.f) ? 1.f : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cs_s[r] = fmaf(ws, a1[r], cs_s[r]); cs_t[r] = fmaf(wt, a1[r], cs_t[r]); }
    if (fh == 0) { cnt_s += ws; cnt_t += wt; }
    mx = fmaxf(mx, __shfl_xor(mx, 32));       // the row's two lane halves share the scale
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
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[kb], al[kb], acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l[kb], ah[kb], acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[kb], ah[kb], acc2, 0, 0, 0);
    }
    // accumulator register i of lane (fr, fh) is output 8*(i/4) + 4*fh + i%4 of row fr: lane half 0 holds outputs 0..3 and
    // 8..11 (8, 9 are the gate products), lane half 1 outputs 4..7
    const float s2 = inv2 * cinvW2;
    float* r2 = &red[slot][cw][fr * R2];
    if (fh == 0) {
      *reinterpret_cast<float4*>(r2) = make_float4(acc2[0] * s2, acc2[1] * s2, acc2[2] * s2, acc2[3] * s2);
      *reinterpret_cast<float4*>(r2 + 8) = make_float4(acc2[4] * s2, acc2[5] * s2, 0.f, 0.f);
    } else {
      *reinterpret_cast<float4*>(r2 + 4) = make_float4(acc2[0] * s2, acc2[1] * s2, acc2[2] * s2, acc2[3] * s2);
    }
    TS_MARK(4);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    uint32_t old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&redcnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    const bool last = old == (uint32_t)NA * (round + 1) - 1;
    if (last) {                               // wave-uniform: this wave adds the NA slices of the tile
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      o0 = *reinterpret_cast<const float4*>(&red[slot][0][lane * 4]);
      o1 = lane < 32 ? *reinterpret_cast<const float4*>(&red[slot][0][(lane + 64) * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 1; w < NA; ++w) {
        const float4 u = *reinterpret_cast<const float4*>(&red[slot][w][lane * 4]);
        o0.x += u.x; o0.y += u.y; o0.z += u.z; o0.w += u.w;
        if (lane < 32) {
          const float4 v = *reinterpret_cast<const float4*>(&red[slot][w][(lane + 64) * 4]);
          o1.x += v.x; o1.y += v.y; o1.z += v.z; o1.w += v.w;
        }
      }
    }
    TS_MARK(5);
    return last;
  };
  auto mfma_chain_plain = [&](const unsigned char* sb, f32x16& acc) {
    constexpr int NB = KB16 / 2;
    h8 fa[2][2][2];
    auto fetch = [&](int b, int buf) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[buf][t][0] = *reinterpret_cast<const h8*>(sb + roff(2 * b + t));
        fa[buf][t][1] = *reinterpret_cast<const h8*>(sb + SL::PIECE + roff(2 * b + t));
      }
    };
    fetch(0, 0); fetch(1, 1);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kb = 2 * b + t;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[kb], fa[b & 1][t][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[kb], fa[b & 1][t][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[kb], fa[b & 1][t][0], acc, 0, 0, 0);
      }
      if (b + 2 < NB) fetch(b + 2, b & 1);
    }
  };
  // plain consumption (first / last iterations): wait, chain, post
  auto consume = [&](int64_t i, int slot, uint32_t round, float4& o0, float4& o1) -> bool {
    if (!consumer || i >= nlocal) return false;
    const unsigned char* const sb = ring + slot * SL::BYTES;
    lds_wait_ge(&ready[slot], (uint32_t)NW * (round + 1));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    mfma_chain_plain(sb, acc);
    const float* cf = reinterpret_cast<const float*>(sb + SL::COEF);
    const float s_row = cf[64 + fr], dom = cf[96 + fr];
    lds_arrive(&done[slot]);
    return post(acc, s_row, dom, blockIdx.x + i * (int64_t)gridDim.x, slot, round, o0, o1);
  };
  // steady state: the staging steps of tile i + AHEAD ride behind the MFMAs of tile i
  auto fused = [&](int64_t i, const u32x4 (&rv)[CPT], uint8_t mk, int sslot, uint32_t sround, int cslot, uint32_t cround,
                   uint32_t pf_done, uint32_t pf_ready, float4& o0, float4& o1) -> bool {
    TS_MARK(0);
    lds_wait_ge2(&done[sslot], (uint32_t)NA * sround, pf_done, &ready[cslot], (uint32_t)NW * (cround + 1), pf_ready);
    TS_MARK(1);
    unsigned char* const wb = ring + sslot * SL::BYTES;
    const unsigned char* const sb = ring + cslot * SL::BYTES;
    StageState q;
    stage_init(q, rv, mk);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int NB = KB16 / 2;
    h8 fa[2][2][2];
    auto fetch = [&](int b, int buf) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[buf][t][0] = *reinterpret_cast<const h8*>(sb + roff(2 * b + t));
        fa[buf][t][1] = *reinterpret_cast<const h8*>(sb + SL::PIECE + roff(2 * b + t));
      }
    };
    fetch(0, 0); fetch(1, 1);
    const float* cf = reinterpret_cast<const float*>(sb + SL::COEF);
    const float s_row = cf[64 + fr], dom = cf[96 + fr];
    int step = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kb = 2 * b + t;
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 1 ? wlo[kb] : whi[kb], pr == 0 ? fa[b & 1][t][1] : fa[b & 1][t][0], acc, 0, 0, 0);
          if (step < STAGE_STEPS) stage_step(q, step, wb);
          ++step;
        }
      }
      if (b + 2 < NB) fetch(b + 2, b & 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    lds_inc(&ready[sslot]);
    lds_inc(&done[cslot]);
    TS_MARK(3);
    return post(acc, s_row, dom, blockIdx.x + i * (int64_t)gridDim.x, cslot, cround, o0, o1);
  };
  // the tile's 32 x 12 floats: item `lane` and item `lane + 64` (16 bytes each); empty descriptor unless this wave finished the tile
  int64_t st_left = 0;
  float* st_o = p.raw;
  auto store_raw = [&](bool live, const float4& o0, const float4& o1) {
    const int rows_here = (!live || st_left < 0) ? 0 : (st_left > 32 ? 32 : (int)st_left);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(st_o, (short)0, rows_here * R2 * 4, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), ro, lane * 16, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), ro, (lane + 64) * 16, 0, 0);
    st_left -= tstep; st_o += tstep * R2;
  };

  // ---------------------------------------------------------------- pipeline (see transform_stream_kernel)
  uint32_t pf_done = 0, pf_ready = 0;
  int ss = 0, cs = 0;
  uint32_t sr = 0, cr = 0;
  auto bump = [](int& slot, uint32_t& round) { ++slot; if (slot == SLOTS) { slot = 0; ++round; } };
  int64_t i_first = -(AHEAD + DEPTH);
  asm volatile("" : "+s"(i_first));
  st_left = p.N - ((int64_t)blockIdx.x + i_first * (int64_t)gridDim.x) * 32;
  st_o = p.raw + ((int64_t)blockIdx.x + i_first * (int64_t)gridDim.x) * 32 * R2;
  for (int64_t i0 = i_first; i0 < nlocal; i0 += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int64_t i = i0 + u, s = i + AHEAD;
      u32x4 rv[CPT];
      uint8_t mk;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { unsigned t; asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(ra[u][c][e])); rv[c][e] = t; }
      }
      { unsigned tm; asm volatile("v_mov_b32 %0, %1" : "=v"(tm) : "v"((unsigned)rm[u])); mk = (uint8_t)tm; }
      TS_MARK(0);
      request(u);
      TS_MARK(2);
      float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
      bool live = false;
      const bool my_turn = (int)(i & 1) == team;                 // (wave-uniform)
      if (my_turn && i >= 0 && s < nlocal) {
        live = fused(i, rv, mk, ss, sr, cs, cr, pf_done, pf_ready, o0, o1);
        bump(ss, sr);
        bump(cs, cr);
      } else {
        if (s >= 0) {
          stage(s, rv, mk, ss, sr);
          bump(ss, sr);
        }
        if (i >= 0) {
          if (my_turn) live = consume(i, cs, cr, o0, o1);
          bump(cs, cr);
        }
      }
      TS_MARK(6);
      store_raw(live, o0, o1);
      pf_done = lds_peek(&done[ss]);
      pf_ready = lds_peek(&ready[cs]);
      TS_MARK(7);
    }
  }
#ifdef TS_STAMP
  if (lane == 0 && p.sk_out[0][0] != nullptr) {
    float* o = p.sk_out[0][0] + ((int64_t)blockIdx.x * NW + wave) * 8;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) o[k2] = (float)st_acc[k2];
  }
#endif
  // ---------------------------------------------------------------- per-domain column sums of the activation (+ node counts)
  if (consumer && p.colsum != nullptr) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = cs_s[r], b = cs_t[r];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }     // over the 32 rows of a lane half
      if (fr == 0) {
        const int c = col_base + 8 * (r >> 2) + 4 * fh + (r & 3);
        unsafeAtomicAdd(&p.colsum[c], (double)a);
        unsafeAtomicAdd(&p.colsum[NC + c], (double)b);
      }
    }
    if (cw == 0) {
      float a = cnt_s, b = cnt_t;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
      if (lane == 0) { unsafeAtomicAdd(&p.colsum[2 * NC], (double)a); unsafeAtomicAdd(&p.colsum[2 * NC + 1], (double)b); }
    }
  }
}

}  // namespace

static bool stream2_supported(const GemmParams& p);

bool bgnn_tf_stream_supported(const GemmParams& p, int mode) {
  if (mode == 2) return stream2_supported(p);
  if (mode != 0) return false;
  if (p.n_heads != 1 || p.Din > 128 || p.Din <= 64 || (p.Din & 3) || (p.ldh & 31)) return false;
  if (p.NC != 2 * p.ldh || (p.NC != 256 && p.NC != 128) || p.col_off != 0) return false;
  if (p.ldx * 4 * 32 > 0x7fffffff || p.row_stride * 4 * 36 > 0x7fffffff) return false;     // 32-bit offsets inside a tile
  return true;
}

static bool stream2_supported(const GemmParams& p) {
  return p.NC == 128 && p.Din <= 128 && p.Din > 64 && (p.Din & 3) == 0 && p.mask && p.w2 && p.g2 && p.raw && p.ldx * 4 * 32 <= 0x7fffffff;
}

int bgnn_tf_stream_launch(const GemmParams& p, int mode, hipStream_t st, int n_cu) {
  if (mode == 2) {
    if (!stream2_supported(p)) return BGNN_E_SHAPE;
    const int64_t nt = (p.N + 31) / 32;
    const dim3 g2((unsigned)(nt < n_cu ? nt : n_cu), 1u);
#ifdef TS_STAMP
    {
      GemmParams q = p;
      static float* dbuf2 = nullptr;
      if (!dbuf2) (void)hipMalloc((void**)&dbuf2, sizeof(float) * 256 * 8 * 8);
      q.sk_out[0][0] = dbuf2;
      hipLaunchKernelGGL((transform_stream2_kernel<128, 8, 4, TS_SLOTS, TS_AHEAD, TS_DEPTH>), g2, dim3(512), 0, st, q);
      static int calls2 = 0;
      if (++calls2 % 12 == 0) {
        (void)hipDeviceSynchronize();
        static float host[256 * 8 * 8];
        (void)hipMemcpy(host, dbuf2, sizeof(float) * g2.x * 8 * 8, hipMemcpyDeviceToHost);
        double acc[8] = {0};
        for (unsigned w = 0; w < g2.x * 8; ++w) for (int k = 0; k < 8; ++k) acc[k] += host[w * 8 + k];
        const double d = (double)g2.x * 8 * ((double)nt / g2.x);
        fprintf(stderr, "[ts2 stamp] cycles per tile and wave (mean): other %.0f  flags %.0f  request %.0f  chain(+stage, fused turn) %.0f  post %.0f  reduce/flush %.0f  pre-store %.0f  stores+peek %.0f\n",
                acc[0] / d, acc[1] / d, acc[2] / d, acc[3] / d, acc[4] / d, acc[5] / d, acc[6] / d, acc[7] / d);
      }
      BGNN_LAUNCH_CHECK();
      return 0;
    }
#endif
    hipLaunchKernelGGL((transform_stream2_kernel<128, 8, 4, TS_SLOTS, TS_AHEAD, TS_DEPTH>), g2, dim3(512), 0, st, p);
    BGNN_LAUNCH_CHECK();
    return 0;
  }
  if (!bgnn_tf_stream_supported(p, mode)) return BGNN_E_SHAPE;
  const int64_t ntiles = (p.N + 31) / 32;
  const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu), 1u);
#ifdef TS_STAMP
  GemmParams q = p;
  static float* dbuf = nullptr;
  if (!dbuf) (void)hipMalloc((void**)&dbuf, sizeof(float) * 256 * 8 * 8);
  q.raw = dbuf;
  if (p.NC == 256) hipLaunchKernelGGL((transform_stream_kernel<128, 8, 8, TS_SLOTS, TS_AHEAD, TS_DEPTH, 0>), grid, dim3(512), 0, st, q);
  else hipLaunchKernelGGL((transform_stream_kernel<128, 8, 4, TS_SLOTS, TS_AHEAD, TS_DEPTH, 0>), grid, dim3(512), 0, st, q);
  static int calls = 0;
  if (++calls % 12 == 0) {
    (void)hipDeviceSynchronize();
    static float host[256 * 8 * 8];
    (void)hipMemcpy(host, dbuf, sizeof(float) * grid.x * 8 * 8, hipMemcpyDeviceToHost);
    double acc[8] = {0}, mx[8] = {0};
    for (unsigned w = 0; w < grid.x * 8; ++w)
      for (int k = 0; k < 8; ++k) { acc[k] += host[w * 8 + k]; if (host[w * 8 + k] > mx[k]) mx[k] = host[w * 8 + k]; }
    const double nt = (double)ntiles / grid.x;
    fprintf(stderr, "[ts stamp] cycles per tile and wave (mean | max over waves): other %.0f|%.0f  wait_done %.0f|%.0f  wait_loads+request %.0f|%.0f  stage %.0f|%.0f  wait_ready %.0f|%.0f  mfma %.0f|%.0f  epilogue %.0f|%.0f  stores %.0f|%.0f\n",
            acc[0] / (grid.x * 8) / nt, mx[0] / nt, acc[1] / (grid.x * 8) / nt, mx[1] / nt, acc[2] / (grid.x * 8) / nt, mx[2] / nt, acc[3] / (grid.x * 8) / nt, mx[3] / nt,
            acc[4] / (grid.x * 8) / nt, mx[4] / nt, acc[5] / (grid.x * 8) / nt, mx[5] / nt, acc[6] / (grid.x * 8) / nt, mx[6] / nt, acc[7] / (grid.x * 8) / nt, mx[7] / nt);
  }
#else
  if (p.NC == 256) hipLaunchKernelGGL((transform_stream_kernel<128, 8, 8, TS_SLOTS, TS_AHEAD, TS_DEPTH, 0>), grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL((transform_stream_kernel<128, 8, 4, TS_SLOTS, TS_AHEAD, TS_DEPTH, 0>), grid, dim3(512), 0, st, p);
#endif
  BGNN_LAUNCH_CHECK();
  return 0;
}
