// Score stage, up to 16 queries per corpus pass: f32 corpus with exact f32 math, or an
// f16 corpus with f32 accumulation (batches of more than 16 queries take gemm_tiled.h):
//     scores[j][i] = sum_d M[i,d] * Q[j,d]        j < 16
// The reference has no batched entry (a batch is a loop of np.dot calls,
// src/svs/kb.py:1623); this kernel amortises ONE read of the corpus over 16
// queries, which is what lifts the path from ~1.1 k to > 10 k queries/s
// (BASELINE.json north_star) while staying HBM-bound: 16 x 2 flop per 4 corpus
// bytes = 8 flop/B, far below the f32 ridge.
//
// Two kernels compute it.  gemm_q16r_kernel (second half of this file, 4x4x1 / 4x4x4 MFMA,
// whole-line loads) is the product path; gemm_f32_q16_kernel (16x16x4 MFMA, half-line
// loads) is the first design, kept behind svs_index_set_variant(3) as the measured
// alternative (1.21 vs 1.10 ms per 16 queries at 1M x 1536).
//
// ---- gemm_f32_q16_kernel ----
// Arithmetic: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bit-for-bit an
// fmaf chain, no reduced precision).  Per wave and 16-row tile the A operand is
// the corpus (lane l: row l&15, k-slot l>>4) and the B operand the queries
// (lane l: query l&15, k-slot l>>4).  Each lane loads ONE float4 of its row per
// 16-column step -- 64 contiguous bytes per row per wave instruction -- and the
// four components feed four MFMAs whose k-slots map to columns
// 16 s + 4 (l>>4) + e; the query image in LDS is laid out in that same order, so
// the permutation cancels.  The summation order depends only on d: a row's score
// is the same wherever it is sharded (not bit-equal to the single-query GEMV's
// order; both are within 1e-5 of numpy, tests/test_batch_gpu.py).
//
// Geometry: a workgroup of 8 waves stages the 16 queries once in LDS
// ([d/16][64] float4 = 6 KiB per 1536-d query, 96 KiB in all), then streams a
// contiguous block of rows, each wave taking 32-row tiles (two independent
// accumulators hide the 40-cycle dependent-MFMA latency).  Loads for the next
// PF steps are in flight while the current ones are consumed.  Loads are plain
// (not nontemporal): a 128-byte line is consumed by two consecutive steps, and
// with `nt` the second half was re-fetched (4.5 vs 5.2 TB/s measured).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemv_f32.h"
#include "keys.h"

namespace svs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GQ = 16;        // queries per pass
constexpr int GEMM_WAVES = 8;
constexpr int GEMM_PF = 8;    // k-steps prefetched per tile half
constexpr int GEMM_LCAND = 1024;  // fused epilogue: candidates a workgroup parks in LDS before its one flush
constexpr int GEMM_FUSE_LDS = 8 + GEMM_LCAND * 12;   // counter + keys (u64) + query ids (u32), after the query image

// Q16: [16][ld] f32 (rows >= nq zero).  scores: [16][sstride].
//
// FUSE: no score matrix.  Writing 16 x n scores is only 1 % of the bytes but the writes,
// mixed into a saturating read stream, cost 10-12 % of the kernel (measured: ~1.6 us
// per MB whatever their layout); instead a score is offered to the query's candidate
// list when it reaches thr[q], a proven lower bound of the k-th best (the exact k-th
// best of a prefix of the rows, gemm_tiled.h has the full argument).  Header word 0 =
// candidate count, as select.h expects; row numbers are LOCAL (select_final adds
// row_offset).  A returning global atomic in the row loop would make the wave wait on
// vmcnt(0) -- i.e. drain its prefetched loads -- once per tile (measured: 1.14 ms, slower
// than storing the scores); so candidates are parked in LDS (ds atomics count on lgkmcnt)
// and the workgroup claims its global slots once, at the end.  If the LDS list is
// full (adversarial row order) the lane goes to the global list directly.
template <bool NT, int TILES = 2, int PF = GEMM_PF, bool FUSE = false>
__global__ __launch_bounds__(GEMM_WAVES * 64) void gemm_f32_q16_kernel(
    const float* __restrict__ M, const float* __restrict__ Q16, float* __restrict__ scores,
    int64_t n, int ld, int64_t sstride, int nq, int rows_per_block,
    uint32_t* __restrict__ fstate = nullptr, int fstate_stride = 0, uint64_t* __restrict__ fcand = nullptr,
    uint32_t fcap = 0, const float* __restrict__ fthr = nullptr, int fthr_stride = 0) {
  extern __shared__ v4f qlds[];  // [ld/16][64], then (FUSE) the parked candidates
  const int ksteps = ld >> 4;
  uint32_t* lcount = (uint32_t*)(qlds + (size_t)ksteps * 64);
  uint64_t* lkey = (uint64_t*)(lcount + 2);
  uint32_t* lqid = (uint32_t*)(lkey + GEMM_LCAND);
  if (FUSE && threadIdx.x == 0) lcount[0] = 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // ---- stage the queries in MFMA-B order
  for (int e = threadIdx.x; e < ksteps * 64; e += GEMM_WAVES * 64) {
    const int s = e >> 6, l = e & 63;
    qlds[e] = *(const v4f*)(Q16 + (int64_t)(l & 15) * ld + 16 * s + 4 * (l >> 4));
  }
  __syncthreads();

  // (fused form: the row blocks are taken in a scattered order, as the phased kernel takes its row tiles -- gemm_phased.h tile_desc:
  //  neighbouring blocks of a corpus stored topic by topic send all their candidates to the lists of the same few queries)
  int64_t pblk = blockIdx.x;
  if constexpr (FUSE) {
    const int g_ = (int)gridDim.x;
    int c = 1;
    if (g_ > 2 * 3 && g_ % 3) c = 3;
    if (g_ > 2 * 7 && g_ % 7) c = 7;
    if (g_ > 2 * 31 && g_ % 31) c = 31;
    if (g_ > 2 * 127 && g_ % 127) c = 127;
    if (g_ > 2 * 1009 && g_ % 1009) c = 1009;
    pblk = (pblk * c) % g_;
  }
  const int64_t blk0 = pblk * rows_per_block;
  const int64_t blk1 = blk0 + rows_per_block < n ? blk0 + rows_per_block : n;
  const int r16 = lane & 15, g = lane >> 4;
  constexpr int TROWS = 16 * TILES;
  // The (tile, chunk) walk is one continuous stream: the first chunk of the NEXT tile is
  // requested before the last chunk of this one is multiplied, so a wave never sits with
  // nothing in flight, and the score stores (which share vmcnt with the loads on gfx9
  // and complete in order with them) are never waited for.
  auto tile_ptrs = [&](int64_t row0, const v4f* (&p)[TILES]) {
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      int64_t r = row0 + 16 * t + r16;
      r = r < n ? r : n - 1;  // clamp: never read past the matrix
      p[t] = (const v4f*)(M + r * ld + 4 * g);
    }
  };
  int64_t row0 = blk0 + wave * TROWS;
  if (!FUSE && row0 >= blk1) return;   // (fused: every wave reaches the flush barrier)
  const v4f* p[TILES];
  tile_ptrs(row0, p);
  v4f a[TILES][PF];
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int j = 0; j < PF; ++j) a[t][j] = ldg4<NT>(p[t] + 4 * j);   // step j: columns 16 j + 4 g .. +4
  for (; row0 < blk1; row0 += GEMM_WAVES * TROWS) {
    f32x4 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mul_chunk = [&](int s0) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const v4f qf = qlds[(s0 + j) * 64 + lane];
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j].x, qf.x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j].y, qf.y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j].z, qf.z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j].w, qf.w, acc[t], 0, 0, 0);
      }
    };
    // No conditional around a prefetch anywhere: a branch there makes hipcc wait for the
    // just-issued loads at the join (the same trap as in gemv_f32.h).
    int s0 = 0;
    for (; s0 + PF < ksteps; s0 += PF) {
      v4f nx[TILES][PF];
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) nx[t][j] = ldg4<NT>(p[t] + 4 * (s0 + PF + j));
      mul_chunk(s0);
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) a[t][j] = nx[t][j];
    }
    {
      // last chunk of this tile: the next tile's first chunk goes out first (past the
      // block's end the rows clamp to valid memory and the data is never used)
      const v4f* pn[TILES];
      tile_ptrs(row0 + GEMM_WAVES * TROWS, pn);
      v4f nx[TILES][PF];
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) nx[t][j] = ldg4<NT>(pn[t] + 4 * j);
      mul_chunk(s0);
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        p[t] = pn[t];
#pragma unroll
        for (int j = 0; j < PF; ++j) a[t][j] = nx[t][j];
      }
    }
    // D layout: column (query) = lane & 15, rows 4 g + r of the 16-row tile
    if constexpr (FUSE) {
      if (r16 < nq) {
        uint32_t* hdr = fstate + (int64_t)r16 * fstate_stride;
        uint64_t* cq = fcand + (int64_t)r16 * fcap;
        const float thr = fthr[(int64_t)r16 * fthr_stride];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          const int64_t ob = row0 + 16 * t + 4 * g;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[t][r];
            if (!(v < thr) && ob + r < blk1) {
              const uint64_t key = ((uint64_t)score_key(v) << 32) | (uint32_t)(ob + r);
              const uint32_t ls = atomicAdd(lcount, 1u);
              if (ls < (uint32_t)GEMM_LCAND) {
                lkey[ls] = key;
                lqid[ls] = (uint32_t)r16;
              } else {
                const uint32_t slot = atomicAdd(hdr, 1u);
                if (slot < fcap) cq[slot] = key;
                else if (fthr_stride) *(volatile float*)(fthr + (int64_t)r16 * fthr_stride) = __builtin_inff();   // (full list: see gemm_q16r_kernel)
              }
            }
          }
        }
      }
    } else if (r16 < nq) {
      float* o = scores + (int64_t)r16 * sstride;
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        const int64_t ob = row0 + 16 * t + 4 * g;
        if (ob + 3 < blk1) *(f32x4*)(o + ob) = acc[t];
        else
          for (int r = 0; r < 4; ++r)
            if (ob + r < blk1) o[ob + r] = acc[t][r];
      }
    }
  }
  if constexpr (FUSE) {
    __syncthreads();
    const uint32_t parked = lcount[0] < (uint32_t)GEMM_LCAND ? lcount[0] : (uint32_t)GEMM_LCAND;
    for (uint32_t i = threadIdx.x; i < parked; i += GEMM_WAVES * 64) {
      const uint32_t q = lqid[i];
      const uint32_t slot = atomicAdd(fstate + (int64_t)q * fstate_stride, 1u);
      if (slot < fcap) fcand[(int64_t)q * fcap + slot] = lkey[i];
      // The query's list is full (it will be handed back for a re-run whatever else arrives): its threshold goes to +inf,
      // so that the workgroups that start after this store offer nothing for it -- a corpus sorted by similarity to the
      // queries otherwise sends every score of every block through the atomics above (12 ms instead of 1 per 16 queries)
      else if (fthr_stride) *(volatile float*)(fthr + (int64_t)q * fthr_stride) = __builtin_inff();
    }
  }
}

// ---- the same product on v_mfma_f32_4x4x1_16b_f32: full-line loads ---------------
// The 16x16x4 kernel above gives each lane one float4 of ITS row per instruction: 64
// contiguous bytes per row, half a cache line, which caps the stream at ~6.1 TB/s (and
// rules out nontemporal loads, see above).  The 4x4x1 instruction runs 16 independent
// 4x4 outer products, one per group of 4 lanes ("block" b = lane >> 2): lane (b, i)
// supplies A = row i's element and B = query i's element of block b's column, and
// block b accumulates the partial product of ITS columns.  So lane (b, i) loads float4
// #b of a 64-column step of row i: an instruction covers 4 rows x 256 contiguous bytes,
// whole lines, nontemporal -- 6.9 TB/s for the bare pattern (tools/pattern_bw.hip).
// Same f32 MFMA rate (32 MAC/clk/SIMD for every f32 shape), 4 queries per instruction,
// so 16 queries are 4 query groups m; a wave iteration is RG = 4 row groups (16 rows).
// After the last step the 16 per-block partial sums of every (row, query) are added
// across lanes (2 DPP row rotations + 2 bpermutes per value); the order depends only on
// d.  Query image in LDS: [m][step][lane] float4 = Q[4m + (lane&3)][64 step + 4 (lane>>2) ..].
constexpr int G4_RG = 4;
constexpr int G4_PF = 2;     // 64-column steps per register buffer (ld % 128 == 0)

//
// EB = bytes per element.  The geometry is in BYTES (a step is 256 B of a row: 64 floats or
// 128 halves), so the same kernel serves an f16 corpus (EB = 2: queries rounded to half by
// stage_queries_f16, v_mfma_f32_4x4x4_16b_f16, two per 16-byte load and query group; f32
// accumulate) -- up to 16 queries over a half corpus stream it at the same ~6.5 TB/s
// instead of the tiled kernel's 5.0-5.5.  M, Q16: rows of ld16 16-byte units.
typedef _Float16 h4x __attribute__((ext_vector_type(4)));
typedef float f32x2q __attribute__((ext_vector_type(2)));

template <bool FUSE, int EB = 4, int RG = G4_RG, int PF = G4_PF>
__global__ __launch_bounds__(GEMM_WAVES * 64) void gemm_q16r_kernel(
    const v4f* __restrict__ M, const v4f* __restrict__ Q16, float* __restrict__ scores,
    int64_t n, int ld16, int64_t sstride, int nq, int rows_per_block,
    uint32_t* __restrict__ fstate = nullptr, int fstate_stride = 0, uint64_t* __restrict__ fcand = nullptr,
    uint32_t fcap = 0, const float* __restrict__ fthr = nullptr, int fthr_stride = 0) {
  extern __shared__ v4f qlds[];  // [4][ld16/16][64], then (FUSE) the parked candidates
  const int ksteps = ld16 >> 4;
  uint32_t* lcount = (uint32_t*)(qlds + (size_t)ksteps * 256);
  uint64_t* lkey = (uint64_t*)(lcount + 2);
  uint32_t* lqid = (uint32_t*)(lkey + GEMM_LCAND);
  if (FUSE && threadIdx.x == 0) lcount[0] = 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int e = threadIdx.x; e < ksteps * 256; e += GEMM_WAVES * 64) {
    const int m = e / (ksteps * 64), rem = e - m * ksteps * 64;
    const int s = rem >> 6, l = rem & 63;
    qlds[e] = Q16[(int64_t)(4 * m + (l & 3)) * ld16 + 16 * s + (l >> 2)];
  }
  __syncthreads();

  // (fused form: the row blocks are taken in a scattered order, as the phased kernel takes its row tiles -- gemm_phased.h tile_desc:
  //  neighbouring blocks of a corpus stored topic by topic send all their candidates to the lists of the same few queries)
  int64_t pblk = blockIdx.x;
  if constexpr (FUSE) {
    const int g_ = (int)gridDim.x;
    int c = 1;
    if (g_ > 2 * 3 && g_ % 3) c = 3;
    if (g_ > 2 * 7 && g_ % 7) c = 7;
    if (g_ > 2 * 31 && g_ % 31) c = 31;
    if (g_ > 2 * 127 && g_ % 127) c = 127;
    if (g_ > 2 * 1009 && g_ % 1009) c = 1009;
    pblk = (pblk * c) % g_;
  }
  const int64_t blk0 = pblk * rows_per_block;
  const int64_t blk1 = blk0 + rows_per_block < n ? blk0 + rows_per_block : n;
  const int i4 = lane & 3, b = lane >> 2;
  constexpr int TROWS = 4 * RG;
  auto group_ptrs = [&](int64_t row0, const v4f* (&p)[RG]) {
#pragma unroll
    for (int t = 0; t < RG; ++t) {
      int64_t r = row0 + 4 * t + i4;
      r = r < n ? r : n - 1;  // clamp: never read past the matrix
      p[t] = M + r * ld16 + b;
    }
  };
  int64_t row0 = blk0 + wave * TROWS;
  if (!FUSE && row0 >= blk1) return;   // (fused: every wave reaches the flush barrier)
  const v4f* p[RG];
  group_ptrs(row0, p);
  v4f a[RG][PF];
#pragma unroll
  for (int t = 0; t < RG; ++t)
#pragma unroll
    for (int j = 0; j < PF; ++j) a[t][j] = ldg4<true>(p[t] + 16 * j);
  for (; row0 < blk1; row0 += GEMM_WAVES * TROWS) {
    f32x4 acc[RG][4];
#pragma unroll
    for (int t = 0; t < RG; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mul_chunk = [&](int s0) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        v4f qf[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) qf[m] = qlds[(m * ksteps + s0 + j) * 64 + lane];
        if constexpr (EB == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < RG; ++t)
#pragma unroll
              for (int m = 0; m < 4; ++m)
                acc[t][m] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t][j][e], qf[m][e], acc[t][m], 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e)   // halves 4e .. 4e+3 of the lane's eight
#pragma unroll
            for (int t = 0; t < RG; ++t)
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                const f32x2q xa = {a[t][j][2 * e], a[t][j][2 * e + 1]}, xb = {qf[m][2 * e], qf[m][2 * e + 1]};
                acc[t][m] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h4x, xa), __builtin_bit_cast(h4x, xb), acc[t][m], 0, 0, 0);
              }
        }
      }
    };
    int s0 = 0;
    for (; s0 + PF < ksteps; s0 += PF) {
      v4f nx[RG][PF];
#pragma unroll
      for (int t = 0; t < RG; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) nx[t][j] = ldg4<true>(p[t] + 16 * (s0 + PF + j));
      mul_chunk(s0);
#pragma unroll
      for (int t = 0; t < RG; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) a[t][j] = nx[t][j];
    }
    {
      const v4f* pn[RG];
      group_ptrs(row0 + GEMM_WAVES * TROWS, pn);
      v4f nx[RG][PF];
#pragma unroll
      for (int t = 0; t < RG; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) nx[t][j] = ldg4<true>(pn[t] + 16 * j);
      mul_chunk(s0);
#pragma unroll
      for (int t = 0; t < RG; ++t) {
        p[t] = pn[t];
#pragma unroll
        for (int j = 0; j < PF; ++j) a[t][j] = nx[t][j];
      }
    }
    // acc[t][m][r] in lane (b, j): block b's share of (row 4t + r, query 4m + j).  Sum over b;
    // lane (b, j) then keeps (t, m) = (b >> 2, b & 3): rows 4t .. 4t+3 of query 4m + j.
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    if constexpr (RG == 4) {
      // Reduce-scatter over the four bits of b (lanes l ^ 4, ^ 8, ^ 16, ^ 32): at each level a
      // lane keeps the half of its values whose (t, m) index matches its own bit, hands the
      // other half to its partner and adds what it receives -- 60 exchanged values per lane
      // instead of the 256 of an all-reduce (the epilogue was a quarter of the f32 kernel's
      // instructions and half of the f16 kernel's).
      const bool b0 = (b & 1) != 0, b1 = (b & 2) != 0, b2 = (b & 4) != 0, b3 = (b & 8) != 0;
      f32x4 v8[8], v4[4], v2[2];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const f32x4 lo = acc[(2 * s) >> 2][(2 * s) & 3], hi = acc[(2 * s + 1) >> 2][(2 * s + 1) & 3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float keep = b0 ? hi[r] : lo[r], send = b0 ? lo[r] : hi[r];
          // partner l ^ 4: l + 4 for the even quads of a 16-lane row, l - 4 for the odd ones
          int got = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x12C /* row_ror:12 */, 0xf, 0x5, false);
          got = __builtin_amdgcn_update_dpp(got, __builtin_bit_cast(int, send), 0x124 /* row_ror:4 */, 0xf, 0xa, false);
          v8[s][r] = keep + __builtin_bit_cast(float, got);
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float keep = b1 ? v8[2 * s + 1][r] : v8[2 * s][r], send = b1 ? v8[2 * s][r] : v8[2 * s + 1][r];
          v4[s][r] = keep + dpp_mov<0x128>(send);   // row_ror:8 == l ^ 8
        }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float keep = b2 ? v4[2 * s + 1][r] : v4[2 * s][r], send = b2 ? v4[2 * s][r] : v4[2 * s + 1][r];
          v2[s][r] = keep + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x401F));   // l ^ 16
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float keep = b3 ? v2[1][r] : v2[0][r], send = b3 ? v2[0][r] : v2[1][r];
        out[r] = keep + __shfl_xor(send, 32, 64);
      }
    } else {
#pragma unroll
      for (int t = 0; t < RG; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          f32x4 v = acc[t][m];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = v[r];
            x += dpp_mov<0x124>(x);   // row_ror:4
            x += dpp_mov<0x128>(x);   // row_ror:8
            x += __shfl_xor(x, 16, 64);
            x += __shfl_xor(x, 32, 64);
            v[r] = x;
          }
          out = (b == 4 * t + m) ? v : out;
        }
    }
    const int query = b < 4 * RG ? 4 * (b & 3) + i4 : GQ;   // lanes past the (t, m) combinations hold nothing
    const int64_t ob = row0 + 4 * (b >> 2);
    if constexpr (FUSE) {
      if (query < nq) {
        const float thr = fthr[(int64_t)query * fthr_stride];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = out[r];
          if (!(v < thr) && ob + r < blk1) {
            const uint64_t key = ((uint64_t)score_key(v) << 32) | (uint32_t)(ob + r);
            const uint32_t ls = atomicAdd(lcount, 1u);
            if (ls < (uint32_t)GEMM_LCAND) {
              lkey[ls] = key;
              lqid[ls] = (uint32_t)query;
            } else {
              const uint32_t slot = atomicAdd(fstate + (int64_t)query * fstate_stride, 1u);
              if (slot < fcap) fcand[(int64_t)query * fcap + slot] = key;
              else if (fthr_stride) *(volatile float*)(fthr + (int64_t)query * fthr_stride) = __builtin_inff();   // (full list: see below)
            }
          }
        }
      }
    } else if (query < nq) {
      float* o = scores + (int64_t)query * sstride;
      if (ob + 3 < blk1) *(f32x4*)(o + ob) = out;
      else
        for (int r = 0; r < 4; ++r)
          if (ob + r < blk1) o[ob + r] = out[r];
    }
  }
  if constexpr (FUSE) {
    __syncthreads();
    const uint32_t parked = lcount[0] < (uint32_t)GEMM_LCAND ? lcount[0] : (uint32_t)GEMM_LCAND;
    for (uint32_t i = threadIdx.x; i < parked; i += GEMM_WAVES * 64) {
      const uint32_t q = lqid[i];
      const uint32_t slot = atomicAdd(fstate + (int64_t)q * fstate_stride, 1u);
      if (slot < fcap) fcand[(int64_t)q * fcap + slot] = lkey[i];
      // The query's list is full (it will be handed back for a re-run whatever else arrives): its threshold goes to +inf,
      // so that the workgroups that start after this store offer nothing for it -- a corpus sorted by similarity to the
      // queries otherwise sends every score of every block through the atomics above (12 ms instead of 1 per 16 queries)
      else if (fthr_stride) *(volatile float*)(fthr + (int64_t)q * fthr_stride) = __builtin_inff();
    }
  }
}

}  // namespace svs
