// Score stage, up to 16 queries per corpus pass, f32 corpus, exact f32 math
// (batches of more than 16 queries take gemm_tiled.h with EB = 4, 32 per pass):
//     scores[j][i] = sum_d M[i,d] * Q[j,d]        j < 16
// The reference has no batched entry (a batch is a loop of np.dot calls,
// src/svs/kb.py:1623); this kernel amortises ONE read of the corpus over 16
// queries, which is what lifts the path from ~1.1 k to > 10 k queries/s
// (BASELINE.json north_star) while staying HBM-bound: 16 x 2 flop per 4 corpus
// bytes = 8 flop/B, far below the f32 ridge.
//
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

namespace svs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GQ = 16;        // queries per pass
constexpr int GEMM_WAVES = 8;
constexpr int GEMM_PF = 8;    // k-steps prefetched per tile half

// Q16: [16][ld] f32 (rows >= nq zero).  scores: [16][sstride].
template <bool NT, int TILES = 2, int PF = GEMM_PF>
__global__ __launch_bounds__(GEMM_WAVES * 64) void gemm_f32_q16_kernel(
    const float* __restrict__ M, const float* __restrict__ Q16, float* __restrict__ scores,
    int64_t n, int ld, int64_t sstride, int nq, int rows_per_block) {
  extern __shared__ v4f qlds[];  // [ld/16][64]
  const int ksteps = ld >> 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // ---- stage the queries in MFMA-B order
  for (int e = threadIdx.x; e < ksteps * 64; e += GEMM_WAVES * 64) {
    const int s = e >> 6, l = e & 63;
    qlds[e] = *(const v4f*)(Q16 + (int64_t)(l & 15) * ld + 16 * s + 4 * (l >> 4));
  }
  __syncthreads();

  const int64_t blk0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t blk1 = blk0 + rows_per_block < n ? blk0 + rows_per_block : n;
  const int r16 = lane & 15, g = lane >> 4;
  constexpr int TROWS = 16 * TILES;
  for (int64_t row0 = blk0 + wave * TROWS; row0 < blk1; row0 += GEMM_WAVES * TROWS) {
    const v4f* p[TILES];
    f32x4 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      int64_t r = row0 + 16 * t + r16;
      r = r < n ? r : n - 1;  // clamp: never read past the matrix
      p[t] = (const v4f*)(M + r * ld + 4 * g);
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    v4f a[TILES][PF];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int j = 0; j < PF; ++j) a[t][j] = ldg4<NT>(p[t] + 4 * j);   // step j: columns 16 j + 4 g .. +4
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
    // Steady state has NO conditional around the prefetch: a branch there makes hipcc wait
    // for the just-issued loads at the join (the same trap as in gemv_f32.h); the last
    // chunk is peeled instead.
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
    mul_chunk(s0);
    // D layout: column (query) = lane & 15, rows 4 g + r of the 16-row tile
    if (r16 < nq) {
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
}

}  // namespace svs
