// Score stage, 16*NG queries per corpus pass, f16 corpus and queries, f32
// accumulate, on v_mfma_f32_16x16x32_f16 (BASELINE.json configs[2]: the batched
// MFMA path).  Same structure as gemm_f32.h: the query block is staged once per
// workgroup in LDS in MFMA-B order ([d/32][NG][64] x 16 B), the corpus streams
// from HBM straight into A fragments (lane l: row l&15, halves 8(l>>4)..+8 of the
// 32-wide k-step = one 16-byte load, 64 contiguous bytes per row per wave
// instruction), 2 row tiles x NG query groups of independent accumulators.
// At 32 queries per pass the kernel is still HBM-bound (8 % MFMA utilisation);
// larger batches loop passes.  A tiled GEMM that keeps a 256-query panel
// resident (MFMA-bound at B = 1024) is the next step for this config.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f32.h"
#include "gemv_f16.h"

namespace svs {

constexpr int GEMM16_PF = 8;

// Q: [16*NG][ld16] halves (rows >= nq zero).  scores: [16*NG][sstride].
template <int NG>
__global__ __launch_bounds__(GEMM_WAVES * 64) void gemm_f16_kernel(
    const _Float16* __restrict__ M, const _Float16* __restrict__ Q, float* __restrict__ scores,
    int64_t n, int ld16, int64_t sstride, int nq, int rows_per_block) {
  extern __shared__ u32x4 qlds16[];  // [ld16/32][NG][64]
  const int ksteps = ld16 >> 5;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int e = threadIdx.x; e < ksteps * NG * 64; e += GEMM_WAVES * 64) {
    const int l = e & 63, grp = (e >> 6) % NG, s = (e >> 6) / NG;
    qlds16[e] = *(const u32x4*)(Q + (int64_t)(grp * 16 + (l & 15)) * ld16 + 32 * s + 8 * (l >> 4));
  }
  __syncthreads();

  const int64_t blk0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t blk1 = blk0 + rows_per_block < n ? blk0 + rows_per_block : n;
  const int r16 = lane & 15, g = lane >> 4;
  for (int64_t row0 = blk0 + wave * 32; row0 < blk1; row0 += GEMM_WAVES * 32) {
    int64_t ra = row0 + r16, rb = row0 + 16 + r16;
    ra = ra < n ? ra : n - 1;
    rb = rb < n ? rb : n - 1;
    const u32x4* pa = (const u32x4*)(M + ra * ld16 + 8 * g);   // +4 u32x4 per 32-half step
    const u32x4* pb = (const u32x4*)(M + rb * ld16 + 8 * g);
    f32x4 acc[2][NG];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < NG; ++q) acc[t][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 a0[GEMM16_PF], a1[GEMM16_PF];
#pragma unroll
    for (int j = 0; j < GEMM16_PF; ++j) {
      a0[j] = pa[4 * j];
      a1[j] = pb[4 * j];
    }
    for (int s0 = 0; s0 < ksteps; s0 += GEMM16_PF) {
      u32x4 n0[GEMM16_PF], n1[GEMM16_PF];
      const bool more = s0 + GEMM16_PF < ksteps;
      if (more) {
#pragma unroll
        for (int j = 0; j < GEMM16_PF; ++j) {
          n0[j] = pa[4 * (s0 + GEMM16_PF + j)];
          n1[j] = pb[4 * (s0 + GEMM16_PF + j)];
        }
      }
#pragma unroll
      for (int j = 0; j < GEMM16_PF; ++j) {
        const h8 fa0 = __builtin_bit_cast(h8, a0[j]), fa1 = __builtin_bit_cast(h8, a1[j]);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
          const h8 fb = __builtin_bit_cast(h8, qlds16[((s0 + j) * NG + q) * 64 + lane]);
          acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa0, fb, acc[0][q], 0, 0, 0);
          acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa1, fb, acc[1][q], 0, 0, 0);
        }
      }
      if (more) {
#pragma unroll
        for (int j = 0; j < GEMM16_PF; ++j) {
          a0[j] = n0[j];
          a1[j] = n1[j];
        }
      }
    }
    // D layout: column (query within group) = lane & 15, rows 4 g + r of the tile
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int query = q * 16 + r16;
      if (query < nq) {
        float* o = scores + (int64_t)query * sstride;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int64_t ob = row0 + 16 * t + 4 * g;
          if (ob + 3 < blk1) *(f32x4*)(o + ob) = acc[t][q];
          else
            for (int r = 0; r < 4; ++r)
              if (ob + r < blk1) o[ob + r] = acc[t][q][r];
        }
      }
    }
  }
}

}  // namespace svs
