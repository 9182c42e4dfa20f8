// f16-resident corpus (SVS_DTYPE_F16; BASELINE.json configs[2]/[3]): rows are
// rounded to IEEE half (RNE) once at upload, halving the HBM bytes per query
// (3.07 GB at 1M x 1536).  Queries are rounded to half as well, so every score is
//     sum_d f32(half(M[i,d])) * f32(half(q[d]))     accumulated in f32
// -- products of two halves are exact in f32 -- and the oracle for this dtype is
// numpy's f32 path on the dequantised corpus and query.
// Recall against the f32 corpus is a property of the rounding, not of the kernel.
//
// Single-query score kernel: same one-shot geometry as gemv_f32.h (HBM-bound,
// algorithmic bytes = n * ld * 2).  A row of ld = NSTEP*512 halves is NSTEP
// wave-wide 1 KiB loads (8 halves per lane); the query sits in NSTEP*4 packed
// half2 registers per lane; v_dot2_f32_f16 does two multiply-adds per lane-op.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "gemv_f32.h"

namespace svs {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 8 halves x 8 halves + acc (four v_dot2_f32_f16).  The operands are split with
// shufflevector on the half view: element-indexing the u32 view and bit-casting
// each element made hipcc (ROCm 7.2) reuse element 0 four times.
__device__ __forceinline__ float dot8(u32x4 a, u32x4 b, float acc) {
  const h8 x = __builtin_bit_cast(h8, a), y = __builtin_bit_cast(h8, b);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(x, x, 0, 1), __builtin_shufflevector(y, y, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(x, x, 2, 3), __builtin_shufflevector(y, y, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(x, x, 4, 5), __builtin_shufflevector(y, y, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(x, x, 6, 7), __builtin_shufflevector(y, y, 6, 7), acc, false);
  return acc;
}

// f32 rows (stride src_ld) -> half rows (stride ld16, zero padded).  grid-stride.
__global__ void convert_rows_f16_kernel(const float* __restrict__ src, int64_t n, int d, int64_t src_ld,
                                        _Float16* __restrict__ dst, int ld16) {
  const int64_t total = n * ld16;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / ld16;
    const int c = (int)(i - r * ld16);
    dst[i] = c < d ? (_Float16)src[r * src_ld + c] : (_Float16)0.f;
  }
}

// debug / parity: half rows back as f32, out stride d
__global__ void dequant_rows_f16_kernel(const _Float16* __restrict__ rows, int64_t row0, int64_t nrows, int d,
                                        int ld16, float* __restrict__ out) {
  const int64_t total = nrows * d;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / d;
    out[i] = (float)rows[(row0 + r) * ld16 + (i - r * d)];
  }
}

// queries f32 [nq][d] -> half [nq][ld16] zero padded (tiny)
__global__ void convert_queries_f16_kernel(const float* __restrict__ q, int nq, int d,
                                           _Float16* __restrict__ dst, int ld16) {
  const int total = nq * ld16;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / ld16, c = i - r * ld16;
    dst[i] = c < d ? (_Float16)q[(int64_t)r * d + c] : (_Float16)0.f;
  }
}

// ---- hot kernel: ld16 == NSTEP * 512 halves, one-shot grid, R rows per wave ----
// q is the f32 query (d = ld16 floats, 16-byte aligned): each wave rounds its
// slice to half itself (RNE), saving a conversion launch on the latency path.
template <int NSTEP, int R, int WPB>
__global__ __launch_bounds__(WPB * 64) void gemv_f16_oneshot_kernel(
    const u32x4* __restrict__ M, const v4f* __restrict__ q, float* __restrict__ scores, int64_t n) {
  constexpr int LD4 = NSTEP * 64;  // row stride in 16-byte units
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row0 = ((int64_t)blockIdx.x * WPB + wave) * R;
  if (row0 >= n) return;
  u32x4 buf[R][NSTEP];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int64_t row = row0 + r;
    row = row < n ? row : n - 1;
    const u32x4* p = M + row * LD4 + lane;
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) buf[r][j] = __builtin_nontemporal_load(p + j * 64);
  }
  u32x4 qv[NSTEP];
#pragma unroll
  for (int j = 0; j < NSTEP; ++j) {
    const v4f lo = q[(j * 64 + lane) * 2], hi = q[(j * 64 + lane) * 2 + 1];
    const h2 p0 = {(_Float16)lo.x, (_Float16)lo.y}, p1 = {(_Float16)lo.z, (_Float16)lo.w};
    const h2 p2 = {(_Float16)hi.x, (_Float16)hi.y}, p3 = {(_Float16)hi.z, (_Float16)hi.w};
    qv[j] = (u32x4){__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1),
                    __builtin_bit_cast(uint32_t, p2), __builtin_bit_cast(uint32_t, p3)};
  }
  float out = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) {
      if (j & 1) s1 = dot8(buf[r][j], qv[j], s1);
      else s0 = dot8(buf[r][j], qv[j], s0);
    }
    const float v = wave_sum(s0 + s1);
    out = lane == r ? v : out;
  }
  const int64_t row = row0 + lane;
  if (lane < R && row < n) scores[row] = out;
}

// ---- generic kernel: any d (rows zero padded to ld16 % 8 == 0) ----------------
template <int T>
__global__ __launch_bounds__(256) void gemv_f16_generic_kernel(
    const u32x4* __restrict__ M, const u32x4* __restrict__ q, float* __restrict__ scores, int64_t n,
    int ld8) {
  constexpr int RPW = 64 / T;
  const int lane = threadIdx.x & 63;
  const int sub = lane & (T - 1);
  const int rsub = lane / T;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t W = (int64_t)gridDim.x * 4;
  for (int64_t base = gw * RPW; base < n; base += W * RPW) {
    int64_t row = base + rsub;
    const bool live = row < n;
    row = live ? row : n - 1;
    const u32x4* p = M + row * ld8;
    float acc = 0.f;
    for (int c = sub; c < ld8; c += T) acc = dot8(p[c], q[c], acc);
#pragma unroll
    for (int off = T / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0 && live) scores[row] = acc;
  }
}

}  // namespace svs
