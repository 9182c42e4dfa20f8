// Single-query score stage for row lengths that are NOT a whole number of 1 KiB wave loads
// (f16 d = 768 or 384, fp8 d = 1536, 768, 384, ...: the exact geometries have their own
// kernels in gemv_f16.h / fp8.h).  T lanes share a row (T a power of two >= the row's 16-byte
// chunks, up to the whole wave), so a wave instruction covers 64 / T adjacent rows -- one
// contiguous span, rows being packed -- and rows longer than a wave take NC chunks per lane.
// What the plain loop kernels (gemv_*_generic_kernel) lack is memory-level parallelism: one
// load in flight per wave, 2.7-5.0 TB/s.  Here U row groups are requested before the first is
// used (nontemporal, like every once-read corpus stream) and the lane's query chunks are
// hoisted out of the row loop (they do not depend on the row).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fp8.h"
#include "gemv_f16.h"

namespace svs {

// A Dot functor: prep() turns the lane's 16-byte query chunk into whatever the inner product
// wants (done once per wave), dot() multiplies one 16-byte row chunk, pre() fetches what
// finish() needs about the row (its scale) together with the row, finish() applies it.
struct DotF32 {   // 4 floats per chunk, plain FMAs: the reference's arithmetic
  typedef v4f Q;
  __device__ __forceinline__ Q prep(u32x4 q) const { return __builtin_bit_cast(v4f, q); }
  __device__ __forceinline__ float dot(u32x4 a, const Q& q, float acc) const { return dot4(__builtin_bit_cast(v4f, a), q, acc); }
  __device__ __forceinline__ float pre(int64_t) const { return 1.f; }
  __device__ __forceinline__ float finish(float v, float) const { return v; }
};

struct DotF16 {   // rows and query in halves, f32 accumulate (v_dot2_f32_f16)
  typedef u32x4 Q;
  __device__ __forceinline__ Q prep(u32x4 q) const { return q; }
  __device__ __forceinline__ float dot(u32x4 a, const Q& q, float acc) const { return dot8(a, q, acc); }
  __device__ __forceinline__ float pre(int64_t) const { return 1.f; }
  __device__ __forceinline__ float finish(float v, float) const { return v; }
};

struct DotFp8 {   // rows in e4m3 (16 per chunk); the query chunk is widened to f32 once; scales at the end
  struct Q { float v[16]; };
  const float* row_scales;
  const float* q_scale;
  __device__ __forceinline__ Q prep(u32x4 q) const {
    Q r;
    const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float u[4];
      unpack_fp8x4(qw[w], u);
      r.v[4 * w] = u[0]; r.v[4 * w + 1] = u[1]; r.v[4 * w + 2] = u[2]; r.v[4 * w + 3] = u[3];
    }
    return r;
  }
  __device__ __forceinline__ float dot(u32x4 a, const Q& q, float acc) const {
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float v[4];
      unpack_fp8x4(aw[w], v);
      acc = fmaf(v[0], q.v[4 * w], acc);
      acc = fmaf(v[1], q.v[4 * w + 1], acc);
      acc = fmaf(v[2], q.v[4 * w + 2], acc);
      acc = fmaf(v[3], q.v[4 * w + 3], acc);
    }
    return acc;
  }
  __device__ __forceinline__ float pre(int64_t row) const { return row_scales[row] * q_scale[0]; }   // requested with the row itself
  __device__ __forceinline__ float finish(float v, float scale) const { return v * scale; }
};

// Sum over the T lanes that share a row, result in (at least) the segment's first lane.
// DPP only up to 16 lanes (quad permutes, then the half-row and row mirrors), one LDS-crossbar
// permute for 32, the readlane form for the whole wave: a chain of ds_bpermutes per row was
// what held the first version of this kernel at the loop kernels' speed.
template <int T>
__device__ __forceinline__ float seg_sum(float v) {
  if constexpr (T == 64) return wave_sum(v);
  if constexpr (T >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (T >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (T >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror: lane l <- 7 - l of its 8
  if constexpr (T >= 16) v += dpp_mov<0x140>(v);  // row_mirror: lane l <- 15 - l of its 16
  if constexpr (T >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

// M: rows of ld16 16-byte chunks; q: the staged query, ld16 chunks (zero padded like the rows)
// One-shot grid like the exact-geometry kernels (gemv_f32.h): a wave takes U row groups and
// ends; workgroups are dispatched in row order, which keeps the chip-wide read front compact
// (a grid-stride version of this kernel ran at 4.8 TB/s whatever the unrolling).
constexpr int UNR_WPB = 16;

template <int T, int NC, int U, class Dot>
__global__ __launch_bounds__(UNR_WPB * 64) void gemv_unrolled_kernel(
    const u32x4* __restrict__ M, const u32x4* __restrict__ q, float* __restrict__ scores, int64_t n, int ld16, Dot dot) {
  constexpr int RPW = 64 / T;
  const int lane = threadIdx.x & 63;
  const int sub = lane & (T - 1);
  const int rsub = lane / T;
  const int64_t base = ((int64_t)blockIdx.x * UNR_WPB + (threadIdx.x >> 6)) * (RPW * U);
  if (base >= n) return;
  typename Dot::Q qv[NC];
  int col[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = sub + c * T;
    col[c] = i < ld16 ? i : ld16 - 1;                       // lanes past the row re-read its last chunk
    qv[c] = dot.prep(i < ld16 ? q[i] : (u32x4){0u, 0u, 0u, 0u});   // ... against a zero query chunk
  }
  {
    u32x4 a[U][NC];
    float extra[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t row = base + u * RPW + rsub;
      row = row < n ? row : n - 1;
      const u32x4* p = M + row * ld16;
#pragma unroll
      for (int c = 0; c < NC; ++c) a[u][c] = __builtin_nontemporal_load(p + col[c]);
      extra[u] = dot.pre(row);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) acc = dot.dot(a[u][c], qv[c], acc);
      acc = seg_sum<T>(acc);
      const int64_t row = base + u * RPW + rsub;
      if (sub == 0 && row < n) scores[row] = dot.finish(acc, extra[u]);
    }
  }
}

}  // namespace svs
