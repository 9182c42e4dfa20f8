// Score stage, single query, f32 corpus:  scores[i] = sum_d M[i,d] * q[d]
// Replaces np.dot(embeddings_matrix, query_vec), reference src/svs/kb.py:1623.
//
// Roofline: HBM.  Algorithmic bytes per query = n * ld * 4 (6.144e9 at
// 1M x 1536); 0.5 flop/byte, so the only thing that matters is keeping enough
// 16-byte loads in flight per CU and never touching a corpus byte twice.
//
// Layout (gfx950, wave = 64 lanes):
//   * one wave owns R whole rows per step.  A row of ld = NSTEP*256 floats is
//     NSTEP wave-wide global_load_dwordx4 (1 KiB each, fully coalesced): lane l
//     reads v4f #(j*64 + l) of the row, j = 0..NSTEP-1;
//   * the query lives in NSTEP v4f registers per lane, loaded once per wave;
//   * rows go straight to VGPRs (no LDS round trip: nothing is shared between
//     waves -- cdna_hip_programming.md, "GEMV / M <= 16" row), with the NEXT
//     step's rows requested before the current step is reduced (register
//     double buffering, counted vmcnt waits emitted by the compiler);
//   * per-lane partial sums are reduced with DPP adds inside 16-lane rows and
//     four scalar read-backs (no LDS, no waits).  The summation order depends
//     only on ld, never on the row's position, so a row's score is
//     bit-identical for every sharding of the corpus.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svs {

// native clang vector (HIP's v4f class is not accepted by the nontemporal builtin)
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ v4f ldg4(const v4f* p) {
  if constexpr (NT) {
    return __builtin_nontemporal_load(p);
  } else {
    return *p;
  }
}

__device__ __forceinline__ float dot4(v4f a, v4f b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  acc = fmaf(a.w, b.w, acc);
  return acc;
}

// ---- wave-wide f32 sum without LDS ------------------------------------------
// DPP adds inside each 16-lane row (xor 1, xor 2, rotate 4, rotate 8), then the
// four row sums are read back as scalars.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x128>(v);  // row_ror:8
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}

// ---- hot kernel: ld == NSTEP * 256 floats -----------------------------------
// tile t = rows [t*R, t*R + R).  Wave gw of W takes tiles gw, gw+W, ... when
// CONTIG == false (the whole grid sweeps the matrix front to back), or a
// contiguous run of tiles when CONTIG == true.  All loop control is scalar
// (wave-uniform), so prefetches are unconditional inside the steady-state loop
// and the compiler's vmcnt waits leave the next tile in flight.
template <int NSTEP, int R, int WPB, bool NT, bool CONTIG>
__global__ __launch_bounds__(WPB * 64) void gemv_f32_rows_kernel(
    const v4f* __restrict__ M, const v4f* __restrict__ q, float* __restrict__ scores,
    int64_t n) {
  constexpr int LD4 = NSTEP * 64;  // row stride in v4f
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t gw = (int64_t)blockIdx.x * WPB + wave;
  const int64_t W = (int64_t)gridDim.x * WPB;
  const int64_t ntiles = (n + R - 1) / R;

  int64_t t0, tend, tstep;
  if constexpr (CONTIG) {
    const int64_t per = (ntiles + W - 1) / W;
    t0 = gw * per;
    tend = t0 + per < ntiles ? t0 + per : ntiles;
    tstep = 1;
  } else {
    t0 = gw;
    tend = ntiles;
    tstep = W;
  }
  if (t0 >= tend) return;

  v4f qv[NSTEP];
#pragma unroll
  for (int j = 0; j < NSTEP; ++j) qv[j] = q[j * 64 + lane];

  auto load_tile = [&](v4f (&buf)[R][NSTEP], int64_t t) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t row = t * R + r;
      row = row < n ? row : n - 1;  // clamp: never read past the matrix, result discarded
      const v4f* p = M + row * LD4 + lane;
#pragma unroll
      for (int j = 0; j < NSTEP; ++j) buf[r][j] = ldg4<NT>(p + j * 64);
    }
  };
  auto finish_tile = [&](v4f (&buf)[R][NSTEP], int64_t t) {
    float out = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int j = 0; j < NSTEP; ++j) {
        if (j & 1) s1 = dot4(buf[r][j], qv[j], s1);
        else s0 = dot4(buf[r][j], qv[j], s0);
      }
      const float v = wave_sum(s0 + s1);
      out = lane == r ? v : out;
    }
    const int64_t row = t * R + lane;
    if (lane < R && row < n) scores[row] = out;
  };

  v4f A[R][NSTEP], B[R][NSTEP];
  int64_t t = t0;
  load_tile(A, t);
  while (t + 2 * tstep < tend) {
    load_tile(B, t + tstep);
    finish_tile(A, t);
    load_tile(A, t + 2 * tstep);
    finish_tile(B, t + tstep);
    t += 2 * tstep;
  }
  if (t + tstep < tend) {
    load_tile(B, t + tstep);
    finish_tile(A, t);
    finish_tile(B, t + tstep);
  } else {
    finish_tile(A, t);
  }
}

// ---- one-shot kernel: ld == NSTEP * 256 floats, one tile per wave, no loop ----
// Measured on MI355X (1M x 1536): letting the hardware dispatcher hand out
// 16-wave workgroups in row order keeps the chip-wide read front compact and
// streams at 7.3 TB/s, against 6.8 TB/s for the persistent grid-stride form
// above, whose waves drift apart.  Workgroup b owns rows
// [b*WPB*R, (b+1)*WPB*R); wave w of it owns R consecutive rows.
// QLDS: the query is staged once per workgroup in LDS and read back with
// ds_read_b128 instead of each wave fetching it from L2.
// XCDMAP (experiment): workgroups are dealt round-robin over the 8 XCDs; with the remap
// each XCD streams one contiguous eighth of the corpus instead of every eighth block.
template <int NSTEP, int R, int WPB, bool NT, bool QLDS, bool XCDMAP = false>
__global__ __launch_bounds__(WPB * 64) void gemv_f32_oneshot_kernel(
    const v4f* __restrict__ M, const v4f* __restrict__ q, float* __restrict__ scores,
    int64_t n) {
  constexpr int LD4 = NSTEP * 64;
  __shared__ v4f qs[QLDS ? LD4 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t bid = blockIdx.x;
  if constexpr (XCDMAP) {
    const int64_t nb = gridDim.x, qd = nb / 8, rm = nb % 8, x = bid % 8;
    bid = (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd) + bid / 8;   // bijective for any grid size
  }
  const int64_t row0 = (bid * WPB + wave) * R;

  v4f buf[R][NSTEP];
  if (row0 < n) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t row = row0 + r;
      row = row < n ? row : n - 1;
      const v4f* p = M + row * LD4 + lane;
#pragma unroll
      for (int j = 0; j < NSTEP; ++j) buf[r][j] = ldg4<NT>(p + j * 64);
    }
  }
  v4f qv[NSTEP];
  if constexpr (QLDS) {
    for (int i = threadIdx.x; i < LD4; i += WPB * 64) qs[i] = q[i];
    __syncthreads();
    if (row0 >= n) return;
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) qv[j] = qs[j * 64 + lane];
  } else {
    if (row0 >= n) return;
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) qv[j] = q[j * 64 + lane];
  }
  float out = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) {
      if (j & 1) s1 = dot4(buf[r][j], qv[j], s1);
      else s0 = dot4(buf[r][j], qv[j], s0);
    }
    const float v = wave_sum(s0 + s1);
    out = lane == r ? v : out;
  }
  const int64_t row = row0 + lane;
  if (lane < R && row < n) scores[row] = out;
}

// ---- generic kernel: any d (rows zero-padded to ld % 4 == 0 in HBM) ---------
// T lanes (power of two) cooperate on one row, 64/T rows per wave step.  Used
// for dimensions that are not a multiple of 256 (e.g. the reference's 3-d unit
// tests).  q is read with bounds (d need not be a multiple of 4).
template <int T>
__global__ __launch_bounds__(256) void gemv_f32_generic_kernel(
    const v4f* __restrict__ M, const float* __restrict__ q, float* __restrict__ scores,
    int64_t n, int d, int ld4) {
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
    const v4f* p = M + row * ld4;
    float acc = 0.f;
    for (int c = sub; c < ld4; c += T) {
      const v4f a = p[c];
      const int e = c * 4;
      v4f b;
      b.x = e + 0 < d ? q[e + 0] : 0.f;
      b.y = e + 1 < d ? q[e + 1] : 0.f;
      b.z = e + 2 < d ? q[e + 2] : 0.f;
      b.w = e + 3 < d ? q[e + 3] : 0.f;
      acc = dot4(a, b, acc);
    }
#pragma unroll
    for (int off = T / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0 && live) scores[row] = acc;
  }
}

}  // namespace svs
