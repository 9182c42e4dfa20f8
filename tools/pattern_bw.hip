// Access-pattern microbenchmark (tools only): the 16-query kernel's load pattern
// (lane (r16,g) reads float4 #g of step s of row r16: 16 rows x 64 B per instruction)
// with the MFMAs replaced by adds, against rows-contiguous loads.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
#include "../svs_amd/csrc/gemv_f32.h"
#include "../svs_amd/csrc/gemm_q16.h"
using namespace svs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int TILES, int PF, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void frag_pattern(const float* __restrict__ M, float* __restrict__ out, int64_t n, int ld, int rows_per_block) {
  extern __shared__ float dyn[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, g = lane >> 4, ksteps = ld >> 4;
  const int64_t blk0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t blk1 = blk0 + rows_per_block < n ? blk0 + rows_per_block : n;
  float acc = 0.f;
  for (int64_t row0 = blk0 + wave * 16 * TILES; row0 < blk1; row0 += WAVES * 16 * TILES) {
    const v4f* p[TILES];
    for (int t = 0; t < TILES; ++t) { int64_t r = row0 + 16 * t + r16; r = r < n ? r : n - 1; p[t] = (const v4f*)(M + r * ld + 4 * g); }
    for (int s0 = 0; s0 < ksteps; s0 += PF) {
      v4f a[TILES][PF];
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) a[t][j] = ldg4<NT>(p[t] + 4 * (s0 + j));
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) acc += a[t][j].x + a[t][j].y + a[t][j].z + a[t][j].w;
    }
  }
  if (acc == 12345.678f) out[0] = acc + dyn[0];
}

// frag pattern + the real kernel's MFMAs.  MODE 1: B operand from a register; 2: B from LDS (like the product kernel)
template <int TILES, int PF, int MODE, int STORE = 0>
__global__ __launch_bounds__(512) void frag_mfma(const float* __restrict__ M, float* __restrict__ out, int64_t n, int ld, int rows_per_block, float* __restrict__ S = nullptr) {
  extern __shared__ v4f qlds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, g = lane >> 4, ksteps = ld >> 4;
  if (MODE == 2) { for (int e = threadIdx.x; e < ksteps * 64; e += 512) qlds[e] = (v4f){1.f, 2.f, 3.f, 4.f}; __syncthreads(); }
  const int64_t blk0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t blk1 = blk0 + rows_per_block < n ? blk0 + rows_per_block : n;
  f32x4 tot = {0, 0, 0, 0};
  for (int64_t row0 = blk0 + wave * 16 * TILES; row0 < blk1; row0 += 8 * 16 * TILES) {
    const v4f* p[TILES]; f32x4 acc[TILES];
    for (int t = 0; t < TILES; ++t) { int64_t r = row0 + 16 * t + r16; r = r < n ? r : n - 1; p[t] = (const v4f*)(M + r * ld + 4 * g); acc[t] = (f32x4){0, 0, 0, 0}; }
    v4f a[TILES][PF];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int j = 0; j < PF; ++j) a[t][j] = ldg4<false>(p[t] + 4 * j);
    auto mul = [&](int s0) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        v4f qf = MODE == 2 ? qlds[(s0 + j) * 64 + lane] : (v4f){1.f, 2.f, 3.f, (float)lane};
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
    int s0 = 0;
    for (; s0 + PF < ksteps; s0 += PF) {
      v4f nx[TILES][PF];
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) nx[t][j] = ldg4<false>(p[t] + 4 * (s0 + PF + j));
      mul(s0);
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) a[t][j] = nx[t][j];
    }
    mul(s0);
    for (int t = 0; t < TILES; ++t) tot += acc[t];
    if (STORE == 1) {        // product layout: [q][n], per instruction 16 queries x 64 B
      for (int t = 0; t < TILES; ++t) *(f32x4*)(S + (int64_t)r16 * n + row0 + 16 * t + 4 * g) = acc[t];
    } else if (STORE == 2) { // [q][n], per instruction 8 queries x one full 128 B line (32 rows)
      for (int t = 0; t < TILES; ++t) *(f32x4*)(S + (int64_t)((lane >> 3) + 8 * t) * n + row0 + 4 * (lane & 7)) = acc[t];
    } else if (STORE == 3) { // interleaved [n][16]: per instruction 1 KiB contiguous
      for (int t = 0; t < TILES; ++t) *(f32x4*)(S + (row0 + 16 * t + r16) * 16 + 4 * g) = acc[t];
    } else if (STORE == 4) { // [q][n] nontemporal
      for (int t = 0; t < TILES; ++t) __builtin_nontemporal_store(acc[t], (f32x4*)(S + (int64_t)r16 * n + row0 + 16 * t + 4 * g));
    } else if (STORE == 5) { // interleaved nontemporal
      for (int t = 0; t < TILES; ++t) __builtin_nontemporal_store(acc[t], (f32x4*)(S + (row0 + 16 * t + r16) * 16 + 4 * g));
    }
  }
  if (tot.x == 12345.678f) out[0] = tot.x + tot.y + tot.z + tot.w;
}

// rows-contiguous: wave reads whole rows (1 KiB per instruction), same block geometry
template <int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void row_pattern(const float* __restrict__ M, float* __restrict__ out, int64_t n, int ld, int rows_per_block) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t blk0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t blk1 = blk0 + rows_per_block < n ? blk0 + rows_per_block : n;
  float acc = 0.f;
  for (int64_t row = blk0 + wave; row < blk1; row += WAVES) {
    const v4f* p = (const v4f*)(M + row * ld) + lane;
    v4f a[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) a[j] = ldg4<NT>(p + 64 * j);
#pragma unroll
    for (int j = 0; j < 6; ++j) acc += a[j].x + a[j].y + a[j].z + a[j].w;
  }
  if (acc == 12345.678f) out[0] = acc;
}

// K-slice pattern: wave w of 8 reads columns [ld/8*w, ld/8*(w+1)) of G 16-row tiles per iteration
template <int G, int KS, bool NT>
__global__ __launch_bounds__(512) void kslice_pattern(const float* __restrict__ M, float* __restrict__ out, int64_t n, int ld, int rows_per_block) {
  extern __shared__ float dyn[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int64_t blk0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t blk1 = blk0 + rows_per_block < n ? blk0 + rows_per_block : n;
  float acc = 0.f;
  const int col0 = wave * KS * 16 + 4 * g;
  v4f a[G][KS];
  auto issue = [&](int64_t row0, v4f (&dst)[G][KS]) {
#pragma unroll
    for (int t = 0; t < G; ++t) {
      int64_t r = row0 + 16 * t + r16; r = r < n ? r : n - 1;
      const v4f* p = (const v4f*)(M + r * ld + col0);
#pragma unroll
      for (int j = 0; j < KS; ++j) dst[t][j] = ldg4<NT>(p + 4 * j);
    }
  };
  issue(blk0, a);
  for (int64_t row0 = blk0; row0 < blk1; row0 += 16 * G) {
    v4f nx[G][KS];
    issue(row0 + 16 * G < blk1 ? row0 + 16 * G : row0, nx);
#pragma unroll
    for (int t = 0; t < G; ++t)
#pragma unroll
      for (int j = 0; j < KS; ++j) acc += a[t][j].x + a[t][j].y + a[t][j].z + a[t][j].w;
#pragma unroll
    for (int t = 0; t < G; ++t)
#pragma unroll
      for (int j = 0; j < KS; ++j) a[t][j] = nx[t][j];
  }
  if (acc == 12345.678f) out[0] = acc + dyn[0];
}

// 4x4x1-MFMA-shaped pattern: lane (b = l >> 2, i = l & 3) reads float4 #b of a 64-column step of row i:
// 4 rows x 256 contiguous bytes per instruction, RG row groups per wave, PF steps in flight
template <int RG, int PF, bool NT>
__global__ __launch_bounds__(512) void rows4_pattern(const float* __restrict__ M, float* __restrict__ out, int64_t n, int ld, int rows_per_block) {
  extern __shared__ float dyn[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i4 = lane & 3, b = lane >> 2, ksteps = ld >> 6;
  const int64_t blk0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t blk1 = blk0 + rows_per_block < n ? blk0 + rows_per_block : n;
  float acc = 0.f;
  for (int64_t row0 = blk0 + wave * 4 * RG; row0 < blk1; row0 += 8 * 4 * RG) {
    const v4f* p[RG];
    for (int t = 0; t < RG; ++t) { int64_t r = row0 + 4 * t + i4; r = r < n ? r : n - 1; p[t] = (const v4f*)(M + r * ld + 4 * b); }
    v4f a[RG][PF];
#pragma unroll
    for (int t = 0; t < RG; ++t)
#pragma unroll
      for (int j = 0; j < PF; ++j) a[t][j] = ldg4<NT>(p[t] + 16 * j);
    int s0 = 0;
    for (; s0 + PF < ksteps; s0 += PF) {
      v4f nx[RG][PF];
#pragma unroll
      for (int t = 0; t < RG; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) nx[t][j] = ldg4<NT>(p[t] + 16 * (s0 + PF + j));
#pragma unroll
      for (int t = 0; t < RG; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) { acc += a[t][j].x + a[t][j].y + a[t][j].z + a[t][j].w; a[t][j] = nx[t][j]; }
    }
#pragma unroll
    for (int t = 0; t < RG; ++t)
#pragma unroll
      for (int j = 0; j < PF; ++j) acc += a[t][j].x + a[t][j].y + a[t][j].z + a[t][j].w;
  }
  if (acc == 12345.678f) out[0] = acc + dyn[0];
}

template <class F> void timeit(const char* name, F launch, double bytes) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch();
  std::vector<float> ts;
  for (int i = 0; i < 15; ++i) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
  std::sort(ts.begin(), ts.end());
  printf("%-44s median %7.1f us  %5.2f TB/s\n", name, ts[7] * 1e3, bytes / (ts[7] * 1e-3) / 1e12); fflush(stdout);
}

int main() {
  const int64_t n = 1000000; const int ld = 1536;
  float *M, *out; CK(hipMalloc(&M, (size_t)n * ld * 4)); CK(hipMalloc(&out, 64)); CK(hipMemset(M, 0x11, (size_t)n * ld * 4));
  const double bytes = (double)n * ld * 4;
  for (int ldsb : {0, 40 * 1024, 96 * 1024}) {
    const int rpb = 1024; unsigned blocks = (unsigned)((n + rpb - 1) / rpb); char nm[96];
    CK(hipFuncSetAttribute((const void*)frag_pattern<2, 8, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)frag_pattern<2, 16, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)kslice_pattern<1, 12, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)kslice_pattern<2, 12, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)kslice_pattern<2, 12, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    snprintf(nm, 96, "frag 2 tiles PF8 8w lds=%dK", ldsb / 1024);
    timeit(nm, [&] { hipLaunchKernelGGL((frag_pattern<2, 8, 8, false>), dim3(blocks), dim3(512), ldsb, 0, M, out, n, ld, rpb); }, bytes);
    snprintf(nm, 96, "frag 2 tiles PF16 8w lds=%dK", ldsb / 1024);
    timeit(nm, [&] { hipLaunchKernelGGL((frag_pattern<2, 16, 8, false>), dim3(blocks), dim3(512), ldsb, 0, M, out, n, ld, rpb); }, bytes);
    snprintf(nm, 96, "kslice G1 lds=%dK", ldsb / 1024);
    timeit(nm, [&] { hipLaunchKernelGGL((kslice_pattern<1, 12, false>), dim3(blocks), dim3(512), ldsb, 0, M, out, n, ld, rpb); }, bytes);
    snprintf(nm, 96, "kslice G2 lds=%dK", ldsb / 1024);
    timeit(nm, [&] { hipLaunchKernelGGL((kslice_pattern<2, 12, false>), dim3(blocks), dim3(512), ldsb, 0, M, out, n, ld, rpb); }, bytes);
    snprintf(nm, 96, "kslice G2 nt lds=%dK", ldsb / 1024);
    timeit(nm, [&] { hipLaunchKernelGGL((kslice_pattern<2, 12, true>), dim3(blocks), dim3(512), ldsb, 0, M, out, n, ld, rpb); }, bytes);
  }
  {
    const int rpb = 1024; unsigned blocks = (unsigned)((n + rpb - 1) / rpb);
    CK(hipFuncSetAttribute((const void*)frag_mfma<2, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)frag_mfma<2, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)frag_mfma<1, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    timeit("frag+mfma regB lds=96K", [&] { hipLaunchKernelGGL((frag_mfma<2, 8, 1>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb); }, bytes);
    timeit("frag+mfma regB lds=0", [&] { hipLaunchKernelGGL((frag_mfma<2, 8, 1>), dim3(blocks), dim3(512), 0, 0, M, out, n, ld, rpb); }, bytes);
    timeit("frag+mfma ldsB lds=96K", [&] { hipLaunchKernelGGL((frag_mfma<2, 8, 2>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb); }, bytes);
    timeit("frag+mfma 1tile regB lds=0", [&] { hipLaunchKernelGGL((frag_mfma<1, 8, 1>), dim3(blocks), dim3(512), 0, 0, M, out, n, ld, rpb); }, bytes);
  }
  {
    const int rpb = 1024; unsigned blocks = (unsigned)((n + rpb - 1) / rpb);
    float *Q, *S; CK(hipMalloc(&Q, 16 * ld * 4)); CK(hipMalloc(&S, (size_t)16 * n * 4)); CK(hipMemset(Q, 0x11, 16 * ld * 4));
    CK(hipFuncSetAttribute((const void*)gemm_f32_q16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    timeit("product q16 kernel, constant data", [&] { hipLaunchKernelGGL((gemm_f32_q16_kernel<false>), dim3(blocks), dim3(512), 96 * 1024, 0, M, Q, S, n, ld, n, 16, rpb); }, bytes);
    timeit("product q16 kernel, nq=0 (no stores)", [&] { hipLaunchKernelGGL((gemm_f32_q16_kernel<false>), dim3(blocks), dim3(512), 96 * 1024, 0, M, Q, S, n, ld, n, 0, rpb); }, bytes);
    std::vector<float> h((size_t)64 << 20); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((rand() & 0xffff) - 32768) / 32768.f * 0.03f;
    for (size_t off = 0; off < (size_t)n * ld; off += h.size()) CK(hipMemcpy(M + off, h.data(), std::min(h.size(), (size_t)n * ld - off) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Q, h.data(), 16 * ld * 4, hipMemcpyHostToDevice));
    timeit("product q16 kernel, random data", [&] { hipLaunchKernelGGL((gemm_f32_q16_kernel<false>), dim3(blocks), dim3(512), 96 * 1024, 0, M, Q, S, n, ld, n, 16, rpb); }, bytes);
    timeit("product q16 kernel, random, no stores", [&] { hipLaunchKernelGGL((gemm_f32_q16_kernel<false>), dim3(blocks), dim3(512), 96 * 1024, 0, M, Q, S, n, ld, n, 0, rpb); }, bytes);
    for (int nqv : {1, 4, 8}) { char nm[64]; snprintf(nm, 64, "product q16, stores for nq=%d", nqv);
      timeit(nm, [&] { hipLaunchKernelGGL((gemm_f32_q16_kernel<false>), dim3(blocks), dim3(512), 96 * 1024, 0, M, Q, S, n, ld, n, nqv, rpb); }, bytes); }
    timeit("product q16, sstride=0 (all queries overlap)", [&] { hipLaunchKernelGGL((gemm_f32_q16_kernel<false>), dim3(blocks), dim3(512), 96 * 1024, 0, M, Q, S, n, ld, (int64_t)0, 16, rpb); }, bytes);
    timeit("product q16, sstride=32 (interleaved-ish)", [&] { hipLaunchKernelGGL((gemm_f32_q16_kernel<false>), dim3(blocks), dim3(512), 96 * 1024, 0, M, Q, S, n, ld, (int64_t)32, 16, rpb); }, bytes);
    timeit("product q16, sstride=n+4096+32", [&] { hipLaunchKernelGGL((gemm_f32_q16_kernel<false>), dim3(blocks), dim3(512), 96 * 1024, 0, M, Q, S, n, ld, (int64_t)(1 << 20), 15, rpb); }, bytes);
    CK(hipFuncSetAttribute((const void*)frag_mfma<2, 8, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)frag_mfma<2, 8, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)frag_mfma<2, 8, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)frag_mfma<2, 8, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)frag_mfma<2, 8, 2, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    timeit("frag+mfma store [q][n] 16x64B", [&] { hipLaunchKernelGGL((frag_mfma<2, 8, 2, 1>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb, S); }, bytes);
    timeit("frag+mfma store [q][n] 8x128B", [&] { hipLaunchKernelGGL((frag_mfma<2, 8, 2, 2>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb, S); }, bytes);
    timeit("frag+mfma store [n][16] 1KiB", [&] { hipLaunchKernelGGL((frag_mfma<2, 8, 2, 3>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb, S); }, bytes);
    timeit("frag+mfma store [q][n] 16x64B nt", [&] { hipLaunchKernelGGL((frag_mfma<2, 8, 2, 4>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb, S); }, bytes);
    timeit("frag+mfma store [n][16] 1KiB nt", [&] { hipLaunchKernelGGL((frag_mfma<2, 8, 2, 5>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb, S); }, bytes);
    {
      uint32_t* st; uint64_t* cand; float* thr; const int SW = 4096 + 8;
      CK(hipMalloc(&st, 16 * SW * 4)); CK(hipMalloc(&cand, (size_t)16 * 32768 * 8)); CK(hipMalloc(&thr, 64));
      CK(hipFuncSetAttribute((const void*)gemm_f32_q16_kernel<false, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024 + GEMM_FUSE_LDS));
      for (float tv : {1e30f, 0.0293f, 0.02f}) {
        std::vector<float> th(16, tv); CK(hipMemcpy(thr, th.data(), 64, hipMemcpyHostToDevice));
        char nm[64]; snprintf(nm, 64, "product q16 FUSED thr=%g", tv);
        timeit(nm, [&] { hipMemsetAsync(st, 0, 16 * SW * 4, 0); hipLaunchKernelGGL((gemm_f32_q16_kernel<false, 2, 8, true>), dim3(blocks), dim3(512), 96 * 1024 + GEMM_FUSE_LDS, 0, M, Q, (float*)nullptr, n, ld, (int64_t)0, 16, rpb, st, SW, cand, 32768u, thr, 1); }, bytes);
        uint32_t c0; CK(hipMemcpy(&c0, st, 4, hipMemcpyDeviceToHost)); printf("   candidates of query 0: %u\n", c0);
      }
    }
    CK(hipFuncSetAttribute((const void*)rows4_pattern<4, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)rows4_pattern<4, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)rows4_pattern<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)rows4_pattern<2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)rows4_pattern<8, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    timeit("rows4 RG4 PF4 nt lds=96K", [&] { hipLaunchKernelGGL((rows4_pattern<4, 4, true>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb); }, bytes);
    timeit("rows4 RG4 PF4 nt lds=0", [&] { hipLaunchKernelGGL((rows4_pattern<4, 4, true>), dim3(blocks), dim3(512), 0, 0, M, out, n, ld, rpb); }, bytes);
    timeit("rows4 RG4 PF4 plain lds=96K", [&] { hipLaunchKernelGGL((rows4_pattern<4, 4, false>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb); }, bytes);
    timeit("rows4 RG4 PF2 nt lds=96K", [&] { hipLaunchKernelGGL((rows4_pattern<4, 2, true>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb); }, bytes);
    timeit("rows4 RG2 PF4 nt lds=96K", [&] { hipLaunchKernelGGL((rows4_pattern<2, 4, true>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb); }, bytes);
    timeit("rows4 RG8 PF2 nt lds=96K", [&] { hipLaunchKernelGGL((rows4_pattern<8, 2, true>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb); }, bytes);
    {
      float* S2; CK(hipMalloc(&S2, (size_t)16 * n * 4));
      CK(hipFuncSetAttribute((const void*)gemm_q16r_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      hipLaunchKernelGGL((gemm_f32_q16_kernel<false>), dim3(blocks), dim3(512), 96 * 1024, 0, M, Q, S, n, ld, n, 16, rpb);
      timeit("NEW q16r kernel (4x4x1), stores", [&] { hipLaunchKernelGGL((gemm_q16r_kernel<false, 4>), dim3(blocks), dim3(512), 96 * 1024, 0, (const v4f*)M, (const v4f*)Q, S2, n, ld / 4, n, 16, rpb); }, bytes);
      timeit("NEW q16r kernel (4x4x1), no stores", [&] { hipLaunchKernelGGL((gemm_q16r_kernel<false, 4>), dim3(blocks), dim3(512), 96 * 1024, 0, (const v4f*)M, (const v4f*)Q, S2, n, ld / 4, n, 0, rpb); }, bytes);
      hipLaunchKernelGGL((gemm_q16r_kernel<false, 4>), dim3(blocks), dim3(512), 96 * 1024, 0, (const v4f*)M, (const v4f*)Q, S2, n, ld / 4, n, 16, rpb);
      std::vector<float> h1((size_t)16 * n), h2((size_t)16 * n);
      CK(hipMemcpy(h1.data(), S, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), S2, h2.size() * 4, hipMemcpyDeviceToHost));
      double md = 0, mx = 0; for (size_t i = 0; i < h1.size(); ++i) { md = std::max(md, (double)fabsf(h1[i] - h2[i])); mx = std::max(mx, (double)fabsf(h1[i])); }
      printf("   max |old - new| = %.3g (max |score| %.3g)\n", md, mx);
#define TRY(RGv, PFv) { CK(hipFuncSetAttribute((const void*)gemm_q16r_kernel<false, 4, RGv, PFv>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); \
      timeit("q16r RG=" #RGv " PF=" #PFv " stores", [&] { hipLaunchKernelGGL((gemm_q16r_kernel<false, 4, RGv, PFv>), dim3(blocks), dim3(512), 96 * 1024, 0, (const v4f*)M, (const v4f*)Q, S2, n, ld / 4, n, 16, rpb); }, bytes); \
      timeit("q16r RG=" #RGv " PF=" #PFv " no stores", [&] { hipLaunchKernelGGL((gemm_q16r_kernel<false, 4, RGv, PFv>), dim3(blocks), dim3(512), 96 * 1024, 0, (const v4f*)M, (const v4f*)Q, S2, n, ld / 4, n, 0, rpb); }, bytes); \
      hipLaunchKernelGGL((gemm_q16r_kernel<false, 4, RGv, PFv>), dim3(blocks), dim3(512), 96 * 1024, 0, (const v4f*)M, (const v4f*)Q, S2, n, ld / 4, n, 16, rpb); \
      CK(hipMemcpy(h2.data(), S2, h2.size() * 4, hipMemcpyDeviceToHost)); md = 0; for (size_t i = 0; i < h1.size(); ++i) md = std::max(md, (double)fabsf(h1[i] - h2[i])); printf("   max diff %.3g\n", md); }
      TRY(2, 4)
    }
    timeit("frag+mfma ldsB random data", [&] { hipLaunchKernelGGL((frag_mfma<2, 8, 2>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb); }, bytes);
    timeit("frag adds random data", [&] { hipLaunchKernelGGL((frag_pattern<2, 8, 8, false>), dim3(blocks), dim3(512), 96 * 1024, 0, M, out, n, ld, rpb); }, bytes);
    timeit("rows nt random data", [&] { hipLaunchKernelGGL((row_pattern<8, true>), dim3(blocks), dim3(512), 0, 0, M, out, n, ld, rpb); }, bytes);
  }
  for (int rpb : std::vector<int>{}) {
    unsigned blocks = (unsigned)((n + rpb - 1) / rpb);
    char nm[96];
    snprintf(nm, 96, "frag 2 tiles PF8 8w rpb=%d", rpb);
    timeit(nm, [&] { hipLaunchKernelGGL((frag_pattern<2, 8, 8, false>), dim3(blocks), dim3(512), 0, 0, M, out, n, ld, rpb); }, bytes);
    snprintf(nm, 96, "frag 2 tiles PF8 8w nt rpb=%d", rpb);
    timeit(nm, [&] { hipLaunchKernelGGL((frag_pattern<2, 8, 8, true>), dim3(blocks), dim3(512), 0, 0, M, out, n, ld, rpb); }, bytes);
    snprintf(nm, 96, "frag 1 tile PF16 8w rpb=%d", rpb);
    timeit(nm, [&] { hipLaunchKernelGGL((frag_pattern<1, 16, 8, false>), dim3(blocks), dim3(512), 0, 0, M, out, n, ld, rpb); }, bytes);
    snprintf(nm, 96, "frag 1 tile PF8 16w rpb=%d", rpb);
    timeit(nm, [&] { hipLaunchKernelGGL((frag_pattern<1, 8, 16, false>), dim3(blocks), dim3(1024), 0, 0, M, out, n, ld, rpb); }, bytes);
    snprintf(nm, 96, "rows 8w rpb=%d", rpb);
    timeit(nm, [&] { hipLaunchKernelGGL((row_pattern<8, false>), dim3(blocks), dim3(512), 0, 0, M, out, n, ld, rpb); }, bytes);
    snprintf(nm, 96, "rows 8w nt rpb=%d", rpb);
    timeit(nm, [&] { hipLaunchKernelGGL((row_pattern<8, true>), dim3(blocks), dim3(512), 0, 0, M, out, n, ld, rpb); }, bytes);
  }
  return 0;
}
