// Micro (tools only): would a single-query f16 kernel keep its HBM rate on an 8-row-interleaved corpus image?
//   row-major (today):  addr(row, byte) = row * ld + byte                      -> one wave per row, NSTEP 1 KiB loads
//   interleaved:        addr(row, byte) = (row >> 3) * 8 * ld + (byte >> 7) * 1024 + (row & 7) * 128 + (byte & 127)
//                       -> one wave per GROUP of 8 rows: NK = ld / 128 loads of 1 KiB (8 rows x 128 B each), the query's
//                          16-byte chunk of that k-tile from LDS (8 distinct addresses per read), a 3-step sum over the 8
//                          lanes of a row at the end
// usage: gemv_il [rows=4000000] [d=1536]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../svs_amd/csrc/gemv_f16.h"
using namespace svs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// the same with the lane's NK query chunks held in registers (no LDS, no barrier); NT: nontemporal loads
template <int NK, int WPB, int UB, bool NT>
__global__ __launch_bounds__(WPB * 64) void gemv_f16_il_regq_kernel(const u32x4* __restrict__ M, const u32x4* __restrict__ qh, float* __restrict__ scores, int64_t n) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t group = (int64_t)blockIdx.x * WPB + wave;
  if (group * 8 >= n) return;
  const u32x4* p = M + group * (NK * 64) + lane;
  u32x4 buf[UB];
#pragma unroll
  for (int j = 0; j < UB; ++j) buf[j] = NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
  u32x4 qv[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) qv[k] = qh[k * 8 + (lane & 7)];
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int k0 = 0; k0 < NK; k0 += UB) {
    u32x4 nxt[UB];
    if (k0 + UB < NK) {
#pragma unroll
      for (int j = 0; j < UB; ++j) nxt[j] = NT ? __builtin_nontemporal_load(p + (k0 + UB + j) * 64) : p[(k0 + UB + j) * 64];
    }
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      if (j & 1) s1 = dot8(buf[j], qv[k0 + j], s1); else s0 = dot8(buf[j], qv[k0 + j], s0);
    }
    if (k0 + UB < NK) {
#pragma unroll
      for (int j = 0; j < UB; ++j) buf[j] = nxt[j];
    }
  }
  float s = s0 + s1;
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  const int64_t row = group * 8 + (lane >> 3);
  if ((lane & 7) == 0 && row < n) scores[row] = s;
}

template <int NK, int WPB, int UB>   // UB: loads issued per batch
__global__ __launch_bounds__(WPB * 64) void gemv_f16_il_kernel(const u32x4* __restrict__ M, const u32x4* __restrict__ qh, float* __restrict__ scores, int64_t n) {
  __shared__ u32x4 qs[NK * 8];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < NK * 8; i += WPB * 64) qs[i] = qh[i];
  __syncthreads();
  const int64_t group = (int64_t)blockIdx.x * WPB + wave;
  if (group * 8 >= n) return;
  const u32x4* p = M + group * (NK * 64) + lane;       // the group's 8 rows are one contiguous span of NK KiB
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int k0 = 0; k0 < NK; k0 += UB) {
    u32x4 buf[UB];
#pragma unroll
    for (int j = 0; j < UB; ++j) buf[j] = __builtin_nontemporal_load(p + (k0 + j) * 64);
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const u32x4 qv = qs[(k0 + j) * 8 + (lane & 7)];
      if (j & 1) s1 = dot8(buf[j], qv, s1); else s0 = dot8(buf[j], qv, s0);
    }
  }
  float s = s0 + s1;
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  const int64_t row = group * 8 + (lane >> 3);
  if ((lane & 7) == 0 && row < n) scores[row] = s;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 4000000;
  const int d = argc > 2 ? atoi(argv[2]) : 1536;
  if (d != 1536) { printf("d = 1536 only\n"); return 1; }
  const size_t bytes = (size_t)n * d * 2;
  uint16_t* M; float* qf; uint16_t* qh; float* sc;
  CK(hipMalloc(&M, bytes)); CK(hipMalloc(&qf, d * 4)); CK(hipMalloc(&qh, d * 2)); CK(hipMalloc(&sc, n * 4));
  { std::vector<uint16_t> h((size_t)32 << 20); for (auto& x : h) x = (uint16_t)(0x2000 + (rand() & 0x0fff)); for (size_t off = 0; off < bytes; off += h.size() * 2) CK(hipMemcpy((char*)M + off, h.data(), std::min(h.size() * 2, bytes - off), hipMemcpyHostToDevice));
    std::vector<float> q(d, 0.01f); CK(hipMemcpy(qf, q.data(), d * 4, hipMemcpyHostToDevice)); std::vector<uint16_t> q16(d, 0x211f); CK(hipMemcpy(qh, q16.data(), d * 2, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0)); for (int i = 0; i < 4; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 4); }
    printf("  %-66s %7.3f ms  %5.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
  };
  printf("%lld x %d f16 (%.1f GB), single query\n", (long long)n, d, bytes / 1e9);
  timeit("row-major, gemv_f16_oneshot_kernel<3, 2, 16> (today)", [&] { hipLaunchKernelGGL((gemv_f16_oneshot_kernel<3, 2, 16>), dim3((unsigned)((n + 31) / 32)), dim3(1024), 0, 0, (const u32x4*)M, (const v4f*)qf, sc, n); });
  timeit("interleaved, 16 waves, 6 loads per batch", [&] { hipLaunchKernelGGL((gemv_f16_il_kernel<24, 16, 6>), dim3((unsigned)((n / 8 + 15) / 16)), dim3(1024), 0, 0, (const u32x4*)M, (const u32x4*)qh, sc, n); });
  timeit("interleaved, 16 waves, 12 loads per batch", [&] { hipLaunchKernelGGL((gemv_f16_il_kernel<24, 16, 12>), dim3((unsigned)((n / 8 + 15) / 16)), dim3(1024), 0, 0, (const u32x4*)M, (const u32x4*)qh, sc, n); });
  timeit("interleaved, 16 waves, 24 loads per batch", [&] { hipLaunchKernelGGL((gemv_f16_il_kernel<24, 16, 24>), dim3((unsigned)((n / 8 + 15) / 16)), dim3(1024), 0, 0, (const u32x4*)M, (const u32x4*)qh, sc, n); });
  timeit("interleaved, 8 waves, 12 loads per batch", [&] { hipLaunchKernelGGL((gemv_f16_il_kernel<24, 8, 12>), dim3((unsigned)((n / 8 + 7) / 8)), dim3(512), 0, 0, (const u32x4*)M, (const u32x4*)qh, sc, n); });
  timeit("interleaved, 4 waves, 12 loads per batch", [&] { hipLaunchKernelGGL((gemv_f16_il_kernel<24, 4, 12>), dim3((unsigned)((n / 8 + 3) / 4)), dim3(256), 0, 0, (const u32x4*)M, (const u32x4*)qh, sc, n); });
  timeit("interleaved, query in registers, 16 waves, 12 loads in flight, nt", [&] { hipLaunchKernelGGL((gemv_f16_il_regq_kernel<24, 16, 12, true>), dim3((unsigned)((n / 8 + 15) / 16)), dim3(1024), 0, 0, (const u32x4*)M, (const u32x4*)qh, sc, n); });
  timeit("interleaved, query in registers, 16 waves, 6 loads in flight, nt", [&] { hipLaunchKernelGGL((gemv_f16_il_regq_kernel<24, 16, 6, true>), dim3((unsigned)((n / 8 + 15) / 16)), dim3(1024), 0, 0, (const u32x4*)M, (const u32x4*)qh, sc, n); });
  timeit("interleaved, query in registers, 8 waves, 12 loads in flight, nt", [&] { hipLaunchKernelGGL((gemv_f16_il_regq_kernel<24, 8, 12, true>), dim3((unsigned)((n / 8 + 7) / 8)), dim3(512), 0, 0, (const u32x4*)M, (const u32x4*)qh, sc, n); });
  timeit("interleaved, query in registers, 16 waves, 12 loads in flight, default policy", [&] { hipLaunchKernelGGL((gemv_f16_il_regq_kernel<24, 16, 12, false>), dim3((unsigned)((n / 8 + 15) / 16)), dim3(1024), 0, 0, (const u32x4*)M, (const u32x4*)qh, sc, n); });
  timeit("row-major again", [&] { hipLaunchKernelGGL((gemv_f16_oneshot_kernel<3, 2, 16>), dim3((unsigned)((n + 31) / 32)), dim3(1024), 0, 0, (const u32x4*)M, (const v4f*)qf, sc, n); });
  return 0;
}
