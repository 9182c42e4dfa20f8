// Micro-benchmarks behind gemm_phased.h's schedule (tools only): what a raw s_barrier, a 16-MFMA
// cluster, a fragment-read burst and an LDS-DMA issue cost a 512-thread workgroup per CU, alone and
// combined in the staggered two-group pattern.  One workgroup per CU, cycles from s_memtime around
// the loop (median over workgroups).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// MODE bits: 1 = MFMA cluster (16), 2 = fragment reads (8 x ds_read_b128), 4 = LDS-DMA (2 x 1 KiB, from a 64 KiB
// L2-resident buffer), 8 = staggered (waves 4-7 one barrier behind; the cluster and the loads then alternate),
// 16 = only ONE barrier per iteration (both groups in the same segment order)
// PAT: DMA source of one instruction: 0 = 1 KiB contiguous, from 64 KiB per workgroup (L2 hits); 1 = 8 rows x 128 B at a
// 3,072-byte stride, the same 64 KiB-ish footprint (L2 hits); 2 = the GEMM's real stream: 8 rows x 128 B of a 256-row
// tile, k-tile after k-tile, a new 768 KiB tile every 24 iterations (HBM / MALL)
template <int MODE, int PAT = 0>
__global__ __launch_bounds__(512) void micro(const uint8_t* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ u32x4 lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v4f acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = (v4f){0, 0, 0, 0};
  u32x4 fa[8];
  for (int i = 0; i < 8; ++i) fa[i] = (u32x4){0x3c003c00u + lane, 0x3c003c00u, 0x38003800u, 0x3c003c00u};
  const unsigned adr = (unsigned)(threadIdx.x * 16);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 65536), 0, 65536, 0x00020000);
  int voff = lane * 16 + wave * 1024;
  if (PAT >= 1) voff = ((lane >> 3) + 8 * wave) * 3072 + (lane & 7) * 16;   // rows 8 wave .. 8 wave + 7, one 128-byte line each
  const bool behind = (MODE & 8) && wave >= 4;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (behind) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
    // ---- load segment
    if (MODE & 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[i]) : "v"(adr), "n"(i * 8192));
    }
    if (MODE & 4) {
      if (PAT == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 4096 + wave * 64), 16, voff, (it & 3) * 16384, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 4096 + 512 + wave * 64), 16, voff, (it & 3) * 16384 + 8192, 0, 0);
      } else if (PAT == 1) {   // 128 rows x 3 KiB = 384 KiB footprint per workgroup, 2 k-tiles of it re-read
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 393216), 0, 393216, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (__attribute__((address_space(3))) void*)(lds + 4096 + wave * 64), 16, voff, (it & 1) * 128, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (__attribute__((address_space(3))) void*)(lds + 4096 + 512 + wave * 64), 16, voff + 64 * 3072, (it & 1) * 128, 0, 0);
      } else {                 // a fresh 128-row x 3 KiB half-tile stream: tile changes every 24 iterations
        const size_t tile = (size_t)blockIdx.x + (size_t)(it / 24) * gridDim.x;
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (tile % 2600) * 393216), 0, 393216, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (__attribute__((address_space(3))) void*)(lds + 4096 + wave * 64), 16, voff, (it % 24) * 128, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (__attribute__((address_space(3))) void*)(lds + 4096 + 512 + wave * 64), 16, voff + 64 * 3072, (it % 24) * 128, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    }
    if (!(MODE & 16)) __builtin_amdgcn_s_barrier();
    if (MODE & 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7])::"memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- multiply segment
    if (MODE & 1) {
      __builtin_amdgcn_s_setprio(1);
      if (MODE & 32) {   // the GEMM's order: 8 accumulators of one quadrant, two dependent MFMAs each (k halves), quadrants by turns
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          if ((it & 3) == qd) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int t = 0; t < 8; ++t)
                acc[qd * 8 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa[(t >> 1) * 2 + h]), __builtin_bit_cast(h8, fa[(t & 1) * 2 + h]), acc[qd * 8 + t], 0, 0, 0);
          }
      } else if (MODE & 64) {   // 8 accumulators, two dependent MFMAs each, 8 apart; no quadrant switch
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int t = 0; t < 8; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa[(t >> 1) * 2 + h]), __builtin_bit_cast(h8, fa[(t & 1) * 2 + h]), acc[t], 0, 0, 0);
      } else if (MODE & 128) {  // 16 independent accumulators, the GEMM's operand pattern
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int t = 0; t < 8; ++t)
            acc[h * 8 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa[(t >> 1) * 2 + h]), __builtin_bit_cast(h8, fa[(t & 1) * 2 + h]), acc[h * 8 + t], 0, 0, 0);
      } else if (MODE & 256) {  // dependent pairs ADJACENT (h inner): t, t, t+1, t+1, ...
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa[(t >> 1) * 2 + h]), __builtin_bit_cast(h8, fa[(t & 1) * 2 + h]), acc[t], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa[i & 7]), __builtin_bit_cast(h8, fa[(i + 3) & 7]), acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if ((MODE & 8) && !behind) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float t = 0;
  for (int i = 0; i < 32; ++i) t += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * 512 + threadIdx.x] = t;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int PAT = 0>
void run(const char* name, const uint8_t* src, float* out, unsigned long long* cyc, int cus) {
  const int iters = 20000, lds = 131072;
  CK(hipFuncSetAttribute((const void*)micro<MODE, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((micro<MODE, PAT>), dim3(cus), dim3(512), lds, 0, src, out, cyc, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(cus); CK(hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("  %-72s %7.1f cycles per iteration (min %7.1f, max %7.1f)\n", name, (double)h[cus / 2] / iters, (double)h[0] / iters, (double)h[cus - 1] / iters);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint8_t* src; float* out; unsigned long long* cyc;
  CK(hipMalloc(&src, (size_t)2600 * 393216)); CK(hipMemset(src, 0x3c, (size_t)2600 * 393216));   // 1 GB: past the Infinity Cache
  CK(hipMalloc(&out, (size_t)cus * 512 * 4)); CK(hipMalloc(&cyc, cus * 8));
  printf("512-thread workgroups, one per CU (%d); an iteration = [loads] barrier [16 MFMAs per wave] barrier\n", cus);
  run<0>("two barriers, nothing else", src, out, cyc, cus);
  run<16>("one barrier, nothing else", src, out, cyc, cus);
  run<1>("16 MFMAs per wave (2 waves per SIMD: 512 cycles of matrix pipe)", src, out, cyc, cus);
  run<1 | 16>("16 MFMAs per wave, one barrier", src, out, cyc, cus);
  run<2>("8 ds_read_b128 per wave", src, out, cyc, cus);
  run<4>("2 LDS-DMA instructions per wave (16 KiB per CU and iteration)", src, out, cyc, cus);
  run<1 | 2>("reads + MFMAs, all waves in step", src, out, cyc, cus);
  run<1 | 2 | 4>("reads + DMA + MFMAs, all waves in step", src, out, cyc, cus);
  run<1 | 8>("staggered: MFMAs only (a SIMD's two waves alternate: 256 cycles per half)", src, out, cyc, cus);
  run<1 | 2 | 8>("staggered: reads + MFMAs", src, out, cyc, cus);
  run<1 | 4 | 8>("staggered: DMA + MFMAs", src, out, cyc, cus);
  run<1 | 2 | 4 | 8>("staggered: reads + DMA + MFMAs", src, out, cyc, cus);
  run<2 | 4 | 8>("staggered: reads + DMA, no MFMAs", src, out, cyc, cus);
  run<1 | 2 | 4 | 8 | 32>("staggered: reads + DMA + MFMAs in the GEMM's accumulator order", src, out, cyc, cus);
  run<1 | 8 | 32>("staggered: MFMAs only, the GEMM's accumulator order", src, out, cyc, cus);
  run<1 | 8 | 64>("staggered: MFMAs only, 8 accumulators x 2 dependent (8 apart), no quadrant switch", src, out, cyc, cus);
  run<1 | 8 | 128>("staggered: MFMAs only, 16 independent accumulators, GEMM operand pattern", src, out, cyc, cus);
  run<1 | 8 | 256>("staggered: MFMAs only, dependent pairs adjacent", src, out, cyc, cus);
  run<1 | 64>("in step: 8 accumulators x 2 dependent", src, out, cyc, cus);
  run<1 | 128>("in step: 16 independent, GEMM operand pattern", src, out, cyc, cus);
  printf("DMA source pattern (one instruction = 8 rows x 128 B at a 3 KiB stride instead of 1 KiB contiguous):\n");
  run<4, 1>("2 LDS-DMA per wave, strided rows, L2-resident", src, out, cyc, cus);
  run<4, 2>("2 LDS-DMA per wave, strided rows, streaming 1 GB", src, out, cyc, cus);
  run<1 | 2 | 4 | 8, 1>("staggered: reads + DMA + MFMAs, strided rows, L2-resident", src, out, cyc, cus);
  run<1 | 2 | 4 | 8, 2>("staggered: reads + DMA + MFMAs, strided rows, streaming 1 GB", src, out, cyc, cus);
  run<2 | 4 | 8, 2>("staggered: reads + DMA, no MFMAs, strided rows, streaming 1 GB", src, out, cyc, cus);
  return 0;
}
