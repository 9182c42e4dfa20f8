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
template <int MODE>
__global__ __launch_bounds__(512) void micro(const uint8_t* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ u32x4 lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v4f acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (v4f){0, 0, 0, 0};
  u32x4 fa[8];
  for (int i = 0; i < 8; ++i) fa[i] = (u32x4){0x3c003c00u + lane, 0x3c003c00u, 0x38003800u, 0x3c003c00u};
  const unsigned adr = (unsigned)(threadIdx.x * 16);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 65536), 0, 65536, 0x00020000);
  const int voff = lane * 16 + wave * 1024;
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
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 4096 + wave * 64), 16, voff, (it & 3) * 16384, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 4096 + 512 + wave * 64), 16, voff, (it & 3) * 16384 + 8192, 0, 0);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    }
    if (!(MODE & 16)) __builtin_amdgcn_s_barrier();
    if (MODE & 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7])::"memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- multiply segment
    if (MODE & 1) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa[i & 7]), __builtin_bit_cast(h8, fa[(i + 3) & 7]), acc[i], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if ((MODE & 8) && !behind) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float t = 0;
  for (int i = 0; i < 16; ++i) t += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * 512 + threadIdx.x] = t;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const uint8_t* src, float* out, unsigned long long* cyc, int cus) {
  const int iters = 20000, lds = 131072;
  CK(hipFuncSetAttribute((const void*)micro<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(micro<MODE>, dim3(cus), dim3(512), lds, 0, src, out, cyc, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(cus); CK(hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("  %-72s %7.1f cycles per iteration (min %7.1f, max %7.1f)\n", name, (double)h[cus / 2] / iters, (double)h[0] / iters, (double)h[cus - 1] / iters);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint8_t* src; float* out; unsigned long long* cyc;
  CK(hipMalloc(&src, (size_t)cus * 65536)); CK(hipMemset(src, 0x3c, (size_t)cus * 65536));
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
  return 0;
}
