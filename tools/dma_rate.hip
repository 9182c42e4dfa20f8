// Micro (tools only): how many bytes per clock can ONE CU move from L2 into LDS?
//   mode 0: buffer_load_dwordx4 ... lds (LDS-DMA), 1 KiB per wave-instruction
//   mode 1: buffer_load_dwordx4 to VGPRs + ds_write_b128
//   mode 2: buffer_load_dwordx4 to VGPRs only (the vector L1 path itself)
//   mode 3: LDS-DMA, dword (256 B per wave-instruction)
// Every workgroup (512 threads unless WAVES is given) streams its own 768 KiB window (L2-resident after
// the first pass) in 128-byte row pieces at a 3072-byte row stride, like gemm_phased's A operand;
// DEPTH instructions in flight per wave (counted vmcnt).
//   usage: dma_rate [workgroups=256] [waves=8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void stream(const uint8_t* src, unsigned long long* out, int iters, int ldb, int windows) {
  extern __shared__ u32x4 lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const uint8_t* base = src + (size_t)(blockIdx.x % windows) * 256 * ldb;   // (32 windows: 4 per XCD = 3 MiB of its 4 MiB L2)
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 256 * ldb, 0x00020000);
  const int voff = (wave * 8 + (lane >> 3)) * ldb + (lane & 7) * 16;      // 8 rows x 128 B per instruction
  u32x4 r[DEPTH];
  u32x4 sink = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int kt = it % (ldb / 128);
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const int vo = voff + ((j * nw * 8) % 256) * ldb;
      if constexpr (MODE == 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (j * nw + wave) * 64), 16, vo, kt * 128, 0, 0);
      else if constexpr (MODE == 3)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (j * nw + wave) * 16), 4, voff / 4 * 1 + lane * 0 + vo % 4, kt * 128, 0, 0);
      else
        r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, kt * 128, 0);
    }
    if constexpr (MODE == 0 || MODE == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
    if constexpr (MODE == 1) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) lds[(j * nw + wave) * 64 + lane] = r[j];
    }
    if constexpr (MODE == 2) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) sink += r[j];
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (sink.x == 0x12345678u && MODE == 2) out[blockIdx.x] = 0;
  if (MODE == 1 && lds[threadIdx.x].x == 0x12345678u) out[blockIdx.x] = 1;
}

template <int MODE, int DEPTH>
void run(const char* name, const uint8_t* src, unsigned long long* out, int wgs, int waves, int ldb, int windows) {
  const int iters = 400;
  CK(hipFuncSetAttribute((const void*)stream<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((stream<MODE, DEPTH>), dim3(wgs), dim3(waves * 64), 131072, 0, src, out, iters, ldb, windows);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(wgs); CK(hipMemcpy(h.data(), out, wgs * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double bytes = (double)iters * DEPTH * waves * (MODE == 3 ? 256 : 1024);
  printf("  %-52s depth %2d: %6.1f B/clk/CU (median workgroup; %5.1f cycles per wave-instruction)\n", name, DEPTH, bytes / (double)h[wgs / 2],
         (double)h[wgs / 2] / (iters * DEPTH * waves));
}

// HBM streaming: every workgroup reads `tiles` consecutive 256-row tiles ONCE (nothing is reused), LDS-DMA, DEPTH
// instructions in flight per wave.  PANEL == 0: the corpus is row-major (a k-tile of a row tile = 256 pieces of
// 128 B at a row stride of ldb bytes, what gemm_phased streams); PANEL == 1: tile-major (the k-tile is one
// contiguous 32 KiB block).
template <int PANEL, int DEPTH>
__global__ __launch_bounds__(512) void hbm_stream(const uint8_t* src, unsigned long long* out, int tiles, int ldb) {
  extern __shared__ u32x4 lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t tile_bytes = (size_t)256 * ldb;
  const int KT = ldb / 128;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    const uint8_t* base = src + ((size_t)blockIdx.x * tiles + t) * tile_bytes;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)tile_bytes, 0x00020000);
    for (int kt = 0; kt < KT; kt += DEPTH / 4) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {            // 4 instructions of a wave per k-tile (8 waves x 4 KiB = 32 KiB)
        const int k = kt + j / 4, piece = wave * 4 + (j & 3);   // piece: 8 rows (row-major) or 1 KiB (tile-major)
        int vo, so;
        if constexpr (PANEL == 0) { vo = (piece * 8 + (lane >> 3)) * ldb + (lane & 7) * 16; so = k * 128; }
        else { vo = piece * 1024 + lane * 16; so = k * 32768; }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (j * 8 + wave) * 64), 16, vo, so, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int PANEL, int DEPTH>
void run_hbm(const char* name, const uint8_t* src, unsigned long long* out, int tiles, int ldb) {
  CK(hipFuncSetAttribute((const void*)hbm_stream<PANEL, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((hbm_stream<PANEL, DEPTH>), dim3(256), dim3(512), 131072, 0, src, out, tiles, ldb);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  const double bytes = 256.0 * tiles * 256 * ldb;
  printf("  %-40s depth %2d: %6.2f TB/s (%.2f ms for %.1f GB)\n", name, DEPTH, bytes / (best * 1e-3) / 1e12, best, bytes / 1e9);
}

int main(int argc, char** argv) {
  if (argc > 1 && atoi(argv[1]) == 0) {   // dma_rate 0: HBM streaming patterns
    const int ldb = 3072, tiles = 48;     // 256 workgroups x 48 tiles x 768 KiB = 9.66 GB
    uint8_t* src; unsigned long long* out;
    CK(hipMalloc(&src, (size_t)256 * tiles * 256 * ldb)); CK(hipMemset(src, 1, (size_t)256 * tiles * 256 * ldb)); CK(hipMalloc(&out, 256 * 8));
    printf("HBM streaming, 256 workgroups, LDS-DMA, every byte read once\n");
    run_hbm<0, 8>("row-major: 8 rows x 128 B per piece", src, out, tiles, ldb);
    run_hbm<0, 16>("row-major: 8 rows x 128 B per piece", src, out, tiles, ldb);
    run_hbm<1, 8>("tile-major: 1 KiB contiguous per piece", src, out, tiles, ldb);
    run_hbm<1, 16>("tile-major: 1 KiB contiguous per piece", src, out, tiles, ldb);
    return 0;
  }
  const int wgs = argc > 1 ? atoi(argv[1]) : 256, waves = argc > 2 ? atoi(argv[2]) : 8, ldb = 3072, windows = argc > 3 ? atoi(argv[3]) : 32;
  uint8_t* src; unsigned long long* out;
  CK(hipMalloc(&src, (size_t)wgs * 256 * ldb)); CK(hipMemset(src, 1, (size_t)wgs * 256 * ldb)); CK(hipMalloc(&out, wgs * 8));
  printf("%d workgroups x %d waves, %d windows of 768 KiB (32: L2-resident; = workgroups: 192 MB, Infinity Cache)\n", wgs, waves, windows);
  run<0, 4>("LDS-DMA dwordx4", src, out, wgs, waves, ldb, windows);
  run<0, 8>("LDS-DMA dwordx4", src, out, wgs, waves, ldb, windows);
  run<0, 16>("LDS-DMA dwordx4", src, out, wgs, waves, ldb, windows);
  run<1, 4>("buffer_load_dwordx4 -> VGPR -> ds_write_b128", src, out, wgs, waves, ldb, windows);
  run<1, 8>("buffer_load_dwordx4 -> VGPR -> ds_write_b128", src, out, wgs, waves, ldb, windows);
  run<2, 8>("buffer_load_dwordx4 -> VGPR", src, out, wgs, waves, ldb, windows);
  run<2, 16>("buffer_load_dwordx4 -> VGPR", src, out, wgs, waves, ldb, windows);
  return 0;
}
