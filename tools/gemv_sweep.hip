// Standalone A/B harness for the f32 score kernel (tools only; not shipped).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemv_sweep.hip -o tools/gemv_sweep
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../svs_amd/csrc/gemv_f32.h"
using namespace svs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void fill_kernel(float* p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = ((int)(x & 0xffff) - 32768) * (1.0f / 32768.f) * 0.0255f;
  }
}

struct Cfg { const char* name; int R, WPB; bool NT, CONTIG; void (*fn)(const v4f*, const v4f*, float*, int64_t); };
#define O(R, WPB, NT, QL) Cfg{"one" #R "r" #WPB "w" #NT "n" #QL "q", R, WPB, NT, false, gemv_f32_oneshot_kernel<6, R, WPB, NT, QL>}
#define K(R, WPB, NT, CG) Cfg{#R "r" #WPB "w" #NT "n" #CG "c", R, WPB, NT, CG, gemv_f32_rows_kernel<6, R, WPB, NT, CG>}

int main(int argc, char** argv) {
  int64_t n = argc > 1 ? atoll(argv[1]) : 1000000; const int d = 1536; int iters = argc > 2 ? atoi(argv[2]) : 20;
  float *M, *q, *s; CK(hipMalloc(&M, (size_t)n * d * 4)); CK(hipMalloc(&q, d * 4)); CK(hipMalloc(&s, n * 4));
  fill_kernel<<<4096, 256>>>(M, (size_t)n * d, 1u); fill_kernel<<<8, 256>>>(q, d, 7u); CK(hipDeviceSynchronize());
  std::vector<Cfg> cfgs = {
    O(1,16,true,false), Cfg{"one1r16w-xcdmap", 1, 16, true, false, gemv_f32_oneshot_kernel<6, 1, 16, true, false, true>},
    O(2,16,true,false), Cfg{"one2r16w-xcdmap", 2, 16, true, false, gemv_f32_oneshot_kernel<6, 2, 16, true, false, true>},
    O(1,8,true,false), Cfg{"one1r8w-xcdmap", 1, 8, true, false, gemv_f32_oneshot_kernel<6, 1, 8, true, false, true>},
    O(4,16,true,false), O(1,16,true,false) };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int bpcs[] = {9999};
  for (auto& c : cfgs) {
    for (int bpc : bpcs) {
      int64_t tiles = (n + c.R - 1) / c.R; int64_t maxb = (tiles + c.WPB - 1) / c.WPB;
      int blocks = (int)std::min<int64_t>(maxb, (int64_t)256 * bpc);
      if (bpc != 9999 && blocks * c.WPB > 256 * 32 * 2) continue;
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(c.WPB * 64), 0, 0, (const v4f*)M, (const v4f*)q, s, n);
      std::vector<float> ts;
      for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(c.WPB * 64), 0, 0, (const v4f*)M, (const v4f*)q, s, n); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end());
      double med = ts[ts.size() / 2], mn = ts[0];
      printf("%-14s blocks/CU %4d (%6d blocks): median %7.1f us %5.2f TB/s   min %7.1f us %5.2f TB/s\n", c.name, bpc, blocks, med * 1e3,
             (double)n * d * 4 / (med * 1e-3) / 1e12, mn * 1e3, (double)n * d * 4 / (mn * 1e-3) / 1e12);
      fflush(stdout);
    }
  }
  return 0;
}
