// Sustained MFMA throughput probe (tools only): every SIMD issues back-to-back f16 / fp8 MFMAs on
// register operands (no memory traffic) for ~20 ms, with constant or random operand bits.
// Shows what the chip sustains under its power limit, i.e. the practical ceiling for the GEMM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
template <int KIND>
__global__ __launch_bounds__(512) void burn(const uint32_t* __restrict__ seed, float* out, int iters) {
  const uint32_t* s = seed + (blockIdx.x * blockDim.x + threadIdx.x) * 16;
  uint32_t w[16]; for (int i = 0; i < 16; ++i) w[i] = s[i];
  v4f acc[8]; for (int i = 0; i < 8; ++i) acc[i] = (v4f){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) {
        h8 a = __builtin_bit_cast(h8, *(const __attribute__((ext_vector_type(4))) uint32_t*)&w[(i & 1) * 4]);
        h8 b = __builtin_bit_cast(h8, *(const __attribute__((ext_vector_type(4))) uint32_t*)&w[8 + (i & 1) * 4]);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
      } else {
        i32x8 a = {(int)w[0], (int)w[1], (int)w[2], (int)w[3], (int)w[4], (int)w[5], (int)w[6], (int)w[7]};
        i32x8 b = {(int)w[8], (int)w[9], (int)w[10], (int)w[11], (int)w[12], (int)w[13], (int)w[14], (int)w[15]};
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0, 0, 0);
      }
    }
  }
  float t = 0; for (int i = 0; i < 8; ++i) t += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
int main() {
  const int blocks = 256, threads = 512;   // 2 waves per SIMD
  uint32_t* seed; float* out; CK(hipMalloc(&seed, blocks * threads * 64)); CK(hipMalloc(&out, blocks * threads * 4));
  std::vector<uint32_t> h(blocks * threads * 16);
  for (int kind = 0; kind < 2; ++kind) for (int rnd = 0; rnd < 2; ++rnd) {
    for (auto& x : h) x = rnd ? ((uint32_t)rand() & (kind == 0 ? 0x37ff37ffu : 0x37373737u)) : (kind == 0 ? 0x3c003c00u : 0x38383838u);
    CK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int iters = kind == 0 ? 400000 : 200000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (kind == 0) hipLaunchKernelGGL(burn<0>, dim3(blocks), dim3(threads), 0, 0, seed, out, iters);
      else hipLaunchKernelGGL(burn<1>, dim3(blocks), dim3(threads), 0, 0, seed, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double macs = (double)blocks * (threads / 64) * iters * 8 * (kind == 0 ? 8192.0 : 32768.0);
    const double cyc = (double)iters * 8 * 2 * (kind == 0 ? 16 : 32);   // per SIMD: 2 waves
    printf("%s, %s operands: %.1f ms, %.0f TFLOP/s, implied clock %.2f GHz if the pipe never idles\n", kind == 0 ? "f16 16x16x32" : "fp8 16x16x128 f8f6f4",
           rnd ? "random" : "constant", ms, 2 * macs / (ms * 1e-3) / 1e12, cyc / (ms * 1e-3) / 1e9);
  }
  return 0;
}
