// MFMA issue-rate probe (tools only): cycles per instruction for a few f32 shapes, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
template <int SHAPE>
__global__ void probe(float* out, long long* cyc, int iters) {
  v4f acc[8]; for (int i = 0; i < 8; ++i) acc[i] = (v4f){0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  }
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; long long* cyc; CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&cyc, 8));
  const int iters = 20000;
  for (int shape = 0; shape < 2; ++shape) for (int waves : {1, 2, 4, 8}) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double n_per_simd = (double)iters * 8 * (waves + 3) / 4;  // waves per SIMD (rounded up roughly)
    printf("%s waves/CU=%d: %.3f ms, clock64 delta %lld -> %.2f ns per MFMA per wave, %.2f ns per MFMA per SIMD-slot\n", shape == 0 ? "4x4x1 " : "16x16x4", waves, ms,
           c, ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0) / ((waves + 3) / 4));
  }
  return 0;
}
