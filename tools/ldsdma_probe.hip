// Probe (tools only): does `buffer_load_dword / dwordx4 ... lds` reach LDS offsets beyond 64 / 128 KiB?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int SIZE>
__global__ __launch_bounds__(64) void probe(const uint32_t* src, uint32_t* out, int lds_off) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 40960; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4096, 0x00020000);
  if constexpr (SIZE == 4)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((char*)lds + lds_off), 4, (int)threadIdx.x * 4, 0, 0, 0);
  else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((char*)lds + lds_off), 16, (int)threadIdx.x * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // where did the first element land?  scan the whole LDS for the marker values
  for (int i = threadIdx.x; i < 40960; i += 64)
    if (lds[i] == 0x1000u) out[0] = (uint32_t)i * 4;
  if (threadIdx.x == 0) out[1] = lds[lds_off / 4];
}
int main() {
  uint32_t *src, *out;
  CK(hipMalloc(&src, 4096)); CK(hipMalloc(&out, 64));
  std::vector<uint32_t> h(1024); for (int i = 0; i < 1024; ++i) h[i] = 0x1000u + i;
  CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  CK(hipFuncSetAttribute((const void*)probe<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  for (int off : {0, 4096, 65536, 65536 + 1024, 131072 - 1024, 131072, 131072 + 2048, 140000 / 16 * 16, 163840 - 1024}) {
    for (int size : {4, 16}) {
      uint32_t z[2] = {0xffffffffu, 0}; CK(hipMemcpy(out, z, 8, hipMemcpyHostToDevice));
      if (size == 4) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 163840, 0, src, out, off);
      else hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), 163840, 0, src, out, off);
      CK(hipDeviceSynchronize());
      uint32_t r[2]; CK(hipMemcpy(r, out, 8, hipMemcpyDeviceToHost));
      printf("size %2d  wanted LDS offset %6d: first element found at %6d (value at wanted offset 0x%x)%s\n", size, off, (int)r[0], r[1], (int)r[0] == off ? "" : "   <-- NOT where asked");
    }
  }
  return 0;
}
