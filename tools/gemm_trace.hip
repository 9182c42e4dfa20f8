// Phase timeline of the tiled GEMM's k-loop (tools only).  Build with -DTG_TRACE.
// stamps per (wave, k-step): 0 loop top, 1 after the DMA wait, 2 after the barrier, 3 after the MFMAs were issued
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdlib>
#include "../svs_amd/csrc/select.h"
#include "../svs_amd/csrc/gemm_tiled.h"
using namespace svs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
template <int BN, int EB, int BM, bool FUSE = false>
int run(const char* name, int64_t n, int d, int nq, bool random_data = false) {
  const int64_t ldb = (int64_t)d * EB;
  uint8_t *M, *Q; float* S; long long* tr;
  CK(hipMalloc(&M, n * ldb)); CK(hipMalloc(&Q, (size_t)nq * ldb)); CK(hipMalloc(&S, (size_t)nq * n * 4)); CK(hipMalloc(&tr, (8 * 64 * 4 + 64) * 8));
  CK(hipMemset(M, 0x3c, n * ldb)); CK(hipMemset(Q, 0x3c, nq * ldb));
  if (random_data) {   // random bytes with the exponent bits kept small: realistic operand toggling
    std::vector<uint8_t> h((size_t)64 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(rand() & (EB == 2 ? ((i & 1) ? 0xa7 : 0xff) : 0xb7));
    for (size_t off = 0; off < (size_t)(n * ldb); off += h.size()) CK(hipMemcpy(M + off, h.data(), std::min(h.size(), (size_t)(n * ldb) - off), hipMemcpyHostToDevice));
    CK(hipMemcpy(Q, h.data(), (size_t)nq * ldb, hipMemcpyHostToDevice));
  } CK(hipMemset(tr, 0, (8 * 64 * 4 + 64) * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(tg_trace_buf), &tr, sizeof(tr)));
  const int lds = tg_lds_bytes(BM, BN);
  CK(hipFuncSetAttribute((const void*)gemm_tiled_kernel<BN, FUSE, EB, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  float* rs; CK(hipMalloc(&rs, n * 4)); CK(hipMemset(rs, 0, n * 4));
  uint32_t* st; uint64_t* cand; float* thr;
  CK(hipMalloc(&st, (size_t)nq * SCR_WORDS * 4)); CK(hipMemset(st, 0, (size_t)nq * SCR_WORDS * 4));
  CK(hipMalloc(&cand, (size_t)nq * CAND_CAP * 8)); CK(hipMalloc(&thr, nq * 4));
  { std::vector<float> t(nq, 1e30f); CK(hipMemcpy(thr, t.data(), nq * 4, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    if (rep == 2) CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gemm_tiled_kernel<BN, FUSE, EB, BM>), dim3((unsigned)((n + BM - 1) / BM), (nq + BN - 1) / BN), dim3(512), lds, 0,
                       M, Q, S, n, ldb, n, nq, st, (int)SCR_WORDS, cand, (uint32_t)CAND_CAP, thr, 1, rs, rs);
    if (rep == 2) CK(hipEventRecord(e1));
  }
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(8 * 64 * 4 + 64); CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
  const int ksteps = (int)(ldb / 128);
  printf("%s: ksteps %d\n", name, ksteps);
  for (int w : {0, 7}) {
    printf(" wave %d: per k-step cycles [wait DMA | barrier | issue DMA+reads+MFMA] (loop period)\n", w);
    for (int s = 1; s < ksteps && s < 14; ++s) {
      const long long* a = &h[(w * 64 + s) * 4]; const long long* p = &h[(w * 64 + s - 1) * 4];
      printf("   s=%2d  %6lld | %6lld | %6lld   (%lld)\n", s, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[0] - p[0]);
    }
  }
  printf(" whole k-loop, wave 0: %lld cycles\n", h[(0 * 64 + ksteps - 1) * 4 + 3] - h[0]);
  const long long* b = &h[8 * 64 * 4];
  printf(" block phases, wave 0: prologue %lld | k-loop %lld | epilogue %lld cycles;  kernel %.3f ms = %.0f cycles per block-round at 2.4 GHz (%lld blocks, %.2f rounds)\n",
         b[1] - b[0], b[2] - b[1], b[3] - b[2], ms, ms * 2.4e6 / (((n + BM - 1) / BM) * ((nq + BN - 1) / BN) / 256.0),
         (long long)(((n + BM - 1) / BM) * ((nq + BN - 1) / BN)), ((n + BM - 1) / BM) * ((nq + BN - 1) / BN) / 256.0);
  hipFree(M); hipFree(Q); hipFree(S); hipFree(tr); hipFree(rs);
  return 0;
}
int main() {
  if (run<256, 2, 256, true>("f16 256x256 fused, random operands", 1000000, 1536, 1024, true)) return 1;
  if (run<256, 2, 256, true>("f16 256x256 fused, constant operands", 1000000, 1536, 1024, false)) return 1;
  if (run<256, 1, 256, true>("fp8 256x256 fused, random operands", 1000000, 1536, 1024, true)) return 1;
  if (run<256, 1, 256, true>("fp8 256x256 fused, constant operands", 1000000, 1536, 1024, false)) return 1;
  return 0;
}
