// Phase timeline of gemm_phased_kernel (tools only; builds the kernel with -DPG_TRACE stamps).
// Prints, for the two waves that share SIMD 0's... rather: for waves w and w + 4 (one SIMD), 16
// consecutive phases of one workgroup's third output tile: cycles from each phase's start (release
// of the previous barrier) to: loads issued + waited | barrier 1 passed + fragments in | MFMAs
// issued | barrier 2 passed.
//   usage: gemm_phased_trace [n=1000000] [d=1536] [nq=1024] [eb=2]
#define PG_TRACE
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../svs_amd/csrc/select.h"
#include "../svs_amd/csrc/gemm_phased.h"
using namespace svs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int EB, int EXP>
int run(int64_t n, int d, int nq) {
  const int ldb = d * EB, nq_pad = (nq + 255) / 256 * 256;
  uint8_t *M, *Q; uint32_t* st; uint64_t* cand; float *thr, *rs; unsigned long long* tb;
  CK(hipMalloc(&M, n * ldb)); CK(hipMalloc(&Q, (size_t)nq_pad * ldb));
  std::vector<uint8_t> h((size_t)64 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(rand() & (EB == 2 ? ((i & 1) ? 0xa7 : 0xff) : 0xb7));
  for (size_t off = 0; off < (size_t)(n * ldb); off += h.size()) CK(hipMemcpy(M + off, h.data(), std::min(h.size(), (size_t)(n * ldb) - off), hipMemcpyHostToDevice));
  CK(hipMemcpy(Q, h.data() + 12346, (size_t)nq_pad * ldb, hipMemcpyHostToDevice));
  CK(hipMalloc(&st, (size_t)nq * SCR_WORDS * 4)); CK(hipMemset(st, 0, (size_t)nq * SCR_WORDS * 4));
  CK(hipMalloc(&cand, (size_t)nq * CAND_CAP * 8)); CK(hipMalloc(&thr, nq * 4)); CK(hipMalloc(&rs, n * 4)); CK(hipMemset(rs, 0, n * 4));
  { std::vector<float> t(nq, 1e30f); CK(hipMemcpy(thr, t.data(), nq * 4, hipMemcpyHostToDevice)); }
  const int NST = 8 * PG_TRACE_KTS * 4 * 4;
  CK(hipMalloc(&tb, NST * 8)); CK(hipMemset(tb, 0, NST * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(pg_trace_buf), &tb, sizeof(tb)));
  CK(hipDeviceSynchronize());
  const int lds = PG_LDS_TOTAL + NST * 8;
  CK(hipFuncSetAttribute((const void*)gemm_phased_kernel<true, EB, EXP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int gx = (int)((n + 255) / 256), gy = (nq + 255) / 256;
  const unsigned grid = (unsigned)std::min<int64_t>((int64_t)gx * gy, prop.multiProcessorCount);
  for (int rep = 0; rep < 3; ++rep)
    hipLaunchKernelGGL((gemm_phased_kernel<true, EB, EXP>), dim3(grid), dim3(PG_THREADS), lds, 0, M, Q, (float*)nullptr, n, ldb, n, nq, gx, gy,
                       st, (int)SCR_WORDS, cand, (uint32_t)CAND_CAP, thr, 1, rs, rs);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> t(NST); CK(hipMemcpy(t.data(), tb, NST * 8, hipMemcpyDeviceToHost));
  auto at = [&](int w, int ph, int slot) { return (long long)t[(w * PG_TRACE_KTS * 4 + ph) * 4 + slot]; };
  const long long base = at(0, 0, 0);
  printf("EB=%d EXP=%d: %d phases of workgroup %d, tile %d.  Per phase: absolute cycle of [ready for barrier 1 | fragments in | MFMAs issued | past barrier 2]\n",
         EB, EXP, PG_TRACE_KTS * 4, PG_TRACE_BLOCK, PG_TRACE_TILE);
  for (int w : {0, 4, 1, 5}) {
    printf(" wave %d:\n", w);
    for (int ph = 0; ph < PG_TRACE_KTS * 4; ++ph) {
      const long long s0 = at(w, ph, 0) - base, s1 = at(w, ph, 1) - base, s2 = at(w, ph, 2) - base, s3 = at(w, ph, 3) - base;
      const long long prev = ph ? at(w, ph - 1, 3) - base : s0;
      printf("   ph %2d (%d): loads+waits %5lld | barrier1+land %5lld | mfma %5lld | barrier2 %5lld    [abs %6lld %6lld %6lld %6lld]\n", ph, ph & 3,
             s0 - prev, s1 - s0, s2 - s1, s3 - s2, s0, s1, s2, s3);
    }
  }
  // barrier view: global barrier b is (phase, barrier 1) = 2 ph, (phase, barrier 2) = 2 ph + 1 for waves 0-3 and one
  // later for waves 4-7.  Arrival of every wave (cycles before the LAST arrival), and release - last arrival.
  printf(" barriers: arrival of waves 0..7 relative to the last one to arrive | release after the last arrival (min over waves)\n");
  for (int b = 1; b < 2 * PG_TRACE_KTS * 4 - 1; ++b) {
    long long arr[8], rel[8];
    for (int w = 0; w < 8; ++w) {
      const int bb = w < 4 ? b : b - 1, ph = bb >> 1, second = bb & 1;
      arr[w] = at(w, ph, second ? 2 : 0) - base;
      rel[w] = at(w, ph, second ? 3 : 1) - base;   // (barrier 1: includes the wait for the fragments)
    }
    long long last = arr[0], first_rel = rel[0];
    for (int w = 1; w < 8; ++w) { last = std::max(last, arr[w]); first_rel = std::min(first_rel, rel[w]); }
    printf("   b %2d (A: ph %d b%d): ", b, (b >> 1) & 3, (b & 1) + 1);
    for (int w = 0; w < 8; ++w) printf("%5lld", arr[w] - last);
    printf(" | %4lld   [last arrival at %6lld]\n", first_rel - last, last);
  }
  const long long span = at(0, PG_TRACE_KTS * 4 - 1, 3) - at(0, 0, 3);
  printf(" wave 0: %lld cycles for %d phases = %lld per k-tile\n", span, PG_TRACE_KTS * 4 - 1, span * 4 / (PG_TRACE_KTS * 4 - 1));
  return 0;
}
int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
  const int d = argc > 2 ? atoi(argv[2]) : 1536, nq = argc > 3 ? atoi(argv[3]) : 1024, eb = argc > 4 ? atoi(argv[4]) : 2;
  if (eb == 2) { run<2, 0>(n, d, nq); run<2, 14>(n, d, nq); }
  else run<1, 0>(n, d, nq);
  return 0;
}
