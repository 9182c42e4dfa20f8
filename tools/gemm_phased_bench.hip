// Timing harness for the 256 x 256 MFMA kernels (tools only): the round-1 kernel
// (gemm_tiled_kernel<256, true, EB, 256>) against gemm_phased_kernel and its timing-only
// ablations (EXP 1..5, gemm_phased.h), configs[2] shape by default, random operand bits.
//   usage: gemm_phased_bench [n=1000000] [d=1536] [nq=1024] [eb=2] [rounds=3]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define PG_CLOCKS
#include "../svs_amd/csrc/select.h"
#include "../svs_amd/csrc/gemm_phased.h"
using namespace svs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Args {
  uint8_t *M, *Q; float* S; int64_t n; int ldb, nq; uint32_t* st; uint64_t* cand; float *thr, *rs;
};
template <int EB, int EXP, int QT = 256>
void launch_phased(const Args& a, int cus) {
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_phased_kernel<true, EB, EXP, QT>, hipFuncAttributeMaxDynamicSharedMemorySize, PG_LDS_TOTAL)); once = true; }
  const int gx = (int)((a.n + 255) / 256), gy = (a.nq + QT - 1) / QT;
  const unsigned grid = (unsigned)std::min<int64_t>((int64_t)gx * gy, cus);
  if (getenv("PGB_REAL")) CK(hipMemsetAsync(a.st, 0, (size_t)a.nq * SCR_WORDS * 4, 0));   // (the candidate lists start empty, like in a search)
  hipLaunchKernelGGL((gemm_phased_kernel<true, EB, EXP, QT>), dim3(grid), dim3(PG_THREADS), PG_LDS_TOTAL, 0, a.M, a.Q, a.S, a.n, a.ldb, a.n, a.nq, gx, gy,
                     a.st, (int)SCR_WORDS, a.cand, (uint32_t)CAND_CAP, a.thr, 1, a.rs, a.rs);
}
template <int EB>
void launch_tiled(const Args& a, int) {
  static bool once = false;
  const int lds = tg_lds_bytes(256, 256);
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_tiled_kernel<256, true, EB, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); once = true; }
  if (getenv("PGB_REAL")) CK(hipMemsetAsync(a.st, 0, (size_t)a.nq * SCR_WORDS * 4, 0));
  hipLaunchKernelGGL((gemm_tiled_kernel<256, true, EB, 256>), dim3((unsigned)((a.n + 255) / 256), (a.nq + 255) / 256), dim3(512), lds, 0,
                     a.M, a.Q, a.S, a.n, (int64_t)a.ldb, a.n, a.nq, a.st, (int)SCR_WORDS, a.cand, (uint32_t)CAND_CAP, a.thr, 1, a.rs, a.rs);
}

// which SIMD each wave of a 512-thread, 128 KiB-LDS workgroup runs on (HW_REG_HW_ID bits 5:4)
__global__ __launch_bounds__(512) void census_kernel(int* out) {
  extern __shared__ int dummy[];
  const int wave = threadIdx.x >> 6;
  const int simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = simd;
  if (threadIdx.x == 0) dummy[0] = 0;
}

template <int EB>
int run(int64_t n, int d, int nq, int rounds) {
  {
    int* o; CK(hipMalloc(&o, 256 * 8 * 4));
    CK(hipFuncSetAttribute((const void*)census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PG_LDS_BYTES));
    hipLaunchKernelGGL(census_kernel, dim3(256), dim3(512), PG_LDS_BYTES, 0, o);
    std::vector<int> h(256 * 8); CK(hipMemcpy(h.data(), o, h.size() * 4, hipMemcpyDeviceToHost));
    printf("SIMD of waves 0..7, first workgroups:");
    for (int b = 0; b < 6; ++b) { printf("  ["); for (int w = 0; w < 8; ++w) printf("%d", h[b * 8 + w]); printf("]"); }
    int same = 0; for (int b = 0; b < 256; ++b) { bool ok = true; for (int w = 0; w < 4; ++w) ok &= h[b * 8 + w] == h[b * 8 + w + 4]; same += ok; }
    printf("   waves w and w+4 share a SIMD in %d of 256 workgroups\n", same);
    hipFree(o);
  }
  Args a;
  a.n = n; a.ldb = d * EB; a.nq = nq;
  const int nq_pad = (nq + 255) / 256 * 256;
  CK(hipMalloc(&a.M, n * a.ldb)); CK(hipMalloc(&a.Q, (size_t)nq_pad * a.ldb)); a.S = nullptr;
  std::vector<uint8_t> h((size_t)64 << 20);
  // PGB_REAL=1 (f16): rows like the parity corpus -- Gaussian, unit norm in expectation -- and the thresholds the
  // prefix pass would hand the epilogue (k = 100 of n / 64 rows: 2.49 sigma), ~420 survivors per output tile
  const bool real = getenv("PGB_REAL") != nullptr;
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(rand() & (EB == 2 ? ((i & 1) ? 0xa7 : 0xff) : 0xb7));   // exponent bits kept small
  if (real && EB == 1) {   // e4m3 bytes of Gaussian rows scaled so that ~3.6 sigma -> 448 (what quantize_rows_fp8 stores)
    uint64_t x = 88172645463325252ull;
    auto u01 = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (float)((x >> 11) * (1.0 / 9007199254740992.0)) + 1e-12f; };
    auto e4m3 = [](float f) -> uint8_t {   // round to nearest e4m3fn (good enough for a timing corpus)
      const uint8_t sgn = f < 0 ? 0x80 : 0; f = fabsf(f);
      if (f >= 448.f) return sgn | 0x7e;
      if (f < 0.0009765625f) return sgn;
      int e; float mnt = frexpf(f, &e);            // f = mnt * 2^e, mnt in [0.5, 1)
      int E = e - 1 + 7;                           // biased exponent of 1.xxx * 2^(e-1)
      if (E <= 0) { const int q = (int)lrintf(f * 512.f); return sgn | (uint8_t)std::min(q, 8); }   // subnormals: units of 2^-9
      int q = (int)lrintf((mnt * 2.f - 1.f) * 8.f);
      if (q == 8) { q = 0; ++E; }
      return sgn | (uint8_t)((E << 3) | q);
    };
    for (size_t i = 0; i + 1 < h.size(); i += 2) {
      const float r = sqrtf(-2.f * logf(u01())), t = 6.2831853f * u01();
      h[i] = e4m3(124.f * r * cosf(t)); h[i + 1] = e4m3(124.f * r * sinf(t));
    }
  }
  if (real && EB == 2) {
    _Float16* hh = (_Float16*)h.data();
    const float sd = 1.0f / sqrtf((float)d);
    uint64_t x = 88172645463325252ull;
    auto u01 = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (float)((x >> 11) * (1.0 / 9007199254740992.0)) + 1e-12f; };
    for (size_t i = 0; i + 1 < h.size() / 2; i += 2) {
      const float r = sqrtf(-2.f * logf(u01())), t = 6.2831853f * u01();
      hh[i] = (_Float16)(sd * r * cosf(t)); hh[i + 1] = (_Float16)(sd * r * sinf(t));
    }
  }
  for (size_t off = 0; off < (size_t)(n * a.ldb); off += h.size()) CK(hipMemcpy(a.M + off, h.data(), std::min(h.size(), (size_t)(n * a.ldb) - off), hipMemcpyHostToDevice));
  // PGB_ZERO=1 (round 4): an all-zero corpus -- the same instructions and bytes, matrix-pipe operands that toggle nothing: the
  // kernels' speed when the chip's power budget is not what limits them (timing only)
  if (getenv("PGB_ZERO")) { CK(hipMemset(a.M, 0, (size_t)(n * a.ldb))); printf("corpus: all zeros\n"); }
  CK(hipMemcpy(a.Q, h.data() + 12346, (size_t)nq_pad * a.ldb, hipMemcpyHostToDevice));   // (an EVEN offset: the exponent mask sits on the odd bytes; an odd one fills the queries with inf / NaN)
  CK(hipMalloc(&a.st, (size_t)nq * SCR_WORDS * 4)); CK(hipMemset(a.st, 0, (size_t)nq * SCR_WORDS * 4));
  CK(hipMalloc(&a.cand, (size_t)nq * CAND_CAP * 8)); CK(hipMalloc(&a.thr, nq * 4)); CK(hipMalloc(&a.rs, n * 4)); CK(hipMemset(a.rs, 0, n * 4));
  {
    // the threshold the prefix pass would hand the epilogue: the k-th best (k = 100) of max(16384, n / 64) rows,
    // in sigmas of the score distribution (2.49 at a 16,384-row prefix, 3.22 at configs[4]'s 156,250)
    const double prefix = getenv("PGB_PREFIX") ? atof(getenv("PGB_PREFIX")) : std::max(16384.0, (double)n / 64.0), p = 100.0 / prefix;   // (PGB_PREFIX: thresholds as tight as a prefix of that many rows would give)
    const double t = sqrt(-2.0 * log(p)), z = t - (2.515517 + 0.802853 * t + 0.010328 * t * t) / (1.0 + 1.432788 * t + 0.189269 * t * t + 0.001308 * t * t * t);
    printf("thresholds: %.2f sigma (prefix %.0f rows)\n", z, prefix);
    std::vector<float> th(nq, real ? (float)z / sqrtf((float)d) : 1e30f); CK(hipMemcpy(a.thr, th.data(), nq * 4, hipMemcpyHostToDevice));
  }
  if (real && EB == 1) {   // row / query scales: raw e4m3 dot products come out ~ N(0, d * 124^4)
    std::vector<float> sc((size_t)n, 1.0f / (124.f * sqrtf((float)d)));
    CK(hipMemcpy(a.rs, sc.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = getenv("PGB_CUS") ? atoi(getenv("PGB_CUS")) : prop.multiProcessorCount;   // (PGB_CUS: fewer workgroups, the same work each: per-CU rates)
  // every phased launch stamps its clocks: the buffer must exist before the first one
  unsigned long long* cb; CK(hipMalloc(&cb, (size_t)std::max(cus, 256) * 8 * 8)); CK(hipMemset(cb, 0, (size_t)std::max(cus, 256) * 8 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(pg_clock_buf), &cb, sizeof(cb)));
  CK(hipDeviceSynchronize());
  typedef void (*Fn)(const Args&, int);
  struct V { const char* name; Fn fn; };
#ifdef PGB_Q128   // panels of up to 128 queries: the 128-query tiles (run with nq <= 128)
  const V vs[] = {{"round-1 tiled 256x256 (reference only)", launch_tiled<EB>}, {"phased QT=128, corpus nontemporal (what ships)", launch_phased<EB, 20, 128>},
                  {"phased QT=128, default policy", launch_phased<EB, 0, 128>}, {"phased QT=128, no epilogue", launch_phased<EB, 14, 128>},
                  {"phased QT=128, LDS-DMA + barriers only", launch_phased<EB, 5, 128>}, {"phased QT=128, tile-major, LDS-DMA + barriers only", launch_phased<EB, 45, 128>},
                  {"phased QT=128, 8-row-interleaved image (timing only)", launch_phased<EB, 41, 128>}, {"phased QT=128, 8-row-interleaved, LDS-DMA + barriers only", launch_phased<EB, 46, 128>}};
#elif defined(PGB_ORDER)   // round 4: the order of a quadrant's MFMAs (which operand changes from one MFMA to the next): an energy question
  const V vs[] = {{"round-1 tiled 256x256 (reference only)", launch_tiled<EB>}, {"phased (corpus fragment outer: what ships for f16)", launch_phased<EB, 0>},
                  {"phased, snake: one operand changes per MFMA", launch_phased<EB, 60>}, {"phased, query fragment outer, snake", launch_phased<EB, 61>},
                  {"phased nt (what ships for fp8)", launch_phased<EB, 20>}, {"phased nt, snake", launch_phased<EB, 62>}, {"phased nt, query fragment outer, snake", launch_phased<EB, 63>}};
#elif defined(PGB_FULL)   // every ablation (slow to compile: fourteen instantiations per operand type)
  const V vs[] = {{"round-1 tiled 256x256", launch_tiled<EB>}, {"phased", launch_phased<EB, 0>}, {"phased, no LDS-DMA in loop", launch_phased<EB, 1>},
                  {"phased, no fragment reads", launch_phased<EB, 2>}, {"phased, no MFMA", launch_phased<EB, 3>},
                  {"phased, no stagger", launch_phased<EB, 7>}, {"phased, LDS-DMA + barriers only", launch_phased<EB, 5>}, {"phased, DMA pieces between the MFMAs", launch_phased<EB, 30>}, {"phased, always k-tile 0 (L2 hits)", launch_phased<EB, 21>}, {"phased, DMA + barriers only, always k-tile 0", launch_phased<EB, 25>}, {"phased, barriers only", launch_phased<EB, 26>}, {"phased, epilogue with a branch per register", launch_phased<EB, 31>}, {"phased, corpus pieces nontemporal", launch_phased<EB, 20>}, {"phased, no epilogue", launch_phased<EB, 14>}};
#else             // round 3's working set
  const V vs[] = {{"round-1 tiled 256x256", launch_tiled<EB>}, {"phased", launch_phased<EB, 0>}, {"phased, corpus pieces nontemporal", launch_phased<EB, 20>},
                  {"phased, two-sweep epilogue on every tile", launch_phased<EB, 32>}, {"phased, no epilogue", launch_phased<EB, 14>},
                  {"phased, LDS-DMA + barriers only", launch_phased<EB, 5>},
                  {"phased, nt, corpus addressed tile-major (timing only)", launch_phased<EB, 40>},
                  {"phased, tile-major, LDS-DMA + barriers only", launch_phased<EB, 45>},
                  {"phased, XCD map: 32 row tiles x 1 query tile", launch_phased<EB, 50>},
                  {"phased, XCD map: 16 row tiles x 2 query tiles", launch_phased<EB, 51>},
                  {"phased, XCD map 32 x 1, corpus nontemporal", launch_phased<EB, 52>},
                  {"phased, phase 0 reads 8 fragments instead of 12 (timing only)", launch_phased<EB, 53>},
                  {"phased, nt, 8-row-interleaved image (timing only)", launch_phased<EB, 41>}, {"phased, 8-row-interleaved, LDS-DMA + barriers only", launch_phased<EB, 46>}};
#endif
  const int NVALL = sizeof(vs) / sizeof(vs[0]);
  const char* only = getenv("PGB_ONLY");          // e.g. PGB_ONLY=2 runs variant 2 alone (fault hunting)
  V sel[20]; int NV = 0;
  for (int v = 0; v < NVALL; ++v) if (!only || atoi(only) == v) sel[NV++] = vs[v];
#define vs sel
  std::vector<std::vector<float>> ms(NV);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  setvbuf(stdout, nullptr, _IONBF, 0);
  for (int v = 0; v < NV; ++v) {   // warm-up, one variant at a time (a faulting variant is the last one named)
    printf("warm-up: %s\n", vs[v].name);
    vs[v].fn(a, cus); vs[v].fn(a, cus);
    CK(hipDeviceSynchronize());
  }
  for (int r = 0; r < rounds; ++r)
    for (int v = 0; v < NV; ++v) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 4; ++i) vs[v].fn(a, cus);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[v].push_back(t / 4);
    }
  // in-kernel clock of the phased variants: shader cycles / 100 MHz ticks, median over workgroups
  std::vector<double> ghz(NV, 0.0), kcyc(NV, 0.0), lcyc(NV, 0.0), ecyc(NV, 0.0);
  for (int v = only ? 0 : 1; v < NV; ++v) {
    if (only && atoi(only) == 0) break;
    for (int i = 0; i < 3; ++i) vs[v].fn(a, cus);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hc(256 * 8); CK(hipMemcpy(hc.data(), cb, hc.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> g, c, l, ep;
    for (int b = 0; b < cus && b < 256; ++b) {
      const double cyc = (double)(hc[b * 8 + 1] - hc[b * 8]), tick = (double)(hc[b * 8 + 3] - hc[b * 8 + 2]);
      if (tick > 0) { g.push_back(cyc / tick * 0.1); c.push_back(cyc); l.push_back((double)hc[b * 8 + 4]); ep.push_back((double)hc[b * 8 + 5]); }
    }
    std::sort(g.begin(), g.end()); std::sort(c.begin(), c.end()); std::sort(l.begin(), l.end()); std::sort(ep.begin(), ep.end());
    if (!g.empty()) { ghz[v] = g[g.size() / 2]; kcyc[v] = c[c.size() / 2]; lcyc[v] = l[l.size() / 2]; ecyc[v] = ep[ep.size() / 2]; }
  }
  const double flop = 2.0 * n * d * nq;
  printf("EB=%d  %lld x %d, %d queries (%d CUs), per launch: median [min] of %d rounds of 4\n", EB, (long long)n, d, nq, cus, rounds);
  for (int v = 0; v < NV; ++v) {
    std::sort(ms[v].begin(), ms[v].end());
    const float med = ms[v][ms[v].size() / 2], mn = ms[v][0];
    const double ktiles = (double)((n + 255) / 256) * ((nq + 255) / 256) / cus * (d * EB / 128);
    printf("  %-44s %7.3f ms [%7.3f] %7.1f TFLOP/s  %.2f GHz, %5.0f cycles per k-tile (%5.0f inside the k loop), epilogue %6.0f cycles per tile\n", vs[v].name, med, mn,
           flop / (med * 1e-3) / 1e12, ghz[v], kcyc[v] / ktiles, lcyc[v] / ktiles, ecyc[v] / (ktiles / (d * EB / 128)));
  }
  fflush(stdout);
  hipFree(a.M); hipFree(a.Q); hipFree(a.st); hipFree(a.cand); hipFree(a.thr); hipFree(a.rs);
  return 0;
}
int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
  const int d = argc > 2 ? atoi(argv[2]) : 1536, nq = argc > 3 ? atoi(argv[3]) : 1024, eb = argc > 4 ? atoi(argv[4]) : 2;
  const int rounds = argc > 5 ? atoi(argv[5]) : 3;
#if defined(PGB_EB) && PGB_EB == 1
  return run<1>(n, d, nq, rounds);
#elif defined(PGB_EB) && PGB_EB == 2
  return run<2>(n, d, nq, rounds);
#else
  return eb == 2 ? run<2>(n, d, nq, rounds) : run<1>(n, d, nq, rounds);
#endif
}
