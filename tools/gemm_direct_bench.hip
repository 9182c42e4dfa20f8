// Harness for gemm_direct_kernel (tools only): parity against gemm_tiled_kernel<128, ., EB, 256> on a small corpus
// (materialised scores bit for bit; fused candidate sets), then timing on the configs-sized corpus against the shipped
// 128-query phased kernel.
//   usage: gemm_direct_bench [n=1000000] [d=1536] [nq=128] [rounds=3]      (-DGDB_EB=1 for fp8)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <vector>
#define GD_CLOCKS
#include "../svs_amd/csrc/select.h"
#include "../svs_amd/csrc/gemm_phased.h"
#include "gemm_direct.h"
using namespace svs;
#ifndef GDB_EB
#define GDB_EB 2
#endif
constexpr int EB = GDB_EB;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Args { uint8_t *M, *Q; float* S; int64_t n; int ldb, nq; uint32_t* st; uint64_t* cand; float *thr, *rs, *qsc; };

template <bool FUSE, int EXP = 0, int PF = 2> void launch_direct(const Args& a, int cus, int64_t sstride) {
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_direct_kernel<FUSE, EB, EXP, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, GD_LDS_TOTAL)); once = true; }
  const int tiles = (int)((a.n + GD_ROWS - 1) / GD_ROWS);
  hipLaunchKernelGGL((gemm_direct_kernel<FUSE, EB, EXP, PF>), dim3(std::min(tiles, cus)), dim3(GD_WAVES * 64), GD_LDS_TOTAL, 0, a.M, a.Q, a.S, a.n, a.ldb, sstride, a.nq, 128, tiles,
                     a.st, (int)SCR_WORDS, a.cand, (uint32_t)CAND_CAP, a.thr, 1, a.rs, a.qsc);
}
template <bool FUSE> void launch_tiled(const Args& a, int, int64_t sstride) {
  static bool once = false;
  const int lds = tg_lds_bytes(256, 128);
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_tiled_kernel<128, FUSE, EB, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); once = true; }
  hipLaunchKernelGGL((gemm_tiled_kernel<128, FUSE, EB, 256>), dim3((unsigned)((a.n + 255) / 256), 1), dim3(512), lds, 0,
                     a.M, a.Q, a.S, a.n, (int64_t)a.ldb, sstride, a.nq, a.st, (int)SCR_WORDS, a.cand, (uint32_t)CAND_CAP, a.thr, 1, a.rs, a.qsc);
}
void launch_phased128(const Args& a, int cus, int64_t) {
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_phased_kernel<true, EB, 20, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, PG_LDS_TOTAL)); once = true; }
  const int gx = (int)((a.n + 255) / 256);
  hipLaunchKernelGGL((gemm_phased_kernel<true, EB, 20, 128>), dim3(std::min(gx, cus)), dim3(PG_THREADS), PG_LDS_TOTAL, 0, a.M, a.Q, a.S, a.n, a.ldb, a.n, a.nq, gx, 1,
                     a.st, (int)SCR_WORDS, a.cand, (uint32_t)CAND_CAP, a.thr, 1, a.rs, a.qsc);
}

static void fill(std::vector<uint8_t>& h, int d) {
  uint64_t x = 88172645463325252ull;
  auto u01 = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (float)((x >> 11) * (1.0 / 9007199254740992.0)) + 1e-12f; };
  if (EB == 2) {
    _Float16* hh = (_Float16*)h.data();
    const float sd = 1.0f / sqrtf((float)d);
    for (size_t i = 0; i + 1 < h.size() / 2; i += 2) {
      const float r = sqrtf(-2.f * logf(u01())), t = 6.2831853f * u01();
      hh[i] = (_Float16)(sd * r * cosf(t)); hh[i + 1] = (_Float16)(sd * r * sinf(t));
    }
  } else {
    auto e4m3 = [](float f) -> uint8_t {
      const uint8_t sgn = f < 0 ? 0x80 : 0; f = fabsf(f);
      if (f >= 448.f) return sgn | 0x7e;
      if (f < 0.0009765625f) return sgn;
      int e; float mnt = frexpf(f, &e);
      int E = e - 1 + 7;
      if (E <= 0) { const int q = (int)lrintf(f * 512.f); return sgn | (uint8_t)std::min(q, 8); }
      int q = (int)lrintf((mnt * 2.f - 1.f) * 8.f);
      if (q == 8) { q = 0; ++E; }
      return sgn | (uint8_t)((E << 3) | q);
    };
    for (size_t i = 0; i + 1 < h.size(); i += 2) {
      const float r = sqrtf(-2.f * logf(u01())), t = 6.2831853f * u01();
      h[i] = e4m3(124.f * r * cosf(t)); h[i + 1] = e4m3(124.f * r * sinf(t));
    }
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
  const int d = argc > 2 ? atoi(argv[2]) : 1536, nq = argc > 3 ? atoi(argv[3]) : 128, rounds = argc > 4 ? atoi(argv[4]) : 3;
  Args a; a.n = n; a.ldb = d * EB; a.nq = nq;
  if (!gd_shape_ok(a.ldb, nq)) { printf("shape not supported\n"); return 1; }
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = getenv("GDB_CUS") ? atoi(getenv("GDB_CUS")) : prop.multiProcessorCount;
  unsigned long long* cb; CK(hipMalloc(&cb, (size_t)std::max(cus, 256) * 4 * 8)); CK(hipMemset(cb, 0, (size_t)std::max(cus, 256) * 4 * 8));   // (every direct launch stamps its clocks)
  CK(hipMemcpyToSymbol(HIP_SYMBOL(gd_clock_buf), &cb, sizeof(cb)));
  CK(hipMalloc(&a.M, n * a.ldb)); CK(hipMalloc(&a.Q, (size_t)128 * a.ldb));
  std::vector<uint8_t> h((size_t)64 << 20);
  fill(h, d);
  for (size_t off = 0; off < (size_t)(n * a.ldb); off += h.size()) CK(hipMemcpy(a.M + off, h.data(), std::min(h.size(), (size_t)(n * a.ldb) - off), hipMemcpyHostToDevice));
  // GDB_ZERO=1: an all-zero corpus (timing only; the parity section still runs on it): the same instructions and the same bytes
  // moved, but matrix-pipe operands that toggle nothing -- what the kernels do when the chip's power budget is not the limit
  if (getenv("GDB_ZERO")) { CK(hipMemset(a.M, 0, (size_t)(n * a.ldb))); printf("corpus: all zeros\n"); }
  CK(hipMemset(a.Q, 0, (size_t)128 * a.ldb));
  CK(hipMemcpy(a.Q, h.data() + 1234560, (size_t)nq * a.ldb, hipMemcpyHostToDevice));
  CK(hipMalloc(&a.st, (size_t)128 * SCR_WORDS * 4)); CK(hipMalloc(&a.cand, (size_t)128 * CAND_CAP * 8));
  CK(hipMalloc(&a.thr, 128 * 4)); CK(hipMalloc(&a.rs, n * 4)); CK(hipMalloc(&a.qsc, 128 * 4));
  {
    const double prefix = std::max(16384.0, (double)n / 64.0), p = 100.0 / prefix;
    const double t = sqrt(-2.0 * log(p)), z = t - (2.515517 + 0.802853 * t + 0.010328 * t * t) / (1.0 + 1.432788 * t + 0.189269 * t * t + 0.001308 * t * t * t);
    const float sc = EB == 1 ? 1.0f / (124.f * sqrtf((float)d)) : 1.0f;
    std::vector<float> th(128, (float)z / sqrtf((float)d)), rsv((size_t)n, sc), qsv(128, EB == 1 ? sc * 124.f * sqrtf((float)d) * sc : 1.0f);
    if (EB == 1) for (auto& v : qsv) v = sc;   // scores = raw dot * sc * sc ~ N(0, 1/d)... (raw ~ N(0, d 124^4))
    CK(hipMemcpy(a.thr, th.data(), 128 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(a.rs, rsv.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(a.qsc, qsv.data(), 128 * 4, hipMemcpyHostToDevice));
    printf("thresholds %.2f sigma\n", z);
  }
  // ---- parity on the first np rows (a partly filled last tile on purpose), for the kernel and for the variant whose timing is quoted ----
  typedef void (*Lf)(const Args&, int, int64_t);
  auto parity = [&](const char* what, Lf mat, Lf fus) -> bool {
    printf("parity of %s:\n", what);
    Args b = a; b.n = std::min<int64_t>(n, 70000 + 37);
    const int64_t ss = (b.n + 3) & ~(int64_t)3;
    float *s1, *s2; CK(hipMalloc(&s1, (size_t)nq * ss * 4)); CK(hipMalloc(&s2, (size_t)nq * ss * 4));
    CK(hipMemset(s1, 0xff, (size_t)nq * ss * 4)); CK(hipMemset(s2, 0xff, (size_t)nq * ss * 4));
    b.S = s1; launch_tiled<false>(b, cus, ss);
    b.S = s2; mat(b, cus, ss);
    CK(hipDeviceSynchronize());
    std::vector<float> h1((size_t)nq * ss), h2((size_t)nq * ss);
    CK(hipMemcpy(h1.data(), s1, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), s2, h2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; double maxd = 0;
    for (int q = 0; q < nq; ++q)
      for (int64_t i = 0; i < b.n; ++i) {
        const float x = h1[q * ss + i], y = h2[q * ss + i];
        maxd = std::max(maxd, (double)fabsf(x - y));
        if (!(fabsf(x - y) <= 2e-6f)) { if (bad < 5) printf("  differ q %d row %lld: tiled %.9g direct %.9g\n", q, (long long)i, x, y); ++bad; }
      }
    printf("materialised scores, %lld rows x %d queries: %zu of %zu further than 2e-6 from gemm_tiled_kernel's (max |diff| %.3g: another k order)\n", (long long)b.n, nq, bad, (size_t)nq * b.n, maxd);
    // fused: the same candidate multiset per query
    std::vector<std::vector<uint64_t>> got[2];
    for (int v = 0; v < 2; ++v) {
      CK(hipMemset(b.st, 0, (size_t)128 * SCR_WORDS * 4));
      if (v == 0) launch_tiled<true>(b, cus, 0); else fus(b, cus, 0);
      CK(hipDeviceSynchronize());
      std::vector<uint32_t> hs((size_t)128 * SCR_WORDS); CK(hipMemcpy(hs.data(), b.st, hs.size() * 4, hipMemcpyDeviceToHost));
      got[v].resize(nq);
      for (int q = 0; q < nq; ++q) {
        const uint32_t c = std::min<uint32_t>(hs[(size_t)q * SCR_WORDS], CAND_CAP);
        got[v][q].resize(c);
        CK(hipMemcpy(got[v][q].data(), b.cand + (size_t)q * CAND_CAP, (size_t)c * 8, hipMemcpyDeviceToHost));
        std::sort(got[v][q].begin(), got[v][q].end());
      }
    }
    // the two kernels' candidate ROWS must agree except where a score sits within 2e-6 of the threshold
    std::vector<float> hthr(128); CK(hipMemcpy(hthr.data(), b.thr, 128 * 4, hipMemcpyDeviceToHost));
    size_t tot = 0, qbad = 0, edge = 0;
    for (int q = 0; q < nq; ++q) {
      tot += got[0][q].size();
      std::vector<uint32_t> r0, r1, d;
      for (auto k : got[0][q]) r0.push_back((uint32_t)k);
      for (auto k : got[1][q]) r1.push_back((uint32_t)k);
      std::sort(r0.begin(), r0.end()); std::sort(r1.begin(), r1.end());
      std::set_symmetric_difference(r0.begin(), r0.end(), r1.begin(), r1.end(), std::back_inserter(d));
      bool ok = true;
      for (auto row : d) { if (fabsf(h1[q * ss + row] - hthr[q]) <= 2e-6f) ++edge; else ok = false; }
      if (!ok) { if (qbad < 5) printf("  query %d: %zu vs %zu candidates, %zu rows differ\n", q, r0.size(), r1.size(), d.size()); ++qbad; }
    }
    printf("fused candidates: %zu in all (%.1f per query), %zu of %d queries differ from gemm_tiled_kernel beyond scores at the threshold (%zu such rows)\n", tot, (double)tot / nq, qbad, nq, edge);
    CK(hipFree(s1)); CK(hipFree(s2));
    if (bad || qbad) { printf("PARITY FAILED\n"); return false; }
    return true;
  };
  if (!parity("the kernel (two register sets, next slab transposed in the MFMAs' shadow)", launch_direct<false>, launch_direct<true>)) return 2;
  if (!parity("EXP 8 (three sets, loads at the slab's top)", launch_direct<false, 8, 3>, launch_direct<true, 8, 3>)) return 2;
  // ---- timing ---------------------------------------------------------------------------------------------------------
  a.S = nullptr;
  typedef void (*Fn)(const Args&, int, int64_t);
  struct V { const char* name; Fn fn; } vs[] = {{"phased QT=128 (round 3, ships)", launch_phased128}, {"direct (corpus register-direct)", launch_direct<true>},
                  {"direct, loads + waits + barriers only", launch_direct<true, 1>}, {"direct, no corpus loads", launch_direct<true, 2>},
                  {"direct, corpus loads not nontemporal", launch_direct<true, 3>},
                  {"direct, three slabs in flight (PF = 3)", launch_direct<true, 0, 3>},
                  {"direct, loads (issued 4k cycles late) + waits only", launch_direct<true, 7>}, {"direct, PF = 3, late loads + waits only", launch_direct<true, 7, 3>},
                  {"direct, 3 sets, loads at the slab's top, own transposition up front", launch_direct<true, 8, 3>}};
  const int NV = sizeof(vs) / sizeof(vs[0]);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<std::vector<float>> ms(NV);
  for (int v = 0; v < NV; ++v) { CK(hipMemset(a.st, 0, (size_t)128 * SCR_WORDS * 4)); vs[v].fn(a, cus, 0); vs[v].fn(a, cus, 0); CK(hipDeviceSynchronize()); }
  for (int r = 0; r < rounds; ++r)
    for (int v = 0; v < NV; ++v) {
      CK(hipMemset(a.st, 0, (size_t)128 * SCR_WORDS * 4));
      CK(hipEventRecord(e0));
      for (int i = 0; i < 4; ++i) vs[v].fn(a, cus, 0);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[v].push_back(t / 4);
    }
  // in-kernel clock of the direct variants: shader cycles / 100 MHz ticks, median over workgroups
  std::vector<double> ghz(NV, 0.0);
  {
    for (int v = 1; v < NV; ++v) {
      for (int i = 0; i < 3; ++i) vs[v].fn(a, cus, 0);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> hc((size_t)cus * 4); CK(hipMemcpy(hc.data(), cb, hc.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> g;
      for (int b = 0; b < cus; ++b) { const double cyc = (double)(hc[b * 4 + 1] - hc[b * 4]), tick = (double)(hc[b * 4 + 3] - hc[b * 4 + 2]); if (tick > 0) g.push_back(cyc / tick * 0.1); }
      std::sort(g.begin(), g.end());
      if (!g.empty()) ghz[v] = g[g.size() / 2];
    }
  }
  const double bytes = (double)n * a.ldb;
  printf("EB=%d  %lld x %d, %d queries (%d CUs): median [min] of %d rounds of 4 launches\n", EB, (long long)n, d, nq, cus, rounds);
  for (int v = 0; v < NV; ++v) {
    std::sort(ms[v].begin(), ms[v].end());
    const float med = ms[v][ms[v].size() / 2], mn = ms[v][0];
    printf("  %-70s %7.3f ms [%7.3f]  %6.2f TB/s of corpus  %7.1f TFLOP/s  %.2f GHz\n", vs[v].name, med, mn, bytes / (med * 1e-3) / 1e12, 2.0 * n * d * nq / (med * 1e-3) / 1e12, ghz[v]);
  }
  return 0;
}
