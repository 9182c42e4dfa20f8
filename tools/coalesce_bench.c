/* Concurrent single-query callers of the C ABI (tools only): T threads call svs_index_search(nq = 1)
 * back to back for a few seconds, without and with svs_index_set_coalesce.
 *   build: gcc -O2 -I include tools/coalesce_bench.c -o tools/coalesce_bench_c -L svs_amd/lib -lsvs_amd -lpthread -lm -Wl,-rpath,$PWD/svs_amd/lib -Wl,-rpath,/opt/rocm/lib
 *   usage: coalesce_bench_c [rows=1000000] [dim=1536] [seconds=3] [threads=1,4,16,64,256] [modes=012]
 *          (modes: 0 solo, 1 coalesced, 2 coalesced without the whole-tile cut; bench.py runs "64 1" and reads the
 *           RESULT lines) */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "svs_amd.h"

enum { NQ = 256, K = 100 };
static svs_index* g_idx;
static float* g_q;
static int g_d;
static volatile int g_stop;
static long g_done[512];

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void* caller(void* arg) {
  const long t = (long)arg;
  float s[K];
  int64_t r[K];
  int32_t count;
  long i = t, n = 0;
  while (!g_stop) {
    if (svs_index_search(g_idx, g_q + (size_t)(i % NQ) * g_d, 1, g_d, K, s, r, &count) != SVS_OK) { fprintf(stderr, "search: %s\n", svs_last_error()); exit(5); }
    i += 7; ++n;
  }
  g_done[t] = n;
  return NULL;
}

int main(int argc, char** argv) {
  const long rows = argc > 1 ? atol(argv[1]) : 1000000;
  const int d = argc > 2 ? atoi(argv[2]) : 1536;
  const double secs = argc > 3 ? atof(argv[3]) : 3.0;
  int threads[16] = {1, 4, 16, 64, 256}, nthreads = 5;
  const char* modes = argc > 5 ? argv[5] : "012";
  float* m = (float*)malloc(sizeof(float) * (size_t)rows * d);
  unsigned long long x = 88172645463325252ull;
  size_t i;
  int ti, mode;
  g_d = d;
  if (argc > 4) {
    char* p = argv[4];
    nthreads = 0;
    while (*p && nthreads < 16) { threads[nthreads] = (int)strtol(p, &p, 10); if (threads[nthreads] > 0 && threads[nthreads] <= 512) ++nthreads; if (*p == ',') ++p; else break; }
  }
  g_q = (float*)malloc(sizeof(float) * NQ * d);
  if (!m || !g_q) return 2;
  for (i = 0; i < (size_t)rows * d; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; m[i] = ((float)(x >> 40) / 8388608.0f - 1.0f) * 0.044f; }
  for (i = 0; i < (size_t)NQ * d; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; g_q[i] = ((float)(x >> 40) / 8388608.0f - 1.0f) * 0.044f; }
  if (svs_index_create(m, rows, d, SVS_DTYPE_F32, 0, 0, &g_idx) != SVS_OK) { fprintf(stderr, "create: %s\n", svs_last_error()); return 4; }
  free(m);
  for (ti = 0; ti < nthreads; ++ti)
    for (mode = 0; mode < 3; ++mode) {
      if (!strchr(modes, '0' + mode)) continue;
      pthread_t th[512];
      const int T = threads[ti];
      long t, total = 0;
      int64_t p0, q0, p1, q1;
      double t0, dt;
      svs_index_set_coalesce(g_idx, mode);
      svs_index_coalesce_stats(g_idx, &p0, &q0);
      g_stop = 0;
      t0 = now();
      for (t = 0; t < T; ++t) pthread_create(&th[t], NULL, caller, (void*)t);
      while (now() - t0 < secs) { struct timespec ts = {0, 20000000}; nanosleep(&ts, NULL); }
      g_stop = 1;
      for (t = 0; t < T; ++t) { pthread_join(th[t], NULL); total += g_done[t]; }
      dt = now() - t0;
      svs_index_coalesce_stats(g_idx, &p1, &q1);
      printf("%3d threads %-9s: %9.0f queries/s  (mean latency %.2f ms", T, mode == 0 ? "solo" : (mode == 1 ? "coalesced" : "coal. all"), total / dt, 1e3 * dt * T / (double)total);
      if (mode) printf(", %.1f queries per corpus pass", (double)(q1 - q0) / (double)(p1 - p0 > 0 ? p1 - p0 : 1));
      printf(")\n");
      printf("RESULT %d %d %.1f %.4f %.2f\n", T, mode, total / dt, 1e3 * dt * T / (double)total, mode ? (double)(q1 - q0) / (double)(p1 - p0 > 0 ? p1 - p0 : 1) : 1.0);
      fflush(stdout);
    }
  svs_index_release(g_idx);
  return 0;
}
