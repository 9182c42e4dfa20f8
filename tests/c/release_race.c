/* Pure-C caller of the ABI (no Python wrapper holding an extra reference):
 * thread A runs a long search while thread B drops the ONLY owner reference
 * (include/svs_amd.h: "release() ... may race with an in-flight ... search").
 * The search must finish with correct results and the index must be destroyed
 * afterwards by the search's own reference -- with the retain guard declared
 * after the geometry lock (round-1 bug) the lock's destructor ran on freed memory.
 *   usage: release_race ROUNDS      prints "ok <rounds>" */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "svs_amd.h"

enum { N = 200000, D = 256, NQ = 64, K = 10 };
static svs_index* g_idx;
static volatile int g_started;
static float* g_q;
static float g_s[NQ * K];
static int64_t g_r[NQ * K];
static int g_rc;

static void* searcher(void* arg) {
  int32_t count = 0;
  (void)arg;
  g_started = 1;
  g_rc = svs_index_search(g_idx, g_q, NQ, D, K, g_s, g_r, &count);
  if (g_rc == SVS_OK && count != K) g_rc = -100;
  return NULL;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  float* m = (float*)malloc(sizeof(float) * (size_t)N * D);
  float exp_s[NQ * K];
  int64_t exp_r[NQ * K];
  int r, i;
  unsigned s = 12345u;
  g_q = (float*)malloc(sizeof(float) * NQ * D);
  if (!m || !g_q) return 2;
  for (i = 0; i < N * D; ++i) { s = s * 1664525u + 1013904223u; m[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.0625f; }
  for (i = 0; i < NQ * D; ++i) { s = s * 1664525u + 1013904223u; g_q[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.0625f; }
  if (svs_device_count() <= 0) { fprintf(stderr, "no device\n"); return 3; }
  for (r = 0; r <= rounds; ++r) {
    pthread_t t;
    struct timespec ts;
    if (svs_index_create(m, N, D, SVS_DTYPE_F32, 0, 0, &g_idx) != SVS_OK) { fprintf(stderr, "create: %s\n", svs_last_error()); return 4; }
    if (r == 0) {   /* reference run, nothing racing */
      int32_t count = 0;
      if (svs_index_search(g_idx, g_q, NQ, D, K, exp_s, exp_r, &count) != SVS_OK) return 5;
      svs_index_release(g_idx);
      continue;
    }
    g_started = 0;
    memset(g_r, 0xff, sizeof g_r);
    pthread_create(&t, NULL, searcher, NULL);
    while (!g_started) {}
    ts.tv_sec = 0;
    ts.tv_nsec = 150000L * (long)(1 + r % 4);   /* 0.15 .. 0.6 ms into the call (it has retained by then) */
    nanosleep(&ts, NULL);
    svs_index_release(g_idx);               /* the only owner lets go while the search runs */
    pthread_join(t, NULL);
    if (g_rc != SVS_OK) { fprintf(stderr, "search rc %d: %s\n", g_rc, svs_last_error()); return 6; }
    if (memcmp(g_r, exp_r, sizeof g_r) || memcmp(g_s, exp_s, sizeof g_s)) { fprintf(stderr, "results differ in round %d\n", r); return 7; }
  }
  printf("ok %d\n", rounds);
  free(m);
  free(g_q);
  return 0;
}
