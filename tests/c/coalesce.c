/* Pure-C callers of svs_index_set_coalesce: 24 threads search ONE handle one query at a time, each
 * with its own k; every answer must have the rows of the solo search in the same order (scores within the
 * f32 summation noise: coalesced queries are answered by the batched kernels), wrong-dimension calls made
 * alongside must fail on their own, and fewer corpus passes than searches must have been made.
 *   usage: coalesce      prints "ok <passes> <queries>" */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svs_amd.h"

enum { N = 150000, D = 384, NQ = 96, KMAX = 64, T = 24, REPS = 4 };
static svs_index* g_idx;
static float* g_q;
static float g_s[NQ][KMAX];
static int64_t g_r[NQ][KMAX];
static int g_bad;

static int k_of(int qi) { return 3 + qi % (KMAX - 3); }

static void* caller(void* arg) {
  const int t = (int)(long)arg;
  int rep, qi, i;
  for (rep = 0; rep < REPS; ++rep)
    for (qi = t; qi < NQ; qi += T) {
      float s[KMAX];
      int64_t r[KMAX];
      int32_t count = 0;
      const int k = k_of(qi);
      if (svs_index_search(g_idx, g_q + (size_t)qi * D, 1, D, k, s, r, &count) != SVS_OK || count != k) { __sync_fetch_and_add(&g_bad, 1); continue; }
      for (i = 0; i < k; ++i)
        if (r[i] != g_r[qi][i] || fabsf(s[i] - g_s[qi][i]) > 1e-5f) { __sync_fetch_and_add(&g_bad, 1); break; }
      if (svs_index_search(g_idx, g_q, 1, D + 1, k, s, r, &count) != SVS_ERR_SHAPE) __sync_fetch_and_add(&g_bad, 1);
    }
  return NULL;
}

int main(void) {
  float* m = (float*)malloc(sizeof(float) * (size_t)N * D);
  pthread_t th[T];
  unsigned s = 4242u;
  int64_t passes = 0, queries = 0;
  size_t i;
  int qi, t;
  g_q = (float*)malloc(sizeof(float) * NQ * D);
  if (!m || !g_q) return 2;
  for (i = 0; i < (size_t)N * D; ++i) { s = s * 1664525u + 1013904223u; m[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.08f; }
  for (i = 0; i < (size_t)NQ * D; ++i) { s = s * 1664525u + 1013904223u; g_q[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.08f; }
  if (svs_device_count() <= 0) { fprintf(stderr, "no device\n"); return 3; }
  if (svs_index_create(m, N, D, SVS_DTYPE_F32, 0, 0, &g_idx) != SVS_OK) { fprintf(stderr, "create: %s\n", svs_last_error()); return 4; }
  free(m);
  for (qi = 0; qi < NQ; ++qi) {   /* solo answers */
    int32_t count = 0;
    if (svs_index_search(g_idx, g_q + (size_t)qi * D, 1, D, k_of(qi), g_s[qi], g_r[qi], &count) != SVS_OK) return 5;
  }
  if (svs_index_set_coalesce(g_idx, 1) != SVS_OK) return 6;
  for (t = 0; t < T; ++t) pthread_create(&th[t], NULL, caller, (void*)(long)t);
  for (t = 0; t < T; ++t) pthread_join(th[t], NULL);
  if (g_bad) { fprintf(stderr, "%d wrong answers\n", g_bad); return 7; }
  if (svs_index_coalesce_stats(g_idx, &passes, &queries) != SVS_OK) return 8;
  if (queries != (int64_t)NQ * REPS || passes >= queries) { fprintf(stderr, "passes %lld queries %lld\n", (long long)passes, (long long)queries); return 9; }
  svs_index_release(g_idx);
  printf("ok %lld %lld\n", (long long)passes, (long long)queries);
  return 0;
}
