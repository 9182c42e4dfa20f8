/* Pure-C caller of svs_index_scores_n while another thread appends rows to the same handle.
 * The reader sizes its buffer from svs_index_info(), then calls -- the append may land in between
 * (round 3's capacity-less svs_index_scores overflowed the caller's heap in exactly that window:
 * gpurun_out/r3c_tests.log).  Every call must either fill exactly the rows the handle held when it
 * ran (and leave the guard words behind the buffer untouched) or fail with SVS_ERR_INVALID and report
 * the row count to retry with; never write past out_capacity.
 *   usage: scores_race ROUNDS      prints "ok <calls> <refused>" */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svs_amd.h"

enum { N0 = 20000, D = 64, STEP = 1500, APPENDS = 40, GUARD = 1024 };
static svs_index* g_idx;
static float* g_block;
static volatile int g_done;

static void* appender(void* arg) {
  int i;
  (void)arg;
  for (i = 0; i < APPENDS; ++i)
    if (svs_index_append(g_idx, g_block, STEP) != SVS_OK) { fprintf(stderr, "append: %s\n", svs_last_error()); break; }
  g_done = 1;
  return NULL;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3;
  float* m = (float*)malloc(sizeof(float) * (size_t)N0 * D);
  float q[D];
  long calls = 0, refused = 0;
  int r, i;
  unsigned s = 777u;
  g_block = (float*)malloc(sizeof(float) * (size_t)STEP * D);
  if (!m || !g_block) return 2;
  for (i = 0; i < N0 * D; ++i) { s = s * 1664525u + 1013904223u; m[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.125f; }
  for (i = 0; i < STEP * D; ++i) { s = s * 1664525u + 1013904223u; g_block[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.125f; }
  for (i = 0; i < D; ++i) { s = s * 1664525u + 1013904223u; q[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.125f; }
  if (svs_device_count() <= 0) { fprintf(stderr, "no device\n"); return 3; }
  for (r = 0; r < rounds; ++r) {
    pthread_t t;
    if (svs_index_create(m, N0, D, SVS_DTYPE_F32, 0, 0, &g_idx) != SVS_OK) { fprintf(stderr, "create: %s\n", svs_last_error()); return 4; }
    g_done = 0;
    pthread_create(&t, NULL, appender, NULL);
    while (!g_done) {
      svs_index_info_t info;
      int64_t now = -1, cap;
      float* out;
      int rc, g;
      if (svs_index_info(g_idx, &info) != SVS_OK) return 5;
      cap = info.n;                                   /* sized here ... */
      out = (float*)malloc(sizeof(float) * (size_t)(cap + GUARD));
      if (!out) return 2;
      for (g = 0; g < GUARD; ++g) memcpy(&out[cap + g], "\xde\xc0\xad\xde", 4);
      rc = svs_index_scores_n(g_idx, q, D, out, cap, &now);   /* ... used here: the handle may have grown */
      ++calls;
      for (g = 0; g < GUARD; ++g)
        if (memcmp(&out[cap + g], "\xde\xc0\xad\xde", 4)) { fprintf(stderr, "wrote past the capacity (guard word %d)\n", g); return 6; }
      if (rc == SVS_OK) {
        if (now != cap) { fprintf(stderr, "ok with %lld rows for a %lld-row buffer\n", (long long)now, (long long)cap); return 7; }
      } else if (rc == SVS_ERR_INVALID) {
        ++refused;
        if (now <= cap) { fprintf(stderr, "refused although %lld rows fit %lld\n", (long long)now, (long long)cap); return 8; }
      } else {
        fprintf(stderr, "scores_n rc %d: %s\n", rc, svs_last_error());
        return 9;
      }
      free(out);
    }
    pthread_join(t, NULL);
    {   /* quiescent: a correctly sized call succeeds and a short one is refused */
      svs_index_info_t info;
      int64_t now = 0;
      float* out;
      if (svs_index_info(g_idx, &info) != SVS_OK || info.n != N0 + (int64_t)APPENDS * STEP) return 10;
      out = (float*)malloc(sizeof(float) * (size_t)info.n);
      if (svs_index_scores_n(g_idx, q, D, out, info.n, &now) != SVS_OK || now != info.n) return 11;
      if (svs_index_scores_n(g_idx, q, D, out, info.n - 1, &now) != SVS_ERR_INVALID || now != info.n) return 12;
      free(out);
    }
    svs_index_release(g_idx);
  }
  printf("ok %ld %ld\n", calls, refused);
  free(m);
  free(g_block);
  return 0;
}
