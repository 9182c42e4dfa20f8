/* Pure-C caller: svs_index_mask_rows on shard handles (svs_multi_shard) WHILE svs_multi_search calls
 * that ask for every row are in flight on other threads.  The multi-device search sizes its answer from a
 * snapshot of the shards' live rows; a shard that lost rows since then returns fewer, and the search must
 * answer with what there is -- as ONE svs_index does under the same interleaving (count = min(k, live rows)
 * at the time of its search; the reference analogue is bulk_del_docs racing retrieve(), src/svs/kb.py:1541,
 * :1180-1190) -- not fail with "shards returned fewer rows than they hold" (ADVICE r2).
 *   usage: multi_mask_race      prints "ok ..." */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svs_amd.h"

enum { N = 240, D = 64, SHARDS = 4, SEARCHERS = 3, KILLS = 160 };
static svs_multi* mu;
static float q[D];
static volatile int stop_flag = 0;
static int failures = 0, searches = 0, shrunk = 0;
static pthread_mutex_t lk = PTHREAD_MUTEX_INITIALIZER;

static void* searcher(void* arg) {
  static __thread float s[N];
  static __thread int64_t r[N];
  (void)arg;
  while (!stop_flag) {
    int32_t count = -1;
    int i, bad = 0;
    const int rc = svs_multi_search(mu, q, 1, D, N, s, r, &count);
    if (rc != SVS_OK) { fprintf(stderr, "search failed: %s\n", svs_last_error()); bad = 1; }
    else {
      if (count < N - KILLS || count > N) bad = 1;
      for (i = 1; i < count; ++i)
        if (s[i] > s[i - 1] || (s[i] == s[i - 1] && r[i] > r[i - 1])) bad = 1;   /* (score desc, row desc) */
    }
    pthread_mutex_lock(&lk);
    failures += bad; searches += 1; shrunk += (rc == SVS_OK && count < N);
    pthread_mutex_unlock(&lk);
  }
  return NULL;
}

int main(void) {
  float* m = (float*)malloc(sizeof(float) * N * D);
  int32_t devices[SHARDS] = {0, 0, 0, 0};
  pthread_t th[SEARCHERS];
  unsigned s = 4242u;
  int i;
  if (!m) return 2;
  for (i = 0; i < N * D; ++i) { s = s * 1664525u + 1013904223u; m[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.1f; }
  for (i = 0; i < D; ++i) { s = s * 1664525u + 1013904223u; q[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.1f; }
  if (svs_device_count() <= 0) { fprintf(stderr, "no device\n"); return 3; }
  if (svs_multi_create(m, N, D, SVS_DTYPE_F32, devices, SHARDS, &mu) != SVS_OK) { fprintf(stderr, "create: %s\n", svs_last_error()); return 4; }
  for (i = 0; i < SEARCHERS; ++i) pthread_create(&th[i], NULL, searcher, NULL);
  for (i = 0; i < KILLS; ++i) {      /* one row at a time, walking over the shards */
    svs_index* sh = NULL;
    const int64_t row = (int64_t)((i * 61) % N);
    int64_t kill[1];
    kill[0] = row;
    if (svs_multi_shard(mu, (int32_t)(row / (N / SHARDS)), &sh) != SVS_OK) { fprintf(stderr, "shard: %s\n", svs_last_error()); return 5; }
    if (svs_index_mask_rows(sh, kill, 1) != SVS_OK) { fprintf(stderr, "mask: %s\n", svs_last_error()); return 6; }
    svs_index_release(sh);
  }
  stop_flag = 1;
  for (i = 0; i < SEARCHERS; ++i) pthread_join(th[i], NULL);
  {
    static float sf[N];
    static int64_t rf[N];
    int32_t count = -1;
    if (svs_multi_search(mu, q, 1, D, N, sf, rf, &count) != SVS_OK || count != N - KILLS) { fprintf(stderr, "final count %d\n", count); return 7; }
  }
  svs_multi_release(mu);
  free(m);
  if (failures) { fprintf(stderr, "%d of %d searches failed\n", failures, searches); return 8; }
  printf("ok: %d searches while %d rows were masked (%d of them saw a shrunken corpus)\n", searches, KILLS, shrunk);
  return 0;
}
