/* Pure-C caller of the multi-device entries (include/svs_amd.h, svs_multi_*): the same matrix behind
 * ONE svs_index and behind svs_multi handles of 2, 3 and 5 shards (all on device 0 on a one-GPU box)
 * must give the same scores and rows, in the same order -- including a k larger than a shard holds,
 * k = 0, a wrong dimension (SVS_ERR_SHAPE) and a shard handle borrowed for svs_index_mask_rows.
 *   usage: multi_device      prints "ok" */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svs_amd.h"

enum { N = 30011, D = 192, NQ = 9, K = 40 };

static int same(const float* a, const int64_t* ra, const float* b, const int64_t* rb, int n) {
  int i;
  for (i = 0; i < n; ++i)
    if (a[i] != b[i] || ra[i] != rb[i]) return 0;
  return 1;
}

int main(void) {
  float* m = (float*)malloc(sizeof(float) * (size_t)N * D);
  float* q = (float*)malloc(sizeof(float) * NQ * D);
  static float s1[NQ * K], s2[NQ * K];
  static int64_t r1[NQ * K], r2[NQ * K];
  const int32_t shards[3] = {2, 3, 5};
  int32_t devices[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  svs_index* one = NULL;
  unsigned s = 777u;
  int i, t;
  int32_t count = 0;
  if (!m || !q) return 2;
  for (i = 0; i < N * D; ++i) { s = s * 1664525u + 1013904223u; m[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.07f; }
  for (i = 0; i < NQ * D; ++i) { s = s * 1664525u + 1013904223u; q[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.07f; }
  memcpy(m + (size_t)20000 * D, m + (size_t)5 * D, sizeof(float) * D);   /* an exact tie across shards */
  if (svs_device_count() <= 0) { fprintf(stderr, "no device\n"); return 3; }
  if (svs_index_create(m, N, D, SVS_DTYPE_F32, 0, 0, &one) != SVS_OK) { fprintf(stderr, "create: %s\n", svs_last_error()); return 4; }
  if (svs_index_search(one, q, NQ, D, K, s1, r1, &count) != SVS_OK || count != K) return 5;
  for (t = 0; t < 3; ++t) {
    svs_multi* mu = NULL;
    svs_index* sh = NULL;
    int32_t g = 0, d = 0;
    int64_t n = 0, dead = 0, kill[2];
    if (svs_multi_create(m, N, D, SVS_DTYPE_F32, devices, shards[t], &mu) != SVS_OK) { fprintf(stderr, "multi_create: %s\n", svs_last_error()); return 6; }
    if (svs_multi_info(mu, &g, &n, &d, &dead) != SVS_OK || g != shards[t] || n != N || d != D || dead != 0) return 7;
    memset(r2, 0xff, sizeof r2);
    if (svs_multi_search(mu, q, NQ, D, K, s2, r2, &count) != SVS_OK || count != K) { fprintf(stderr, "multi_search: %s\n", svs_last_error()); return 8; }
    if (!same(s1, r1, s2, r2, NQ * K)) { fprintf(stderr, "%d shards: results differ from one index\n", shards[t]); return 9; }
    if (svs_multi_search(mu, q, NQ, D, 0, s2, r2, &count) != SVS_OK || count != 0) return 10;
    if (svs_multi_search(mu, q, NQ, D + 1, K, s2, r2, &count) != SVS_ERR_SHAPE) return 11;
    /* tombstone the best row of query 0 through the shard that holds it */
    kill[0] = r1[0];
    if (svs_multi_shard(mu, (int32_t)(kill[0] / ((N + shards[t] - 1) / shards[t])), &sh) != SVS_OK) return 12;
    if (svs_index_mask_rows(sh, kill, 1) != SVS_OK) { fprintf(stderr, "mask: %s\n", svs_last_error()); return 13; }
    svs_index_release(sh);
    if (svs_multi_search(mu, q, NQ, D, K, s2, r2, &count) != SVS_OK || count != K) return 14;   /* (the same batch shape: the same kernels, bit for bit) */
    if (r2[0] != r1[1] || s2[0] != s1[1]) { fprintf(stderr, "masked row still first: %lld %lld\n", (long long)r2[0], (long long)r1[1]); return 15; }
    if (svs_multi_retain(mu) != SVS_OK) return 16;
    svs_multi_release(mu);
    svs_multi_release(mu);
  }
  svs_index_release(one);
  free(m);
  free(q);
  printf("ok\n");
  return 0;
}
