/*
 * svs_amd.h -- C ABI of the MI355X (gfx950) brute-force similarity backend for SVS.
 *
 * This is the drop-in boundary for the ONE hot path of Rhobota/svs v0.7.4: the
 * cosine-similarity + top-k search inside KB.retrieve()/AsyncKB.retrieve().
 * The reference has no FFI of its own (it is pure Python + NumPy), so each entry
 * point below cites the reference lines it replaces; INTEGRATION.md shows the
 * ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain pointers and sizes only; no exceptions / abort() cross this boundary;
 *   - every function returns SVS_OK (0) or a negative svs_status; the message for
 *     the calling thread's last failure is svs_last_error() (thread-local);
 *   - status -> Python exception the host wrapper raises, matching the reference:
 *       SVS_ERR_SHAPE   ValueError  (numpy: "shapes (N,D) and (d,) not aligned",
 *                                    src/svs/kb.py:1623 with a wrong-sized query)
 *       SVS_ERR_INVALID ValueError  (bad argument)
 *       SVS_ERR_DEVICE  RuntimeError (HIP failure)     SVS_ERR_NOMEM MemoryError
 *   - an index handle is reference counted: create() returns it with one
 *     reference; search calls hold a reference for their duration, so release()
 *     from _EmbeddingsMatrix.invalidate() (src/svs/kb.py:861-864) may race with
 *     an in-flight AsyncKB search on an executor thread (src/svs/kb.py:1184-1190);
 *   - all search entry points are re-entrant on one handle.
 */
#ifndef SVS_AMD_H
#define SVS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svs_index svs_index;

typedef enum svs_status {
  SVS_OK = 0,
  SVS_ERR_INVALID = -1,
  SVS_ERR_SHAPE = -2,
  SVS_ERR_DEVICE = -3,
  SVS_ERR_NOMEM = -4,
  SVS_ERR_UNSUPPORTED = -5
} svs_status;

/* element type of the HBM-resident corpus (the arithmetic stays f32-accumulate) */
typedef enum svs_dtype {
  SVS_DTYPE_F32 = 0, /* reference layout: np.zeros((n, m), float32), src/svs/kb.py:600 */
  SVS_DTYPE_F16 = 1, /* extension for BASELINE.json configs[2]/[3]: rows AND queries rounded to IEEE
                        half (RNE), products exact in f32, f32 accumulate.  Parity oracle: numpy's
                        f32 path on the dequantised corpus and query. */
  SVS_DTYPE_FP8 = 2  /* extension for BASELINE.json configs[4]: OCP e4m3fn rows with one f32 scale per
                        row (max|x| -> 448), queries quantised the same way; score = scale_row *
                        scale_query * sum(q8 * q8) accumulated in f32.  Same oracle rule.  (Batches of more
                        than 16 queries run on the fp8 matrix instruction, whose internal sum of 128 products
                        keeps ~13 bits below the largest one: scores within 3e-6 of the exact dot product of the
                        quantised values on unit-norm data, up to 1.4e-4 relative when a single product
                        dominates; single queries use f32 FMAs: 2e-7.) */
} svs_dtype;

typedef struct svs_index_info_t {
  int64_t n;          /* rows held by this handle (one shard)                 */
  int32_t d;          /* logical dimension                                     */
  int32_t ld;         /* padded row stride in elements (HBM layout)            */
  int32_t dtype;      /* svs_dtype                                             */
  int32_t device;     /* HIP device ordinal                                    */
  int64_t row_offset; /* global row index of local row 0 (row sharding, 8(e))  */
  int64_t hbm_bytes;  /* bytes of HBM held by the corpus                       */
  int64_t n_masked;   /* rows tombstoned by svs_index_mask_rows (still counted in n) */
} svs_index_info_t;

/* Stages timed with HIP events on the stream the kernels run on (bench.py roofline). */
typedef struct svs_timing_t {
  double score_ms_sum;  /* dominant kernel: query x corpus GEMV/GEMM            */
  double select_ms_sum; /* top-k select + order kernels                          */
  int64_t launches;     /* number of searches accumulated                        */
  double dominant_ms_sum; /* the ONE dominant kernel launch inside the score stage: for a fused batched
                             search the whole-corpus GEMM (without query staging and the prefix pass that
                             seeds its thresholds); otherwise equal to score_ms_sum              */
} svs_timing_t;

/* ---- library ------------------------------------------------------------- */
const char* svs_version(void);
const char* svs_last_error(void);
/* Number of visible HIP devices (0 when none; never fails). */
int32_t svs_device_count(void);
/* Free / total HBM of a device in bytes (capacity planning, leak tests). */
int32_t svs_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes);

/* ---- index lifetime: replaces the cached (embeddings_matrix, emb_id_lookup)
 *      pair of _EmbeddingsMatrix, src/svs/kb.py:856-893 --------------------- */

/* One pinned-staged H2D copy of a C-contiguous f32 (n, d) host matrix -- the
 * array _Querier.build_embeddings_matrix returns (src/svs/kb.py:573-618) -- into
 * HBM on `device`.  The library copies, never aliases: the caller keeps owning
 * host_rows.  n == 0 or d == 0 is representable (search then reports
 * SVS_ERR_SHAPE, like numpy on a (0,0) matrix).  row_offset is added to every
 * returned row index (row-sharded corpora; 0 for a whole corpus). */
int32_t svs_index_create(const float* host_rows, int64_t n, int32_t d, int32_t store_dtype,
                         int32_t device, int64_t row_offset, svs_index** out);

/* Same, from rows already in device memory on `device` (f32, row stride
 * src_ld elements): synthetic corpora generated on the GPU, shards staged
 * GPU->GPU.  Copies; does not alias. */
int32_t svs_index_create_from_device(const float* dev_rows, int64_t n, int32_t d, int64_t src_ld,
                                     int32_t store_dtype, int32_t device, int64_t row_offset,
                                     svs_index** out);

/* ---- incremental update: instead of dropping the whole cached matrix on every
 *      bulk_add_docs / bulk_del_docs (invalidate(), src/svs/kb.py:1523, :1541, :1062, :1086)
 *      and re-uploading 6-300 GB, the HBM copy is edited in place. */

/* Appends n_new rows (f32, C-contiguous (n_new, d), host) behind the existing ones:
 * new local rows n .. n+n_new-1, exactly where `SELECT id, embedding FROM embeddings`
 * (src/svs/kb.py:603-609) puts newly inserted embeddings.  Grows the HBM buffers by
 * 1.5x when needed.  Blocks searches on this handle while it runs. */
int32_t svs_index_append(svs_index* idx, const float* host_rows, int64_t n_new);

/* Same, from rows already in device memory on the index's device (f32, row stride src_ld
 * elements): corpora generated or staged on the GPU block by block, so that the f32 source of a
 * 10M x 3072 fp8 corpus (123 GB) never has to exist at once.  With svs_index_create(NULL, 0, d, ...)
 * + svs_index_reserve() this builds an index of any size from device blocks without a
 * reallocation.  Copies; does not alias. */
int32_t svs_index_append_from_device(svs_index* idx, const float* dev_rows, int64_t n_new, int64_t src_ld);

/* Grows the HBM buffers to hold rows_capacity rows (no-op when they already do), so that the
 * appends that follow neither reallocate nor copy.  The matrix the reference builds is sized
 * from SELECT COUNT(*) up front the same way (src/svs/kb.py:574-601). */
int32_t svs_index_reserve(svs_index* idx, int64_t rows_capacity);

/* ---- cold start without a host matrix: replaces the row loop of _Querier.build_embeddings_matrix,
 *      src/svs/kb.py:603-615 (98.7 s per 1M rows published) ---------------
 * The caller decodes SQLite BLOBs (little-endian f32, src/svs/embeddings/util.py:15-23) STRAIGHT
 * into pinned staging memory owned by the library and commits block after block; the DMA of block i
 * overlaps the filling of block i + 1 (two 32 MiB blocks).  No (n, m) host matrix, no second host
 * copy.  One producer at a time per handle; searches issued before svs_index_staging_finish() wait
 * for the pending copies first.
 *   acquire: a pinned block of *rows_cap rows x d floats the caller may fill (blocks until the DMA
 *            that last read this block has finished);
 *   commit:  appends the first n_rows rows of the block last acquired behind the existing rows
 *            (as svs_index_append) and returns once the copy is ENQUEUED;
 *   finish:  waits for every pending copy and frees the staging blocks. */
int32_t svs_index_staging_acquire(svs_index* idx, float** host_block, int64_t* rows_cap);
int32_t svs_index_staging_commit(svs_index* idx, int64_t n_rows);
int32_t svs_index_staging_finish(svs_index* idx);

/* Tombstones rows (GLOBAL indices, i.e. row_offset + local): they keep their index
 * (later rows do not shift, so the caller's emb_id_lookup stays valid) but can never
 * be returned again; count = min(k, n - masked).  Relative order of the surviving rows
 * is unchanged, so results equal those of a rebuilt matrix up to the row numbering. */
int32_t svs_index_mask_rows(svs_index* idx, const int64_t* rows, int64_t count);

int32_t svs_index_retain(svs_index* idx);
/* Drops one reference; HBM is freed when the last holder (including in-flight
 * searches) lets go.  Called from invalidate(), src/svs/kb.py:861-864. */
int32_t svs_index_release(svs_index* idx);
int32_t svs_index_info(const svs_index* idx, svs_index_info_t* out);

/* ---- search: replaces np.dot + get_top_k of superheavy(),
 *      src/svs/kb.py:1622-1627 (sync) / :1184-1189 (async),
 *      src/svs/util.py:190-203 ----------------------------------------------- */

/* nq queries (f32, C-contiguous (nq, d), host memory).  Per query:
 *   count = min(max(k, 0), n)            (src/svs/util.py:198-201)
 *   out_scores[i*k .. i*k+count)  f32 dot products, descending
 *   out_rows  [i*k .. i*k+count)  row_offset + local row; order is
 *                                 (score desc, row desc)  (src/svs/util.py:203)
 * Rows are ROW indices; the caller applies emb_id_lookup (src/svs/kb.py:1626).
 * d != index d -> SVS_ERR_SHAPE.  Blocks until the results are in the output
 * buffers.  The corpus is NOT re-normalised (vectors are validated unit-norm by
 * the reference, src/svs/embeddings/util.py:26-41, so cosine == dot). */
int32_t svs_index_search(svs_index* idx, const float* queries, int32_t nq, int32_t d, int32_t k,
                         float* out_scores, int64_t* out_rows, int32_t* out_count);

/* Coalescing of concurrent single-query searches (off by default).  AsyncKB.retrieve runs its np.dot on
 * an executor thread outside the KB lock (src/svs/kb.py:1184-1190), so a server has many threads inside the
 * search at once; with enable != 0, svs_index_search calls with nq == 1 that arrive while the device is busy
 * are queued and answered TOGETHER by one batched pass over the corpus when it is free (up to 256; a call that
 * finds the device idle runs at once, alone).  The answer is the SAME total order (score desc, row desc,
 * src/svs/util.py:203) applied to scores that can differ from the single-query kernels' in the last bits
 * (the batched kernels sum in another order, far inside 1e-5): two rows whose scores are closer than that
 * rounding noise may therefore come out swapped, exactly as between two runs of numpy's own sgemv with
 * different blocking (hard part 1).  On every golden corpus recorded from the reference the
 * coalesced rows equal the reference's position by position (tests/test_search_gpu.py::test_search_golden).  Errors stay with the
 * call that made them.  svs_index_coalesce_stats: passes made / queries answered through this path;
 * svs_index_coalesce_sizes: out[s] = passes that carried exactly s queries, s < cap (cap <= 257). */
int32_t svs_index_set_coalesce(svs_index* idx, int32_t enable);
int32_t svs_index_coalesce_stats(svs_index* idx, int64_t* passes, int64_t* queries);
int32_t svs_index_coalesce_sizes(svs_index* idx, int64_t* out, int32_t cap);
/* Device-resident variant for pipelines and the multi-GPU gather (8(e)): queries
 * and outputs are device pointers on the index's device, work is enqueued on
 * `hip_stream` (a hipStream_t; NULL = the default stream) and the call returns
 * without synchronising.  Output layout as above with stride k; entries past
 * count are filled with score = -inf, row = -1.  *out_count is written on the
 * host immediately (it depends only on k and n). */
int32_t svs_index_search_device(svs_index* idx, const float* dev_queries, int32_t nq, int32_t d,
                                int32_t k, float* dev_out_scores, int64_t* dev_out_rows,
                                int32_t* out_count, void* hip_stream);

/* All scores of one query, f32 (n) to host: the raw `np.dot(M, q)` vector
 * (src/svs/kb.py:1623) for callers that want it and for parity tests.  The call writes one float per
 * row the handle holds WHEN IT RUNS and never more than out_capacity: a handle that has grown past the
 * caller's buffer (svs_index_append / svs_index_staging_commit from another thread between sizing the
 * buffer and this call) is SVS_ERR_INVALID, not an overflow -- the row count is read under the same
 * lock that appends take exclusively.  *out_n (may be NULL) = the rows the handle holds, written in
 * every case, so a caller can re-size and retry.  (Replaces round 3's svs_index_scores(idx, q, d, out),
 * which had no capacity and overflowed the caller's heap in exactly that race; the old symbol is gone
 * on purpose: a stale binding fails at load time instead of corrupting memory.) */
int32_t svs_index_scores_n(svs_index* idx, const float* query, int32_t d, float* out_scores,
                           int64_t out_capacity, int64_t* out_n);

/* ---- pairwise: replaces np.dot(M, M.T) + get_top_pairs of
 *      document_top_pairwise_scores, src/svs/kb.py:1642-1671, src/svs/util.py:206-233 --
 * The k best-scoring row PAIRS (i < j; diagonal and lower triangle ignored), ordered
 * (score desc, then i desc, then j desc -- the reference's flat upper-triangle index,
 * descending).  count = min(max(k,0), n(n-1)/2).  Up to n*n = 2^32 scores (65,536 rows) the
 * n x n matrix is materialised in HBM like the reference materialises it in RAM.  Beyond that
 * nothing of that size exists: the exact k-th best pair score of a prefix block bounds the
 * answer from below, the tiled MFMA GEMM runs with the i < j mask and that bound in its
 * epilogue, and only the surviving pairs are ordered (same order key, same result).  Needs rows
 * of whole 128-byte lines there; SVS_ERR_UNSUPPORTED if millions of pairs pass the bound
 * (near-duplicate documents en masse) or k needs a prefix block past 2^32 scores. */
int32_t svs_index_top_pairs(svs_index* idx, int32_t k, float* out_scores, int64_t* out_i, int64_t* out_j,
                            int32_t* out_count);

/* ---- one process, several GPUs: the `devices, ndev` form of the create/search pair (SURVEY.md
 *      8(b), 8(e)).  Replaces the same reference lines as svs_index_create / svs_index_search
 *      (src/svs/kb.py:875-876, :1623-1626, src/svs/util.py:190-203) for a KB process that owns a whole
 *      node: the (n, d) matrix is row-sharded, shard g = rows [g * ceil(n / ndev), ...) on devices[g]
 *      (a device may be listed more than once); a search runs on every shard at once (one worker
 *      thread per shard inside the library), each returns its local top-k with GLOBAL rows, and the
 *      calling thread merges them under the same total order -- scores, rows and order are those of
 *      ONE index over the whole matrix, for any ndev.  No RCCL: the exchange is ndev * k * 12 bytes
 *      of host memory.  Reference counted and re-entrant like svs_index. ------------------------- */
typedef struct svs_multi svs_multi;
int32_t svs_multi_create(const float* host_rows, int64_t n, int32_t d, int32_t store_dtype,
                         const int32_t* devices, int32_t ndev, svs_multi** out);
/* Same contract as svs_index_search (count = min(max(k, 0), live rows of ALL shards)). */
int32_t svs_multi_search(svs_multi* m, const float* queries, int32_t nq, int32_t d, int32_t k,
                         float* out_scores, int64_t* out_rows, int32_t* out_count);
int32_t svs_multi_retain(svs_multi* m);
int32_t svs_multi_release(svs_multi* m);
/* Shard count, total rows (masked ones included), dimension, masked rows; any pointer may be NULL. */
int32_t svs_multi_info(svs_multi* m, int32_t* ndev, int64_t* n, int32_t* d, int64_t* n_masked);
/* As svs_index_set_coalesce, in front of the shards: concurrent single-query svs_multi_search calls are
 * answered together by one batched search per shard. */
int32_t svs_multi_set_coalesce(svs_multi* m, int32_t enable);
int32_t svs_multi_coalesce_stats(svs_multi* m, int64_t* passes, int64_t* queries);
/* Shard g as an ordinary index handle (one more reference: release it with svs_index_release),
 * e.g. for svs_index_mask_rows / svs_index_info on the shard that holds a row. */
int32_t svs_multi_shard(svs_multi* m, int32_t g, svs_index** out);

/* ---- parity support (tests): what the index really holds -------------------- */
/* Rows [row0, row0 + nrows) exactly as stored, dequantised to f32 (nrows x d,
 * C-contiguous, host).  f32: the rows; f16: half-rounded; fp8: e4m3 * row scale. */
int32_t svs_index_debug_dequant(svs_index* idx, int64_t row0, int64_t nrows, float* out);
/* The query as the kernels of this index's dtype see it (f32: unchanged; f16:
 * half-rounded; fp8: quantised with its own scale and dequantised), d floats, host. */
int32_t svs_index_debug_query(svs_index* idx, const float* query, int32_t d, float* out);

/* ---- measurement --------------------------------------------------------- */
/* enable = N > 0: record HIP events around the score and select stages of every N-th
 * subsequent search on this handle and accumulate them (N = 1: every search; the three
 * event records cost ~10 us of stream time per timed search).  0: off. */
int32_t svs_index_set_timing(svs_index* idx, int32_t enable);
/* Waits for outstanding timed searches, returns the sums, and resets them. */
int32_t svs_index_get_timing(svs_index* idx, svs_timing_t* out);

/* Tuning knob for A/B runs (bench/profiling only): selects a kernel variant of
 * the score stage; 0 = library default.  Returns SVS_ERR_INVALID if unknown. */
int32_t svs_index_set_variant(svs_index* idx, int32_t variant);

#ifdef __cplusplus
}
#endif
#endif /* SVS_AMD_H */
