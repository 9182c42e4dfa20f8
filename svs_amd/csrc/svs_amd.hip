// C ABI of the MI355X similarity backend (see include/svs_amd.h).
// Host side: HBM-resident corpus handle, per-call search contexts (stream +
// scratch, so searches are re-entrant), launch sequencing of the score stage
// (gemv_f32.h) and the top-k stage (select.h).  gfx950 only.
#include "../../include/svs_amd.h"
#include "internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "fp8.h"
#include "gemm_tiled.h"
#include "gemm_phased.h"
#include "gemm_q16.h"
#include "gemv_unrolled.h"
#include "gemv_f16.h"
#include "gemv_f32.h"
#include "select.h"

namespace {

using namespace svs;

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(e_ == hipErrorOutOfMemory ? SVS_ERR_NOMEM : SVS_ERR_DEVICE, "%s: %s (%s:%d)", \
                  #expr, hipGetErrorString(e_), __FILE__, __LINE__);                        \
  } while (0)

// ---- internal tunables (svs_internal_tune: tools and tests; not part of the ABI) --------------------
std::atomic<int64_t> g_tune_prefix_div{64};   // fused path: rows of the threshold prefix = n / this (>= FUSE_PREFIX_MIN)
std::atomic<int64_t> g_tune_upload{0};        // host batches: 0 = f16 batches are PULLED from pinned memory by the staging kernel, chunk by chunk
                                              // (no DMA, no f32 copy in HBM); 1 = round 3's staging + DMA for every dtype
std::atomic<int64_t> g_tune_spread{1};        // fused path: 1 = thresholds from a sample spread over the whole corpus (prefix_image), 0 = from its first rows (rounds 1-3)
thread_local double g_host_phase[6];          // svs_internal_host_phases: seconds since the call began (last svs_index_search on this thread)

struct EvTriple {
  hipEvent_t e0, e1, e2;
  hipEvent_t d0 = nullptr, d1 = nullptr;   // around the dominant kernel of a fused search (else unset)
};

// One in-flight search: stream, device scratch, pinned staging.
struct Ctx {
  hipStream_t stream = nullptr;
  float* q_dev = nullptr;       size_t q_cap = 0;        // floats
  float* q16 = nullptr;         size_t q16_cap = 0;      // [16][ld] zero-padded query group
  _Float16* qh = nullptr;       size_t qh_cap = 0;       // half queries, [rows][ld] zero padded
  uint8_t* q8 = nullptr;        size_t q8_cap = 0;       // e4m3 queries [rows][ld], zero padded
  float* q8f = nullptr;         size_t q8f_cap = 0;      // the same values as f32 (single-query kernel)
  float* q8s = nullptr;         size_t q8s_cap = 0;      // query scales
  const float* q_f32 = nullptr;                          // staged f32 queries (q16 or the caller's buffer)
  float* pref_s = nullptr;      size_t pref_s_cap = 0;   // fused GEMM: top-k of the prefix rows (thresholds)
  int64_t* pref_r = nullptr;    size_t pref_r_cap = 0;
  float* scores = nullptr;      size_t scores_cap = 0;   // floats
  uint32_t* hist = nullptr;     size_t hist_cap = 0;     // queries
  uint64_t* cand = nullptr;                               // counters live behind hist
  uint64_t* keys = nullptr;     size_t keys_cap = 0;     // u64
  float* q_pin = nullptr;       size_t q_pin_cap = 0;
  float* out_s_pin = nullptr;   int64_t* out_r_pin = nullptr; size_t out_pin_cap = 0;
  float* redo_s_pin = nullptr;  int64_t* redo_r_pin = nullptr; size_t redo_pin_cap = 0;   // results of re-run queries (search_host)
  float* redo_q_pin = nullptr;  size_t redo_q_cap = 0;                                      // ... and the queries themselves, gathered
  // Scratch is reused in stream order.  A context stays with the stream that
  // last used it; handing it to ANOTHER stream first drains the old one.
  hipStream_t last_stream = nullptr;
  bool async_pending = false;
};

}  // namespace

struct svs_index {
  std::atomic<int> refs{1};
  int device = 0;
  int64_t n = 0;
  int d = 0, ld = 0, dtype = SVS_DTYPE_F32;
  int64_t row_offset = 0;
  int64_t cap = 0;               // rows the buffers can hold (append grows them)
  void* rows = nullptr;
  float* row_scales = nullptr;   // fp8 only: one f32 per row
  size_t bytes = 0;
  // Corpus geometry lock: searches hold it shared while they enqueue (and, for the
  // host API, until their results are back); append / mask_rows take it exclusive.
  std::shared_mutex rw;
  std::vector<uint8_t> dead_flag;      // host, one per row
  std::vector<uint32_t> dead_list;     // host copy of the masked (tombstoned) local rows
  uint32_t* dead_dev = nullptr;        // device copy
  size_t dead_dev_cap = 0;
  std::vector<uint32_t> dead_bits;     // host bitmap, one bit per row (bit r & 31 of word r >> 5)
  uint32_t* dead_bits_dev = nullptr;   // device copy: the fused top-k path drops masked candidates with it
  size_t dead_bits_cap = 0;            // words
  int cu_count = 256;
  // The fused batch path's threshold sample ("prefix image", prefix_image()): pfx_nmat rows copied out of the corpus in
  // blocks of PFX_BLOCK rows taken every pfx_stride rows, as one contiguous matrix the batched kernels can run over.
  // Valid while pfx_n == n (an append / reserve / staging commit moves or extends the rows: they reset pfx_n).
  void* pfx_rows = nullptr;
  float* pfx_scales = nullptr;
  size_t pfx_cap = 0;                  // rows the two buffers hold
  int64_t pfx_n = -1, pfx_nmat = 0, pfx_stride = 0;
  const void* pfx_src = nullptr;       // idx->rows when the image was taken (a reallocation moves the rows)
  std::mutex pfx_mu;

  std::mutex mu;
  std::condition_variable cv;
  std::vector<Ctx*> free_ctx;
  int n_ctx = 0;
  static constexpr int kMaxCtx = 8;

  // cold-start staging (svs_index_staging_*): two pinned blocks, DMA'd on their own stream
  struct Staging {
    void* pin[2] = {nullptr, nullptr};
    float* dstage[2] = {nullptr, nullptr};   // f16 / fp8: the f32 block lands here, a kernel converts it
    hipEvent_t done[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    int cur = 1;                             // block handed out by the last acquire
    int64_t rows_cap = 0;
    bool active = false;
  } stg;
  std::mutex stg_mu;
  std::atomic<bool> staging_pending{false};

  // svs_index_set_coalesce: single-query host searches that are in flight together share corpus passes
  struct Waiter {
    const float* q;
    int k, count = 0, rc = SVS_OK;
    float* out_s;
    int64_t* out_r;
    std::string err;
    bool done = false, lead = false;
    std::condition_variable cv;
  };
  std::atomic<bool> coalesce{false};
  std::atomic<bool> co_round{true};
  std::atomic<int64_t> co_sizes[257] = {};   // co_sizes[s]: passes that carried s queries
  int co_hold = 0;                           // svs_index_coalesce_hold: the next pass waits for this many queued callers (under co_mu)
  std::condition_variable co_hold_cv;
  std::mutex co_mu;
  std::vector<Waiter*> co_pending;
  bool co_busy = false;
  std::atomic<int64_t> co_passes{0}, co_queries{0};

  std::atomic<int> timing{0};          // 0 off, N: time every N-th search
  std::atomic<uint32_t> timing_seq{0};
  std::atomic<int> variant{0};
  std::vector<EvTriple> evs;  // guarded by mu
};

namespace {

void ctx_destroy(Ctx* c) {
  if (!c) return;
  if (c->async_pending) (void)hipStreamSynchronize(c->last_stream);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  (void)hipFree(c->q_dev);
  (void)hipFree(c->q16);
  (void)hipFree(c->qh);
  (void)hipFree(c->q8);
  (void)hipFree(c->q8f);
  (void)hipFree(c->q8s);
  (void)hipFree(c->pref_s);
  (void)hipFree(c->pref_r);
  if (c->redo_q_pin) (void)hipHostFree(c->redo_q_pin);
  if (c->redo_s_pin) (void)hipHostFree(c->redo_s_pin);
  if (c->redo_r_pin) (void)hipHostFree(c->redo_r_pin);
  (void)hipFree(c->scores);
  (void)hipFree(c->hist);
  (void)hipFree(c->cand);
  (void)hipFree(c->keys);
  (void)hipHostFree(c->q_pin);
  (void)hipHostFree(c->out_s_pin);
  (void)hipHostFree(c->out_r_pin);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

void staging_free(svs_index* idx) {
  auto& g = idx->stg;
  if (g.st) (void)hipStreamSynchronize(g.st);
  for (int i = 0; i < 2; ++i) {
    if (g.pin[i]) (void)hipHostFree(g.pin[i]);
    if (g.dstage[i]) (void)hipFree(g.dstage[i]);
    if (g.done[i]) (void)hipEventDestroy(g.done[i]);
    g.pin[i] = nullptr; g.dstage[i] = nullptr; g.done[i] = nullptr;
  }
  if (g.st) (void)hipStreamDestroy(g.st);
  g.st = nullptr;
  g.active = false;
  idx->staging_pending.store(false);
}

void index_destroy(svs_index* idx) {
  (void)hipSetDevice(idx->device);
  staging_free(idx);
  for (Ctx* c : idx->free_ctx) ctx_destroy(c);
  for (auto& t : idx->evs) {
    (void)hipEventDestroy(t.e0);
    (void)hipEventDestroy(t.e1);
    (void)hipEventDestroy(t.e2);
    if (t.d0) (void)hipEventDestroy(t.d0);
    if (t.d1) (void)hipEventDestroy(t.d1);
  }
  (void)hipFree(idx->rows);
  (void)hipFree(idx->row_scales);
  (void)hipFree(idx->pfx_rows);
  (void)hipFree(idx->pfx_scales);
  (void)hipFree(idx->dead_dev);
  (void)hipFree(idx->dead_bits_dev);
  delete idx;
}

// svs_index_staging_commit publishes idx->n while its H2D copy and conversion are still queued on the staging
// stream: EVERY entry point that reads idx->rows (searches, scores, pairwise, debug read-back) waits here first.
int staging_wait(svs_index* idx) {
  if (!idx->staging_pending.load()) return SVS_OK;
  std::lock_guard<std::mutex> lk(idx->stg_mu);
  if (idx->stg.st) HIP_TRY(hipStreamSynchronize(idx->stg.st));
  idx->staging_pending.store(false);
  return SVS_OK;
}

// `want`: the stream the caller will enqueue on (nullptr = the context's own).
int ctx_acquire(svs_index* idx, hipStream_t want, bool own_stream, Ctx** out) {
  Ctx* c = nullptr;
  {
    std::unique_lock<std::mutex> lk(idx->mu);
    for (;;) {
      int pick = -1;
      for (int i = (int)idx->free_ctx.size() - 1; i >= 0; --i) {
        Ctx* f = idx->free_ctx[i];
        if (own_stream ? !f->async_pending : (f->async_pending && f->last_stream == want)) {
          pick = i;
          break;
        }
      }
      if (pick < 0 && idx->n_ctx < svs_index::kMaxCtx) {
        idx->n_ctx++;
        lk.unlock();
        c = new (std::nothrow) Ctx();
        hipError_t e = c ? hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) : hipErrorOutOfMemory;
        if (e != hipSuccess) {
          ctx_destroy(c);
          lk.lock();
          idx->n_ctx--;
          return fail(SVS_ERR_DEVICE, "search context: %s", hipGetErrorString(e));
        }
        *out = c;
        return SVS_OK;
      }
      if (pick < 0 && !idx->free_ctx.empty()) pick = (int)idx->free_ctx.size() - 1;
      if (pick >= 0) {
        c = idx->free_ctx[pick];
        idx->free_ctx.erase(idx->free_ctx.begin() + pick);
        break;
      }
      idx->cv.wait(lk);
    }
  }
  // migrating between streams: drain the previous user of this scratch (rare)
  if (c->async_pending && (own_stream || c->last_stream != want)) {
    (void)hipStreamSynchronize(c->last_stream);  // a destroyed stream has already drained
    c->async_pending = false;
  }
  *out = c;
  return SVS_OK;
}

void ctx_release(svs_index* idx, Ctx* c) {
  {
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->free_ctx.push_back(c);
  }
  idx->cv.notify_one();
}

template <typename T>
int grow_dev(T** p, size_t* cap, size_t need) {
  if (need <= *cap) return SVS_OK;
  if (*p) HIP_TRY(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  HIP_TRY(hipMalloc((void**)p, need * sizeof(T)));
  *cap = need;
  return SVS_OK;
}

int next_pow2_i64(int64_t v, int64_t* out) {
  int64_t p = 2;
  while (p < v) p <<= 1;
  *out = p;
  return 0;
}

// ---- score stage launch -----------------------------------------------------
template <int NSTEP, int R, int WPB, bool NT>
void launch_oneshot(const svs_index* idx, const float* q, float* scores, hipStream_t st) {
  const int64_t rows_per_block = (int64_t)R * WPB;
  const int64_t blocks = (idx->n + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL((gemv_f32_oneshot_kernel<NSTEP, R, WPB, NT, false>), dim3((unsigned)blocks), dim3(WPB * 64), 0, st,
                     (const v4f*)idx->rows, (const v4f*)q, scores, idx->n);
}

template <int NSTEP, int R, bool NT>
void launch_persistent(const svs_index* idx, const float* q, float* scores, hipStream_t st) {
  constexpr int WPB = 4;
  const int64_t tiles = (idx->n + R - 1) / R;
  const int blocks = (int)std::min<int64_t>((tiles + WPB - 1) / WPB, (int64_t)idx->cu_count * 4);
  hipLaunchKernelGGL((gemv_f32_rows_kernel<NSTEP, R, WPB, NT, false>), dim3(blocks), dim3(WPB * 64), 0, st,
                     (const v4f*)idx->rows, (const v4f*)q, scores, idx->n);
}

// Default geometry per row length (measured at NSTEP = 6: one-shot, 16-wave
// workgroups, one row per wave, nontemporal loads: 7.2 TB/s on MI355X).
// Short rows take several rows per wave so a wave still has >= 4 KiB in flight.
template <int NSTEP>
void launch_rows(const svs_index* idx, const float* q, float* scores, hipStream_t st, int variant) {
  switch (variant) {
    case 1: launch_persistent<NSTEP, 1, false>(idx, q, scores, st); return;
    case 2: launch_persistent<NSTEP, 2, true>(idx, q, scores, st); return;
    case 3: launch_oneshot<NSTEP, 2, 16, true>(idx, q, scores, st); return;
    case 4: launch_oneshot<NSTEP, 1, 16, false>(idx, q, scores, st); return;
    case 5: launch_oneshot<NSTEP, 1, 8, true>(idx, q, scores, st); return;
    default: break;
  }
  if constexpr (NSTEP <= 2) launch_oneshot<NSTEP, 4, 16, true>(idx, q, scores, st);
  else if constexpr (NSTEP <= 4) launch_oneshot<NSTEP, 2, 16, true>(idx, q, scores, st);
  else if constexpr (NSTEP <= 6) launch_oneshot<NSTEP, 1, 16, true>(idx, q, scores, st);
  else launch_oneshot<NSTEP, 1, 8, true>(idx, q, scores, st);
}

template <int T>
void launch_generic(const svs_index* idx, const float* q, float* scores, hipStream_t st) {
  constexpr int RPW = 64 / T;
  int64_t waves = (idx->n + RPW - 1) / RPW;
  int blocks = (int)std::min<int64_t>((waves + 3) / 4, (int64_t)idx->cu_count * 8);
  hipLaunchKernelGGL((gemv_f32_generic_kernel<T>), dim3(blocks), dim3(256), 0, st,
                     (const v4f*)idx->rows, q, scores, idx->n, idx->d, idx->ld / 4);
}

template <int NSTEP>
void launch_rows_f16(const svs_index* idx, const float* q, float* scores, hipStream_t st) {
  constexpr int R = NSTEP <= 1 ? 4 : (NSTEP <= 3 ? 2 : 1), WPB = NSTEP <= 6 ? 16 : 8;
  const int64_t rows_per_block = (int64_t)R * WPB;
  const int64_t blocks = (idx->n + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL((gemv_f16_oneshot_kernel<NSTEP, R, WPB>), dim3((unsigned)blocks), dim3(WPB * 64), 0, st,
                     (const u32x4*)idx->rows, (const v4f*)q, scores, idx->n);
}

template <int T>
void launch_generic_f16(const svs_index* idx, const _Float16* qh, float* scores, hipStream_t st) {
  constexpr int RPW = 64 / T;
  int64_t waves = (idx->n + RPW - 1) / RPW;
  int blocks = (int)std::min<int64_t>((waves + 3) / 4, (int64_t)idx->cu_count * 8);
  hipLaunchKernelGGL((gemv_f16_generic_kernel<T>), dim3(blocks), dim3(256), 0, st, (const u32x4*)idx->rows,
                     (const u32x4*)qh, scores, idx->n, idx->ld / 8);
}

// rounds nq f32 queries to half into c->qh ([rows_alloc][ld], rows >= nq zero)
// (row0 > 0: a later chunk of a batch staged piece by piece -- search_host; the buffer was grown by the first chunk's
//  call, which passes the whole batch's row count as rows_alloc, and only the last chunk zeroes the padding rows)
int stage_queries_f16(const svs_index* idx, Ctx* c, const float* q, int nq, int rows_alloc, hipStream_t st, int row0 = 0) {
  int rc = grow_dev(&c->qh, &c->qh_cap, (size_t)rows_alloc * idx->ld);
  if (rc != SVS_OK) return rc;
  _Float16* dst = c->qh + (size_t)row0 * idx->ld;
  rows_alloc -= row0;
  // (the kernel writes whole padded rows: only the rows behind the queries need zeroing)
  if (rows_alloc > nq) HIP_TRY(hipMemsetAsync(dst + (size_t)nq * idx->ld, 0, (size_t)(rows_alloc - nq) * idx->ld * sizeof(_Float16), st));
  hipLaunchKernelGGL(convert_queries_f16_kernel, dim3((unsigned)std::min<int64_t>(2048, ((int64_t)nq * idx->ld + 255) / 256)), dim3(256), 0, st, q, nq, idx->d, dst, idx->ld);
  return SVS_OK;
}

// quantises nq f32 queries to e4m3 into c->q8 ([rows_alloc][ld] bytes, rows >= nq zero) with
// scales c->q8s; want_f32 also fills c->q8f with the quantised values as f32
int stage_queries_fp8(const svs_index* idx, Ctx* c, const float* q, int nq, int rows_alloc, bool want_f32, hipStream_t st) {
  int rc;
  if ((rc = grow_dev(&c->q8, &c->q8_cap, (size_t)rows_alloc * idx->ld)) != SVS_OK) return rc;
  if ((rc = grow_dev(&c->q8s, &c->q8s_cap, (size_t)rows_alloc)) != SVS_OK) return rc;
  if (want_f32 && (rc = grow_dev(&c->q8f, &c->q8f_cap, (size_t)rows_alloc * idx->ld)) != SVS_OK) return rc;
  if (rows_alloc > nq) {   // (the kernel writes whole padded rows: only the rows behind the queries need zeroing)
    HIP_TRY(hipMemsetAsync(c->q8 + (size_t)nq * idx->ld, 0, (size_t)(rows_alloc - nq) * idx->ld, st));
    HIP_TRY(hipMemsetAsync(c->q8s + nq, 0, (size_t)(rows_alloc - nq) * sizeof(float), st));
  }
  hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3((nq + 3) / 4), dim3(256), 0, st, q, (int64_t)nq, idx->d, (int64_t)idx->d,
                     c->q8, idx->ld, c->q8s, want_f32 ? c->q8f : (float*)nullptr);
  return SVS_OK;
}

template <int T>
void launch_gemv_fp8(const svs_index* idx, Ctx* c, float* scores, hipStream_t st) {
  constexpr int RPW = 64 / T;
  int64_t waves = (idx->n + RPW - 1) / RPW;
  int blocks = (int)std::min<int64_t>((waves + 3) / 4, (int64_t)idx->cu_count * 8);
  hipLaunchKernelGGL((gemv_fp8_kernel<T>), dim3(blocks), dim3(256), 0, st, (const u32x4_t*)idx->rows, idx->row_scales,
                     (const v4f*)c->q8f, c->q8s, scores, idx->n, idx->ld / 16);
}

// Rows that are not whole 1 KiB wave loads (gemv_unrolled.h); false: longer than 16 KiB
template <class Dot>
bool launch_unrolled(const svs_index* idx, const void* q_staged, int ld16, float* scores, hipStream_t st, Dot dot) {
  const u32x4* M = (const u32x4*)idx->rows;
  const u32x4* q = (const u32x4*)q_staged;
#define SVS_UNROLLED(T, NC, U)                                                                                   \
  do {                                                                                                           \
    const int64_t groups = (idx->n + (64 / (T)) * (U) - 1) / ((64 / (T)) * (U));                                   \
    const int64_t blocks = (groups + UNR_WPB - 1) / UNR_WPB;                                                       \
    hipLaunchKernelGGL((gemv_unrolled_kernel<T, NC, U, Dot>), dim3((unsigned)blocks), dim3(UNR_WPB * 64), 0, st, M, q, scores, idx->n, ld16, dot); \
    return true;                                                                                                 \
  } while (0)
  if (ld16 <= 1) SVS_UNROLLED(1, 1, 8);
  if (ld16 <= 2) SVS_UNROLLED(2, 1, 8);
  if (ld16 <= 4) SVS_UNROLLED(4, 1, 8);
  if (ld16 <= 8) SVS_UNROLLED(8, 1, 8);
  if (ld16 <= 16) SVS_UNROLLED(16, 1, 8);
  if (ld16 <= 32) SVS_UNROLLED(32, 1, 8);
  if (ld16 <= 64) SVS_UNROLLED(64, 1, 8);
  if (ld16 <= 128) SVS_UNROLLED(64, 2, 4);
  if (ld16 <= 192) SVS_UNROLLED(64, 3, 2);
  if (ld16 <= 256) SVS_UNROLLED(64, 4, 2);
  if (ld16 <= 384) SVS_UNROLLED(64, 6, 1);
  if (ld16 <= 512) SVS_UNROLLED(64, 8, 1);
  if (ld16 <= 768) SVS_UNROLLED(64, 12, 1);
  if (ld16 <= 1024) SVS_UNROLLED(64, 16, 1);
#undef SVS_UNROLLED
  return false;
}

// q: device, d floats (unpadded); scores: device, n floats
int launch_scores(const svs_index* idx, Ctx* c, const float* q, float* scores, hipStream_t st) {
  const int variant = idx->variant.load();
  const bool q_aligned = (((uintptr_t)q) & 15) == 0;
  if (idx->dtype == SVS_DTYPE_FP8) {
    int rc = stage_queries_fp8(idx, c, q, 1, 1, true, st);
    if (rc != SVS_OK) return rc;
    // hot geometries: one-shot grid, 16 waves, nontemporal row loads (as gemv_f32.h)
#define SVS_FP8_HOT(NSTEP, LB, R)                                                                               \
  do {                                                                                                          \
    const int64_t blocks = (idx->n + (R) * 16 - 1) / ((R) * 16);                                                \
    hipLaunchKernelGGL((gemv_fp8_oneshot_kernel<NSTEP, LB, R, 16>), dim3((unsigned)blocks), dim3(16 * 64), 0, st, \
                       (const uint8_t*)idx->rows, idx->row_scales, (const uint8_t*)c->q8, c->q8s, scores, idx->n);     \
    return SVS_OK;                                                                                              \
  } while (0)
    switch (idx->ld) {
      // the query stays packed (converted next to the row bytes), so several short rows per
      // wave cost no registers (58-66 VGPRs)
      case 1024: SVS_FP8_HOT(1, 16, 4);   // >= 4 KiB per wave: 6.5 vs 4.2 TB/s with one row per wave
      case 2048: SVS_FP8_HOT(2, 16, 2);   // 6.5 vs 5.8
      case 3072: SVS_FP8_HOT(3, 16, 1);
      case 4096: SVS_FP8_HOT(4, 16, 1);
      case 512: SVS_FP8_HOT(1, 8, 4);    // 8-byte loads: several rows per wave keep enough bytes in flight
      case 1536: SVS_FP8_HOT(3, 8, 2);
      default: break;
    }
#undef SVS_FP8_HOT
    const int ld16 = idx->ld / 16;
    if (variant != 4 && launch_unrolled(idx, c->q8, ld16, scores, st, DotFp8{idx->row_scales, c->q8s})) return SVS_OK;
    if (ld16 <= 1) launch_gemv_fp8<1>(idx, c, scores, st);
    else if (ld16 <= 2) launch_gemv_fp8<2>(idx, c, scores, st);
    else if (ld16 <= 4) launch_gemv_fp8<4>(idx, c, scores, st);
    else if (ld16 <= 8) launch_gemv_fp8<8>(idx, c, scores, st);
    else if (ld16 <= 16) launch_gemv_fp8<16>(idx, c, scores, st);
    else if (ld16 <= 32) launch_gemv_fp8<32>(idx, c, scores, st);
    else launch_gemv_fp8<64>(idx, c, scores, st);
    return SVS_OK;
  }
  if (idx->dtype == SVS_DTYPE_F16) {
    if (idx->ld % 512 == 0 && idx->ld <= 4096) {
      const float* qq = q;   // the kernel rounds ld query floats itself: pad them when rows are padded
      if (idx->ld != idx->d || !q_aligned) {
        int rc = grow_dev(&c->q16, &c->q16_cap, (size_t)GQ * idx->ld);
        if (rc != SVS_OK) return rc;
        HIP_TRY(hipMemsetAsync(c->q16, 0, (size_t)idx->ld * sizeof(float), st));
        HIP_TRY(hipMemcpyAsync(c->q16, q, (size_t)idx->d * sizeof(float), hipMemcpyDeviceToDevice, st));
        qq = c->q16;
      }
      switch (idx->ld / 512) {
#define SVS_ROWS_CASE(N) case N: launch_rows_f16<N>(idx, qq, scores, st); return SVS_OK;
        SVS_ROWS_CASE(1) SVS_ROWS_CASE(2) SVS_ROWS_CASE(3) SVS_ROWS_CASE(4) SVS_ROWS_CASE(5) SVS_ROWS_CASE(6)
        SVS_ROWS_CASE(7) SVS_ROWS_CASE(8)
#undef SVS_ROWS_CASE
        default: break;
      }
    }
    int rc = stage_queries_f16(idx, c, q, 1, 1, st);
    if (rc != SVS_OK) return rc;
    const int ld8 = idx->ld / 8;
    if (variant != 4 && launch_unrolled(idx, c->qh, ld8, scores, st, DotF16{})) return SVS_OK;
    if (ld8 <= 1) launch_generic_f16<1>(idx, c->qh, scores, st);
    else if (ld8 <= 2) launch_generic_f16<2>(idx, c->qh, scores, st);
    else if (ld8 <= 4) launch_generic_f16<4>(idx, c->qh, scores, st);
    else if (ld8 <= 8) launch_generic_f16<8>(idx, c->qh, scores, st);
    else if (ld8 <= 16) launch_generic_f16<16>(idx, c->qh, scores, st);
    else if (ld8 <= 32) launch_generic_f16<32>(idx, c->qh, scores, st);
    else launch_generic_f16<64>(idx, c->qh, scores, st);
    return SVS_OK;
  }
  if (idx->ld % 256 == 0 && idx->ld <= 4096) {
    // rows padded beyond d (choose_ld): the kernel reads ld query floats, so pad the query too
    const float* qq = q;
    if (idx->ld != idx->d || !q_aligned) {
      int rc = grow_dev(&c->q16, &c->q16_cap, (size_t)GQ * idx->ld);
      if (rc != SVS_OK) return rc;
      HIP_TRY(hipMemsetAsync(c->q16, 0, (size_t)idx->ld * sizeof(float), st));
      HIP_TRY(hipMemcpyAsync(c->q16, q, (size_t)idx->d * sizeof(float), hipMemcpyDeviceToDevice, st));
      qq = c->q16;
    }
    switch (idx->ld / 256) {
#define SVS_ROWS_CASE(N) case N: launch_rows<N>(idx, qq, scores, st, variant); return SVS_OK;
      SVS_ROWS_CASE(1) SVS_ROWS_CASE(2) SVS_ROWS_CASE(3) SVS_ROWS_CASE(4) SVS_ROWS_CASE(5) SVS_ROWS_CASE(6)
      SVS_ROWS_CASE(7) SVS_ROWS_CASE(8) SVS_ROWS_CASE(9) SVS_ROWS_CASE(10) SVS_ROWS_CASE(11) SVS_ROWS_CASE(12)
      SVS_ROWS_CASE(13) SVS_ROWS_CASE(14) SVS_ROWS_CASE(15) SVS_ROWS_CASE(16)
#undef SVS_ROWS_CASE
      default: break;
    }
  }
  const int ld4 = idx->ld / 4;
  if (variant != 4 && ld4 <= 1024) {
    // rows of up to 16 KiB that are not whole wave loads (gemv_unrolled.h): the query is read in
    // 16-byte chunks of the padded row, so it is padded (and aligned) the same way
    const float* qq = q;
    if (idx->ld != idx->d || !q_aligned) {
      int rc = grow_dev(&c->q16, &c->q16_cap, (size_t)GQ * idx->ld);
      if (rc != SVS_OK) return rc;
      HIP_TRY(hipMemsetAsync(c->q16, 0, (size_t)idx->ld * sizeof(float), st));
      HIP_TRY(hipMemcpyAsync(c->q16, q, (size_t)idx->d * sizeof(float), hipMemcpyDeviceToDevice, st));
      qq = c->q16;
    }
    if (launch_unrolled(idx, qq, ld4, scores, st, DotF32{})) return SVS_OK;
  }
  if (ld4 <= 1) launch_generic<1>(idx, q, scores, st);
  else if (ld4 <= 2) launch_generic<2>(idx, q, scores, st);
  else if (ld4 <= 4) launch_generic<4>(idx, q, scores, st);
  else if (ld4 <= 8) launch_generic<8>(idx, q, scores, st);
  else if (ld4 <= 16) launch_generic<16>(idx, q, scores, st);
  else if (ld4 <= 32) launch_generic<32>(idx, q, scores, st);
  else launch_generic<64>(idx, q, scores, st);
  return SVS_OK;
}

// ---- up to 16 queries per corpus pass (gemm_q16.h) ---------------------------
size_t elem_bytes(const svs_index* idx) { return idx->dtype == SVS_DTYPE_F32 ? 4 : (idx->dtype == SVS_DTYPE_F16 ? 2 : 1); }

bool batch_kernel_ok(const svs_index* idx) {
  if (idx->variant.load() == 7) return false;
  // the query image (16 x row bytes) must fit the LDS beside the fused candidate list
  if (idx->dtype == SVS_DTYPE_F32) return idx->ld % 128 == 0 && idx->ld <= 2304;
  if (idx->dtype == SVS_DTYPE_F16) return idx->ld % 256 == 0 && idx->ld <= 4608;   // whole pairs of 256-byte steps
  return false;
}

struct FuseLaunch {   // non-null state: fused top-k epilogue, no score matrix
  uint32_t* state = nullptr;
  uint64_t* cand = nullptr;
  const float* thr = nullptr;   // thr[q * thr_stride]: lower bound of query q's k-th best score
  int thr_stride = 0;
  TgPairs pairs{};              // pair mode (svs_index_top_pairs, tiled kernels only): see gemm_tiled.h
  const void* rows = nullptr;   // non-null: the row operand is THIS matrix (the prefix image), not idx->rows
  const float* row_scales = nullptr;   // ... and its fp8 row scales
  FuseLaunch at(int q0) const {
    if (!state) return *this;
    return FuseLaunch{state + (size_t)q0 * SCR_WORDS, cand + (size_t)q0 * CAND_CAP, thr + (size_t)q0 * thr_stride, thr_stride, pairs, rows, row_scales};
  }
  const void* rows_of(const svs_index* idx) const;
  const float* scales_of(const svs_index* idx) const;
};

inline const void* FuseLaunch::rows_of(const svs_index* idx) const { return rows ? rows : idx->rows; }
inline const float* FuseLaunch::scales_of(const svs_index* idx) const { return rows ? row_scales : idx->row_scales; }

// f32 queries as the batched kernels read them: [nq rounded up to `group`][ld], zero padded.
// Returns the caller's buffer itself when it already has that shape.
int stage_queries_f32(const svs_index* idx, Ctx* c, const float* q_dev, int nq, int group, const float** out, hipStream_t st) {
  const int ld = idx->ld;
  const int nq_pad = (nq + group - 1) / group * group;
  if (nq_pad == nq && ld == idx->d && !(((uintptr_t)q_dev) & 15)) { *out = q_dev; return SVS_OK; }
  int rc = grow_dev(&c->q16, &c->q16_cap, (size_t)nq_pad * ld);
  if (rc != SVS_OK) return rc;
  HIP_TRY(hipMemsetAsync(c->q16, 0, (size_t)nq_pad * ld * sizeof(float), st));
  HIP_TRY(hipMemcpy2DAsync(c->q16, (size_t)ld * sizeof(float), q_dev, (size_t)idx->d * sizeof(float),
                           (size_t)idx->d * sizeof(float), (size_t)nq, hipMemcpyDeviceToDevice, st));
  *out = c->q16;
  return SVS_OK;
}

template <class K, class P>
void launch_q16_kernel(K kernel, const svs_index* idx, const P* rows, const P* q16, int ld_units, size_t row_bytes, int nq_g,
                       int64_t n_rows, float* scores, int64_t sstride, int rows_per_block, FuseLaunch fl, hipStream_t st) {
  const unsigned blocks = (unsigned)((n_rows + rows_per_block - 1) / rows_per_block);
  const size_t lds = row_bytes * 16 + (fl.state ? GEMM_FUSE_LDS : 0);   // 16 queries x row bytes
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(GEMM_WAVES * 64), lds, st,
                     rows, q16, scores, n_rows, ld_units, sstride, nq_g, rows_per_block,
                     fl.state, (int)SCR_WORDS, fl.cand, (uint32_t)CAND_CAP, fl.thr, fl.thr_stride);
}

// q16: 16 staged queries ([16][ld] in the corpus dtype: f32, or halves for an f16 corpus); rows [0, n_rows)
int launch_scores_q16(const svs_index* idx, const void* q16, int nq_g, int64_t n_rows, float* scores,
                      int64_t sstride, FuseLaunch fl, hipStream_t st) {
  static std::once_flag once;
  std::call_once(once, [] {
    const int lds = 2304 * 64 + GEMM_FUSE_LDS;
    (void)hipFuncSetAttribute((const void*)gemm_q16r_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_q16r_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_q16r_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_q16r_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_f32_q16_kernel<false, 2, GEMM_PF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_f32_q16_kernel<false, 2, GEMM_PF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  const int variant = idx->variant.load();
  // 1024 rows (6 MB at d = 1536) per workgroup amortise the 96 KiB query staging; short
  // row ranges (the fused path's prefix) take smaller blocks so that every CU gets one.
  int rows_per_block = variant == 4 ? 2048 : (variant == 5 ? 512 : 1024);
  while (rows_per_block > 64 && (n_rows + rows_per_block - 1) / rows_per_block < 512) rows_per_block /= 2;
  const size_t row_bytes = (size_t)idx->ld * elem_bytes(idx);
  const int ld16 = (int)(row_bytes / 16);
  if (idx->dtype == SVS_DTYPE_F16) {
    if (fl.state) launch_q16_kernel(gemm_q16r_kernel<true, 2>, idx, (const v4f*)fl.rows_of(idx), (const v4f*)q16, ld16, row_bytes, nq_g, n_rows, scores, sstride, rows_per_block, fl, st);
    else launch_q16_kernel(gemm_q16r_kernel<false, 2>, idx, (const v4f*)fl.rows_of(idx), (const v4f*)q16, ld16, row_bytes, nq_g, n_rows, scores, sstride, rows_per_block, fl, st);
  } else if (variant == 3) {   // A/B: the 16x16x4 kernel (half-line loads)
    if (fl.state) launch_q16_kernel(gemm_f32_q16_kernel<false, 2, GEMM_PF, true>, idx, (const float*)fl.rows_of(idx), (const float*)q16, idx->ld, row_bytes, nq_g, n_rows, scores, sstride, rows_per_block, fl, st);
    else launch_q16_kernel(gemm_f32_q16_kernel<false, 2, GEMM_PF, false>, idx, (const float*)fl.rows_of(idx), (const float*)q16, idx->ld, row_bytes, nq_g, n_rows, scores, sstride, rows_per_block, fl, st);
  } else {
    if (fl.state) launch_q16_kernel(gemm_q16r_kernel<true, 4>, idx, (const v4f*)fl.rows_of(idx), (const v4f*)q16, ld16, row_bytes, nq_g, n_rows, scores, sstride, rows_per_block, fl, st);
    else launch_q16_kernel(gemm_q16r_kernel<false, 4>, idx, (const v4f*)fl.rows_of(idx), (const v4f*)q16, ld16, row_bytes, nq_g, n_rows, scores, sstride, rows_per_block, fl, st);
  }
  return SVS_OK;
}

// ---- LDS-tiled MFMA GEMM, f16 / fp8 corpus (gemm_tiled.h) ----------------------
bool tiled_ok(const svs_index* idx) {
  if (idx->variant.load() == 7) return false;
  if (idx->dtype == SVS_DTYPE_F32) return (idx->ld * 4) % TG_BKB == 0 && idx->variant.load() != 5;
  if (idx->dtype == SVS_DTYPE_F16) return (idx->ld * 2) % TG_BKB == 0;
  if (idx->dtype == SVS_DTYPE_FP8) return idx->ld % TG_BKB == 0;
  return false;
}

template <int BN, bool FUSE, int EB, int BM = TG_BM>
int launch_tiled_bn(const svs_index* idx, Ctx* c, int64_t n_rows, int nq, float* scores, int64_t sstride,
                    FuseLaunch fl, hipStream_t st) {
  static std::once_flag once;
  const size_t lds = (size_t)tg_lds_bytes(BM, BN);
  std::call_once(once, [] {
    (void)hipFuncSetAttribute((const void*)gemm_tiled_kernel<BN, FUSE, EB, BM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              tg_lds_bytes(BM, BN));
  });
  const unsigned gx = (unsigned)((n_rows + BM - 1) / BM), gy = (unsigned)((nq + BN - 1) / BN);
  const uint8_t* Q = EB == 2 ? (const uint8_t*)c->qh : (EB == 1 ? (const uint8_t*)c->q8 : (const uint8_t*)c->q_f32);
  // pair mode: the row operand starts at global row fl.pairs.row_base (n_rows counts from there)
  const int64_t rb = fl.pairs.on ? fl.pairs.row_base : 0;
  hipLaunchKernelGGL((gemm_tiled_kernel<BN, FUSE, EB, BM>), dim3(gx, gy), dim3(TG_WAVES * 64), lds, st,
                     (const uint8_t*)fl.rows_of(idx) + (size_t)rb * idx->ld * EB, Q, scores, n_rows, (int64_t)idx->ld * EB, sstride, nq,
                     fl.state, (int)SCR_WORDS, fl.cand, (uint32_t)CAND_CAP, fl.thr, fl.thr_stride,
                     (const float*)(fl.scales_of(idx) ? fl.scales_of(idx) + rb : nullptr), (const float*)c->q8s, fl.pairs);
  return SVS_OK;
}

// 256 x 256 output tiles, persistent workgroups, four-phase k-tiles (gemm_phased.h)
template <bool FUSE, int EB, int EXP = 0, int QT = PG_TILE>
int launch_phased(const svs_index* idx, Ctx* c, int64_t n_rows, int nq, float* scores, int64_t sstride,
                  FuseLaunch fl, hipStream_t st) {
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute((const void*)gemm_phased_kernel<FUSE, EB, EXP, QT>, hipFuncAttributeMaxDynamicSharedMemorySize, PG_LDS_TOTAL);
  });
  const int gx = (int)((n_rows + PG_TILE - 1) / PG_TILE), gy = (nq + QT - 1) / QT;
  const int64_t total = (int64_t)gx * gy;
  const unsigned grid = (unsigned)std::min<int64_t>(total, idx->cu_count);   // one persistent workgroup per CU
  const uint8_t* Q = EB == 2 ? (const uint8_t*)c->qh : (const uint8_t*)c->q8;
  // pair mode: the row operand starts at global row fl.pairs.row_base (n_rows counts from there)
  const int64_t rb = fl.pairs.on ? fl.pairs.row_base : 0;
  hipLaunchKernelGGL((gemm_phased_kernel<FUSE, EB, EXP, QT>), dim3(grid), dim3(PG_THREADS), PG_LDS_TOTAL, st,
                     (const uint8_t*)fl.rows_of(idx) + (size_t)rb * idx->ld * EB, Q, scores, n_rows, (int)(idx->ld * EB), sstride, nq, gx, gy,
                     fl.state, (int)SCR_WORDS, fl.cand, (uint32_t)CAND_CAP, fl.thr, fl.thr_stride,
                     (const float*)(fl.scales_of(idx) ? fl.scales_of(idx) + rb : nullptr), (const float*)c->q8s, fl.pairs);
  return SVS_OK;
}

// the phased kernel's preconditions: an even number (>= 6) of 128-byte k-tiles per row, tile bytes
// and tile counts inside 32-bit descriptors / ints
bool phased_ok(const svs_index* idx, int64_t n_rows, int nq) {
  const int64_t ldb = (int64_t)idx->ld * (int64_t)elem_bytes(idx);
  return idx->variant.load() != 2 && (idx->dtype == SVS_DTYPE_F16 || idx->dtype == SVS_DTYPE_FP8) && ldb % (2 * TG_BKB) == 0 && ldb >= PG_MIN_KT * TG_BKB && ldb <= (1 << 20) &&
         ((n_rows + PG_TILE - 1) / PG_TILE) * ((nq + PG_TILE - 1) / PG_TILE) < (1ll << 30);
}

template <int EB>
int launch_tiled_eb(const svs_index* idx, Ctx* c, int64_t n_rows, int nq, int bn, float* scores, int64_t sstride,
                    FuseLaunch fl, hipStream_t st) {
  const bool f = fl.state != nullptr;
  if constexpr (EB != 4) {
    if (bn == 256 && phased_ok(idx, n_rows, nq)) {
      if (f && idx->variant.load() == 8)   // A/B: LDS-DMA pieces issued with the fragment reads
        return launch_phased<true, EB, 30>(idx, c, n_rows, nq, scores, sstride, fl, st);
      if (f && idx->variant.load() == 9)   // A/B: fused epilogue with a branch per register
        return launch_phased<true, EB, 31>(idx, c, n_rows, nq, scores, sstride, fl, st);
      // One query tile (up to 256 queries): every corpus byte is used by exactly one workgroup, so its LDS-DMA
      // pieces are issued nontemporal (EXP 20) and leave the L2 to the queries (configs[4]: 7.63 vs 7.89-8.0 ms in
      // tools/gemm_phased_bench; with several query tiles the corpus tile is SHARED through the L2: not there).
      if (f && (nq <= PG_TILE || idx->variant.load() == 10) && idx->variant.load() != 9)   // (variant 10: nontemporal with several query tiles too, A/B)
        return launch_phased<true, EB, 20>(idx, c, n_rows, nq, scores, sstride, fl, st);
      return f ? launch_phased<true, EB>(idx, c, n_rows, nq, scores, sstride, fl, st)
               : launch_phased<false, EB>(idx, c, n_rows, nq, scores, sstride, fl, st);
    }
  }
  if constexpr (EB != 4) {
    // 65 .. 128 queries (one query tile: the batches a coalescer forms on a reduced-precision index): the phased
    // kernel at 128-query tiles, corpus pieces nontemporal -- HBM-bound, every corpus byte used once
    if (bn == 128 && !fl.pairs.on && idx->variant.load() != 4 && phased_ok(idx, n_rows, nq))
      return f ? launch_phased<true, EB, 20, 128>(idx, c, n_rows, nq, scores, sstride, fl, st)
               : launch_phased<false, EB, 0, 128>(idx, c, n_rows, nq, scores, sstride, fl, st);
  }
  switch (bn) {
    case 32: return f ? launch_tiled_bn<32, true, EB>(idx, c, n_rows, nq, scores, sstride, fl, st) : launch_tiled_bn<32, false, EB>(idx, c, n_rows, nq, scores, sstride, fl, st);
    case 64:   // (256-row tiles measured 2-5 % slower here)
      return f ? launch_tiled_bn<64, true, EB>(idx, c, n_rows, nq, scores, sstride, fl, st) : launch_tiled_bn<64, false, EB>(idx, c, n_rows, nq, scores, sstride, fl, st);
    case 128:
      if (idx->variant.load() == 4)   // A/B: 128-row tiles (0.87 vs 0.76 ms at 1M x 1536 f16)
        return f ? launch_tiled_bn<128, true, EB>(idx, c, n_rows, nq, scores, sstride, fl, st) : launch_tiled_bn<128, false, EB>(idx, c, n_rows, nq, scores, sstride, fl, st);
      return f ? launch_tiled_bn<128, true, EB, 256>(idx, c, n_rows, nq, scores, sstride, fl, st) : launch_tiled_bn<128, false, EB, 256>(idx, c, n_rows, nq, scores, sstride, fl, st);
    default:
      if (idx->variant.load() == 4)   // A/B: 128-row tiles, three-stage ring
        return f ? launch_tiled_bn<256, true, EB>(idx, c, n_rows, nq, scores, sstride, fl, st) : launch_tiled_bn<256, false, EB>(idx, c, n_rows, nq, scores, sstride, fl, st);
      return f ? launch_tiled_bn<256, true, EB, 256>(idx, c, n_rows, nq, scores, sstride, fl, st)
               : launch_tiled_bn<256, false, EB, 256>(idx, c, n_rows, nq, scores, sstride, fl, st);
  }
}

// rows [0, n_rows) of the corpus; restage == false reuses the quantised queries already staged in the context
int launch_scores_tiled(const svs_index* idx, Ctx* c, const float* q_dev, int64_t n_rows, int nq, float* scores,
                        int64_t sstride, FuseLaunch fl, hipStream_t st, bool restage = true) {
  if (idx->dtype == SVS_DTYPE_F32) {
    // exact-f32 MFMA runs at the vector rate: 32 queries per pass keep the kernel HBM-bound
    // (62 % of the matrix pipe); larger batches are more query tiles of the same launch (grid y).
    // 64 queries per tile from 33 queries up (256 queries: 6.7 vs 7.3 ms with 32; 128-query tiles
    // measured the same as 64: ~125 TFLOP/s of the 157 TF f32 MFMA peak)
    const bool wide = nq > 32 && idx->variant.load() != 4;   // (variant 4: 32-query tiles, A/B)
    if (restage) {
      const float* qs = nullptr;
      int rc = stage_queries_f32(idx, c, q_dev, nq, wide ? 64 : 32, &qs, st);
      if (rc != SVS_OK) return rc;
      c->q_f32 = qs;
    }
    if (wide)
      return fl.state ? launch_tiled_bn<64, true, 4>(idx, c, n_rows, nq, scores, sstride, fl, st)
                      : launch_tiled_bn<64, false, 4>(idx, c, n_rows, nq, scores, sstride, fl, st);
    return fl.state ? launch_tiled_bn<32, true, 4>(idx, c, n_rows, nq, scores, sstride, fl, st)
                    : launch_tiled_bn<32, false, 4>(idx, c, n_rows, nq, scores, sstride, fl, st);
  }
  const int bn = nq <= 32 ? 32 : (nq <= 64 ? 64 : (nq <= 128 ? 128 : 256));
  const int nq_pad = (nq + bn - 1) / bn * bn;
  if (restage) {
    int rc = idx->dtype == SVS_DTYPE_F16 ? stage_queries_f16(idx, c, q_dev, nq, nq_pad, st)
                                         : stage_queries_fp8(idx, c, q_dev, nq, nq_pad, false, st);
    if (rc != SVS_OK) return rc;
  }
  return idx->dtype == SVS_DTYPE_F16 ? launch_tiled_eb<2>(idx, c, n_rows, nq, bn, scores, sstride, fl, st)
                                     : launch_tiled_eb<1>(idx, c, n_rows, nq, bn, scores, sstride, fl, st);
}

// Top-k stage over a materialised score matrix scores[nq][sstride] with n_eff rows.
int run_select(svs_index* idx, Ctx* c, const float* scores, int64_t n_eff, int64_t sstride, int nq, int k,
               int count, float* out_s, int64_t* out_r, hipStream_t st, int64_t row_offset) {
  int rc;
  uint32_t* hist = c->hist;
  uint64_t* cand = c->cand;
  if (n_eff <= SORT_CAP) {
    hipLaunchKernelGGL(select_final_kernel, dim3(nq), dim3(FINAL_THREADS), 0, st, scores, n_eff, sstride, k, count, 1,
                       (uint32_t*)nullptr, (uint64_t*)nullptr, row_offset, out_s, out_r, (const uint32_t*)nullptr);
  } else if (count <= SEL_KMAX) {
    const int64_t per_block = (int64_t)FA_THREADS * SEL_VPT * 4;
    const unsigned blocks = (unsigned)((n_eff + per_block - 1) / per_block);
    hipLaunchKernelGGL(select_window_hist_kernel, dim3(blocks, nq), dim3(FA_THREADS), 0, st, scores, n_eff, sstride, hist);
    hipLaunchKernelGGL(select_window_filter_kernel, dim3(blocks, nq), dim3(FA_THREADS), 0, st, scores, n_eff, sstride,
                       (uint32_t)count, hist, cand);
    hipLaunchKernelGGL(select_final_kernel, dim3(nq), dim3(FINAL_THREADS), 0, st, scores, n_eff, sstride, k, count, 0,
                       hist, cand, row_offset, out_s, out_r, (const uint32_t*)nullptr);
  } else {
    int64_t npad;
    next_pow2_i64(n_eff, &npad);
    if ((rc = grow_dev(&c->keys, &c->keys_cap, (size_t)nq * (size_t)npad)) != SVS_OK) return rc;
    int gb = (int)std::min<int64_t>((npad + 255) / 256, 4096);
    hipLaunchKernelGGL(keys_build_kernel, dim3(gb, nq), dim3(256), 0, st, scores, n_eff, sstride, npad, c->keys);
    const int64_t chunk = std::min<int64_t>(npad, SORT_CAP);
    hipLaunchKernelGGL(bitonic_local_kernel, dim3((unsigned)(npad / chunk), nq), dim3(SORT_THREADS), 0, st, c->keys, npad, 0, 1);
    for (int64_t size = 2 * (int64_t)SORT_CAP; size <= npad; size <<= 1) {
      for (int64_t stride = size >> 1; stride >= SORT_CAP; stride >>= 1) {
        int g2 = (int)std::min<int64_t>(((npad >> 1) + 255) / 256, 8192);
        hipLaunchKernelGGL(bitonic_global_kernel, dim3(g2, nq), dim3(256), 0, st, c->keys, npad, size, stride);
      }
      hipLaunchKernelGGL(bitonic_local_kernel, dim3((unsigned)(npad / chunk), nq), dim3(SORT_THREADS), 0, st, c->keys, npad, size, 0);
    }
    int ge = std::min((k + 255) / 256, 1024);
    hipLaunchKernelGGL(keys_emit_kernel, dim3(ge, nq), dim3(256), 0, st, c->keys, npad, k, count,
                       row_offset, out_s, out_r);
  }
  return SVS_OK;
}

// rows whose exact k-th best seeds the fused epilogue's thresholds: about k * n / prefix
// candidates per query survive, so the prefix grows with n (n/64 -> ~64 k survivors)
// Materialised scores of nq queries: scores[q][sstride] (the non-fused score stage).
bool uses_q16(const svs_index* idx, int nq) {
  if (nq < 2 || !batch_kernel_ok(idx)) return false;
  if (idx->dtype == SVS_DTYPE_F16) return nq <= GQ && idx->variant.load() != 5;   // (variant 5: tiled kernel, A/B)
  return nq <= GQ || idx->variant.load() == 5 || !tiled_ok(idx);
}

// Rows of the staged (half / e4m3) query image the batched kernels read for nq queries: the batch padded to the
// query tile of the kernel that will take it (launch_scores_any / launch_scores_tiled use the same rule).
int staged_rows(const svs_index* idx, int nq) {
  if (uses_q16(idx, nq)) return (nq + GQ - 1) / GQ * GQ;
  const int bn = nq <= 32 ? 32 : (nq <= 64 ? 64 : (nq <= 128 ? 128 : 256));
  return (nq + bn - 1) / bn * bn;
}

// rows [0, n_rows); fl.state != null: fused epilogue (batched kernels only); restage == false
// reuses the queries staged by the previous call on this context.
int launch_scores_any(svs_index* idx, Ctx* c, const float* q_dev, int64_t n_rows, int nq, float* scores, int64_t sstride,
                      FuseLaunch fl, hipStream_t st, bool restage = true) {
  int rc;
  // f32: up to 16 queries -> the 16-query streaming kernel; more -> the tiled kernel at 32
  // queries per pass (bound by the f32 MFMA rate)
  if (uses_q16(idx, nq)) {
    const bool f16 = idx->dtype == SVS_DTYPE_F16;
    if (restage) {
      if (f16) {
        if ((rc = stage_queries_f16(idx, c, q_dev, nq, (nq + GQ - 1) / GQ * GQ, st)) != SVS_OK) return rc;
      } else {
        const float* qs = nullptr;
        if ((rc = stage_queries_f32(idx, c, q_dev, nq, GQ, &qs, st)) != SVS_OK) return rc;
        c->q_f32 = qs;
      }
    }
    for (int q0 = 0; q0 < nq; q0 += GQ) {
      const void* qg = f16 ? (const void*)((const _Float16*)c->qh + (size_t)q0 * idx->ld) : (const void*)(c->q_f32 + (size_t)q0 * idx->ld);
      rc = launch_scores_q16(idx, qg, std::min(GQ, nq - q0), n_rows,
                             scores ? scores + (size_t)q0 * sstride : nullptr, sstride, fl.at(q0), st);
      if (rc != SVS_OK) return rc;
    }
  } else if (nq >= 2 && tiled_ok(idx)) {
    if ((rc = launch_scores_tiled(idx, c, q_dev, n_rows, nq, scores, sstride, fl, st, restage)) != SVS_OK) return rc;
  } else {
    if (fl.state || n_rows != idx->n) return fail(SVS_ERR_INVALID, "internal: single-query kernels have no fused / prefix form");
    for (int qi = 0; qi < nq; ++qi) {
      rc = launch_scores(idx, c, q_dev + (size_t)qi * idx->d, scores + (size_t)qi * sstride, st);
      if (rc != SVS_OK) return rc;
    }
  }
  return SVS_OK;
}

constexpr int64_t FUSE_PREFIX_MIN = 16384;
constexpr int64_t PFX_BLOCK = 256;   // rows per block of the threshold sample (one row tile of the MFMA kernels)
inline int64_t fuse_prefix_rows(int64_t n) {
  const int64_t p = std::max<int64_t>(FUSE_PREFIX_MIN, n / std::max<int64_t>(g_tune_prefix_div.load(), 1));
  return (p + PFX_BLOCK - 1) / PFX_BLOCK * PFX_BLOCK;
}

// The rows the fused path takes its thresholds from.  Rounds 1-3 used the FIRST n / 64 rows: a corpus whose first rows
// are unlike the rest (sorted by similarity to what is asked, or drifting) then gives thresholds that cut nothing, every
// candidate list overflows and the batch falls back to the materialised path.  Any n_mat rows of the corpus give a valid
// bound (the k-th best of a subset is never above the k-th best of the whole), so the sample is taken in blocks of 256
// rows EVERY n / (n_mat / 256) rows -- one strided device-to-device copy into a contiguous image the batched kernels run
// over unchanged -- and is as good a picture of a sorted corpus as of a shuffled one.  1.6 % more HBM; rebuilt (20 us at
// 1M x 1536 f16) when rows were appended.  Rebuilding frees nothing a kernel may still read: the row count only changes
// under the exclusive geometry lock, i.e. after every fused search has drained (the host entry synchronises).
int prefix_image(svs_index* idx, int64_t n_mat, hipStream_t st) {
  std::lock_guard<std::mutex> lk(idx->pfx_mu);
  if (idx->pfx_n == idx->n && idx->pfx_nmat == n_mat && idx->pfx_src == idx->rows) return SVS_OK;
  const int64_t nblk = n_mat / PFX_BLOCK, stride = idx->n / nblk;   // (stride >= PFX_BLOCK: n_mat <= n)
  const size_t rowb = (size_t)idx->ld * elem_bytes(idx);
  if ((size_t)n_mat > idx->pfx_cap) {
    if (idx->pfx_rows) HIP_TRY(hipFree(idx->pfx_rows));
    if (idx->pfx_scales) HIP_TRY(hipFree(idx->pfx_scales));
    idx->pfx_rows = nullptr; idx->pfx_scales = nullptr; idx->pfx_cap = 0; idx->pfx_n = -1;
    if (hipMalloc(&idx->pfx_rows, (size_t)n_mat * rowb) != hipSuccess)
      return fail(SVS_ERR_NOMEM, "out of HBM for the %lld-row threshold sample", (long long)n_mat);
    if (idx->row_scales && hipMalloc((void**)&idx->pfx_scales, (size_t)n_mat * sizeof(float)) != hipSuccess)
      return fail(SVS_ERR_NOMEM, "out of HBM for the threshold sample's row scales");
    idx->pfx_cap = (size_t)n_mat;
  }
  HIP_TRY(hipMemcpy2DAsync(idx->pfx_rows, (size_t)PFX_BLOCK * rowb, idx->rows, (size_t)stride * rowb, (size_t)PFX_BLOCK * rowb,
                           (size_t)nblk, hipMemcpyDeviceToDevice, st));
  if (idx->row_scales)
    HIP_TRY(hipMemcpy2DAsync(idx->pfx_scales, (size_t)PFX_BLOCK * sizeof(float), idx->row_scales, (size_t)stride * sizeof(float),
                             (size_t)PFX_BLOCK * sizeof(float), (size_t)nblk, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));   // (searches on other streams use the image as soon as the lock is gone)
  idx->pfx_n = idx->n; idx->pfx_nmat = n_mat; idx->pfx_stride = stride; idx->pfx_src = idx->rows;
  return SVS_OK;
}

// One search, enqueued in three steps (the host API sizes scratch and starts the timing events BEFORE it stages
// the queries, so that everything the device does for a call sits between the call's events):
//   plan_search     decisions (materialised / fused, prefix size), scratch, the timing start event
//   enqueue_prefix  fused only: the queries staged in the corpus dtype, their scores over the prefix rows, their
//                   thresholds (exact k-th best of the prefix)
//   enqueue_main    fused: the whole-corpus pass with the threshold epilogue + final select; otherwise the
//                   materialised score stage + top-k stage
// (Round 4 built and measured the prefix step PER QUERY TILE -- tile t's staging, prefix GEMM and thresholds enqueued
//  behind its DMA while tile t + 1 was still being copied: configs[2] 3.26 ms per call against 3.04.  One query tile's
//  prefix GEMM is 64 output tiles, a quarter of the CUs, and takes as long as the whole batch's 256; and every
//  copy -> kernel -> copy alternation on the stream is a hand-over between the DMA engine and the compute queue.)
struct SearchPlan {
  int nq = 0, k = 0, count = 0;
  bool path_a = false, fused = false, kth = false, timed = false;
  bool staged = false;   // the caller has staged the queries in the corpus dtype already (search_host, chunk by chunk)
  int64_t n_mat = 0, sstride = 0;
  EvTriple ev{};
};

int plan_search(svs_index* idx, Ctx* c, int nq, int k, int count, hipStream_t st, bool allow_fused, SearchPlan* p) {
  const int64_t n = idx->n;
  int rc;
  if ((rc = staging_wait(idx)) != SVS_OK) return rc;
  p->nq = nq; p->k = k; p->count = count;
  p->path_a = k > 0 && n > SORT_CAP && count <= SEL_KMAX;
  // Fused top-k epilogue (no score matrix) for the batched kernels; a query whose
  // candidate list overflows comes back marked and is re-run by the caller.  The prefix
  // pass costs ~60 us whatever the batch: measured break-even is 16 queries (f32: 13.2 k vs
  // 12.7 k queries/s at 16, 6.4 k vs 6.5 k at 8; f16 at 32: 49 k vs 41 k; fp8 at 32: 77 k vs 64 k).
  const bool batched = uses_q16(idx, nq) || (nq >= 2 && tiled_ok(idx));
  p->fused = allow_fused && p->path_a && batched && nq >= 16 &&
             n >= 8 * FUSE_PREFIX_MIN && (int64_t)n < ((int64_t)1 << 32) &&
             count <= 256 && idx->variant.load() != 6;
  p->n_mat = p->fused ? fuse_prefix_rows(n) : n;       // rows of the materialised score matrix
  p->sstride = (p->n_mat + 3) & ~(int64_t)3;           // float4-aligned score vectors
  // thresholds: many queries over a short prefix -> one k-th-value kernel (47 vs 62 us at 1024 x 16,384);
  // otherwise the ordinary three-launch top-k, whose kernels spread one query over many workgroups
  // (16 queries: 20 vs 33 us; 256 x 156,250 rows: 119 vs 284 us).
  p->kth = p->fused && nq >= 256 && p->n_mat <= 32768;
  if ((rc = grow_dev(&c->scores, &c->scores_cap, (size_t)nq * (size_t)p->sstride)) != SVS_OK) return rc;
  if (p->path_a && (size_t)nq > c->hist_cap) {
    if (c->hist) HIP_TRY(hipFree(c->hist));
    if (c->cand) HIP_TRY(hipFree(c->cand));
    c->hist = nullptr; c->cand = nullptr; c->hist_cap = 0;
    const size_t scr_bytes = (size_t)nq * SCR_WORDS * sizeof(uint32_t);
    HIP_TRY(hipMalloc((void**)&c->hist, scr_bytes));
    HIP_TRY(hipMalloc((void**)&c->cand, (size_t)nq * CAND_CAP * sizeof(uint64_t)));
    // zeroed once; select_final_kernel leaves it zeroed after every search
    HIP_TRY(hipMemsetAsync(c->hist, 0, scr_bytes, st));
    c->hist_cap = nq;
  }
  if (p->fused) {
    if ((rc = grow_dev(&c->pref_s, &c->pref_s_cap, (size_t)nq * (p->kth ? 1 : count))) != SVS_OK) return rc;
    if (!p->kth && (rc = grow_dev(&c->pref_r, &c->pref_r_cap, (size_t)nq * count)) != SVS_OK) return rc;
  }
  const int tevery = idx->timing.load();
  p->timed = tevery > 0 && (idx->timing_seq.fetch_add(1) % (uint32_t)tevery) == 0;
  if (p->timed) {
    HIP_TRY(hipEventCreate(&p->ev.e0));
    HIP_TRY(hipEventCreate(&p->ev.e1));
    HIP_TRY(hipEventCreate(&p->ev.e2));
    HIP_TRY(hipEventRecord(p->ev.e0, st));
  }
  return SVS_OK;
}

int enqueue_prefix(svs_index* idx, Ctx* c, const SearchPlan& p, const float* q_dev, hipStream_t st) {
  if (!p.fused) return SVS_OK;
  int rc;
  const int nq = p.nq, count = p.count;
  FuseLaunch sample{};
  int64_t blk_stride = 0;
  if (g_tune_spread.load()) {
    if ((rc = prefix_image(idx, p.n_mat, st)) != SVS_OK) return rc;
    sample.rows = idx->pfx_rows;
    sample.row_scales = idx->pfx_scales;
    blk_stride = idx->pfx_stride;
  }
  if ((rc = launch_scores_any(idx, c, q_dev, p.n_mat, nq, c->scores, p.sstride, sample, st, !p.staged)) != SVS_OK) return rc;
  if (!idx->dead_list.empty())   // thresholds must come from LIVE rows: masked sample rows -> -inf (rows outside the sample are skipped)
    hipLaunchKernelGGL(mask_dead_rows_kernel, dim3(64), dim3(256), 0, st, c->scores, p.sstride, nq, idx->dead_dev,
                       (int64_t)idx->dead_list.size(), p.n_mat, blk_stride);
  if (p.kth)
    hipLaunchKernelGGL(prefix_kth_kernel, dim3(nq), dim3(FINAL_THREADS), 0, st, (const float*)c->scores, p.n_mat, p.sstride, count, c->pref_s);
  else if ((rc = run_select(idx, c, c->scores, p.n_mat, p.sstride, nq, count, count, c->pref_s, c->pref_r, st, idx->row_offset)) != SVS_OK)
    return rc;
  return SVS_OK;
}

int enqueue_main(svs_index* idx, Ctx* c, SearchPlan& p, const float* q_dev, float* out_s, int64_t* out_r, hipStream_t st) {
  const int64_t n = idx->n;
  const int nq = p.nq, k = p.k, count = p.count;
  int rc;
  EvTriple& ev = p.ev;
  if (p.fused) {
    // the whole corpus, keeping only scores >= threshold (the queries are staged: enqueue_prefix)
    FuseLaunch fl{c->hist, c->cand, p.kth ? c->pref_s : c->pref_s + (count - 1), p.kth ? 1 : count};
    if (p.timed) {
      HIP_TRY(hipEventCreate(&ev.d0));
      HIP_TRY(hipEventCreate(&ev.d1));
      HIP_TRY(hipEventRecord(ev.d0, st));
    }
    if ((rc = launch_scores_any(idx, c, q_dev, n, nq, nullptr, 0, fl, st, false)) != SVS_OK) return rc;
    if (p.timed) HIP_TRY(hipEventRecord(ev.d1, st));
    if (p.timed) HIP_TRY(hipEventRecord(ev.e1, st));
    hipLaunchKernelGGL(select_final_kernel, dim3(nq), dim3(FINAL_THREADS), 0, st, (const float*)nullptr, n, (int64_t)0, k, count, 3,
                       c->hist, c->cand, idx->row_offset, out_s, out_r,
                       (const uint32_t*)(idx->dead_list.empty() ? nullptr : idx->dead_bits_dev));
  } else {
    if ((rc = launch_scores_any(idx, c, q_dev, n, nq, c->scores, p.sstride, FuseLaunch{}, st, !p.staged)) != SVS_OK) return rc;
    if (!idx->dead_list.empty())   // tombstoned rows can never be returned
      hipLaunchKernelGGL(mask_dead_rows_kernel, dim3(64), dim3(256), 0, st, c->scores, p.sstride, nq, idx->dead_dev,
                         (int64_t)idx->dead_list.size(), n, (int64_t)0);
    if (p.timed) HIP_TRY(hipEventRecord(ev.e1, st));
    if (k > 0 && (rc = run_select(idx, c, c->scores, n, p.sstride, nq, k, count, out_s, out_r, st, idx->row_offset)) != SVS_OK) return rc;
  }
  HIP_TRY(hipGetLastError());
  if (p.timed) {
    HIP_TRY(hipEventRecord(ev.e2, st));
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->evs.push_back(ev);
  }
  return SVS_OK;
}

int enqueue_search(svs_index* idx, Ctx* c, const float* q_dev, int nq, int k, int count,
                   float* out_s, int64_t* out_r, hipStream_t st, bool allow_fused = false) {
  SearchPlan p;
  int rc;
  if ((rc = plan_search(idx, c, nq, k, count, st, allow_fused, &p)) != SVS_OK) return rc;
  if ((rc = enqueue_prefix(idx, c, p, q_dev, st)) != SVS_OK) return rc;
  return enqueue_main(idx, c, p, q_dev, out_s, out_r, st);
}

// Holds one reference for the duration of a call.  Declared BEFORE the geometry lock in every
// entry point: locals unwind in reverse order, so the lock is dropped first and only then the
// reference -- if a concurrent svs_index_release() made ours the last one, index_destroy() must
// not run while this call still holds idx->rw.
struct RefGuard {
  svs_index* i;
  explicit RefGuard(svs_index* idx) : i(idx) { i->refs.fetch_add(1); }
  ~RefGuard() { svs_index_release(i); }
  RefGuard(const RefGuard&) = delete;
  RefGuard& operator=(const RefGuard&) = delete;
};

// ---- pairwise (document_top_pairwise_scores, src/svs/kb.py:1642-1671) -----------------------
// rows [r0, r0 + nrows) of the corpus as f32 queries (what the index holds), [nrows][d]
int dequant_rows_to(svs_index* idx, int64_t r0, int64_t nrows, float* out, hipStream_t st) {
  if (idx->dtype == SVS_DTYPE_F32)
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)idx->d * sizeof(float), (const float*)idx->rows + r0 * idx->ld, (size_t)idx->ld * sizeof(float),
                             (size_t)idx->d * sizeof(float), (size_t)nrows, hipMemcpyDeviceToDevice, st));
  else if (idx->dtype == SVS_DTYPE_F16)
    hipLaunchKernelGGL(dequant_rows_f16_kernel, dim3(2048), dim3(256), 0, st, (const _Float16*)idx->rows, r0, nrows, idx->d, idx->ld, out);
  else
    hipLaunchKernelGGL(dequant_rows_fp8_kernel, dim3(2048), dim3(256), 0, st, (const uint8_t*)idx->rows, idx->row_scales, r0, nrows,
                       idx->d, idx->ld, out);
  return SVS_OK;
}

// Top `count` pairs among rows [0, ns) (ns * ns_padded <= 2^32): S = M M^T materialised, strict upper
// triangle, the ordinary top-k stage over the flattened matrix (flat index i * np + j reproduces the
// reference's tie order).  Results (scores, flat indices) stay on the device.
int pairs_block_device(svs_index* idx, Ctx* c, int64_t ns, int count, float* S, float* qbuf, float* d_s, int64_t* d_r, hipStream_t st) {
  const int64_t np = (ns + 3) & ~(int64_t)3;
  int rc;
  const int chunk = 1024;
  for (int64_t q0 = 0; q0 < ns; q0 += chunk) {
    const int nq = (int)std::min<int64_t>(chunk, ns - q0);
    if ((rc = dequant_rows_to(idx, q0, nq, qbuf, st)) != SVS_OK) return rc;
    if ((rc = launch_scores_any(idx, c, qbuf, ns, nq, S + q0 * np, np, FuseLaunch{}, st)) != SVS_OK) return rc;
  }
  hipLaunchKernelGGL(mask_upper_triangle_kernel, dim3(4096), dim3(256), 0, st, S, ns, np);
  if (!idx->dead_list.empty())
    hipLaunchKernelGGL(mask_dead_pairs_kernel, dim3(1024), dim3(256), 0, st, S, ns, np, idx->dead_dev, (int64_t)idx->dead_list.size());
  const int64_t flat = ns * np;
  const bool path_a = flat > SORT_CAP && count <= SEL_KMAX;
  if (path_a && c->hist_cap < 1) {
    HIP_TRY(hipMalloc((void**)&c->hist, (size_t)SCR_WORDS * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&c->cand, (size_t)CAND_CAP * sizeof(uint64_t)));
    HIP_TRY(hipMemsetAsync(c->hist, 0, (size_t)SCR_WORDS * sizeof(uint32_t), st));
    c->hist_cap = 1;
  }
  if ((rc = run_select(idx, c, S, flat, flat, 1, count, count, d_s, d_r, st, /*row_offset=*/0)) != SVS_OK) return rc;
  HIP_TRY(hipGetLastError());
  return SVS_OK;
}

struct DevTmp {   // frees on scope exit
  std::vector<void*> p;
  template <class T> hipError_t alloc(T** out, size_t bytes) {
    hipError_t e = hipMalloc((void**)out, bytes);
    if (e == hipSuccess) p.push_back((void*)*out);
    return e;
  }
  ~DevTmp() { for (void* q : p) (void)hipFree(q); }
};

int top_pairs_materialised(svs_index* idx, Ctx* c, int64_t n, int count, float* out_scores, int64_t* out_i, int64_t* out_j) {
  const int64_t np = (n + 3) & ~(int64_t)3;
  hipStream_t st = c->stream;
  DevTmp tmp;
  float *qbuf = nullptr, *S = nullptr, *d_s = nullptr;
  int64_t* d_r = nullptr;
  HIP_TRY(tmp.alloc(&qbuf, (size_t)1024 * idx->d * sizeof(float)));
  HIP_TRY(tmp.alloc(&S, (size_t)n * np * sizeof(float)));
  HIP_TRY(tmp.alloc(&d_s, (size_t)count * sizeof(float)));
  HIP_TRY(tmp.alloc(&d_r, (size_t)count * sizeof(int64_t)));
  int rc = pairs_block_device(idx, c, n, count, S, qbuf, d_s, d_r, st);
  if (rc != SVS_OK) return rc;
  std::vector<int64_t> flat_rows((size_t)count);
  HIP_TRY(hipMemcpyAsync(out_scores, d_s, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(flat_rows.data(), d_r, (size_t)count * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  for (int t = 0; t < count; ++t) {
    out_i[t] = flat_rows[t] / np + idx->row_offset;
    out_j[t] = flat_rows[t] % np + idx->row_offset;
  }
  return SVS_OK;
}

// Corpora whose n x n scores cannot be materialised (n^2 > 2^32: beyond ~65k rows; the reference
// itself stops at what fits its RAM).  (1) The exact count-th best pair score of a PREFIX block
// (rows [0, P), materialised as above) is a lower bound of the global count-th best.  (2) The
// tiled MFMA GEMM runs over chunks of query rows x the rows above them with the fused epilogue in
// pair mode: scores >= that bound and j > i become per-query candidates, nothing else is written;
// after each chunk they are appended to one global (score, i, j) list.  (3) The host orders the
// list by (score desc, i desc, j desc) -- the reference's flat upper-triangle index, descending --
// and keeps `count`.  Exact; the list holds about count * (n / P)^2 pairs.
int top_pairs_tiled(svs_index* idx, Ctx* c, int count, float* out_scores, int64_t* out_i, int64_t* out_j) {
  const int64_t n = idx->n;
  if (!tiled_ok(idx)) return fail(SVS_ERR_UNSUPPORTED, "pairwise scores over %lld rows need rows of whole 128-byte lines (d = %d)", (long long)n, idx->d);
  hipStream_t st = c->stream;
  int rc;
  // prefix block: at least `count` live pairs inside it
  int64_t P = std::min<int64_t>(n, count > SEL_KMAX ? 8192 : 16384);
  while (P < n && (P - (int64_t)idx->dead_list.size()) * (P - (int64_t)idx->dead_list.size() - 1) / 2 < count) P = std::min<int64_t>(n, P * 2);
  const int64_t Pp = (P + 3) & ~(int64_t)3;
  if (P * Pp > 0xffffffffll) return fail(SVS_ERR_UNSUPPORTED, "top_pairs: k = %d needs a prefix block past 2^32 scores", count);
  constexpr int QC = 1024;                 // query rows per chunk
  constexpr uint32_t LIST_CAP = 8u << 20;  // global pair list (96 MB)
  DevTmp tmp;
  float *qbuf = nullptr, *S = nullptr, *d_s = nullptr;
  int64_t* d_r = nullptr;
  uint32_t *gstate = nullptr, *l_key = nullptr, *l_i = nullptr, *l_j = nullptr;
  HIP_TRY(tmp.alloc(&qbuf, (size_t)QC * idx->d * sizeof(float)));
  HIP_TRY(tmp.alloc(&S, (size_t)P * Pp * sizeof(float)));
  HIP_TRY(tmp.alloc(&d_s, (size_t)count * sizeof(float)));
  HIP_TRY(tmp.alloc(&d_r, (size_t)count * sizeof(int64_t)));
  HIP_TRY(tmp.alloc(&gstate, 16));
  HIP_TRY(tmp.alloc(&l_key, (size_t)LIST_CAP * 4));
  HIP_TRY(tmp.alloc(&l_i, (size_t)LIST_CAP * 4));
  HIP_TRY(tmp.alloc(&l_j, (size_t)LIST_CAP * 4));
  HIP_TRY(hipMemsetAsync(gstate, 0, 16, st));
  if ((rc = pairs_block_device(idx, c, P, count, S, qbuf, d_s, d_r, st)) != SVS_OK) return rc;
  const float* thr = d_s + (count - 1);    // the bound, read by the epilogue with stride 0
  // per-query candidate scratch for one chunk
  if ((size_t)QC > c->hist_cap) {
    if (c->hist) HIP_TRY(hipFree(c->hist));
    if (c->cand) HIP_TRY(hipFree(c->cand));
    c->hist = nullptr; c->cand = nullptr; c->hist_cap = 0;
    HIP_TRY(hipMalloc((void**)&c->hist, (size_t)QC * SCR_WORDS * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&c->cand, (size_t)QC * CAND_CAP * sizeof(uint64_t)));
    HIP_TRY(hipMemsetAsync(c->hist, 0, (size_t)QC * SCR_WORDS * sizeof(uint32_t), st));
    c->hist_cap = QC;
  }
  for (int64_t q0 = 0; q0 < n - 1; q0 += QC) {
    // the last chunk is moved back so that it still holds QC rows (its first rows were done: first_query)
    const int64_t qs = std::max<int64_t>(0, std::min<int64_t>(q0, n - QC));
    const int nq = (int)std::min<int64_t>(QC, n - qs);
    const int64_t row_base = (qs + 1) & ~(int64_t)3;          // rows above the chunk's first query (4-aligned: fp8 row scales)
    if ((rc = dequant_rows_to(idx, qs, nq, qbuf, st)) != SVS_OK) return rc;
    FuseLaunch fl{c->hist, c->cand, thr, 0, TgPairs{(long long)qs, (long long)row_base, (long long)q0, 1}};
    if ((rc = launch_scores_any(idx, c, qbuf, n - row_base, nq, nullptr, 0, fl, st)) != SVS_OK) return rc;
    hipLaunchKernelGGL(collect_pairs_kernel, dim3(nq), dim3(256), 0, st, c->hist, (const uint64_t*)c->cand, (long long)qs, (long long)q0,
                       (const uint32_t*)(idx->dead_list.empty() ? nullptr : idx->dead_bits_dev), gstate, LIST_CAP, l_key, l_i, l_j);
  }
  HIP_TRY(hipGetLastError());
  uint32_t hs[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(hs, gstate, 16, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (hs[1] || hs[0] > LIST_CAP)
    return fail(SVS_ERR_UNSUPPORTED, "top_pairs: more than %u pairs (or 32,768 for one row) score at least the %d-th best of the first %lld rows: "
                                      "near-duplicate documents en masse", LIST_CAP, count, (long long)P);
  const uint32_t L = hs[0];
  if ((int64_t)L < count) return fail(SVS_ERR_DEVICE, "internal: pair list holds %u < %d entries", L, count);
  std::vector<uint32_t> hk(L), hi(L), hj(L);
  HIP_TRY(hipMemcpy(hk.data(), l_key, (size_t)L * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hi.data(), l_i, (size_t)L * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hj.data(), l_j, (size_t)L * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> order(L);
  for (uint32_t t = 0; t < L; ++t) order[t] = t;
  auto before = [&](uint32_t a, uint32_t b) {   // (score desc, i desc, j desc)
    if (hk[a] != hk[b]) return hk[a] > hk[b];
    if (hi[a] != hi[b]) return hi[a] > hi[b];
    return hj[a] > hj[b];
  };
  std::partial_sort(order.begin(), order.begin() + count, order.end(), before);
  for (int t = 0; t < count; ++t) {
    out_scores[t] = key_score(hk[order[t]]);
    out_i[t] = (int64_t)hi[order[t]] + idx->row_offset;
    out_j[t] = (int64_t)hj[order[t]] + idx->row_offset;
  }
  return SVS_OK;
}

int check_query_args(const svs_index* idx, const void* q, int nq, int d) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  if (nq < 0) return fail(SVS_ERR_INVALID, "nq must be >= 0");
  if (d != idx->d || idx->n == 0 || idx->d == 0)
    return fail(SVS_ERR_SHAPE, "shapes (%lld,%d) and (%d,) not aligned: %d (dim 1) != %d (dim 0)",
                (long long)idx->n, idx->d, d, idx->d, d);
  if (nq > 0 && !q) return fail(SVS_ERR_INVALID, "null queries");
  return SVS_OK;
}

// Host rows [0, nrows) (f32, C-contiguous, d floats each) -> HBM rows [row0, row0+nrows).
// Pinned double-buffered staging: host memcpy of chunk i+1 overlaps the DMA of chunk i.
// f32 corpus: the DMA writes the padded HBM layout directly (2D copy).
// f16 / fp8 corpus: the DMA lands in a device staging buffer and a kernel converts it.
hipError_t upload_host_rows(svs_index* idx, const float* host_rows, int64_t nrows, int64_t row0) {
  const int d = idx->d;
  const int64_t n = nrows;
  const bool f16 = idx->dtype != SVS_DTYPE_F32;   // f16 and fp8: convert on the device
  const size_t row_b = (size_t)d * sizeof(float);
  const size_t esz = idx->dtype == SVS_DTYPE_F32 ? 4 : (idx->dtype == SVS_DTYPE_F16 ? 2 : 1);
  const size_t chunk_rows = std::max<size_t>(1, std::min<size_t>((32u << 20) / row_b, (size_t)n));
  void* pin[2] = {nullptr, nullptr};
  float* dstage[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  hipStream_t st = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) {
    e = hipHostMalloc(&pin[i], chunk_rows * row_b, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    if (e == hipSuccess && f16) e = hipMalloc((void**)&dstage[i], chunk_rows * row_b);
  }
  if (e == hipSuccess && !f16 && idx->ld != d)
    e = hipMemsetAsync((char*)idx->rows + (size_t)row0 * idx->ld * esz, 0, (size_t)n * idx->ld * esz, st);
  int b = 0;
  for (size_t r0 = 0; r0 < (size_t)n && e == hipSuccess; r0 += chunk_rows, b ^= 1) {
    const size_t rows = std::min(chunk_rows, (size_t)n - r0);
    const size_t dr = (size_t)row0 + r0;   // destination row
    e = hipEventSynchronize(done[b]);
    if (e != hipSuccess) break;
    memcpy(pin[b], host_rows + r0 * (size_t)d, rows * row_b);
    if (f16) {
      e = hipMemcpyAsync(dstage[b], pin[b], rows * row_b, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) {
        if (idx->dtype == SVS_DTYPE_F16)
          hipLaunchKernelGGL(convert_rows_f16_kernel, dim3(2048), dim3(256), 0, st, (const float*)dstage[b],
                             (int64_t)rows, d, (int64_t)d, (_Float16*)idx->rows + dr * (size_t)idx->ld, idx->ld);
        else
          hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(2048), dim3(256), 0, st, (const float*)dstage[b],
                             (int64_t)rows, d, (int64_t)d, (uint8_t*)idx->rows + dr * (size_t)idx->ld, idx->ld,
                             idx->row_scales + dr, (float*)nullptr);
        e = hipGetLastError();
      }
    } else if (idx->ld == d) {
      e = hipMemcpyAsync((float*)idx->rows + dr * (size_t)d, pin[b], rows * row_b, hipMemcpyHostToDevice, st);
    } else {
      e = hipMemcpy2DAsync((float*)idx->rows + dr * (size_t)idx->ld, (size_t)idx->ld * sizeof(float), pin[b], row_b,
                           row_b, rows, hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess) e = hipEventRecord(done[b], st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  for (int i = 0; i < 2; ++i) {
    if (done[i]) (void)hipEventDestroy(done[i]);
    if (pin[i]) (void)hipHostFree(pin[i]);
    if (dstage[i]) (void)hipFree(dstage[i]);
  }
  if (st) (void)hipStreamDestroy(st);
  return e;
}


// Row stride in elements.  Rows are always 16-byte aligned (4 floats / 8 halves / 16 fp8).
//  1. A whole number of 1 KiB wave loads when that costs at most an eighth more bytes (zero
//     columns): the exact-geometry streaming kernels, ~7 TB/s (d = 1000 -> 1024: +2.4 %).
//  2. Otherwise whole 128-byte lines per row, again for at most an eighth more: rows then never
//     share a cache line (the streams are nontemporal), the tiled kernels' 128-byte k-steps
//     apply, and gemv_unrolled.h streams such rows at 6.6-7.1 TB/s (d = 384 f16: 768-byte rows;
//     the earlier rule padded them to 1 KiB, +33 % bytes).
//  3. Otherwise tight.
int choose_ld(int d, int dtype) {
  const int align = dtype == SVS_DTYPE_F32 ? 4 : (dtype == SVS_DTYPE_F16 ? 8 : 16);   // elements per 16 bytes
  const int tight = (d + align - 1) / align * align;
  if (d <= 0) return tight;
  const int wave = 64 * align, line = 8 * align;
  const int waved = (d + wave - 1) / wave * wave, lined = (d + line - 1) / line * line;
  if ((int64_t)waved * 8 <= (int64_t)tight * 9) return waved;
  if ((int64_t)lined * 8 <= (int64_t)tight * 9) return lined;
  return tight;
}

// Device rows [0, nrows) (f32, stride src_ld) -> HBM rows [row0, row0 + nrows) of the index's layout.
hipError_t copy_device_rows(svs_index* idx, const float* dev_rows, int64_t nrows, int64_t src_ld, int64_t row0) {
  const int d = idx->d;
  hipError_t e = hipSuccess;
  if (idx->dtype == SVS_DTYPE_F16) {
    hipLaunchKernelGGL(convert_rows_f16_kernel, dim3(4096), dim3(256), 0, 0, dev_rows, nrows, d, src_ld,
                       (_Float16*)idx->rows + (size_t)row0 * idx->ld, idx->ld);
    e = hipGetLastError();
  } else if (idx->dtype == SVS_DTYPE_FP8) {
    hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(4096), dim3(256), 0, 0, dev_rows, nrows, d, src_ld,
                       (uint8_t*)idx->rows + (size_t)row0 * idx->ld, idx->ld, idx->row_scales + row0, (float*)nullptr);
    e = hipGetLastError();
  } else {
    float* dst = (float*)idx->rows + (size_t)row0 * idx->ld;
    if (idx->ld != d) e = hipMemset(dst, 0, (size_t)nrows * idx->ld * sizeof(float));
    if (e == hipSuccess)
      e = hipMemcpy2D(dst, (size_t)idx->ld * sizeof(float), dev_rows, (size_t)src_ld * sizeof(float),
                      (size_t)d * sizeof(float), (size_t)nrows, hipMemcpyDeviceToDevice);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  return e;
}

// Device bitmap of the masked rows, one bit per row of the current capacity (caller holds the
// geometry lock exclusively).  Nothing is allocated while no row is masked.
int sync_dead_bits(svs_index* idx) {
  if (idx->dead_list.empty()) return SVS_OK;
  const size_t words = (size_t)((std::max(idx->cap, idx->n) + 31) / 32);
  idx->dead_bits.assign(words, 0u);
  for (uint32_t r : idx->dead_list) idx->dead_bits[r >> 5] |= 1u << (r & 31);
  if (words > idx->dead_bits_cap) {
    HIP_TRY(hipDeviceSynchronize());   // enqueued searches may still read the old bitmap
    (void)hipFree(idx->dead_bits_dev);
    idx->dead_bits_dev = nullptr;
    idx->dead_bits_cap = 0;
    HIP_TRY(hipMalloc((void**)&idx->dead_bits_dev, words * sizeof(uint32_t)));
    idx->dead_bits_cap = words;
  }
  HIP_TRY(hipMemcpy(idx->dead_bits_dev, idx->dead_bits.data(), words * sizeof(uint32_t), hipMemcpyHostToDevice));
  return SVS_OK;
}

// Makes room for `rows` rows (caller holds the geometry lock exclusively).  exact == false grows
// by 1.5x (amortised appends); exact == true (svs_index_reserve) allocates just `rows`.
int ensure_capacity(svs_index* idx, int64_t rows, bool exact) {
  if (rows > 0xffffffffll) return fail(SVS_ERR_INVALID, "at most 2^32 rows per handle; shard the corpus");
  if (rows <= idx->cap) return SVS_OK;
  const size_t esz = idx->dtype == SVS_DTYPE_F32 ? 4 : (idx->dtype == SVS_DTYPE_F16 ? 2 : 1), row_b = (size_t)idx->ld * esz;
  const int64_t new_cap = exact ? rows : std::max<int64_t>(rows, idx->cap + idx->cap / 2 + 1024);
  void* nrows = nullptr;
  float* nscales = nullptr;
  HIP_TRY(hipMalloc(&nrows, (size_t)new_cap * row_b));
  if (idx->dtype == SVS_DTYPE_FP8) {
    hipError_t e = hipMalloc((void**)&nscales, (size_t)new_cap * sizeof(float));
    if (e != hipSuccess) {
      (void)hipFree(nrows);
      return fail(SVS_ERR_NOMEM, "hipMalloc(row scales): %s", hipGetErrorString(e));
    }
  }
  // (the new buffers are the caller's problem only once they are installed: every failure until then frees them)
  struct Fresh {
    void* rows; float* scales; bool keep = false;
    ~Fresh() { if (!keep) { (void)hipFree(rows); (void)hipFree(scales); } }
  } fresh{nrows, nscales};
  // work already enqueued by the device API may still read the old buffers
  HIP_TRY(hipDeviceSynchronize());
  const int64_t n_old = idx->n;
  if (n_old) HIP_TRY(hipMemcpy(nrows, idx->rows, (size_t)n_old * row_b, hipMemcpyDeviceToDevice));
  if (n_old && nscales) HIP_TRY(hipMemcpy(nscales, idx->row_scales, (size_t)n_old * sizeof(float), hipMemcpyDeviceToDevice));
  fresh.keep = true;
  (void)hipFree(idx->rows);
  (void)hipFree(idx->row_scales);
  idx->rows = nrows;
  idx->row_scales = nscales;
  idx->cap = new_cap;
  idx->bytes = (size_t)new_cap * row_b + (idx->dtype == SVS_DTYPE_FP8 ? (size_t)new_cap * sizeof(float) : 0);
  return SVS_OK;
}

int create_common(int64_t n, int32_t d, int32_t store_dtype, int32_t device, int64_t row_offset,
                  svs_index** out, svs_index** made) {
  if (!out) return fail(SVS_ERR_INVALID, "null out");
  *out = nullptr;
  if (n < 0 || d < 0) return fail(SVS_ERR_INVALID, "negative shape (%lld, %d)", (long long)n, d);
  if (store_dtype != SVS_DTYPE_F32 && store_dtype != SVS_DTYPE_F16 && store_dtype != SVS_DTYPE_FP8)
    return fail(SVS_ERR_UNSUPPORTED, "unknown store dtype %d", store_dtype);
  if (n > 0xffffffffll) return fail(SVS_ERR_INVALID, "at most 2^32 rows per handle; shard the corpus");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SVS_ERR_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(SVS_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  svs_index* idx = new (std::nothrow) svs_index();
  if (!idx) return fail(SVS_ERR_NOMEM, "host allocation failed");
  idx->device = device;
  idx->n = n;
  idx->cap = n;
  idx->d = d;
  idx->dtype = store_dtype;
  idx->ld = choose_ld(d, store_dtype);
  idx->row_offset = row_offset;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) idx->cu_count = prop.multiProcessorCount;
  idx->bytes = (size_t)n * (size_t)idx->ld * (store_dtype == SVS_DTYPE_F16 ? 2 : (store_dtype == SVS_DTYPE_FP8 ? 1 : 4));
  if (idx->bytes) {
    hipError_t e = hipMalloc(&idx->rows, idx->bytes);
    if (e != hipSuccess) {
      delete idx;
      return fail(e == hipErrorOutOfMemory ? SVS_ERR_NOMEM : SVS_ERR_DEVICE, "hipMalloc(%zu): %s", idx->bytes, hipGetErrorString(e));
    }
    if (store_dtype == SVS_DTYPE_FP8) {
      e = hipMalloc((void**)&idx->row_scales, (size_t)n * sizeof(float));
      if (e != hipSuccess) {
        (void)hipFree(idx->rows);
        delete idx;
        return fail(SVS_ERR_NOMEM, "hipMalloc(row scales): %s", hipGetErrorString(e));
      }
      idx->bytes += (size_t)n * sizeof(float);
    }
  }
  idx->dead_flag.assign((size_t)n, 0);
  *made = idx;
  return SVS_OK;
}

}  // namespace

extern "C" {

const char* svs_version(void) { return "svs_amd 0.3.0 (gfx950)"; }
const char* svs_last_error(void) { return g_err.c_str(); }
// (multi.hip: carries a worker thread's message over to the caller's thread; not part of the ABI)
int32_t svs_internal_set_error(int32_t code, const char* msg) { return fail(code, "%s", msg ? msg : ""); }

int32_t svs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int32_t svs_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes) {
  HIP_TRY(hipSetDevice(device));
  size_t f = 0, t = 0;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return SVS_OK;
}

int32_t svs_index_create(const float* host_rows, int64_t n, int32_t d, int32_t store_dtype,
                         int32_t device, int64_t row_offset, svs_index** out) {
  svs_index* idx = nullptr;
  int rc = create_common(n, d, store_dtype, device, row_offset, out, &idx);
  if (rc != SVS_OK) return rc;
  if (idx->bytes) {
    if (!host_rows) {
      index_destroy(idx);
      return fail(SVS_ERR_INVALID, "null host_rows");
    }
    hipError_t e = upload_host_rows(idx, host_rows, n, 0);
    if (e != hipSuccess) {
      index_destroy(idx);
      return fail(SVS_ERR_DEVICE, "corpus upload: %s", hipGetErrorString(e));
    }
  }
  *out = idx;
  return SVS_OK;
}

int32_t svs_index_create_from_device(const float* dev_rows, int64_t n, int32_t d, int64_t src_ld,
                                     int32_t store_dtype, int32_t device, int64_t row_offset,
                                     svs_index** out) {
  svs_index* idx = nullptr;
  int rc = create_common(n, d, store_dtype, device, row_offset, out, &idx);
  if (rc != SVS_OK) return rc;
  if (idx->bytes) {
    if (!dev_rows || src_ld < d) {
      index_destroy(idx);
      return fail(SVS_ERR_INVALID, "bad device source (ptr %p, ld %lld)", (const void*)dev_rows, (long long)src_ld);
    }
    hipError_t e = copy_device_rows(idx, dev_rows, n, src_ld, 0);
    if (e != hipSuccess) {
      index_destroy(idx);
      return fail(SVS_ERR_DEVICE, "device corpus copy: %s", hipGetErrorString(e));
    }
  }
  *out = idx;
  return SVS_OK;
}

int32_t svs_index_append(svs_index* idx, const float* host_rows, int64_t n_new) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  if (n_new < 0) return fail(SVS_ERR_INVALID, "negative row count");
  if (n_new == 0) return SVS_OK;
  if (!host_rows) return fail(SVS_ERR_INVALID, "null host_rows");
  if (idx->d == 0) return fail(SVS_ERR_SHAPE, "cannot append to a zero-dimensional index");
  std::unique_lock<std::shared_mutex> geo(idx->rw);   // no search is enqueuing while the geometry changes
  HIP_TRY(hipSetDevice(idx->device));
  const int64_t n_old = idx->n, n_tot = n_old + n_new;
  int rc = ensure_capacity(idx, n_tot, false);
  if (rc != SVS_OK) return rc;
  hipError_t e = upload_host_rows(idx, host_rows, n_new, n_old);
  if (e != hipSuccess) return fail(SVS_ERR_DEVICE, "append upload: %s", hipGetErrorString(e));
  idx->n = n_tot;
  idx->dead_flag.resize((size_t)n_tot, 0);
  return sync_dead_bits(idx);
}

int32_t svs_index_reserve(svs_index* idx, int64_t rows_capacity) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  if (idx->d == 0) return fail(SVS_ERR_SHAPE, "cannot reserve rows of a zero-dimensional index");
  std::unique_lock<std::shared_mutex> geo(idx->rw);
  HIP_TRY(hipSetDevice(idx->device));
  return ensure_capacity(idx, rows_capacity, true);
}

int32_t svs_index_append_from_device(svs_index* idx, const float* dev_rows, int64_t n_new, int64_t src_ld) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  if (n_new < 0) return fail(SVS_ERR_INVALID, "negative row count");
  if (n_new == 0) return SVS_OK;
  if (idx->d == 0) return fail(SVS_ERR_SHAPE, "cannot append to a zero-dimensional index");
  if (!dev_rows || src_ld < idx->d) return fail(SVS_ERR_INVALID, "bad device source (ptr %p, ld %lld)", (const void*)dev_rows, (long long)src_ld);
  std::unique_lock<std::shared_mutex> geo(idx->rw);
  HIP_TRY(hipSetDevice(idx->device));
  const int64_t n_old = idx->n, n_tot = n_old + n_new;
  int rc = ensure_capacity(idx, n_tot, false);
  if (rc != SVS_OK) return rc;
  hipError_t e = copy_device_rows(idx, dev_rows, n_new, src_ld, n_old);
  if (e != hipSuccess) return fail(SVS_ERR_DEVICE, "device append: %s", hipGetErrorString(e));
  idx->n = n_tot;
  idx->dead_flag.resize((size_t)n_tot, 0);
  return sync_dead_bits(idx);
}

int32_t svs_index_staging_acquire(svs_index* idx, float** host_block, int64_t* rows_cap) {
  if (!idx || !host_block || !rows_cap) return fail(SVS_ERR_INVALID, "null argument");
  if (idx->d == 0) return fail(SVS_ERR_SHAPE, "cannot stage rows of a zero-dimensional index");
  HIP_TRY(hipSetDevice(idx->device));
  std::lock_guard<std::mutex> lk(idx->stg_mu);
  auto& g = idx->stg;
  if (!g.active) {
    const size_t row_b = (size_t)idx->d * sizeof(float);
    g.rows_cap = (int64_t)std::max<size_t>(1, ((size_t)32 << 20) / row_b);
    // (a failure half way through the first-time set-up must not strand the stream and the blocks made so far:
    //  the next acquire would make them again)
    hipError_t e = hipStreamCreateWithFlags(&g.st, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
      e = hipHostMalloc(&g.pin[i], (size_t)g.rows_cap * row_b, hipHostMallocDefault);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&g.done[i], hipEventDisableTiming);
      if (e == hipSuccess && idx->dtype != SVS_DTYPE_F32) e = hipMalloc((void**)&g.dstage[i], (size_t)g.rows_cap * row_b);
    }
    if (e != hipSuccess) {
      staging_free(idx);
      return fail(e == hipErrorOutOfMemory ? SVS_ERR_NOMEM : SVS_ERR_DEVICE, "staging blocks: %s", hipGetErrorString(e));
    }
    g.cur = 1;
    g.active = true;
  }
  g.cur ^= 1;
  HIP_TRY(hipEventSynchronize(g.done[g.cur]));   // the DMA that last read this block (never recorded: returns at once)
  *host_block = (float*)g.pin[g.cur];
  *rows_cap = g.rows_cap;
  return SVS_OK;
}

int32_t svs_index_staging_commit(svs_index* idx, int64_t n_rows) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  if (n_rows < 0) return fail(SVS_ERR_INVALID, "negative row count");
  if (n_rows == 0) return SVS_OK;
  std::unique_lock<std::shared_mutex> geo(idx->rw);
  std::lock_guard<std::mutex> lk(idx->stg_mu);
  auto& g = idx->stg;
  if (!g.active) return fail(SVS_ERR_INVALID, "svs_index_staging_commit without svs_index_staging_acquire");
  if (n_rows > g.rows_cap) return fail(SVS_ERR_INVALID, "%lld rows do not fit the staging block (%lld)", (long long)n_rows, (long long)g.rows_cap);
  HIP_TRY(hipSetDevice(idx->device));
  const int64_t n_old = idx->n, n_tot = n_old + n_rows;
  if (n_tot > idx->cap) {   // growing reallocates: drain our own copies first (ensure_capacity drains the rest)
    HIP_TRY(hipStreamSynchronize(g.st));
    int rc = ensure_capacity(idx, n_tot, false);
    if (rc != SVS_OK) return rc;
  }
  const int d = idx->d, b = g.cur;
  const size_t row_b = (size_t)d * sizeof(float);
  if (idx->dtype == SVS_DTYPE_F32) {
    float* dst = (float*)idx->rows + (size_t)n_old * idx->ld;
    if (idx->ld == d) HIP_TRY(hipMemcpyAsync(dst, g.pin[b], (size_t)n_rows * row_b, hipMemcpyHostToDevice, g.st));
    else {
      HIP_TRY(hipMemsetAsync(dst, 0, (size_t)n_rows * idx->ld * sizeof(float), g.st));
      HIP_TRY(hipMemcpy2DAsync(dst, (size_t)idx->ld * sizeof(float), g.pin[b], row_b, row_b, (size_t)n_rows, hipMemcpyHostToDevice, g.st));
    }
  } else {
    HIP_TRY(hipMemcpyAsync(g.dstage[b], g.pin[b], (size_t)n_rows * row_b, hipMemcpyHostToDevice, g.st));
    if (idx->dtype == SVS_DTYPE_F16)
      hipLaunchKernelGGL(convert_rows_f16_kernel, dim3(2048), dim3(256), 0, g.st, (const float*)g.dstage[b], n_rows, d, (int64_t)d,
                         (_Float16*)idx->rows + (size_t)n_old * idx->ld, idx->ld);
    else
      hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(2048), dim3(256), 0, g.st, (const float*)g.dstage[b], n_rows, d, (int64_t)d,
                         (uint8_t*)idx->rows + (size_t)n_old * idx->ld, idx->ld, idx->row_scales + n_old, (float*)nullptr);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(g.done[b], g.st));
  idx->staging_pending.store(true);
  idx->n = n_tot;
  idx->dead_flag.resize((size_t)n_tot, 0);
  return sync_dead_bits(idx);
}

int32_t svs_index_staging_finish(svs_index* idx) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  HIP_TRY(hipSetDevice(idx->device));
  std::lock_guard<std::mutex> lk(idx->stg_mu);
  if (idx->stg.st) HIP_TRY(hipStreamSynchronize(idx->stg.st));
  staging_free(idx);
  return SVS_OK;
}

int32_t svs_index_mask_rows(svs_index* idx, const int64_t* rows, int64_t count) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  if (count < 0 || (count > 0 && !rows)) return fail(SVS_ERR_INVALID, "bad row list");
  std::unique_lock<std::shared_mutex> geo(idx->rw);
  for (int64_t t = 0; t < count; ++t) {
    const int64_t r = rows[t] - idx->row_offset;
    if (r < 0 || r >= idx->n) return fail(SVS_ERR_INVALID, "row %lld out of range", (long long)rows[t]);
  }
  bool changed = false;
  for (int64_t t = 0; t < count; ++t) {
    const int64_t r = rows[t] - idx->row_offset;
    if (!idx->dead_flag[(size_t)r]) {
      idx->dead_flag[(size_t)r] = 1;
      idx->dead_list.push_back((uint32_t)r);
      changed = true;
    }
  }
  if (!changed) return SVS_OK;
  HIP_TRY(hipSetDevice(idx->device));
  {
    int rc = sync_dead_bits(idx);
    if (rc != SVS_OK) return rc;
  }
  if (idx->dead_list.size() > idx->dead_dev_cap) {
    HIP_TRY(hipDeviceSynchronize());   // enqueued searches may still read the old list
    (void)hipFree(idx->dead_dev);
    idx->dead_dev = nullptr;
    idx->dead_dev_cap = 0;
    const size_t cap = idx->dead_list.size() * 2 + 64;
    HIP_TRY(hipMalloc((void**)&idx->dead_dev, cap * sizeof(uint32_t)));
    idx->dead_dev_cap = cap;
  }
  HIP_TRY(hipMemcpy(idx->dead_dev, idx->dead_list.data(), idx->dead_list.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  return SVS_OK;
}

int32_t svs_index_retain(svs_index* idx) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  idx->refs.fetch_add(1);
  return SVS_OK;
}

int32_t svs_index_release(svs_index* idx) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  if (idx->refs.fetch_sub(1) == 1) index_destroy(idx);
  return SVS_OK;
}

int32_t svs_index_info(const svs_index* idx, svs_index_info_t* out) {
  if (!idx || !out) return fail(SVS_ERR_INVALID, "null argument");
  out->n = idx->n;
  out->d = idx->d;
  out->ld = idx->ld;
  out->dtype = idx->dtype;
  out->device = idx->device;
  out->row_offset = idx->row_offset;
  out->hbm_bytes = (int64_t)idx->bytes;
  out->n_masked = (int64_t)idx->dead_list.size();
  return SVS_OK;
}

static int32_t search_host(svs_index* idx, const float* queries, int32_t nq, int32_t d, int32_t k,
                           float* out_scores, int64_t* out_rows, int32_t* out_count) {
  RefGuard guard(idx);
  std::shared_lock<std::shared_mutex> geo(idx->rw);
  int rc = check_query_args(idx, queries, nq, d);
  if (rc != SVS_OK) return rc;
  const int count = (int)std::min<int64_t>(std::max(k, 0), idx->n - (int64_t)idx->dead_list.size());
  if (out_count) *out_count = count;
  if (nq == 0 || count == 0) return SVS_OK;
  if (!out_scores || !out_rows) return fail(SVS_ERR_INVALID, "null output");
  HIP_TRY(hipSetDevice(idx->device));
  Ctx* c = nullptr;
  if ((rc = ctx_acquire(idx, nullptr, true, &c)) != SVS_OK) return rc;
  struct CtxGuard { svs_index* i; Ctx* c; ~CtxGuard() { ctx_release(i, c); } } cg{idx, c};
  const size_t qn = (size_t)nq * (size_t)d, on = (size_t)nq * (size_t)count;
  if ((rc = grow_dev(&c->q_dev, &c->q_cap, qn)) != SVS_OK) return rc;
  if (qn > c->q_pin_cap) {
    if (c->q_pin) HIP_TRY(hipHostFree(c->q_pin));
    c->q_pin = nullptr; c->q_pin_cap = 0;
    HIP_TRY(hipHostMalloc((void**)&c->q_pin, qn * sizeof(float), hipHostMallocDefault));
    c->q_pin_cap = qn;
  }
  if (on > c->out_pin_cap) {
    if (c->out_s_pin) HIP_TRY(hipHostFree(c->out_s_pin));
    if (c->out_r_pin) HIP_TRY(hipHostFree(c->out_r_pin));
    c->out_s_pin = nullptr; c->out_r_pin = nullptr; c->out_pin_cap = 0;
    HIP_TRY(hipHostMalloc((void**)&c->out_s_pin, on * sizeof(float), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&c->out_r_pin, on * sizeof(int64_t), hipHostMallocDefault));
    c->out_pin_cap = on;
  }
  // The final top-k kernel stores its k results straight into the pinned host
  // buffers (device-visible, zero-copy): no D2H copies on the latency path.
  const auto t_begin = std::chrono::steady_clock::now();
  auto stamp = [&](int i) { g_host_phase[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
  SearchPlan plan;
  if ((rc = plan_search(idx, c, nq, count, count, c->stream, true, &plan)) != SVS_OK) return rc;
  stamp(0);
  const int64_t mode = g_tune_upload.load();
  // f16 batches: the staging kernel (convert_queries_f16) reads the f32 queries STRAIGHT out of the pinned buffer, over PCIe,
  // once, and writes the half image the GEMM kernels use -- no copy-engine transfer, no f32 copy of the queries in HBM, no
  // hand-over between the DMA engine and the compute queue in front of the first kernel -- CHUNK BY CHUNK (<= 1 MiB of whole
  // queries): chunk j is pulled over the bus while the host copies chunk j + 1 into pinned memory.  Measured on configs[2]
  // (6.3 MB of queries; tools/call_breakdown.py): 0.088 ms until everything is enqueued against 0.15 with the DMA calls in
  // between.  f32 indexes keep the staged DMA (their kernels read the f32 queries themselves, many times), and so do fp8
  // ones: quantize_rows_fp8 reads its source twice -- row maximum, then the bytes -- and over the bus that cost configs[4]
  // 0.03-0.05 ms per call.  (Also measured and dropped: helper threads sharing the host copy -- their hand-offs cost what
  // they saved, 0.112 vs 0.088 ms to fill the pinned buffer; and the prefix pass per query tile, see SearchPlan.)
  const bool pull = mode == 0 && idx->dtype == SVS_DTYPE_F16 && (uses_q16(idx, nq) || (nq >= 2 && tiled_ok(idx)));
  const float* q_src = pull ? c->q_pin : c->q_dev;
  if (pull) {
    const int rows_total = staged_rows(idx, nq);            // the image of the WHOLE batch: sized once, before the first chunk
    rc = grow_dev(&c->qh, &c->qh_cap, (size_t)rows_total * idx->ld);
    int cq = 256;                                           // queries per chunk: <= 1 MiB, a power of two
    while (cq > 1 && (size_t)cq * d * sizeof(float) > ((size_t)1 << 20)) cq >>= 1;
    for (int q0 = 0; q0 < nq && rc == SVS_OK; q0 += cq) {
      const int nc = std::min(cq, nq - q0);
      const size_t off = (size_t)q0 * d;
      memcpy(c->q_pin + off, queries + off, (size_t)nc * d * sizeof(float));
      // (rows [q0, q0 + nc) of the image; the last chunk also zeroes the padding rows behind the batch)
      rc = stage_queries_f16(idx, c, c->q_pin + off, nc, q0 + nc == nq ? rows_total : q0 + nc, c->stream, q0);
    }
    plan.staged = true;
  } else {
    // queries -> pinned staging -> HBM, in 1 MiB pieces: the DMA of piece i runs under the host copy of piece i + 1
    for (size_t off = 0; off < qn; off += (size_t)262144) {
      const size_t len = std::min((size_t)262144, qn - off);
      memcpy(c->q_pin + off, queries + off, len * sizeof(float));
      HIP_TRY(hipMemcpyAsync(c->q_dev + off, c->q_pin + off, len * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
  }
  stamp(1);
  if (rc == SVS_OK) rc = enqueue_prefix(idx, c, plan, q_src, c->stream);
  if (rc == SVS_OK) rc = enqueue_main(idx, c, plan, q_src, c->out_s_pin, c->out_r_pin, c->stream);
  stamp(2);
  if (rc != SVS_OK) {
    (void)hipStreamSynchronize(c->stream);
    if (plan.timed && plan.ev.e0) {   // (a failed search keeps no events)
      std::lock_guard<std::mutex> lk(idx->mu);
      bool kept = false;
      for (auto& t : idx->evs) kept = kept || t.e0 == plan.ev.e0;
      if (!kept) {
        (void)hipEventDestroy(plan.ev.e0); (void)hipEventDestroy(plan.ev.e1); (void)hipEventDestroy(plan.ev.e2);
        if (plan.ev.d0) (void)hipEventDestroy(plan.ev.d0);
        if (plan.ev.d1) (void)hipEventDestroy(plan.ev.d1);
      }
    }
    return rc;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  stamp(3);
  // Queries whose fused candidate list overflowed (marked row -2; rows ordered by similarity to the query, so that the
  // prefix's threshold cuts nothing): exact re-run through the MATERIALISED path, up to REDO_BATCH of them per pass.
  // (Rounds 2-3 re-ran them one by one through the single-query kernels: 0.9 ms each at 1M rows, ~1 s for a 1024-query
  //  batch over a corpus sorted that way, against 3 ms normally; in passes of 64 the same call took 55 ms, of 256: see
  //  DESIGN 4.)  A pass's score matrix is kept under 2 GiB: 256 queries at 1M rows, 53 at 10M.
  constexpr int REDO_BATCH = 256;
  const int redo_max = (int)std::min<int64_t>(REDO_BATCH, std::max<int64_t>(1, ((int64_t)2 << 30) / (4 * std::max<int64_t>(idx->n, 1))));
  int n_redo = 0;
  for (int q0 = 0; q0 < nq;) {
    int grp[REDO_BATCH], m = 0;
    for (; q0 < nq && m < redo_max; ++q0)
      if (c->out_r_pin[(size_t)q0 * count] == -2) grp[m++] = q0;
    if (m == 0) break;
    n_redo += m;
    if ((size_t)REDO_BATCH * count > c->redo_pin_cap) {
      if (c->redo_s_pin) HIP_TRY(hipHostFree(c->redo_s_pin));
      if (c->redo_r_pin) HIP_TRY(hipHostFree(c->redo_r_pin));
      c->redo_s_pin = nullptr; c->redo_r_pin = nullptr; c->redo_pin_cap = 0;
      HIP_TRY(hipHostMalloc((void**)&c->redo_s_pin, (size_t)REDO_BATCH * count * sizeof(float), hipHostMallocDefault));
      HIP_TRY(hipHostMalloc((void**)&c->redo_r_pin, (size_t)REDO_BATCH * count * sizeof(int64_t), hipHostMallocDefault));
      c->redo_pin_cap = (size_t)REDO_BATCH * count;
    }
    // (the group's queries go to the front of q_dev in ONE copy -- whatever the main pass kept there is no longer needed,
    //  the pinned buffer still holds every query of the call; a group that is not one run of queries is gathered first)
    const float* src = c->q_pin + (size_t)grp[0] * d;
    if (grp[m - 1] - grp[0] != m - 1) {
      if ((size_t)REDO_BATCH * d > c->redo_q_cap) {
        if (c->redo_q_pin) HIP_TRY(hipHostFree(c->redo_q_pin));
        c->redo_q_pin = nullptr; c->redo_q_cap = 0;
        HIP_TRY(hipHostMalloc((void**)&c->redo_q_pin, (size_t)REDO_BATCH * d * sizeof(float), hipHostMallocDefault));
        c->redo_q_cap = (size_t)REDO_BATCH * d;
      }
      for (int j = 0; j < m; ++j) memcpy(c->redo_q_pin + (size_t)j * d, c->q_pin + (size_t)grp[j] * d, (size_t)d * sizeof(float));
      src = c->redo_q_pin;
    }
    HIP_TRY(hipMemcpyAsync(c->q_dev, src, (size_t)m * d * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if ((rc = enqueue_search(idx, c, c->q_dev, m, count, count, c->redo_s_pin, c->redo_r_pin, c->stream, false)) != SVS_OK) {
      (void)hipStreamSynchronize(c->stream);
      return rc;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int j = 0; j < m; ++j) {
      memcpy(c->out_s_pin + (size_t)grp[j] * count, c->redo_s_pin + (size_t)j * count, (size_t)count * sizeof(float));
      memcpy(c->out_r_pin + (size_t)grp[j] * count, c->redo_r_pin + (size_t)j * count, (size_t)count * sizeof(int64_t));
    }
  }
  g_host_phase[5] = (double)n_redo;   // (svs_internal_host_phases: queries of this call that were re-run)
  // device layout has stride `count`; the caller's has stride k
  if (count == k) {   // (one piece each)
    memcpy(out_scores, c->out_s_pin, on * sizeof(float));
    memcpy(out_rows, c->out_r_pin, on * sizeof(int64_t));
    stamp(4);
    return SVS_OK;
  }
  for (int qi = 0; qi < nq; ++qi) {
    memcpy(out_scores + (size_t)qi * k, c->out_s_pin + (size_t)qi * count, (size_t)count * sizeof(float));
    memcpy(out_rows + (size_t)qi * k, c->out_r_pin + (size_t)qi * count, (size_t)count * sizeof(int64_t));
  }
  stamp(4);
  return SVS_OK;
}

// One pass for everything that queued up while the device was busy.  The calling thread (the leader)
// owns the pass: queries gathered into one batch, k = the largest asked for (a top-k list's first n
// entries are the top-n list), results handed back to the waiters' own buffers.
static void coalesced_pass(svs_index* idx, std::vector<svs_index::Waiter*>& batch, int d) {
  const int nb = (int)batch.size();
  int kmax = 0;
  for (auto* w : batch) kmax = std::max(kmax, w->k);
  int rc = SVS_OK;
  int32_t count = 0;
  std::vector<float> qs, ss;
  std::vector<int64_t> rr;
  try {
    qs.resize((size_t)nb * d);
    ss.resize((size_t)nb * kmax);
    rr.resize((size_t)nb * kmax);
  } catch (const std::bad_alloc&) {
    rc = fail(SVS_ERR_NOMEM, "out of host memory for a coalesced pass");
  }
  if (rc == SVS_OK) {
    for (int i = 0; i < nb; ++i) memcpy(qs.data() + (size_t)i * d, batch[i]->q, (size_t)d * sizeof(float));
    rc = search_host(idx, qs.data(), nb, d, kmax, ss.data(), rr.data(), &count);
  }
  const std::string err = rc == SVS_OK ? std::string() : std::string(svs_last_error());
  idx->co_passes.fetch_add(1);
  idx->co_queries.fetch_add(nb);
  idx->co_sizes[std::min(nb, 256)].fetch_add(1);
  std::lock_guard<std::mutex> lk(idx->co_mu);
  for (int i = 0; i < nb; ++i) {
    svs_index::Waiter* w = batch[i];
    w->rc = rc;
    if (rc == SVS_OK) {
      w->count = std::min(w->k, (int)count);
      memcpy(w->out_s, ss.data() + (size_t)i * kmax, (size_t)w->count * sizeof(float));
      memcpy(w->out_r, rr.data() + (size_t)i * kmax, (size_t)w->count * sizeof(int64_t));
    } else {
      w->err = err;
    }
    w->done = true;
    if (!w->lead) w->cv.notify_one();
  }
}

int32_t svs_index_search(svs_index* idx, const float* queries, int32_t nq, int32_t d, int32_t k,
                         float* out_scores, int64_t* out_rows, int32_t* out_count) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  // (only well-formed single queries are coalesced: every error stays with the call that made it)
  // (and only ordinary k: a "rank everything" call must not size a whole pass's buffers)
  if (!(idx->coalesce.load() && nq == 1 && k > 0 && k <= 2048 && d == idx->d && queries && out_scores && out_rows && idx->n > 0))
    return search_host(idx, queries, nq, d, k, out_scores, out_rows, out_count);
  RefGuard guard(idx);
  svs_index::Waiter me;
  me.q = queries; me.k = k; me.out_s = out_scores; me.out_r = out_rows;
  {
    std::unique_lock<std::mutex> lk(idx->co_mu);
    idx->co_pending.push_back(&me);
    if (idx->co_hold > 0) idx->co_hold_cv.notify_all();
    if (!idx->co_busy) { idx->co_busy = true; me.lead = true; }
    else me.cv.wait(lk, [&] { return me.done || me.lead; });
  }
  if (me.lead) {
    // drive the device until this call's own query is answered, then hand over
    std::vector<svs_index::Waiter*> batch;
    while (!me.done) {
      {
        std::unique_lock<std::mutex> lk(idx->co_mu);
        if (idx->co_hold > 0) {   // (tests / benchmarks: a pass of a chosen size; bounded, so a miscounted test cannot hang)
          const int want = idx->co_hold;
          idx->co_hold_cv.wait_for(lk, std::chrono::seconds(5), [&] { return (int)idx->co_pending.size() >= want; });
          if (idx->co_hold == want) idx->co_hold = 0;   // (one shot; a hold set by another thread meanwhile stays)
        }
        // f32: whole kernel tiles -- the exact-f32 MFMA kernels cost the same for 33 queries as for 64 (1.8 vs 1.2 ms
        // for 32), so a queue that does not fill the next tile size leaves its tail for the following pass.
        // f16 / fp8: the passes are HBM-bound up to 128 queries and cost almost the same whatever they carry
        // (1M x 1536 f16: 0.57 / 0.60 / 0.64 ms at 16 / 32 / 64 queries; fp8 0.32 / 0.33 / 0.40): cutting 40
        // queued callers into 32 + 8 would cost two passes for the price of one, so everything queued goes out.
        size_t take = std::min<size_t>(idx->co_pending.size(), 256);
        if (idx->co_round.load() && idx->dtype == SVS_DTYPE_F32)
          for (size_t g : {(size_t)128, (size_t)64, (size_t)32, (size_t)16})
            if (take > g && take < 2 * g) { take = g; break; }
        batch.assign(idx->co_pending.begin(), idx->co_pending.begin() + take);
        idx->co_pending.erase(idx->co_pending.begin(), idx->co_pending.begin() + take);
      }
      coalesced_pass(idx, batch, d);
    }
    std::lock_guard<std::mutex> lk(idx->co_mu);
    if (!idx->co_pending.empty()) {
      idx->co_pending.front()->lead = true;   // (stays queued: its own loop takes it out)
      idx->co_pending.front()->cv.notify_one();
    } else {
      idx->co_busy = false;
    }
  }
  if (me.rc != SVS_OK) return fail(me.rc, "%s", me.err.c_str());
  if (out_count) *out_count = me.count;
  return SVS_OK;
}

int32_t svs_index_set_coalesce(svs_index* idx, int32_t enable) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  idx->coalesce.store(enable != 0);
  idx->co_round.store(enable != 2);   // (2: take everything that is queued, whatever the tile sizes -- A/B)
  return SVS_OK;
}

int32_t svs_index_coalesce_stats(svs_index* idx, int64_t* passes, int64_t* queries) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  if (passes) *passes = idx->co_passes.load();
  if (queries) *queries = idx->co_queries.load();
  return SVS_OK;
}

// (tests and tools only -- csrc/internal.h, not in include/svs_amd.h: the NEXT coalesced pass waits, at most 5 s,
//  until n single-query calls are queued, so that a pass of a chosen size can be formed on purpose)
int32_t svs_internal_coalesce_hold(svs_index* idx, int32_t n) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  if (n < 0 || n > 256) return fail(SVS_ERR_INVALID, "svs_internal_coalesce_hold: 0 <= n <= 256");
  std::lock_guard<std::mutex> lk(idx->co_mu);
  idx->co_hold = n;
  return SVS_OK;
}

int32_t svs_index_coalesce_sizes(svs_index* idx, int64_t* out, int32_t cap) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  if (!out || cap < 0 || cap > 257) return fail(SVS_ERR_INVALID, "svs_index_coalesce_sizes: out must hold cap <= 257 counters");
  for (int s = 0; s < cap; ++s) out[s] = idx->co_sizes[s].load();
  return SVS_OK;
}

int32_t svs_index_search_device(svs_index* idx, const float* dev_queries, int32_t nq, int32_t d,
                                int32_t k, float* dev_out_scores, int64_t* dev_out_rows,
                                int32_t* out_count, void* hip_stream) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  RefGuard guard(idx);
  std::shared_lock<std::shared_mutex> geo(idx->rw);
  int rc = check_query_args(idx, dev_queries, nq, d);
  if (rc != SVS_OK) return rc;
  const int count = (int)std::min<int64_t>(std::max(k, 0), idx->n - (int64_t)idx->dead_list.size());
  if (out_count) *out_count = count;
  if (nq == 0 || k <= 0) return SVS_OK;
  if (!dev_out_scores || !dev_out_rows) return fail(SVS_ERR_INVALID, "null output");
  HIP_TRY(hipSetDevice(idx->device));
  hipStream_t st = (hipStream_t)hip_stream;
  Ctx* c = nullptr;
  if ((rc = ctx_acquire(idx, st, false, &c)) != SVS_OK) return rc;
  struct CtxGuard { svs_index* i; Ctx* c; ~CtxGuard() { ctx_release(i, c); } } cg{idx, c};
  // growing scratch frees buffers that earlier work on this stream may still read
  const size_t need_scores = (size_t)nq * (size_t)((idx->n + 3) & ~(int64_t)3);
  if (c->async_pending && (need_scores > c->scores_cap || (size_t)nq > c->hist_cap)) HIP_TRY(hipStreamSynchronize(st));
  rc = enqueue_search(idx, c, dev_queries, nq, k, count, dev_out_scores, dev_out_rows, st);
  c->last_stream = st;
  c->async_pending = true;
  return rc;
}

int32_t svs_index_scores_n(svs_index* idx, const float* query, int32_t d, float* out_scores, int64_t out_capacity,
                           int64_t* out_n) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  RefGuard guard(idx);
  std::shared_lock<std::shared_mutex> geo(idx->rw);   // (appends / commits take it exclusively: idx->n cannot move below)
  if (out_n) *out_n = idx->n;
  int rc = check_query_args(idx, query, 1, d);
  if (rc != SVS_OK) return rc;
  if (!out_scores) return fail(SVS_ERR_INVALID, "null output");
  if (out_capacity < idx->n)
    return fail(SVS_ERR_INVALID, "svs_index_scores_n: the index holds %lld rows, the output buffer %lld floats "
                                 "(rows were appended since it was sized?)", (long long)idx->n, (long long)out_capacity);
  HIP_TRY(hipSetDevice(idx->device));
  if ((rc = staging_wait(idx)) != SVS_OK) return rc;
  Ctx* c = nullptr;
  if ((rc = ctx_acquire(idx, nullptr, true, &c)) != SVS_OK) return rc;
  struct CtxGuard { svs_index* i; Ctx* c; ~CtxGuard() { ctx_release(i, c); } } cg{idx, c};
  if ((rc = grow_dev(&c->q_dev, &c->q_cap, (size_t)d)) != SVS_OK) return rc;
  if ((rc = grow_dev(&c->scores, &c->scores_cap, (size_t)((idx->n + 3) & ~(int64_t)3))) != SVS_OK) return rc;
  HIP_TRY(hipMemcpyAsync(c->q_dev, query, (size_t)d * sizeof(float), hipMemcpyHostToDevice, c->stream));
  if ((rc = launch_scores(idx, c, c->q_dev, c->scores, c->stream)) != SVS_OK) return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out_scores, c->scores, (size_t)idx->n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return SVS_OK;
}

int32_t svs_index_top_pairs(svs_index* idx, int32_t k, float* out_scores, int64_t* out_i, int64_t* out_j,
                            int32_t* out_count) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  RefGuard guard(idx);
  std::shared_lock<std::shared_mutex> geo(idx->rw);
  const int64_t n = idx->n;
  const int64_t np = (n + 3) & ~(int64_t)3;
  const int64_t n_live = n - (int64_t)idx->dead_list.size();
  const int64_t pairs = n_live * (n_live - 1) / 2;
  const int count = (int)std::min<int64_t>(std::max(k, 0), pairs);
  if (out_count) *out_count = count;
  if (count == 0) return SVS_OK;
  if (!out_scores || !out_i || !out_j) return fail(SVS_ERR_INVALID, "null output");
  HIP_TRY(hipSetDevice(idx->device));
  Ctx* c = nullptr;
  int rc;
  if ((rc = staging_wait(idx)) != SVS_OK) return rc;
  if ((rc = ctx_acquire(idx, nullptr, true, &c)) != SVS_OK) return rc;
  struct CtxGuard { svs_index* i; Ctx* c; ~CtxGuard() { ctx_release(i, c); } } cg{idx, c};
  // Small corpora: materialise n x n like the reference (src/svs/kb.py:1651).  Past n^2 = 2^32 scores
  // (or with variant 1, for A/B and tests): tiled GEMM with the i < j mask and a threshold in the
  // fused epilogue, nothing materialised but a prefix block.
  if (n * np <= 0xffffffffll && idx->variant.load() != 1)
    return top_pairs_materialised(idx, c, n, count, out_scores, out_i, out_j);
  return top_pairs_tiled(idx, c, count, out_scores, out_i, out_j);
}

int32_t svs_index_debug_dequant(svs_index* idx, int64_t row0, int64_t nrows, float* out) {
  if (!idx || !out) return fail(SVS_ERR_INVALID, "null argument");
  if (row0 < 0 || nrows < 0 || row0 + nrows > idx->n) return fail(SVS_ERR_INVALID, "row range out of bounds");
  if (nrows == 0 || idx->d == 0) return SVS_OK;
  HIP_TRY(hipSetDevice(idx->device));
  { int rc = staging_wait(idx); if (rc != SVS_OK) return rc; }
  const size_t cnt = (size_t)nrows * idx->d;
  if (idx->dtype == SVS_DTYPE_F32) {
    HIP_TRY(hipMemcpy2D(out, (size_t)idx->d * sizeof(float), (const float*)idx->rows + row0 * idx->ld,
                        (size_t)idx->ld * sizeof(float), (size_t)idx->d * sizeof(float), (size_t)nrows, hipMemcpyDeviceToHost));
    return SVS_OK;
  }
  float* tmp = nullptr;
  HIP_TRY(hipMalloc((void**)&tmp, cnt * sizeof(float)));
  if (idx->dtype == SVS_DTYPE_F16)
    hipLaunchKernelGGL(dequant_rows_f16_kernel, dim3(2048), dim3(256), 0, 0, (const _Float16*)idx->rows, row0, nrows, idx->d, idx->ld, tmp);
  else
    hipLaunchKernelGGL(dequant_rows_fp8_kernel, dim3(2048), dim3(256), 0, 0, (const uint8_t*)idx->rows, idx->row_scales, row0, nrows,
                       idx->d, idx->ld, tmp);
  hipError_t e = hipMemcpy(out, tmp, cnt * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail(SVS_ERR_DEVICE, "dequant read-back: %s", hipGetErrorString(e));
  return SVS_OK;
}

int32_t svs_index_debug_query(svs_index* idx, const float* query, int32_t d, float* out) {
  if (!idx || !query || !out) return fail(SVS_ERR_INVALID, "null argument");
  if (d != idx->d) return fail(SVS_ERR_SHAPE, "query dim %d != index dim %d", d, idx->d);
  if (idx->dtype == SVS_DTYPE_F32 || d == 0) {
    memcpy(out, query, (size_t)d * sizeof(float));
    return SVS_OK;
  }
  // build a one-row index of the same dtype from the query and read it back: the
  // row path and the query path share their rounding / quantisation kernels
  svs_index* tmp = nullptr;
  int rc = svs_index_create(query, 1, d, idx->dtype, idx->device, 0, &tmp);
  if (rc != SVS_OK) return rc;
  rc = svs_index_debug_dequant(tmp, 0, 1, out);
  svs_index_release(tmp);
  return rc;
}

int32_t svs_index_set_timing(svs_index* idx, int32_t enable) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  idx->timing.store(enable > 0 ? enable : 0);
  idx->timing_seq.store(0);
  return SVS_OK;
}

int32_t svs_index_get_timing(svs_index* idx, svs_timing_t* out) {
  if (!idx || !out) return fail(SVS_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(idx->device));
  std::vector<EvTriple> evs;
  {
    std::lock_guard<std::mutex> lk(idx->mu);
    evs.swap(idx->evs);
  }
  out->score_ms_sum = 0;
  out->select_ms_sum = 0;
  out->launches = 0;
  out->dominant_ms_sum = 0;
  int rc = SVS_OK;
  for (auto& t : evs) {
    float a = 0, b = 0;
    hipError_t e = hipEventSynchronize(t.e2);
    if (e == hipSuccess) e = hipEventElapsedTime(&a, t.e0, t.e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&b, t.e1, t.e2);
    float dm = a;
    if (e == hipSuccess && t.d0 && t.d1) e = hipEventElapsedTime(&dm, t.d0, t.d1);
    if (e == hipSuccess) {
      out->score_ms_sum += a;
      out->select_ms_sum += b;
      out->dominant_ms_sum += dm;
      out->launches++;
    } else {
      rc = fail(SVS_ERR_DEVICE, "timing events: %s", hipGetErrorString(e));
    }
    (void)hipEventDestroy(t.e0);
    (void)hipEventDestroy(t.e1);
    (void)hipEventDestroy(t.e2);
    if (t.d0) (void)hipEventDestroy(t.d0);
    if (t.d1) (void)hipEventDestroy(t.d1);
  }
  return rc;
}

int32_t svs_internal_tune(int32_t what, int64_t value) {
  switch (what) {
    case 0: if (value < 1) break; g_tune_prefix_div.store(value); return SVS_OK;
    case 1: if (value < 0 || value > 1) break; g_tune_upload.store(value); return SVS_OK;
    case 2: if (value < 0 || value > 1) break; g_tune_spread.store(value); return SVS_OK;
    default: break;
  }
  return fail(SVS_ERR_INVALID, "svs_internal_tune(%d, %lld): unknown knob or value", what, (long long)value);
}

int32_t svs_internal_host_phases(double* out, int32_t n) {
  for (int i = 0; i < n && i < 6; ++i) out[i] = g_host_phase[i];
  return SVS_OK;
}

int32_t svs_index_set_variant(svs_index* idx, int32_t variant) {
  if (!idx) return fail(SVS_ERR_INVALID, "null index");
  if (variant < 0 || variant > 10) return fail(SVS_ERR_INVALID, "unknown variant %d", variant);
  idx->variant.store(variant);
  return SVS_OK;
}

}  // extern "C"
