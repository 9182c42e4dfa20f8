// svs_multi: ONE process owning several MI355X, the corpus row-sharded across them -- the
// `devices, ndev` form of the C-ABI sketch in SURVEY.md 8(b) / 8(e).  Built on the single-device
// entries of svs_amd.hip (every shard is an ordinary svs_index with its row_offset): shard g holds
// the contiguous rows [g * ceil(n / G), ...), a search runs on all shards at once (one worker thread per
// shard: each call blocks in its device's stream), every shard returns its local top-k with GLOBAL
// rows, and the caller's thread merges them under the library's total order (score desc, row desc;
// NaN largest) -- the result is the single-device result, whatever G is.  No RCCL: the exchange is
// G * k * 12 bytes of pinned host memory.  (One process PER GPU over torch.distributed:
// svs_amd/sharded.py.)
// Reference lines this replaces: the same as svs_index_create / svs_index_search
// (src/svs/kb.py:875-876, :1623-1626, src/svs/util.py:190-203) for a KB whose matrix is larger
// than one card, or whose latency should drop with the card count.
#include "../../include/svs_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "keys.h"

namespace {

// one worker per shard: jobs for a shard run in order, callers wait on their own latch
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false;
  void run() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        job = std::move(q.front());
        q.pop_front();
      }
      job();
    }
  }
  void post(std::function<void()> f) {
    { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); }
    cv.notify_one();
  }
};

struct Latch {
  std::mutex mu;
  std::condition_variable cv;
  int left;
  explicit Latch(int n) : left(n) {}
  void done() { std::lock_guard<std::mutex> lk(mu); if (--left == 0) cv.notify_all(); }
  void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return left == 0; }); }
};

}  // namespace

struct MultiWaiter {   // one queued single-query call (svs_multi_set_coalesce)
  const float* q;
  int k, count = 0, rc = SVS_OK;
  float* out_s;
  int64_t* out_r;
  std::string err;
  bool done = false, lead = false;
  std::condition_variable cv;
};

struct svs_multi {
  std::vector<svs_index*> shards;
  std::vector<Worker*> workers;
  std::atomic<int> refs{1};
  int32_t d = 0;
  std::atomic<bool> coalesce{false};
  std::mutex co_mu;
  std::vector<MultiWaiter*> co_pending;
  bool co_busy = false;
  std::atomic<int64_t> co_passes{0}, co_queries{0};
};

namespace {

// (the single-device entries report through svs_last_error() of the thread that ran them: a worker's
//  message is carried over to the caller's thread by failing again with the same text)
int refail(int code, const std::string& msg);

void multi_destroy(svs_multi* m) {
  for (Worker* w : m->workers) {
    { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
  }
  for (svs_index* s : m->shards)
    if (s) svs_index_release(s);
  delete m;
}

}  // namespace

// svs_amd.hip: sets the calling thread's svs_last_error()
extern "C" int32_t svs_internal_set_error(int32_t code, const char* msg);
namespace {
int refail(int code, const std::string& msg) { return svs_internal_set_error(code, msg.c_str()); }
}  // namespace

extern "C" {

int32_t svs_multi_create(const float* host_rows, int64_t n, int32_t d, int32_t store_dtype, const int32_t* devices,
                         int32_t ndev, svs_multi** out) {
  if (!out) return refail(SVS_ERR_INVALID, "null out");
  *out = nullptr;
  if (!devices || ndev < 1 || ndev > 64) return refail(SVS_ERR_INVALID, "devices: 1..64 ordinals expected");
  if (n < 0 || d < 0 || (n > 0 && d > 0 && !host_rows)) return refail(SVS_ERR_INVALID, "bad matrix arguments");
  svs_multi* m = new (std::nothrow) svs_multi();
  if (!m) return refail(SVS_ERR_NOMEM, "out of host memory");
  m->d = d;
  m->shards.assign((size_t)ndev, nullptr);
  const int64_t per = (n + ndev - 1) / ndev;   // rank g holds [g * per, min((g + 1) * per, n)): svs_amd/sharded.py shard_bounds
  std::vector<int> rc((size_t)ndev, SVS_OK);
  std::vector<std::string> msg((size_t)ndev);
  {
    // uploads run concurrently: every shard has its own pinned staging buffers and stream
    std::vector<std::thread> up;
    for (int g = 0; g < ndev; ++g)
      up.emplace_back([&, g] {
        const int64_t lo = std::min<int64_t>((int64_t)g * per, n), hi = std::min<int64_t>(lo + per, n);
        rc[g] = svs_index_create(hi > lo ? host_rows + lo * (int64_t)d : nullptr, hi - lo, d, store_dtype, devices[g], lo, &m->shards[g]);
        if (rc[g] != SVS_OK) msg[g] = svs_last_error();
      });
    for (auto& t : up) t.join();
  }
  for (int g = 0; g < ndev; ++g)
    if (rc[g] != SVS_OK) {
      const int code = rc[g];
      const std::string text = "shard " + std::to_string(g) + " (device " + std::to_string(devices[g]) + "): " + msg[g];
      multi_destroy(m);
      return refail(code, text);
    }
  for (int g = 0; g < ndev; ++g) {
    Worker* w = new (std::nothrow) Worker();
    if (!w) { multi_destroy(m); return refail(SVS_ERR_NOMEM, "out of host memory"); }
    m->workers.push_back(w);
    w->th = std::thread([w] { w->run(); });
  }
  *out = m;
  return SVS_OK;
}

int32_t svs_multi_retain(svs_multi* m) {
  if (!m) return refail(SVS_ERR_INVALID, "null handle");
  m->refs.fetch_add(1);
  return SVS_OK;
}

int32_t svs_multi_release(svs_multi* m) {
  if (!m) return refail(SVS_ERR_INVALID, "null handle");
  if (m->refs.fetch_sub(1) == 1) multi_destroy(m);
  return SVS_OK;
}

int32_t svs_multi_info(svs_multi* m, int32_t* ndev, int64_t* n, int32_t* d, int64_t* n_masked) {
  if (!m) return refail(SVS_ERR_INVALID, "null handle");
  int64_t rows = 0, dead = 0;
  for (svs_index* s : m->shards) {
    svs_index_info_t info;
    const int rc = svs_index_info(s, &info);
    if (rc != SVS_OK) return rc;
    rows += info.n;
    dead += info.n_masked;
  }
  if (ndev) *ndev = (int32_t)m->shards.size();
  if (n) *n = rows;
  if (d) *d = m->d;
  if (n_masked) *n_masked = dead;
  return SVS_OK;
}

int32_t svs_multi_shard(svs_multi* m, int32_t g, svs_index** out) {
  if (!m || !out) return refail(SVS_ERR_INVALID, "null argument");
  if (g < 0 || g >= (int32_t)m->shards.size()) return refail(SVS_ERR_INVALID, "shard " + std::to_string(g) + " of " + std::to_string(m->shards.size()));
  const int rc = svs_index_retain(m->shards[g]);
  if (rc != SVS_OK) return rc;
  *out = m->shards[g];
  return SVS_OK;
}

static int32_t multi_search_impl(svs_multi* m, const float* queries, int32_t nq, int32_t d, int32_t k, float* out_scores,
                                 int64_t* out_rows, int32_t* out_count) {
  m->refs.fetch_add(1);
  struct Unref { svs_multi* m; ~Unref() { svs_multi_release(m); } } unref{m};
  const int G = (int)m->shards.size();
  // shards that hold rows; an empty corpus (or a wrong d) reports exactly what one index reports
  std::vector<int> live;
  std::vector<svs_index_info_t> info((size_t)G);
  int64_t rows = 0, dead = 0;
  for (int g = 0; g < G; ++g) {
    const int rc = svs_index_info(m->shards[g], &info[g]);
    if (rc != SVS_OK) return rc;
    rows += info[g].n;
    dead += info[g].n_masked;
    if (info[g].n > 0) live.push_back(g);
  }
  if (live.empty() || d != m->d || nq < 0 || (nq > 0 && !queries))
    return svs_index_search(m->shards[0], queries, nq, d, k, out_scores, out_rows, out_count);
  const int kk = std::max(k, 0);
  int count = (int)std::min<int64_t>(kk, rows - dead);
  if (out_count) *out_count = count;
  if (nq == 0 || count == 0) return SVS_OK;
  if (!out_scores || !out_rows) return refail(SVS_ERR_INVALID, "null output");
  const int L = (int)live.size();
  struct Part { std::vector<float> s; std::vector<int64_t> r; int32_t cnt = 0; int stride = 0; int rc = SVS_OK; std::string msg; };
  // (shared: the last worker is still inside done() when the caller wakes up and returns)
  auto latch = std::make_shared<Latch>(L);
  auto parts = std::make_shared<std::vector<Part>>((size_t)L);
  std::vector<Part>& part = *parts;
  for (int t = 0; t < L; ++t) {
    const int g = live[t];
    Part* p = &part[t];
    // every shard is asked for min(k, its live rows): svs_index_search's output rows are k apart
    // (include/svs_amd.h), so that is also the stride of its part -- never nq * k for an absurd k
    const int kg = (int)std::min<int64_t>(kk, info[g].n - info[g].n_masked);
    p->stride = kg;
    try {
      p->s.resize((size_t)nq * (size_t)kg);
      p->r.resize((size_t)nq * (size_t)kg);
    } catch (const std::bad_alloc&) {
      // (jobs already posted hold `parts` alive and finish on their own)
      for (int u = t; u < L; ++u) latch->done();
      latch->wait();
      return refail(SVS_ERR_NOMEM, "out of host memory for the shard results");
    }
    svs_index* shard = m->shards[g];
    m->workers[g]->post([=] {
      (void)parts;   // keeps the result buffers alive as long as a job can touch them
      p->rc = svs_index_search(shard, queries, nq, d, kg, p->s.data(), p->r.data(), &p->cnt);
      if (p->rc != SVS_OK) p->msg = svs_last_error();
      latch->done();
    });
  }
  latch->wait();
  for (int t = 0; t < L; ++t)
    if (part[t].rc != SVS_OK) return refail(part[t].rc, "shard " + std::to_string(live[t]) + ": " + part[t].msg);
  // The snapshot above was taken without a lock: a concurrent svs_index_mask_rows on a shard handle
  // (svs_multi_shard) can have shrunk a shard since.  What the shards RETURNED is what there is -- one index
  // under the same interleaving answers min(k, live rows at the time of its search) too.
  {
    int64_t got = 0;
    for (int t = 0; t < L; ++t) got += part[t].cnt;
    if (got < count) {
      count = (int)got;
      if (out_count) *out_count = count;
    }
  }
  // L sorted lists per query -> the best `count` under (score key desc, row desc)
  std::vector<int> head((size_t)L);
  for (int qi = 0; qi < nq; ++qi) {
    std::fill(head.begin(), head.end(), 0);
    for (int o = 0; o < count; ++o) {
      int best = -1;
      uint32_t bkey = 0;
      int64_t brow = 0;
      for (int t = 0; t < L; ++t) {
        if (head[t] >= part[t].cnt) continue;
        const size_t at = (size_t)qi * (size_t)part[t].stride + (size_t)head[t];
        const uint32_t key = svs::score_key(part[t].s[at]);
        const int64_t row = part[t].r[at];
        if (best < 0 || key > bkey || (key == bkey && row > brow)) { best = t; bkey = key; brow = row; }
      }
      if (best < 0) return refail(SVS_ERR_DEVICE, "internal: shards returned fewer rows than they hold");
      const size_t at = (size_t)qi * (size_t)part[best].stride + (size_t)head[best]++;
      out_scores[(size_t)qi * kk + o] = part[best].s[at];
      out_rows[(size_t)qi * kk + o] = part[best].r[at];
    }
  }
  return SVS_OK;
}

// Concurrent single-query callers share passes over ALL shards: the protocol of svs_index_set_coalesce
// (svs_amd.hip), one level up -- the queue is in front of the shard workers, so one batched search per
// shard serves everybody who queued up while the devices were busy.
static void multi_coalesced_pass(svs_multi* m, std::vector<MultiWaiter*>& batch, int d) {
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
    rc = refail(SVS_ERR_NOMEM, "out of host memory for a coalesced pass");
  }
  if (rc == SVS_OK) {
    for (int i = 0; i < nb; ++i) memcpy(qs.data() + (size_t)i * d, batch[i]->q, (size_t)d * sizeof(float));
    rc = multi_search_impl(m, qs.data(), nb, d, kmax, ss.data(), rr.data(), &count);
  }
  const std::string err = rc == SVS_OK ? std::string() : std::string(svs_last_error());
  m->co_passes.fetch_add(1);
  m->co_queries.fetch_add(nb);
  std::lock_guard<std::mutex> lk(m->co_mu);
  for (int i = 0; i < nb; ++i) {
    MultiWaiter* w = batch[i];
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

int32_t svs_multi_search(svs_multi* m, const float* queries, int32_t nq, int32_t d, int32_t k, float* out_scores,
                         int64_t* out_rows, int32_t* out_count) {
  if (!m) return refail(SVS_ERR_INVALID, "null handle");
  if (!(m->coalesce.load() && nq == 1 && k > 0 && k <= 2048 && d == m->d && queries && out_scores && out_rows))
    return multi_search_impl(m, queries, nq, d, k, out_scores, out_rows, out_count);
  m->refs.fetch_add(1);
  struct Unref { svs_multi* m; ~Unref() { svs_multi_release(m); } } unref{m};
  MultiWaiter me;
  me.q = queries; me.k = k; me.out_s = out_scores; me.out_r = out_rows;
  {
    std::unique_lock<std::mutex> lk(m->co_mu);
    m->co_pending.push_back(&me);
    if (!m->co_busy) { m->co_busy = true; me.lead = true; }
    else me.cv.wait(lk, [&] { return me.done || me.lead; });
  }
  if (me.lead) {
    std::vector<MultiWaiter*> batch;
    while (!me.done) {
      {
        std::lock_guard<std::mutex> lk(m->co_mu);
        size_t take = std::min<size_t>(m->co_pending.size(), 256);
        for (size_t g : {(size_t)128, (size_t)64, (size_t)32, (size_t)16})   // whole kernel tiles (svs_amd.hip)
          if (take > g && take < 2 * g) { take = g; break; }
        batch.assign(m->co_pending.begin(), m->co_pending.begin() + take);
        m->co_pending.erase(m->co_pending.begin(), m->co_pending.begin() + take);
      }
      multi_coalesced_pass(m, batch, d);
    }
    std::lock_guard<std::mutex> lk(m->co_mu);
    if (!m->co_pending.empty()) {
      m->co_pending.front()->lead = true;
      m->co_pending.front()->cv.notify_one();
    } else {
      m->co_busy = false;
    }
  }
  if (me.rc != SVS_OK) return refail(me.rc, me.err);
  if (out_count) *out_count = me.count;
  return SVS_OK;
}

int32_t svs_multi_set_coalesce(svs_multi* m, int32_t enable) {
  if (!m) return refail(SVS_ERR_INVALID, "null handle");
  m->coalesce.store(enable != 0);
  return SVS_OK;
}

int32_t svs_multi_coalesce_stats(svs_multi* m, int64_t* passes, int64_t* queries) {
  if (!m) return refail(SVS_ERR_INVALID, "null handle");
  if (passes) *passes = m->co_passes.load();
  if (queries) *queries = m->co_queries.load();
  return SVS_OK;
}

}  // extern "C"
