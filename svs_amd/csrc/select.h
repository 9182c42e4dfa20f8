// Top-k stage: replaces svs.util.get_top_k (reference src/svs/util.py:190-203 --
// np.argpartition + sorted(reverse=True)) on the device.
//
// Everything works on the unique 64-bit keys of keys.h, so "the k largest keys,
// descending" IS the reference's output order (score desc, index desc) and does
// not depend on scheduling.  Byte/integer work, HBM/L2-bound; no matrix cores.
//
// Path A (k <= SEL_KMAX): exact radix select of the k-th largest 32-bit score
//   key in three histogram passes over the score vector (11 + 11 + 10 bits;
//   4 MB per pass at 1M rows, L2/MALL resident right after the score stage),
//   one filter pass that compacts the keys above the threshold (wave-aggregated
//   appends) plus the ties AT the threshold, and a one-workgroup bitonic sort of
//   the <= SORT_CAP survivors in LDS.  Each pass re-derives the previous
//   passes' bucket choice from their histograms, so there are no tiny "pick"
//   launches and no host round trips.
// Path D (n <= SORT_CAP): the sort workgroup reads the scores directly.
// Path B (any k): all n keys are sorted by a global bitonic network (LDS for
//   strides < SORT_CAP, one launch per larger stride).  Used for k > SEL_KMAX,
//   e.g. the reference's "rank the whole KB" call (n = 10,548).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keys.h"

namespace svs {

constexpr int SEL_KMAX = 1024;    // path A handles k <= SEL_KMAX
constexpr int SORT_CAP = 4096;    // keys sorted in LDS by one workgroup (32 KiB)
constexpr int TIE_CAP = SORT_CAP - SEL_KMAX;
constexpr int HIST_BINS = 2048;   // 11 bits per pass
constexpr int SEL_THREADS = 256;
constexpr int SORT_THREADS = 1024;

// per-query scratch header, zeroed before the passes
struct SelCounters {
  uint32_t n_gt;   // keys strictly above the threshold appended so far
  uint32_t n_eq;   // ties at the threshold seen (may exceed TIE_CAP)
  uint32_t pad0, pad1;
};

__device__ __forceinline__ int pass_shift(int pass) { return pass == 0 ? 21 : (pass == 1 ? 10 : 0); }
__device__ __forceinline__ int pass_bins(int pass) { return pass == 2 ? 1024 : 2048; }

// The first SEL_THREADS threads of a workgroup find, from a histogram of `bins`
// buckets, the bucket holding the k_rem-th largest element counting from the
// top.  Returns the bucket in *b_out and the rank inside it (1-based from the
// bucket's top) in *k_out.  Every thread of the workgroup must call this (it
// contains workgroup barriers); `sh` is SEL_THREADS+2 words of LDS.
__device__ __forceinline__ void pick_bucket(const uint32_t* __restrict__ hist, int bins,
                                            uint32_t k_rem, uint32_t* sh, uint32_t* b_out,
                                            uint32_t* k_out) {
  const int tid = threadIdx.x;
  const bool act = tid < SEL_THREADS;
  const int per = bins / SEL_THREADS;  // 8 or 4
  // thread t owns buckets [bins - (t+1)*per, bins - t*per): t = 0 is the top
  uint32_t loc[8];
  uint32_t sum = 0;
  const int hi = bins - tid * per;
  if (act) {
    for (int i = 0; i < per; ++i) {
      loc[i] = hist[hi - 1 - i];  // descending bucket order
      sum += loc[i];
    }
    sh[tid] = sum;
  }
  __syncthreads();
  // inclusive scan over SEL_THREADS partials (Hillis-Steele, 8 steps)
  for (int off = 1; off < SEL_THREADS; off <<= 1) {
    const uint32_t v = (act && tid >= off) ? sh[tid - off] : 0;
    __syncthreads();
    if (act) sh[tid] += v;
    __syncthreads();
  }
  if (act) {
    const uint32_t incl = sh[tid];
    const uint32_t excl = incl - sum;
    if (excl < k_rem && k_rem <= incl) {
      uint32_t c = excl;
      for (int i = 0; i < per; ++i) {
        if (c + loc[i] >= k_rem) {
          sh[SEL_THREADS] = (uint32_t)(hi - 1 - i);
          sh[SEL_THREADS + 1] = k_rem - c;
          break;
        }
        c += loc[i];
      }
    }
  }
  __syncthreads();
  *b_out = sh[SEL_THREADS];
  *k_out = sh[SEL_THREADS + 1];
  __syncthreads();
}

// Re-derive (prefix bits, remaining rank) after `npass` completed passes.
__device__ __forceinline__ void derive_prefix(const uint32_t* __restrict__ hist_q, int npass,
                                              uint32_t k, uint32_t* sh, uint32_t* prefix,
                                              uint32_t* k_rem) {
  uint32_t p = 0, kr = k;
  for (int ps = 0; ps < npass; ++ps) {
    uint32_t b, k2;
    pick_bucket(hist_q + ps * HIST_BINS, pass_bins(ps), kr, sh, &b, &k2);
    p |= b << pass_shift(ps);
    kr = k2;
  }
  *prefix = p;
  *k_rem = kr;
}

// grid = (blocks, nq).  hist is [nq][3][HIST_BINS], zeroed by the host before pass 0.
__global__ __launch_bounds__(SEL_THREADS) void select_hist_kernel(
    const float* __restrict__ scores, int64_t n, int64_t score_stride, uint32_t k, int pass,
    uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[HIST_BINS];
  __shared__ uint32_t sh[SEL_THREADS + 2];
  const int qi = blockIdx.y;
  const float* s = scores + (int64_t)qi * score_stride;
  uint32_t* hq = hist + (int64_t)qi * 3 * HIST_BINS;
  for (int i = threadIdx.x; i < HIST_BINS; i += SEL_THREADS) lh[i] = 0;
  uint32_t prefix = 0, k_rem = k;
  derive_prefix(hq, pass, k, sh, &prefix, &k_rem);  // ends with a barrier
  const int shift = pass_shift(pass);
  const uint32_t bmask = (uint32_t)pass_bins(pass) - 1;
  // bits above this pass's field must equal the prefix
  const uint32_t hmask = pass == 0 ? 0u : (pass == 1 ? 0xffe00000u : 0xfffffc00u);
  const int64_t stride = (int64_t)gridDim.x * SEL_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * SEL_THREADS + threadIdx.x; i < n; i += stride) {
    const uint32_t key = score_key(s[i]);
    if ((key & hmask) == prefix) atomicAdd(&lh[(key >> shift) & bmask], 1u);
  }
  __syncthreads();
  uint32_t* hp = hq + pass * HIST_BINS;
  for (int i = threadIdx.x; i < HIST_BINS; i += SEL_THREADS) {
    const uint32_t c = lh[i];
    if (c) atomicAdd(&hp[i], c);
  }
}

// grid = (blocks, nq).  cand is [nq][SORT_CAP] keys: [0, SEL_KMAX) strictly
// greater than the threshold, [SEL_KMAX, SORT_CAP) ties at the threshold.
__global__ __launch_bounds__(SEL_THREADS) void select_filter_kernel(
    const float* __restrict__ scores, int64_t n, int64_t score_stride, uint32_t k,
    const uint32_t* __restrict__ hist, uint64_t* __restrict__ cand,
    SelCounters* __restrict__ counters) {
  __shared__ uint32_t sh[SEL_THREADS + 2];
  const int qi = blockIdx.y;
  const float* s = scores + (int64_t)qi * score_stride;
  uint32_t thr = 0, k_rem = k;
  derive_prefix(hist + (int64_t)qi * 3 * HIST_BINS, 3, k, sh, &thr, &k_rem);
  uint64_t* cq = cand + (int64_t)qi * SORT_CAP;
  SelCounters* cn = counters + qi;
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * SEL_THREADS;
  const int64_t start = (int64_t)blockIdx.x * SEL_THREADS + threadIdx.x;
  // uniform trip count per wave so that ballots see whole waves
  const int64_t iters = (n + stride - 1) / stride;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t i = start + it * stride;
    uint32_t key = 0;
    bool gt = false, eq = false;
    if (i < n) {
      key = score_key(s[i]);
      gt = key > thr;
      eq = key == thr;
    }
    const unsigned long long mg = __ballot(gt);
    const unsigned long long me = __ballot(eq);
    if (mg) {
      uint32_t base = 0;
      const int leader = __ffsll((long long)mg) - 1;
      if (lane == leader) base = atomicAdd(&cn->n_gt, (uint32_t)__popcll(mg));
      base = __shfl(base, leader, 64);
      if (gt) {
        const uint32_t slot = base + (uint32_t)__popcll(mg & ((1ull << lane) - 1));
        if (slot < SEL_KMAX) cq[slot] = ((uint64_t)key << 32) | (uint32_t)i;
      }
    }
    if (me) {
      uint32_t base = 0;
      const int leader = __ffsll((long long)me) - 1;
      if (lane == leader) base = atomicAdd(&cn->n_eq, (uint32_t)__popcll(me));
      base = __shfl(base, leader, 64);
      if (eq) {
        const uint32_t slot = base + (uint32_t)__popcll(me & ((1ull << lane) - 1));
        if (slot < TIE_CAP) cq[SEL_KMAX + slot] = ((uint64_t)key << 32) | (uint32_t)i;
      }
    }
  }
}

// Bitonic sort of S[0, m) (m a power of two <= SORT_CAP) in LDS, descending.
__device__ __forceinline__ void bitonic_sort_lds_desc(uint64_t* S, int m) {
  for (int size = 2; size <= m; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < (m >> 1); i += blockDim.x) {
        const int lo = 2 * i - (i & (stride - 1));
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const uint64_t a = S[lo], b = S[hi];
        if ((a < b) == desc) {
          S[lo] = b;
          S[hi] = a;
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

__device__ __forceinline__ void emit_topk(const uint64_t* S, int count, int k, int64_t row_offset,
                                          float* __restrict__ out_scores,
                                          int64_t* __restrict__ out_rows) {
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    if (i < count) {
      const uint64_t key = S[i];
      out_scores[i] = key_score((uint32_t)(key >> 32));
      out_rows[i] = row_offset + (int64_t)(uint32_t)key;
    } else {
      out_scores[i] = -__builtin_inff();
      out_rows[i] = -1;
    }
  }
}

// grid = nq, one workgroup per query.  mode 0: path A (candidates from the
// filter); mode 1: path D (n <= SORT_CAP, read scores directly).
__global__ __launch_bounds__(SORT_THREADS) void select_sort_kernel(
    const float* __restrict__ scores, int64_t n, int64_t score_stride, int k_out, int count,
    int mode, const uint32_t* __restrict__ hist, const uint64_t* __restrict__ cand,
    const SelCounters* __restrict__ counters, int64_t row_offset, float* __restrict__ out_scores,
    int64_t* __restrict__ out_rows) {
  __shared__ uint64_t S[SORT_CAP];
  __shared__ uint32_t sh[SEL_THREADS + 2];
  __shared__ uint32_t s_cnt;
  const int qi = blockIdx.x;
  const float* s = scores + (int64_t)qi * score_stride;
  float* os = out_scores + (int64_t)qi * k_out;
  int64_t* orow = out_rows + (int64_t)qi * k_out;
  int m = 0;
  if (mode == 1) {
    m = next_pow2((int)n < 2 ? 2 : (int)n);
    for (int i = threadIdx.x; i < m; i += blockDim.x) S[i] = i < n ? make_key(s[i], (uint32_t)i) : 0ull;
  } else {
    const uint64_t* cq = cand + (int64_t)qi * SORT_CAP;
    const uint32_t n_gt = counters[qi].n_gt;   // == count - (rank inside the tie bucket)
    const uint32_t n_eq = counters[qi].n_eq;
    if (n_eq <= (uint32_t)TIE_CAP) {
      const int tot = (int)(n_gt + n_eq);
      m = next_pow2(tot < 2 ? 2 : tot);
      for (int i = threadIdx.x; i < m; i += blockDim.x) {
        uint64_t v = 0ull;
        if (i < (int)n_gt) v = cq[i];
        else if (i < tot) v = cq[SEL_KMAX + (i - (int)n_gt)];
        S[i] = v;
      }
    } else {
      // Slow path: more ties at the threshold than the candidate buffer holds
      // (thousands of bit-identical scores).  The winners among ties are the
      // LARGEST rows, so walk the score vector backwards in SORT_THREADS-row
      // chunks collecting ties until enough are held.
      uint32_t thr = 0, k_rem = (uint32_t)count;
      derive_prefix(hist + (int64_t)qi * 3 * HIST_BINS, 3, (uint32_t)count, sh, &thr, &k_rem);
      if (threadIdx.x == 0) s_cnt = 0;
      for (int i = threadIdx.x; i < SORT_CAP; i += blockDim.x) S[i] = i < (int)n_gt ? cq[i] : 0ull;
      __syncthreads();
      for (int64_t hi_row = n; hi_row > 0; hi_row -= SORT_THREADS) {
        const int64_t i = hi_row - 1 - threadIdx.x;
        if (i >= 0) {
          const uint32_t key = score_key(s[i]);
          if (key == thr) {
            const uint32_t slot = atomicAdd(&s_cnt, 1u);
            if (n_gt + slot < (uint32_t)SORT_CAP) S[n_gt + slot] = ((uint64_t)key << 32) | (uint32_t)i;
          }
        }
        __syncthreads();
        const uint32_t have = s_cnt;
        __syncthreads();
        if (have >= k_rem) break;  // every tie with a row above hi_row - chunk is held
      }
      m = SORT_CAP;
    }
  }
  __syncthreads();
  bitonic_sort_lds_desc(S, m);
  emit_topk(S, count, k_out, row_offset, os, orow);
}

// ---- path B: global bitonic sort of all keys --------------------------------
// keys is [nq][npad], npad a power of two >= max(n, 2).
__global__ void keys_build_kernel(const float* __restrict__ scores, int64_t n,
                                  int64_t score_stride, int64_t npad, uint64_t* __restrict__ keys) {
  const int qi = blockIdx.y;
  const float* s = scores + (int64_t)qi * score_stride;
  uint64_t* kq = keys + (int64_t)qi * npad;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += stride)
    kq[i] = i < n ? make_key(s[i], (uint32_t)i) : 0ull;
}

// Sorts/merges each SORT_CAP-key chunk in LDS.  full == 1: all stages with
// size <= SORT_CAP (initial local sort).  full == 0: the strides < SORT_CAP of
// the merge step of `size` (> SORT_CAP).  grid = (npad / chunk, nq).
__global__ __launch_bounds__(SORT_THREADS) void bitonic_local_kernel(uint64_t* __restrict__ keys,
                                                                      int64_t npad, int64_t size,
                                                                      int full) {
  __shared__ uint64_t S[SORT_CAP];
  const int chunk = npad < SORT_CAP ? (int)npad : SORT_CAP;
  uint64_t* kq = keys + (int64_t)blockIdx.y * npad + (int64_t)blockIdx.x * chunk;
  const int64_t gbase = (int64_t)blockIdx.x * chunk;
  for (int i = threadIdx.x; i < chunk; i += blockDim.x) S[i] = kq[i];
  __syncthreads();
  if (full) {
    for (int64_t sz = 2; sz <= chunk; sz <<= 1) {
      for (int stride = (int)(sz >> 1); stride > 0; stride >>= 1) {
        for (int i = threadIdx.x; i < (chunk >> 1); i += blockDim.x) {
          const int lo = 2 * i - (i & (stride - 1));
          const int hi = lo + stride;
          const bool desc = ((gbase + lo) & sz) == 0;
          const uint64_t a = S[lo], b = S[hi];
          if ((a < b) == desc) { S[lo] = b; S[hi] = a; }
        }
        __syncthreads();
      }
    }
  } else {
    for (int stride = chunk >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < (chunk >> 1); i += blockDim.x) {
        const int lo = 2 * i - (i & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((gbase + lo) & size) == 0;
        const uint64_t a = S[lo], b = S[hi];
        if ((a < b) == desc) { S[lo] = b; S[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < chunk; i += blockDim.x) kq[i] = S[i];
}

// One compare-exchange stage with stride >= SORT_CAP.  grid = (blocks, nq).
__global__ void bitonic_global_kernel(uint64_t* __restrict__ keys, int64_t npad, int64_t size,
                                      int64_t stride) {
  uint64_t* kq = keys + (int64_t)blockIdx.y * npad;
  const int64_t half = npad >> 1;
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += gs) {
    const int64_t lo = 2 * i - (i & (stride - 1));
    const int64_t hi = lo + stride;
    const bool desc = (lo & size) == 0;
    const uint64_t a = kq[lo], b = kq[hi];
    if ((a < b) == desc) { kq[lo] = b; kq[hi] = a; }
  }
}

// grid = (blocks, nq)
__global__ void keys_emit_kernel(const uint64_t* __restrict__ keys, int64_t npad, int k_out,
                                 int count, int64_t row_offset, float* __restrict__ out_scores,
                                 int64_t* __restrict__ out_rows) {
  const int qi = blockIdx.y;
  const uint64_t* kq = keys + (int64_t)qi * npad;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < k_out; i += gridDim.x * blockDim.x) {
    if (i < count) {
      const uint64_t key = kq[i];
      out_scores[(int64_t)qi * k_out + i] = key_score((uint32_t)(key >> 32));
      out_rows[(int64_t)qi * k_out + i] = row_offset + (int64_t)(uint32_t)key;
    } else {
      out_scores[(int64_t)qi * k_out + i] = -__builtin_inff();
      out_rows[(int64_t)qi * k_out + i] = -1;
    }
  }
}

}  // namespace svs
