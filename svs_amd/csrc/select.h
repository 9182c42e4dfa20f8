// Top-k stage: replaces svs.util.get_top_k (reference src/svs/util.py:190-203 --
// np.argpartition + sorted(reverse=True)) on the device.
//
// Everything works on the unique 64-bit keys of keys.h, so "the k largest keys,
// descending" IS the reference's output order (score desc, index desc) and does
// not depend on scheduling.  Byte/integer work over the score vector (4 B per
// corpus row, L2/MALL resident right after the score stage); no matrix cores.
//
// Path A (k <= SEL_KMAX), three launches:
//   1. window histogram: the top 16 bits of the orderable score key are a
//      log-linear binning of the floats (128 bins per octave).  A 4096-bin
//      window [2^-31, 2.0] (scores above 2 clamp into the top bin -- still
//      monotone) is counted per workgroup in LDS and flushed to a global
//      histogram.  One pass, no assumption on the score distribution beyond
//      "the k best are positive"; otherwise step 3 falls back (exactly).
//   2. filter: every workgroup suffix-scans the histogram for the bin holding
//      the k-th best, then compacts the keys at or above that bin
//      (wave-aggregated appends).  Typically k + a few dozen survivors.
//   3. final, one workgroup: <= SORT_CAP survivors are sorted in LDS (bitonic);
//      up to CAND_CAP go through an in-LDS 64-bit radix select first; if the
//      window failed (fewer than k scores inside it, or a bin so crowded that
//      the survivors overflow) the same radix select runs over the raw score
//      vector -- slow (one CU), always exact, ties resolved by row bits.
//      It then zeroes the histogram for the next search (no memset launch).
// Path D (n <= SORT_CAP): the final workgroup reads the scores directly.
// Path B (k > SEL_KMAX): all n keys are sorted by a global bitonic network (LDS
//   for strides < SORT_CAP, one launch per larger stride), e.g. the reference's
//   "rank the whole KB" call (n = 10,548).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keys.h"

namespace svs {

constexpr int SEL_KMAX = 2048;    // path A handles k <= SEL_KMAX
constexpr int SORT_CAP = 4096;    // keys sorted in LDS by one workgroup (32 KiB)
constexpr int CAND_CAP = 32768;   // survivors of the filter, per query
constexpr int WBINS = 4096;       // window histogram bins
constexpr uint32_t WTOP = 0xC000u;              // key16 of 2.0f
constexpr uint32_t WBASE = WTOP - (WBINS - 1);  // key16 of ~2^-31
constexpr int FA_THREADS = 512;   // histogram / filter workgroup
constexpr int SORT_THREADS = 1024;  // global bitonic sort (path B) workgroup
constexpr int FINAL_THREADS = 256;  // final kernel workgroup
constexpr int RS_BINS = 2048;     // in-LDS radix select: 11 bits per pass
constexpr int PICK_MAX_PER = 16;

// per-query scratch: SelHeader followed by WBINS histogram words (all zero between searches)
struct SelHeader {
  uint32_t n_cand;  // keys appended by the filter (may exceed CAND_CAP: then invalid)
  uint32_t flag;    // 0 ok, 1 fewer than k scores inside the window
  uint32_t pad0, pad1;
};
constexpr int SCR_WORDS = sizeof(SelHeader) / 4 + WBINS;

__device__ __forceinline__ int window_bin(uint32_t key32) {
  const uint32_t k16 = key32 >> 16;
  if (k16 < WBASE) return -1;
  return (int)((k16 < WTOP ? k16 : WTOP) - WBASE);
}

// The first ACT threads of a workgroup find, from a histogram of `bins` buckets
// (bins / ACT <= PICK_MAX_PER), the bucket holding the k_rem-th largest element counting
// from the top.  *b_out = bucket (0xffffffff if the histogram holds fewer than
// k_rem elements), *k_out = rank inside it (1-based from the bucket's top).
// Every thread of the workgroup must call this (workgroup barriers inside);
// `sh` is ACT+2 words of LDS.
// The scan over the per-thread sums runs inside each wave on cross-lane moves (no LDS, no barrier)
// and meets the other waves' totals once: three workgroup barriers per call where the Hillis-Steele
// scan over LDS took 2 log2(ACT) + 3 (every workgroup of the filter kernel and every radix pass of the
// final / k-th-value kernels pays for this call).
template <int ACT>
__device__ __forceinline__ void pick_bucket(const uint32_t* hist, int bins, uint32_t k_rem,
                                            uint32_t* sh, uint32_t* b_out, uint32_t* k_out) {
  static_assert(ACT % 64 == 0 && ACT / 64 <= 16, "whole waves");
  const int tid = threadIdx.x;
  const bool act = tid < ACT;
  const int per = bins / ACT;
  // thread t owns buckets [bins - (t+1)*per, bins - t*per): t = 0 is the top
  uint32_t loc[PICK_MAX_PER];
  uint32_t sum = 0;
  const int hi = bins - tid * per;
  if (act) {
    for (int i = 0; i < per; ++i) {
      loc[i] = hist[hi - 1 - i];  // descending bucket order
      sum += loc[i];
    }
  }
  const int lane = tid & 63, w = tid >> 6;
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (act && lane == 63) sh[w] = incl;
  if (tid == 0) sh[ACT] = 0xffffffffu;
  __syncthreads();
  if (act) {
    for (int i = 0; i < w; ++i) incl += sh[i];
    const uint32_t excl = incl - sum;
    if (excl < k_rem && k_rem <= incl) {
      uint32_t c = excl;
      for (int i = 0; i < per; ++i) {
        if (c + loc[i] >= k_rem) {
          sh[ACT] = (uint32_t)(hi - 1 - i);
          sh[ACT + 1] = k_rem - c;
          break;
        }
        c += loc[i];
      }
    }
  }
  __syncthreads();
  *b_out = sh[ACT];
  *k_out = sh[ACT + 1];
  __syncthreads();
}

// k-th largest of ONE 32-bit value per thread of a 256-thread workgroup (k <= 256; values of 0 mean "none").  The values go
// through the LDS once; wave 0 then holds four of them per lane and bisects the key space bit by bit -- 32 rounds of four
// compares, four ballots and scalar counting: no atomics, no further barrier.  Returns 0 when fewer than k values are non-zero.
// Every thread of the workgroup must call this; `sh` is 256 + 1 words of LDS.
__device__ __forceinline__ uint32_t block_kth_of_thread_values(uint32_t v, uint32_t k, uint32_t* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint32_t a = sh[threadIdx.x], b = sh[threadIdx.x + 64], c = sh[threadIdx.x + 128], d = sh[threadIdx.x + 192];
    uint32_t T = 0;
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t cand = T | (1u << bit);
      const uint32_t cnt = (uint32_t)(__builtin_popcountll(__ballot(a >= cand)) + __builtin_popcountll(__ballot(b >= cand)) +
                                      __builtin_popcountll(__ballot(c >= cand)) + __builtin_popcountll(__ballot(d >= cand)));
      if (cnt >= k) T = cand;   // (wave-uniform)
    }
    if (threadIdx.x == 0) sh[256] = T;
  }
  __syncthreads();
  const uint32_t T = sh[256];
  __syncthreads();
  return T;
}

// Each thread of the histogram / filter kernels owns SEL_VPT float4 groups of
// the score vector, all requested before the first is used (the vector is
// L2/MALL resident, so the pass is latency-, not bandwidth-bound).
constexpr int SEL_VPT = 4;
typedef float f32x4_sel __attribute__((ext_vector_type(4)));
typedef float sel_v4f __attribute__((ext_vector_type(4)));

// ---- path A, launch 1.  grid = (blocks, nq); score_stride % 4 == 0 ------------
__global__ __launch_bounds__(FA_THREADS) void select_window_hist_kernel(
    const float* __restrict__ scores, int64_t n, int64_t score_stride,
    uint32_t* __restrict__ scratch) {
  __shared__ uint32_t lh[WBINS];
  const int qi = blockIdx.y;
  const float* s = scores + (int64_t)qi * score_stride;
  const sel_v4f* s4 = (const sel_v4f*)s;
  uint32_t* hist = scratch + (int64_t)qi * SCR_WORDS + sizeof(SelHeader) / 4;
  const int64_t n4 = (n + 3) >> 2;  // the allocation is padded to a multiple of 4
  const int64_t base = ((int64_t)blockIdx.x * SEL_VPT) * FA_THREADS + threadIdx.x;
  sel_v4f v[SEL_VPT];
#pragma unroll
  for (int j = 0; j < SEL_VPT; ++j) {
    const int64_t i4 = base + (int64_t)j * FA_THREADS;
    v[j] = i4 < n4 ? s4[i4] : (sel_v4f){0.f, 0.f, 0.f, 0.f};
  }
  for (int i = threadIdx.x; i < WBINS; i += FA_THREADS) lh[i] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < SEL_VPT; ++j) {
    const int64_t i = (base + (int64_t)j * FA_THREADS) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (i + e < n) {
        const int b = window_bin(score_key(v[j][e]));
        if (b >= 0) atomicAdd(&lh[b], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < WBINS; i += FA_THREADS) {
    const uint32_t c = lh[i];
    if (c) atomicAdd(&hist[i], c);
  }
}

// ---- path A, launch 2.  grid = (blocks, nq); cand is [nq][CAND_CAP] ----------
__global__ __launch_bounds__(FA_THREADS) void select_window_filter_kernel(
    const float* __restrict__ scores, int64_t n, int64_t score_stride, uint32_t k,
    uint32_t* __restrict__ scratch, uint64_t* __restrict__ cand) {
  __shared__ uint32_t sh[FA_THREADS + 2];
  const int qi = blockIdx.y;
  const float* s = scores + (int64_t)qi * score_stride;
  const sel_v4f* s4 = (const sel_v4f*)s;
  SelHeader* hdr = (SelHeader*)(scratch + (int64_t)qi * SCR_WORDS);
  const uint32_t* hist = scratch + (int64_t)qi * SCR_WORDS + sizeof(SelHeader) / 4;
  const int64_t n4 = (n + 3) >> 2;
  const int64_t base = ((int64_t)blockIdx.x * SEL_VPT) * FA_THREADS + threadIdx.x;
  sel_v4f v[SEL_VPT];
#pragma unroll
  for (int j = 0; j < SEL_VPT; ++j) {   // requested before the histogram scan: the latency hides under it
    const int64_t i4 = base + (int64_t)j * FA_THREADS;
    v[j] = i4 < n4 ? s4[i4] : (sel_v4f){0.f, 0.f, 0.f, 0.f};
  }
  uint32_t bstar, krank;
  pick_bucket<FA_THREADS>(hist, WBINS, k, sh, &bstar, &krank);
  if (bstar == 0xffffffffu) {  // fewer than k scores inside the window: exact fallback in the final kernel
    if (blockIdx.x == 0 && threadIdx.x == 0) hdr->flag = 1;
    return;
  }
  uint64_t* cq = cand + (int64_t)qi * CAND_CAP;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < SEL_VPT; ++j) {
    const int64_t i = (base + (int64_t)j * FA_THREADS) * 4;
    uint32_t key[4];
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      key[e] = 0;
      if (i + e < n) {
        const uint32_t kk = score_key(v[j][e]);
        if (window_bin(kk) >= (int)bstar) {
          key[e] = kk;
          ++cnt;
        }
      }
    }
    // wave-aggregated append: one atomic per wave per group that has survivors
    const unsigned long long m = __ballot(cnt > 0);
    if (m) {
      // exclusive prefix of cnt over the wave
      int incl = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      const int total = __shfl(incl, 63, 64);
      uint32_t wbase = 0;
      if (lane == 0) wbase = atomicAdd(&hdr->n_cand, (uint32_t)total);
      wbase = __shfl(wbase, 0, 64);
      uint32_t slot = wbase + (uint32_t)(incl - cnt);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (key[e]) {   // a real key is never 0 (score_key's minimum is 0x007fffff)
          if (slot < (uint32_t)CAND_CAP) cq[slot] = ((uint64_t)key[e] << 32) | (uint32_t)(i + e);
          ++slot;
        }
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

// Lists of up to one key per thread (m <= blockDim.x): ordered WITHOUT sorting.  Real keys are unique (the row number
// is their low half), so a key's position in the descending order is the number of keys above it: every thread
// counts that for its own key over the whole list -- 16-byte LDS reads of ONE address by all lanes (a broadcast:
// no bank conflicts, no barrier between the steps) -- and stores its answer where it belongs.  A bitonic network
// over the same 256 slots is 36 barrier-separated compare-exchange steps whose 8-byte accesses at strides 1-16
// collide pairwise (28 % of select_final_kernel's LDS cycles in round 4's counter pass); this is one pass.
// Keys equal to 0 are padding or struck-out candidates (a real key is never 0) and are not emitted; positions from
// the number of real keys (>= count in every mode: see the callers) up to k are filled as emit_topk fills them.
__device__ __forceinline__ void emit_by_rank(const uint64_t* S, int m, int count, int k, int64_t row_offset,
                                             float* __restrict__ out_scores, int64_t* __restrict__ out_rows) {
  const uint64_t a = (int)threadIdx.x < m ? S[threadIdx.x] : 0ull;
  const ulonglong2* S2 = reinterpret_cast<const ulonglong2*>(S);
  int above = 0;
#pragma unroll 4
  for (int j = 0; j < (m >> 1); ++j) {
    const ulonglong2 p = S2[j];
    above += (p.x > a ? 1 : 0) + (p.y > a ? 1 : 0);
  }
  const int n_real = __syncthreads_count(a != 0ull);
  if (a != 0ull && above < count) {
    out_scores[above] = key_score((uint32_t)(a >> 32));
    out_rows[above] = row_offset + (int64_t)(uint32_t)a;
  }
  for (int i = (n_real < count ? n_real : count) + (int)threadIdx.x; i < k; i += blockDim.x) {
    out_scores[i] = -__builtin_inff();
    out_rows[i] = -1;
  }
}

// One workgroup: exact k-th largest of the M unique 64-bit keys key_at(0..M),
// Histogram increment with the wave's dominant bin aggregated: candidate keys of one query share their
// leading bits, so in the upper radix passes (and for the bulk of a prefix's scores) most lanes of a wave hit
// ONE bin -- 64 LDS atomics on one address are served one after the other.  The lanes that share the first
// active lane's bin are counted with a ballot and added by that lane alone; the rest add for themselves.
// Must be called by all lanes of the wave (valid == false: nothing to add).
__device__ __forceinline__ void hist_add_wave(uint32_t* lh, bool valid, uint32_t bin) {
  const unsigned long long act = __ballot(valid);
  if (act == 0) return;
  const int first = __builtin_ctzll(act);
  const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, first);
  const unsigned long long same = __ballot(valid && bin == b0);
  const int lane = (int)(threadIdx.x & 63);
  if (lane == first) atomicAdd(&lh[b0], (uint32_t)__builtin_popcountll(same));
  else if (valid && bin != b0) atomicAdd(&lh[bin], 1u);
}

// MSB-first radix select with an LDS histogram (11 bits per pass, early exit
// as soon as a whole bucket is taken).  Returns T such that exactly k keys
// satisfy key >= T.  lh: RS_BINS words, sh: 256+2 words of LDS.
template <class KeyAt>
__device__ __forceinline__ uint64_t block_radix_select(KeyAt key_at, int64_t M, uint32_t k,
                                                       uint32_t* lh, uint32_t* sh) {
  uint64_t prefix = 0, pmask = 0;
  uint32_t k_rem = k;
  for (int shift = 53; shift >= 0; shift -= 11) {   // 53,42,31,20,9 then the low 9 bits
    for (int i = threadIdx.x; i < RS_BINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < M; i += blockDim.x) {
      const uint64_t key = key_at(i);
      if ((key & pmask) == prefix) atomicAdd(&lh[(uint32_t)(key >> shift) & (RS_BINS - 1)], 1u);
    }
    __syncthreads();
    uint32_t b, k2;
    pick_bucket<256>(lh, RS_BINS, k_rem, sh, &b, &k2);
    const uint32_t in_bucket = lh[b];
    __syncthreads();
    prefix |= (uint64_t)b << shift;
    pmask |= (uint64_t)(RS_BINS - 1) << shift;
    if (in_bucket == k2) return prefix;  // the whole bucket is selected
    k_rem = k2;
    if (shift == 9) {  // last pass: the remaining 9 low bits
      for (int i = threadIdx.x; i < RS_BINS; i += blockDim.x) lh[i] = 0;
      __syncthreads();
      for (int64_t i = threadIdx.x; i < M; i += blockDim.x) {
        const uint64_t key = key_at(i);
        if ((key & pmask) == prefix) atomicAdd(&lh[(uint32_t)key & 511u], 1u);
      }
      __syncthreads();
      pick_bucket<256>(lh, RS_BINS, k_rem, sh, &b, &k2);
      return prefix | b;
    }
  }
  return prefix;
}

// The same select on keys held in REGISTERS (thread t owns keys t, t + T, t + 2T, ... of the
// list, `mine` of them valid).  For the fused epilogue's lists (thousands of candidates in
// global memory): read once with independent loads, then every pass is LDS-only.  With the
// list left in memory each pass walked it with one dependent L2 load per iteration
// (~12 us per pass, 35 us per query).
template <int NPT>
__device__ __forceinline__ uint64_t block_radix_select_regs(const uint64_t (&kr)[NPT], int mine, uint32_t k,
                                                            uint32_t* lh, uint32_t* sh) {
  uint64_t prefix = 0, pmask = 0;
  uint32_t k_rem = k;
  for (int shift = 53; shift >= 0; shift -= 11) {   // 53,42,31,20,9 then the low 9 bits
    for (int i = threadIdx.x; i < RS_BINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPT; ++j)
      hist_add_wave(lh, j < mine && (kr[j] & pmask) == prefix, (uint32_t)(kr[j] >> shift) & (RS_BINS - 1));
    __syncthreads();
    uint32_t b, k2;
    pick_bucket<256>(lh, RS_BINS, k_rem, sh, &b, &k2);
    const uint32_t in_bucket = lh[b];
    __syncthreads();
    prefix |= (uint64_t)b << shift;
    pmask |= (uint64_t)(RS_BINS - 1) << shift;
    if (in_bucket == k2) return prefix;  // the whole bucket is selected
    k_rem = k2;
    if (shift == 9) {
      for (int i = threadIdx.x; i < RS_BINS; i += blockDim.x) lh[i] = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NPT; ++j)
        hist_add_wave(lh, j < mine && (kr[j] & pmask) == prefix, (uint32_t)kr[j] & 511u);
      __syncthreads();
      pick_bucket<256>(lh, RS_BINS, k_rem, sh, &b, &k2);
      return prefix | b;
    }
  }
  return prefix;
}
constexpr int FINAL_REG_KEYS = 32;   // per thread: lists of up to FINAL_THREADS * 32 = 8192 candidates
constexpr int FINAL_DIRECT = 1024;    // lists this short are sorted as they are

// ---- path A launch 3 / path D.  grid = nq, one workgroup per query ------------
// mode 0: path A (candidates from the filter); mode 1: path D (n <= SORT_CAP);
// mode 3: candidates from the fused GEMM epilogue (no score vector exists): an
// overflowed list cannot be repaired here, so the query is marked for the host
// (out_rows[0] = -2) and re-run through the materialised path.
__global__ __launch_bounds__(FINAL_THREADS) void select_final_kernel(
    const float* __restrict__ scores, int64_t n, int64_t score_stride, int k_out, int count,
    int mode, uint32_t* __restrict__ scratch, uint64_t* cand,
    int64_t row_offset, float* __restrict__ out_scores, int64_t* __restrict__ out_rows,
    const uint32_t* __restrict__ dead_bits = nullptr) {
  __shared__ __attribute__((aligned(16))) uint64_t S[SORT_CAP];
  __shared__ uint32_t lh[WBINS];   // window histogram (WBINS) or radix-select histogram (RS_BINS <= WBINS)
  __shared__ uint32_t sh[256 + 2];
  __shared__ uint32_t s_cnt;
  __shared__ uint32_t s_dead;
  const int qi = blockIdx.x;
  const float* s = scores + (int64_t)qi * score_stride;
  float* os = out_scores + (int64_t)qi * k_out;
  int64_t* orow = out_rows + (int64_t)qi * k_out;
  int m;
  if (mode == 1) {
    m = next_pow2((int)n < 2 ? 2 : (int)n);
    for (int i = threadIdx.x; i < m; i += blockDim.x) S[i] = i < n ? make_key(s[i], (uint32_t)i) : 0ull;
  } else {
    SelHeader* hdr = (SelHeader*)(scratch + (int64_t)qi * SCR_WORDS);
    uint64_t* cq = cand + (int64_t)qi * CAND_CAP;
    const uint32_t flag = hdr->flag;
    const uint32_t n_cand = hdr->n_cand;
    // Tombstoned rows (svs_index_mask_rows) on the fused path: the GEMM epilogue does not know
    // them, so their candidates are struck out here (key 0 sorts below every real key); the
    // thresholds came from a prefix whose masked rows were already at -inf, so at least `count`
    // live candidates remain.
    uint32_t n_live = n_cand;
    if (mode == 3 && dead_bits && n_cand <= (uint32_t)CAND_CAP) {
      if (threadIdx.x == 0) s_dead = 0;
      __syncthreads();
      uint32_t dropped = 0;
      for (uint32_t i = threadIdx.x; i < n_cand; i += blockDim.x) {
        const uint32_t row = (uint32_t)cq[i];
        if ((dead_bits[row >> 5] >> (row & 31)) & 1u) {
          cq[i] = 0ull;
          ++dropped;
        }
      }
      if (dropped) atomicAdd(&s_dead, dropped);
      __threadfence_block();
      __syncthreads();
      n_live = n_cand - s_dead;
    }
    if (mode == 3 && (n_cand > (uint32_t)CAND_CAP || n_live < (uint32_t)count)) {
      __syncthreads();
      uint32_t* w0 = scratch + (int64_t)qi * SCR_WORDS;
      for (int i = threadIdx.x; i < (int)(sizeof(SelHeader) / 4); i += blockDim.x) w0[i] = 0;   // (mode 3 only: see below)
      for (int i = threadIdx.x; i < k_out; i += blockDim.x) {
        os[i] = -__builtin_inff();
        orow[i] = -2;
      }
      return;
    }
    if (flag == 0 && n_cand <= (uint32_t)FINAL_DIRECT) {
      m = next_pow2((int)n_cand < 2 ? 2 : (int)n_cand);
      for (int i = threadIdx.x; i < m; i += blockDim.x) S[i] = i < (int)n_cand ? cq[i] : 0ull;
    } else {
      if (threadIdx.x == 0) s_cnt = 0;
      m = next_pow2(count < 2 ? 2 : count);
      if (flag == 0 && n_cand <= (uint32_t)(FINAL_THREADS * FINAL_REG_KEYS)) {
        // Lists of up to 8192 candidates (the fused GEMM epilogue leaves ~k n / prefix of them): read ONCE into
        // registers, then ONE histogram pass over the window bins of the score (128 bins per octave, the binning of
        // path A's launch 1) finds the bin b* of the count-th best; everything at or above b* -- count plus the few
        // dozen keys that share b* -- goes to the LDS sort.  (Round 3: five to six 11-bit radix passes over the
        // 64-bit keys, whose leading bits the candidates of one query all share: most lanes of a wave added to ONE
        // LDS word per pass, 46-58 % of the kernel's LDS cycles were conflict cycles.)  A window that does not hold
        // `count` keys (negative or zero scores among the best) or a bin too crowded for the sort falls back to
        // that radix select on the same registers: always exact.
        uint64_t kr[FINAL_REG_KEYS];
        const int mine = (int)n_cand > (int)threadIdx.x ? ((int)n_cand - (int)threadIdx.x + FINAL_THREADS - 1) / FINAL_THREADS : 0;
#pragma unroll
        for (int j = 0; j < FINAL_REG_KEYS; ++j) kr[j] = j < mine ? cq[threadIdx.x + j * FINAL_THREADS] : 0ull;
        bool done = false;
        __syncthreads();   // s_cnt = 0 is visible
        if (count <= FINAL_THREADS) {
          // (count <= 256) As in prefix_kth_kernel: the count-th largest of the per-thread maxima of the SCORE half of the keys
          // bounds the count-th best score from below; the few hundred keys at or above it are listed by ballot + prefix count
          // (one LDS atomic per wave) and sorted.  No histogram, no scattered LDS atomics.
          uint32_t tmax = 0;
#pragma unroll
          for (int j = 0; j < FINAL_REG_KEYS; ++j) {
            const uint32_t sk = (uint32_t)(kr[j] >> 32);
            tmax = sk > tmax ? sk : tmax;
          }
          const uint32_t L = block_kth_of_thread_values(tmax, (uint32_t)count, sh);
          if (L != 0u) {
            const int lane = (int)(threadIdx.x & 63);
            uint32_t cw = 0;
#pragma unroll
            for (int j = 0; j < FINAL_REG_KEYS; ++j) cw += (uint32_t)__builtin_popcountll(__ballot((uint32_t)(kr[j] >> 32) >= L));
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt, cw);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
            for (int j = 0; j < FINAL_REG_KEYS; ++j) {
              const bool hit = (uint32_t)(kr[j] >> 32) >= L;
              const unsigned long long mm = __ballot(hit);
              if (hit) {
                const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0));
                if (slot < (uint32_t)SORT_CAP) S[slot] = kr[j];
              }
              base += (uint32_t)__builtin_popcountll(mm);
            }
            __syncthreads();
            const uint32_t total = s_cnt;
            __syncthreads();
            if (total <= (uint32_t)SORT_CAP) {
              m = next_pow2((int)total < 2 ? 2 : (int)total);
              for (int i = (int)total + threadIdx.x; i < m; i += blockDim.x) S[i] = 0ull;
              done = true;
            } else if (threadIdx.x == 0) {
              s_cnt = 0;
            }
          }
        }
        if (!done) {
        for (int i = threadIdx.x; i < WBINS; i += blockDim.x) lh[i] = 0;
        __syncthreads();   // s_cnt = 0 and the cleared histogram are visible
#pragma unroll
        for (int j = 0; j < FINAL_REG_KEYS; ++j) {
          const int b = j < mine ? window_bin((uint32_t)(kr[j] >> 32)) : -1;   // (a struck-out candidate, key 0, is outside the window)
          if (b >= 0) atomicAdd(&lh[b], 1u);
        }
        __syncthreads();
        uint32_t bstar, k2;
        pick_bucket<FINAL_THREADS>(lh, WBINS, (uint32_t)count, sh, &bstar, &k2);
        if (bstar != 0xffffffffu) {
#pragma unroll
          for (int j = 0; j < FINAL_REG_KEYS; ++j)
            if (j < mine && window_bin((uint32_t)(kr[j] >> 32)) >= (int)bstar) {
              const uint32_t slot = atomicAdd(&s_cnt, 1u);
              if (slot < (uint32_t)SORT_CAP) S[slot] = kr[j];
            }
          __syncthreads();
          const uint32_t total = s_cnt;
          __syncthreads();
          if (total <= (uint32_t)SORT_CAP) {
            m = next_pow2((int)total < 2 ? 2 : (int)total);
            for (int i = (int)total + threadIdx.x; i < m; i += blockDim.x) S[i] = 0ull;
            done = true;
          }
        }
        }
        if (!done) {
          if (threadIdx.x == 0) s_cnt = 0;
          for (int i = threadIdx.x; i < m; i += blockDim.x) S[i] = 0ull;
          __syncthreads();
          const uint64_t T = block_radix_select_regs(kr, mine, (uint32_t)count, lh, sh);
#pragma unroll
          for (int j = 0; j < FINAL_REG_KEYS; ++j)
            if (j < mine && kr[j] >= T) S[atomicAdd(&s_cnt, 1u)] = kr[j];
        }
      } else if (flag == 0 && n_cand <= (uint32_t)CAND_CAP) {
        for (int i = threadIdx.x; i < m; i += blockDim.x) S[i] = 0ull;
        auto key_at = [&](int64_t i) { return cq[i]; };
        const uint64_t T = block_radix_select(key_at, (int64_t)n_cand, (uint32_t)count, lh, sh);
        for (int64_t i = threadIdx.x; i < (int64_t)n_cand; i += blockDim.x) {
          const uint64_t key = cq[i];
          if (key >= T) S[atomicAdd(&s_cnt, 1u)] = key;
        }
      } else {  // exact fallback over the raw scores (one CU; rare)
        for (int i = threadIdx.x; i < m; i += blockDim.x) S[i] = 0ull;
        auto key_at = [&](int64_t i) { return make_key(s[i], (uint32_t)i); };
        const uint64_t T = block_radix_select(key_at, n, (uint32_t)count, lh, sh);
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
          const uint64_t key = make_key(s[i], (uint32_t)i);
          if (key >= T) S[atomicAdd(&s_cnt, 1u)] = key;
        }
      }
    }
    // leave the scratch zeroed for the next search on this context (after every
    // thread has read the header)
    __syncthreads();
    uint32_t* w = scratch + (int64_t)qi * SCR_WORDS;
    // (mode 3: the fused epilogue touched the header only -- the 16 KiB histogram behind it is still zero; zeroing it anyway
    //  was 16.8 MB of stores per 1024-query call)
    const int nz = mode == 3 ? (int)(sizeof(SelHeader) / 4) : SCR_WORDS;
    for (int i = threadIdx.x; i < nz; i += blockDim.x) w[i] = 0;
  }
  __syncthreads();
  if (m <= FINAL_THREADS) {   // the usual case: k = 100 leaves 120-200 keys here on every route
    emit_by_rank(S, m, count, k_out, row_offset, os, orow);
  } else {
    bitonic_sort_lds_desc(S, m);
    emit_topk(S, count, k_out, row_offset, os, orow);
  }
}

// ---- fused path, thresholds: thr[q] = exact k-th best score of scores[q][0 .. n) ----
// One workgroup per query.  Replaces histogram + filter + final (three launches, a sorted list nobody
// reads) when only the k-th value is wanted.  Prefixes of up to 16,384 rows are read ONCE into registers
// (64 score keys per thread); then
//   1. one histogram pass over the window bins of path A (the top 16 key bits, 128 bins per octave,
//      [2^-31, 2]; negative scores are outside and cost nothing) finds the bin b* of the k-th best;
//   2. the keys inside b* (a handful) are listed in LDS and the one of rank k' is found by counting.
// (Round 3 ran three 11/11/10-bit radix passes over all keys: the first pass's bins are the sign, the
// exponent and two mantissa bits, so the 16,384 scores of a prefix fell into ~20 LDS words and the atomics
// of a wave were served one lane after the other -- 50 % of the kernel's LDS cycles were conflict cycles.)
// A window that holds fewer than k keys (negative k-th best) or a bin with more than PK_LIST keys falls
// back to that radix select: always exact.
constexpr int PK_LIST = 1024;
__global__ __launch_bounds__(FINAL_THREADS, 4) void prefix_kth_kernel(
    const float* __restrict__ scores, int64_t n, int64_t score_stride, int k, float* __restrict__ thr) {
  __shared__ uint32_t lh[WBINS];
  __shared__ uint32_t sh[256 + 2];
  __shared__ uint32_t list[PK_LIST];
  __shared__ uint32_t s_cnt;
  const float* s = scores + (int64_t)blockIdx.x * score_stride;
  uint32_t prefix = 0, pmask = 0, k_rem = (uint32_t)k;
  const int shifts[3] = {21, 10, 0};
  constexpr int PK_REGS = 64;   // score keys per thread held in registers: prefixes of up to 16,384 rows are read ONCE
  if (n <= (int64_t)FINAL_THREADS * PK_REGS && (score_stride & 3) == 0 && (((uintptr_t)scores) & 15) == 0) {
    uint32_t kr[PK_REGS];
#pragma unroll
    for (int j = 0; j < PK_REGS / 4; ++j) {
      const int64_t i = ((int64_t)j * FINAL_THREADS + threadIdx.x) * 4;
      f32x4_sel q4 = {0.f, 0.f, 0.f, 0.f};
      if (i + 3 < n) q4 = *(const f32x4_sel*)(s + i);
      else {
        if (i < n) q4.x = s[i];
        if (i + 1 < n) q4.y = s[i + 1];
        if (i + 2 < n) q4.z = s[i + 2];
      }
      // key 0 (below every real score's key) marks "no element"
      kr[4 * j] = i < n ? score_key(q4.x) : 0u;
      kr[4 * j + 1] = i + 1 < n ? score_key(q4.y) : 0u;
      kr[4 * j + 2] = i + 2 < n ? score_key(q4.z) : 0u;
      kr[4 * j + 3] = i + 3 < n ? score_key(q4.w) : 0u;
    }
    if (threadIdx.x == 0) s_cnt = 0;
    // 0. (k <= 256) A lower bound without a histogram: the k-th largest of the 256 per-thread maxima is >= k keys' worth of
    //    evidence that the k-th best is at least that large (each maximum is a different element).  The ~1.5 % of the keys
    //    at or above it (a couple of hundred of 16,384) are listed by ballot + prefix count -- four LDS atomics per
    //    workgroup, one per wave -- and the k-th is found among them by counting.  No scattered LDS atomics at all: the
    //    histogram passes (half of whose LDS cycles were bank-conflict cycles, profiles/r3_cfg2_summary.json,
    //    r4 with the window histogram: still 0.50 -- 64 random bins per wave instruction hit 32 banks) remain as the
    //    fallback for k > 256, for keys that are mostly equal, and for prefixes with fewer than k positive keys.
    if (k <= FINAL_THREADS) {
      uint32_t tmax = 0;
#pragma unroll
      for (int j = 0; j < PK_REGS; ++j) tmax = kr[j] > tmax ? kr[j] : tmax;
      const uint32_t L = block_kth_of_thread_values(tmax, (uint32_t)k, sh);
      if (L != 0u) {
        const int lane = (int)(threadIdx.x & 63);
        uint32_t cw = 0;
#pragma unroll
        for (int j = 0; j < PK_REGS; ++j) cw += (uint32_t)__builtin_popcountll(__ballot(kr[j] >= L));
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_cnt, cw);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
        for (int j = 0; j < PK_REGS; ++j) {
          const bool hit = kr[j] >= L;
          const unsigned long long m = __ballot(hit);
          if (hit) {
            const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (slot < (uint32_t)PK_LIST) list[slot] = kr[j];
          }
          base += (uint32_t)__builtin_popcountll(m);
        }
        __syncthreads();
        const uint32_t cnt = s_cnt;
        if (cnt <= (uint32_t)PK_LIST) {
          for (uint32_t t = threadIdx.x; t < cnt; t += blockDim.x) {
            const uint32_t mine = list[t];
            uint32_t gt = 0, ge = 0;
            for (uint32_t u = 0; u < cnt; ++u) {
              const uint32_t o = list[u];
              gt += o > mine ? 1u : 0u;
              ge += o >= mine ? 1u : 0u;
            }
            if (gt < (uint32_t)k && (uint32_t)k <= ge) thr[blockIdx.x] = key_score(mine);
          }
          return;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_cnt = 0;   // (a crowded value: the histogram passes below sort it out)
      }
    }
    for (int i = threadIdx.x; i < WBINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PK_REGS; ++j) {
      const int b = window_bin(kr[j]);   // (key 0 is outside the window)
      if (b >= 0) atomicAdd(&lh[b], 1u);
    }
    __syncthreads();
    uint32_t bstar, k2;
    pick_bucket<FINAL_THREADS>(lh, WBINS, (uint32_t)k, sh, &bstar, &k2);
    if (bstar != 0xffffffffu) {
#pragma unroll
      for (int j = 0; j < PK_REGS; ++j)
        if (window_bin(kr[j]) == (int)bstar) {
          const uint32_t slot = atomicAdd(&s_cnt, 1u);
          if (slot < (uint32_t)PK_LIST) list[slot] = kr[j];
        }
      __syncthreads();
      const uint32_t cnt = s_cnt;
      if (cnt <= (uint32_t)PK_LIST) {
        // the key of rank k2 (1-based from the top) among the cnt keys of the bin; equal scores share a rank range
        for (uint32_t t = threadIdx.x; t < cnt; t += blockDim.x) {
          const uint32_t mine = list[t];
          uint32_t gt = 0, ge = 0;
          for (uint32_t u = 0; u < cnt; ++u) {
            const uint32_t o = list[u];
            gt += o > mine ? 1u : 0u;
            ge += o >= mine ? 1u : 0u;
          }
          if (gt < k2 && k2 <= ge) thr[blockIdx.x] = key_score(mine);
        }
        return;
      }
      __syncthreads();
    }
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = shifts[pass];
      const uint32_t bins = pass == 2 ? 1024u : (uint32_t)RS_BINS;
      for (int i = threadIdx.x; i < RS_BINS; i += blockDim.x) lh[i] = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < PK_REGS; ++j)
        if (kr[j] != 0u && (kr[j] & pmask) == prefix) atomicAdd(&lh[(kr[j] >> shift) & (bins - 1)], 1u);
      __syncthreads();
      uint32_t b, k3;
      pick_bucket<256>(lh, (int)bins, k_rem, sh, &b, &k3);
      prefix |= b << shift;
      pmask |= (bins - 1) << shift;
      k_rem = k3;
    }
    if (threadIdx.x == 0) thr[blockIdx.x] = key_score(prefix);
    return;
  }
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = shifts[pass];
    const uint32_t bins = pass == 2 ? 1024u : (uint32_t)RS_BINS;
    for (int i = threadIdx.x; i < RS_BINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)threadIdx.x * 4; i < n; i += (int64_t)blockDim.x * 4) {
      float v[4];
      if (i + 3 < n) {
        const f32x4_sel q4 = *(const f32x4_sel*)(s + i);
        v[0] = q4.x; v[1] = q4.y; v[2] = q4.z; v[3] = q4.w;
      } else {
        for (int e = 0; e < 4; ++e) v[e] = i + e < n ? s[i + e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t key = score_key(v[e]);
        if (i + e < n && (key & pmask) == prefix) atomicAdd(&lh[(key >> shift) & (bins - 1)], 1u);
      }
    }
    __syncthreads();
    uint32_t b, k2;
    pick_bucket<256>(lh, (int)bins, k_rem, sh, &b, &k2);
    prefix |= b << shift;
    pmask |= (bins - 1) << shift;
    k_rem = k2;
  }
  if (threadIdx.x == 0) thr[blockIdx.x] = key_score(prefix);
}

// ---- pairwise path: keep the strict upper triangle of an N x N score matrix ----
// S is [n][np] (np = padded row stride): entry (i, j) survives iff i < j < n; all
// others become -inf, so the flattened matrix is one score vector whose top-k are
// the reference's top pairs (src/svs/util.py:206-233: np.triu_indices(k=1), then
// get_top_k; flat index order == (i, j) lexicographic, ties largest first).
__global__ void mask_upper_triangle_kernel(float* __restrict__ S, int64_t n, int64_t np) {
  const int64_t total = n * np;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t i = t / np, j = t - i * np;
    if (j <= i || j >= n) S[t] = -__builtin_inff();
  }
}

// ---- pairwise path on large corpora: per-query candidate lists -> one global pair list ----------
// After a pair-mode GEMM over one chunk of query rows (gemm_tiled.h, TgPairs) every query q of the
// chunk has its candidates (score key << 32 | global row j, j > i) in cand[q][0 .. n_cand).  This
// kernel appends them as (score key, i, j) to the global list, drops pairs with a tombstoned row,
// flags a query whose list overflowed, and zeroes the query's header for the next chunk.
// grid = queries of the chunk.  gstate: [0] = pairs appended (may exceed cap), [1] = overflow flag.
__global__ __launch_bounds__(256) void collect_pairs_kernel(
    uint32_t* __restrict__ scratch, const uint64_t* __restrict__ cand, long long query_row0, long long first_query,
    const uint32_t* __restrict__ dead_bits, uint32_t* __restrict__ gstate, uint32_t cap,
    uint32_t* __restrict__ out_key, uint32_t* __restrict__ out_i, uint32_t* __restrict__ out_j) {
  const int q = blockIdx.x;
  SelHeader* hdr = (SelHeader*)(scratch + (int64_t)q * SCR_WORDS);
  const uint32_t n_cand = hdr->n_cand;
  const long long i = query_row0 + q;
  __syncthreads();
  if (threadIdx.x == 0) { hdr->n_cand = 0; hdr->flag = 0; }
  if (i < first_query) return;
  if (n_cand > (uint32_t)CAND_CAP) {
    if (threadIdx.x == 0) atomicExch(&gstate[1], 1u);
    return;
  }
  if (dead_bits && ((dead_bits[i >> 5] >> (i & 31)) & 1u)) return;
  const uint64_t* cq = cand + (int64_t)q * CAND_CAP;
  for (uint32_t c = threadIdx.x; c < n_cand; c += blockDim.x) {
    const uint64_t key = cq[c];
    const uint32_t j = (uint32_t)key;
    if (dead_bits && ((dead_bits[j >> 5] >> (j & 31)) & 1u)) continue;
    const uint32_t slot = atomicAdd(&gstate[0], 1u);
    if (slot < cap) {
      out_key[slot] = (uint32_t)(key >> 32);
      out_i[slot] = (uint32_t)i;
      out_j[slot] = j;
    }
  }
}

// ---- tombstones (svs_index_mask_rows) ------------------------------------------
// scores[q][row] = -inf for every masked (tombstoned) row
// (rows >= n_rows are skipped: the fused path materialises only a prefix of the corpus)
// blk_stride != 0: the score matrix is over the fused path's threshold SAMPLE -- blocks of 256 corpus rows taken every
// blk_stride rows (svs_amd.hip prefix_image): corpus row r is sample row (r / blk_stride) * 256 + r % blk_stride if it
// lies inside a block, and is not in the matrix otherwise.
__global__ void mask_dead_rows_kernel(float* __restrict__ scores, int64_t sstride, int nq,
                                      const uint32_t* __restrict__ dead, int64_t n_dead, int64_t n_rows, int64_t blk_stride) {
  const int64_t total = n_dead * nq;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = t / n_dead;
    int64_t r = dead[t - q * n_dead];
    if (blk_stride) {
      const int64_t b = r / blk_stride, off = r - b * blk_stride;
      if (off >= 256) continue;
      r = b * 256 + off;
    }
    if (r < n_rows) scores[q * sstride + r] = -__builtin_inff();
  }
}
// pairwise matrix S[n][np]: whole row and column of a masked row -> -inf (masked rows >= n: the
// matrix covers a prefix of the corpus, nothing to do)
__global__ void mask_dead_pairs_kernel(float* __restrict__ S, int64_t n, int64_t np,
                                       const uint32_t* __restrict__ dead, int64_t n_dead) {
  const int64_t total = n_dead * n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t di = t / n, j = t - di * n;
    const int64_t r = dead[di];
    if (r >= n) continue;
    S[r * np + j] = -__builtin_inff();
    S[j * np + r] = -__builtin_inff();
  }
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
