// Batched score stage for f16 and fp8 corpora: LDS-tiled MFMA GEMM
//     S[j][i] = sum_d M8/16[i,d] * Q8/16[j,d]   (f32 accumulate; x scales for fp8)
// for BASELINE.json configs[2] (1M x 1536 f16, 1024 queries) and configs[4]
// (10M x 3072 fp8, 256 queries).  EB = bytes per element: 2 -> half operands on
// v_mfma_f32_16x16x32_f16, 1 -> e4m3 operands on v_mfma_f32_16x16x128_f8f6f4 (CDNA4),
// 4 -> exact f32 on v_mfma_f32_16x16x4_f32, 16-32 queries per corpus pass (a
// k-step is always 128 BYTES per row: 32 floats, 64 halves or 128 fp8).  The reference's
// nearest analogue is its np.dot(M, M.T) (src/svs/kb.py:1651); a query batch is
// by definition a loop of np.dot(M, q) calls (src/svs/kb.py:1623).
//
// Roofline: with a BN-query panel per workgroup column the corpus is read
// ceil(nq/BN) times (from HBM once: the panels of a row tile share an XCD's L2):
// HBM-bound for small panels (BN = 32: 8 % MFMA), MFMA-bound from BN = 256 up
// (AI = 256 flop per corpus byte) -- in practice bound by the chip's power limit
// (MfmaUtil 40-46 % at 1.7-1.9 GHz).
//
// Structure (gfx950): workgroup tile = BM (128 or 256) corpus rows x BN queries, BK = 64 halves
// (exactly ONE 128-byte line per row per k-step), 8 waves.  Both operand tiles
// are staged HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, every
// wave instruction moves 8 rows x 128 B of whole lines) through a three-buffer
// ring: the DMA of steps s+1 and s+2 is in flight while step s is multiplied
// (counted vmcnt + raw s_barrier, never vmcnt(0) inside the loop).  The LDS image is
// linear (the DMA writes base + lane*16), so the bank-conflict swizzle is put on
// the SOURCE address: physical 16-byte chunk c of row r holds global chunk
// c ^ ((r >> 1) & 7), and fragment reads apply the same XOR -- conflict-free
// ds_read_b128 for every 16-lane group of v_mfma_f32_16x16x32_f16 operands.
// Wave tile = TM x TN (128 x 64 for the 256 x 256 workgroup tile): A and B fragments
// are read once per k-step and reused across the 8 x 4 MFMA tiles.  With BM = BN = 256
// the two-buffer ring fills the 160 KiB LDS (one workgroup per CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm_q16.h"
#include "fp8.h"
#include "gemv_f16.h"
#include "keys.h"

namespace svs {

typedef int i32x8 __attribute__((ext_vector_type(8)));

#ifdef TG_TRACE   // tools/gemm_trace.hip only: cycle stamps of one wave's k-loop phases
__device__ long long* tg_trace_buf;
#define TG_STAMP(slot) do { if (tg_tr) tg_trace_buf[(wave * 64 + s) * 4 + (slot)] = clock64(); } while (0)
#define TG_STAMP_BLOCK(slot) do { if (blockIdx.x == 1000 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) tg_trace_buf[8 * 64 * 4 + (threadIdx.x >> 6) * 4 + (slot)] = clock64(); } while (0)
#else
#define TG_STAMP(slot) do { } while (0)
#define TG_STAMP_BLOCK(slot) do { } while (0)
#endif

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int TG_BM = 128;  // corpus rows per workgroup tile (256 for the MFMA-bound BN = 256 panels)
constexpr int TG_BKB = 128;  // BYTES per row per k-step (one 128-byte line)
constexpr int TG_WAVES = 8;

__device__ __forceinline__ int tg_swz(int row) { return (row >> 1) & 7; }

// ---- fused top-k epilogue ------------------------------------------------------
// With B = 1024 queries the score matrix would be 4 GB per batch; instead the
// epilogue keeps, per query, only the scores that reach a threshold thr[q] known
// BEFORE the pass: the exact k-th best score of the first `prefix` corpus rows
// (a small materialised GEMM + the ordinary top-k stage).  The k-th best of any
// subset is a lower bound of the global k-th best, so every global winner passes
// (score >= thr[q]) and the result is exact; about k*n/prefix candidates per
// query survive (6k at 1M rows, prefix 16k, k = 100).  select_final_kernel then
// picks the exact k.  If a list overflows (rows ordered by similarity to the
// query) the query is marked and the host re-runs it through the materialised
// path.  (A streaming variant that learned the threshold during the pass, level
// counters + monotone cut, was exact too but flooded the atomics while hundreds
// of workgroups still saw the initial cut: 10.7 ms vs 7.0 ms unfused.)
// Layout of the per-query header matches select.h: word 0 = n_cand, word 1 = flag.
__device__ __forceinline__ void fuse_offer(uint32_t* hdr, uint64_t* cand, uint32_t cap, float thr, float v,
                                           uint32_t row) {
  if (v < thr) return;   // (a NaN score passes: it ranks largest, keys.h)
  const uint32_t slot = atomicAdd(hdr, 1u);
  if (slot < cap) cand[slot] = ((uint64_t)score_key(v) << 32) | row;
}

// Pair mode of the fused epilogue (svs_index_top_pairs on corpora too large to materialise n x n
// scores; reference src/svs/util.py:206-233 keeps the strict upper triangle i < j): the queries of
// a launch ARE corpus rows query_row0 .. (global), the row operand starts at global row row_base,
// and a score is a candidate only for global row > the query's own row (and only for queries from
// first_query on: the last query chunk overlaps the one before it).
struct TgPairs {
  long long query_row0 = 0, row_base = 0, first_query = 0;
  int on = 0;
};

// ---- epilogue shared by gemm_tiled_kernel and gemm_phased_kernel ------------------------
// acc[i][j] = one 16 x 16 f32 tile of the wave: rows row0 + wrow + 16 i + 4 g + r (r = register),
// query q0 + wq + 16 j + (lane & 15).  FUSE == false: scores [nq][sstride] are written.
// FUSE == true: scores that reach fthr[query] are offered to the query's candidate list.
// EB == 1: scores are first multiplied by rscale[row] * qscale[query] (fp8 dequantisation).
template <bool FUSE, int EB, int MT, int NT, int BM>
__device__ __forceinline__ void tg_epilogue(f32x4_t (&acc)[MT][NT], int64_t row0, int q0, int wrow, int wq, int lane,
                                            int64_t n, int nq, float* __restrict__ scores, int64_t sstride,
                                            uint32_t* __restrict__ fstate_words, int fstate_stride,
                                            uint64_t* __restrict__ fcand, uint32_t fcap,
                                            const float* __restrict__ fthr, int fthr_stride,
                                            const float* __restrict__ rscale, const float* __restrict__ qscale,
                                            const TgPairs pairs = TgPairs{}) {
  // (an opaque copy of the lane id: formed from the kernel's own `lane`, the per-lane offsets and codes of this epilogue are
  //  computed BEFORE the k loop and held across it -- at the 256 x 256 fp8 tile that is what does not fit 256 registers)
  asm volatile("" : "+v"(lane));
  const int r16 = lane & 15, g = lane >> 4;
  // fp8 dequantisation, v * (row scale * query scale), one 16-row block at a time: its four row scales and the NT query
  // scales are all that is live beside the accumulators.  (Fetched up front for all MT blocks -- 32 registers at the 256 x 256
  // tile -- the scales pushed gemm_tiled_kernel<256, true, 1, 256> to 96 spilled registers and 52 bytes of scratch.)
  if constexpr (EB == 1) {
    float qsc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int query = q0 + wq + j * 16 + r16;
      qsc[j] = query < nq ? qscale[query] : 0.f;   // (a padded query's accumulators are never read)
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f32x4_t rs;
      const int64_t ob = row0 + wrow + i * 16 + 4 * g;
      if (ob + 3 < n) rs = *(const f32x4_t*)(rscale + ob);   // row tiles start on multiples of 4
      else
        for (int r = 0; r < 4; ++r) rs[r] = rscale[ob + r < n ? ob + r : n - 1];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] *= rs[r] * qsc[j];
    }
  }
  // D layout: column (query) = lane & 15, rows 4 g + r of each 16-row tile
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int query = q0 + wq + j * 16 + r16;
    if (query < nq) {
      if constexpr (FUSE) {
        // A lane holds MT * 4 scores of ONE query here.  It counts its survivors first and
        // claims all their slots with ONE atomic (a returning atomic per surviving score made
        // each wave wait ~40 L2 round trips per tile: 24 of the 72 us a 256x256 tile took).
        // Interior tiles (all but the last row tile) skip the row-bound checks.
        const float thr = fthr[(int64_t)query * fthr_stride];
        const int lr0 = wrow + 4 * g;                          // the lane's first row inside the tile
        int lim = (int)(n - row0 < BM ? n - row0 : BM);        // live rows of this tile
        // pair mode: only rows ABOVE the query's own row count (first local row that does: plo)
        int plo = 0;
        if (pairs.on) {
          const long long qrow = pairs.query_row0 + query;
          const long long first = qrow + 1 - (pairs.row_base + row0);
          plo = first > BM ? BM : (first < 0 ? 0 : (int)first);
          if (qrow < pairs.first_query) plo = BM;
        }
        auto offer = [&](auto FULL) {
          constexpr bool full = decltype(FULL)::value;
          uint32_t cnt = 0;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              cnt += (!(acc[i][j][r] < thr) && (full || (lr0 + i * 16 + r < lim && lr0 + i * 16 + r >= plo))) ? 1u : 0u;
          if (cnt) {
            uint32_t slot = atomicAdd(fstate_words + (int64_t)query * fstate_stride, cnt);
            uint64_t* cq = fcand + (int64_t)query * fcap;
            const uint32_t row_lo = (uint32_t)(pairs.row_base + row0 + lr0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float v = acc[i][j][r];
                if (!(v < thr) && (full || (lr0 + i * 16 + r < lim && lr0 + i * 16 + r >= plo))) {
                  if (slot < fcap) cq[slot] = ((uint64_t)score_key(v) << 32) | (row_lo + (uint32_t)(i * 16 + r));
                  ++slot;
                }
              }
          }
        };
        if (lim == BM && plo == 0) offer(std::true_type{});
        else offer(std::false_type{});
      } else {
        float* o = scores + (int64_t)query * sstride;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int64_t ob = row0 + wrow + i * 16 + 4 * g;
          if (ob + 3 < n) *(f32x4_t*)(o + ob) = acc[i][j];
          else
            for (int r = 0; r < 4; ++r)
              if (ob + r < n) o[ob + r] = acc[i][j][r];
        }
      }
    }
  }
}

// Stage `rows` rows x 128 B starting at k-step `s` into `lds` (linear image),
// rows [row0, row0+rows) of a row-major half matrix with stride ld; rows past
// row_max are clamped (their products are never stored).  One wave instruction
// covers 8 rows; the workgroup's waves take instructions round-robin.
// Every wave issues the SAME number of DMA instructions, max(1, rows/64) (with
// fewer than 8 instructions in all, the surplus waves repeat one -- same bytes to
// the same place), so a counted s_waitcnt vmcnt(N) means the same thing in every wave.
// The per-lane source addresses of k-step 0 and the LDS slots are fixed for the whole
// tile: they are computed once (TgSrc) and each stage only adds s * 128 bytes.
template <int ROWS>
struct TgSrc {
  static constexpr int NI = ROWS / 8;                       // wave instructions in the tile
  static constexpr int PER = NI >= TG_WAVES ? NI / TG_WAVES : 1;
  const uint8_t* src[PER];
  int slot[PER];                                            // LDS offset of the instruction, in u32x4
  __device__ __forceinline__ void init(const uint8_t* __restrict__ base, int64_t row0, int64_t row_max, int64_t ldb,
                                       int wave, int lane) {
    const int r_in = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int i = (wave + t * TG_WAVES) % NI;
      const int r = i * 8 + r_in;
      int64_t gr = row0 + r;
      gr = gr < row_max ? gr : row_max - 1;
      src[t] = base + gr * ldb + (pc ^ tg_swz(r)) * 16;
      slot[t] = i * 64;
    }
  }
  __device__ __forceinline__ void stage(int s, u32x4* lds) const {
#pragma unroll
    for (int t = 0; t < PER; ++t)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + (int64_t)s * TG_BKB),
                                       (__attribute__((address_space(3))) void*)(lds + slot[t]), 16, 0, 0);
  }
};

template <int ROWS> constexpr int tg_stage_count() { return ROWS / 8 >= TG_WAVES ? ROWS / 8 / TG_WAVES : 1; }

// Q: [nq_pad][ld] halves, nq_pad a multiple of BN (rows >= nq zero).
// FUSE == false: scores [nq][sstride] are written.  FUSE == true: nothing is
// materialised; scores >= fthr[q * fthr_stride] go to fcand [nq][fcap], counted in
// the per-query header word fstate_words[q * fstate_stride].
// ldb = row stride in BYTES (multiple of 128) of both M and Q.  EB == 1: the
// accumulators are multiplied by rscale[row] * qscale[query] before use.
constexpr int tg_nbuf(int bm, int bn) { return 3 * (bm + bn) * 128 <= 160 * 1024 ? 3 : 2; }
constexpr int tg_lds_bytes(int bm, int bn) { return tg_nbuf(bm, bn) * (bm + bn) * 128; }

template <int BN, bool FUSE, int EB, int BM = TG_BM>
__global__ __launch_bounds__(TG_WAVES * 64) void gemm_tiled_kernel(
    const uint8_t* __restrict__ M, const uint8_t* __restrict__ Q, float* __restrict__ scores,
    int64_t n, int64_t ldb, int64_t sstride, int nq, uint32_t* __restrict__ fstate_words, int fstate_stride,
    uint64_t* __restrict__ fcand, uint32_t fcap, const float* __restrict__ fthr, int fthr_stride,
    const float* __restrict__ rscale, const float* __restrict__ qscale, const TgPairs pairs = TgPairs{}) {
  constexpr int TN = BN < 64 ? BN : 64;     // queries per wave tile
  constexpr int WN = BN / TN;               // waves along the query axis
  constexpr int WM = TG_WAVES / WN;         // waves along the row axis
  constexpr int TM = BM / WM;               // rows per wave tile
  constexpr int MT = TM / 16, NT = TN / 16; // MFMA tiles per wave
  static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile must be whole MFMA tiles");
  extern __shared__ u32x4 tg_lds[];
  TG_STAMP_BLOCK(0);
  // layout: A buffers [NBUF][128 rows][8 chunks], then B buffers [NBUF][BN rows][8 chunks]
  constexpr int NBUF = tg_nbuf(BM, BN);   // 3 when it fits the 160 KiB LDS, else 2 (256 x 256 tiles)
  auto ldsA = [&](int b) { return tg_lds + b * (BM * 8); };
  auto ldsB = [&](int b) { return tg_lds + NBUF * BM * 8 + b * (BN * 8); };
  constexpr int DMA_PER_STAGE = tg_stage_count<BM>() + tg_stage_count<BN>();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r16 = lane & 15, g = lane >> 4;
  // Workgroups are dispatched in linear order (x fastest) round-robin over the 8 XCDs, each
  // with its own L2.  With several query tiles per row tile (grid y > 1) the tiles that share
  // a row tile are mapped to consecutive slots of ONE XCD, so the corpus tile comes from HBM
  // once and from that L2 for the other query tiles (instead of once per query tile).
  int bx = blockIdx.x, by = blockIdx.y;
  if (gridDim.y > 1) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int id = by * gx + bx;
    const int full = (gx >> 3) << 3;            // row tiles in whole groups of 8
    if (id < full * gy) {
      const int grp = id / (8 * gy), within = id - grp * 8 * gy;
      bx = grp * 8 + (within & 7);
      by = within >> 3;
    } else {
      const int rem = gx - full, t = id - full * gy;
      bx = full + t % rem;
      by = t / rem;
    }
  }
  const int64_t row0 = (int64_t)bx * BM;
  const int q0 = by * BN;
  const int ksteps = (int)(ldb / TG_BKB);

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // Three-stage ring, two k-steps of LDS-DMA in flight while one is multiplied.
  // The DMA is ordered for the readers only by the issuing waves' counted vmcnt
  // followed by a barrier every reader has passed, hence: wait (all but the next
  // step's DMA retired) -> raw s_barrier -> issue step s+2 into the buffer that
  // was read in step s-1 (everyone is past that read: same barrier) -> multiply
  // step s.  A plain __syncthreads() would make hipcc drain vmcnt(0) every step.
  TgSrc<BM> srcA;
  TgSrc<BN> srcB;
  srcA.init(M, row0, n, ldb, wave, lane);
  srcB.init(Q, q0, (int64_t)q0 + BN, ldb, wave, lane);
  constexpr int AHEAD = NBUF - 1;         // k-steps of DMA kept in flight
  srcA.stage(0, ldsA(0));
  srcB.stage(0, ldsB(0));
  if (AHEAD > 1 && ksteps > 1) {
    srcA.stage(1, ldsA(1));
    srcB.stage(1, ldsB(1));
  }
  int cur = 0;
  TG_STAMP_BLOCK(1);
#ifdef TG_TRACE
  const bool tg_tr = blockIdx.x == 1000 && blockIdx.y == 0 && lane == 0 && ksteps <= 64;
#endif
  // A lane's fragment of one 16-row tile and one 128-byte stage: two 16-byte chunks.
  //   f16: chunk 4h + g = halves 8g..8g+7 of the 32-wide k-half h (one 16x16x32 MFMA each)
  //   f32: chunk 4h + g = floats 4g..4g+3 of the 16-wide sub-step h; the four components feed
  //        four exact v_mfma_f32_16x16x4_f32 whose k-slot g maps to column 16h + 4g + e
  //   fp8: the same two chunks are the lane's 32 k bytes of ONE v_mfma_f32_16x16x128_f8f6f4
  //        (CDNA4; e4m3 x e4m3, no block scales), 2x the per-clock rate of the 16x16x32 fp8
  //        form, which runs at the f16 rate on gfx950.  (Chunks 2g, 2g + 1 -- consecutive k --
  //        would be the natural choice, but ds_read_b128 serves lanes in the groups
  //        {0-3, 12-15, 20-27}, ... and that pairing put two lanes of a group on every bank:
  //        SQ_LDS_BANK_CONFLICT was half of the LDS cycles.)
  // The k map is the same for both operands, so any permutation inside it cancels.
  //
  // Schedule of one k-step.  Left alone, hipcc keeps two A fragments live and emits
  // "2 ds_reads, s_waitcnt lgkmcnt(0), 8 MFMAs" eight times per k-step: the LDS latency is
  // exposed before every 128 cycles of matrix work (k-loop trace: 3,360 cycles of a k-step
  // spent issuing 2,048 cycles of MFMA).  So the order is spelled out: all B fragments and
  // the first DEPTH A fragments are requested up front, each A fragment's slot of a small
  // ring is refilled right after the MFMAs that consumed it were issued -- DEPTH - 1
  // fragments (~200 cycles of MFMA) before it is needed -- and a scheduling barrier after
  // every fragment keeps the compiler from folding the reads back next to their use.
  // (hipcc still waits lgkmcnt(0) each time -- with LDS-DMA in flight its waitcnt pass never
  // counts LDS reads -- but a hand-counted version with inline-asm ds_reads measured the same:
  // what remains of the k-step is the barrier and the restart after it, not these waits.)
  constexpr int NA = EB == 1 ? MT : 2 * MT;            // A-fragment units per k-step, in order of use
  constexpr int APER = EB == 1 ? 2 : 1;                // 16-byte reads per unit (fp8: both chunks)
  constexpr int DEPTH = NA < (EB == 1 ? 3 : 4) ? NA : (EB == 1 ? 3 : 4);
  auto compute_step = [&](const u32x4* A, const u32x4* B) {
    u32x4 fb[2][NT];
    u32x4 ring[DEPTH][APER];
    auto read_a = [&](int unit, u32x4 (&dst)[APER]) {
      const int i = EB == 1 ? unit : unit % MT;
      const int r = wm * TM + i * 16 + r16;
#pragma unroll
      for (int c = 0; c < APER; ++c) {
        const int h = EB == 1 ? c : unit / MT;
        dst[c] = A[r * 8 + ((4 * h + g) ^ tg_swz(r))];
      }
    };
    auto read_b = [&](int h) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int r = wn * TN + j * 16 + r16;
        fb[h][j] = B[r * 8 + ((4 * h + g) ^ tg_swz(r))];
      }
    };
    read_b(0);
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) read_a(u, ring[u]);
    read_b(1);
    if constexpr (NA > DEPTH) __builtin_amdgcn_sched_barrier(0);   // (small tiles: nothing to pipeline, the compiler's order is fine)
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int slot = u % DEPTH;
      if constexpr (EB == 2) {
        const int h = u / MT, i = u % MT;
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ring[slot][0]),
                                                             __builtin_bit_cast(h8, fb[h][j]), acc[i][j], 0, 0, 0);
      } else if constexpr (EB == 4) {
        const int h = u / MT, i = u % MT;
        const v4f x = __builtin_bit_cast(v4f, ring[slot][0]);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const v4f y = __builtin_bit_cast(v4f, fb[h][j]);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, y.x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, y.y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, y.z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, y.w, acc[i][j], 0, 0, 0);
        }
      } else {
        const u32x4 al = ring[slot][0], ah = ring[slot][1];
        const i32x8 x = {(int)al.x, (int)al.y, (int)al.z, (int)al.w, (int)ah.x, (int)ah.y, (int)ah.z, (int)ah.w};
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const u32x4 bl = fb[0][j], bh = fb[1][j];
          const i32x8 y = {(int)bl.x, (int)bl.y, (int)bl.z, (int)bl.w, (int)bh.x, (int)bh.y, (int)bh.z, (int)bh.w};
          acc[u][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, y, acc[u][j], 0, 0, 0, 0, 0, 0);
        }
      }
      if (u + DEPTH < NA) read_a(u + DEPTH, ring[slot]);
      if constexpr (NA > DEPTH) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // (Tried: the second wave of each SIMD one phase behind the first -- multiplying the previous
  // step's fragments while the first wave reads -- so that the LDS pipe and the matrix pipe
  // overlap instead of taking turns: -4 % cycles on constant operands, nothing on real data,
  // where the chip is at its power limit and the clock, not the schedule, sets the time.)
  for (int s = 0; s < ksteps; ++s) {
    TG_STAMP(0);
    // retire step s's DMA: everything but the (AHEAD - 1) younger steps
    if (AHEAD > 1 && s + 1 < ksteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_STAGE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TG_STAMP(1);
    __builtin_amdgcn_s_barrier();
    TG_STAMP(2);
    if (s + AHEAD < ksteps) {
      const int nb = cur >= 1 ? cur - 1 : NBUF - 1;   // (s + AHEAD) % NBUF: the buffer read in step s - 1
      srcA.stage(s + AHEAD, ldsA(nb));
      srcB.stage(s + AHEAD, ldsB(nb));
    }
    compute_step(ldsA(cur), ldsB(cur));
    TG_STAMP(3);
    cur = cur + 1 < NBUF ? cur + 1 : 0;
  }

  TG_STAMP_BLOCK(2);
  tg_epilogue<FUSE, EB, MT, NT, BM>(acc, row0, q0, wm * TM, wn * TN, lane, n, nq, scores, sstride, fstate_words, fstate_stride,
                                    fcand, fcap, fthr, fthr_stride, rscale, qscale, pairs);
  TG_STAMP_BLOCK(3);
}

}  // namespace svs
