// Batched score stage, panels of 256 (or, QT = 128, 128) queries x 256 corpus rows per workgroup:
//     S[j][i] = sum_d M[i,d] * Q[j,d]      f16 (EB = 2) or e4m3 (EB = 1) operands, f32 accumulate
// for BASELINE.json configs[2] (1M x 1536 f16, 1024 queries per call) and configs[4]
// (10M x 3072 fp8, 256 queries per call).  Same arithmetic, k order, LDS image, swizzle and
// epilogue as gemm_tiled_kernel<256, FUSE, EB, 256> (gemm_tiled.h) -- results are bit-identical --
// but a different main loop.  The reference's nearest analogue is np.dot(M, M.T)
// (src/svs/kb.py:1651); a query batch is a loop of np.dot(M, q) (src/svs/kb.py:1623).
//
// Why a second kernel: gemm_tiled's loop is "wait, barrier, issue the whole next k-step's LDS-DMA,
// read fragments, 64 MFMAs" -- all eight waves issue DMA, then all read LDS, then all multiply,
// and the matrix pipe idles through the first two (k-step 4,200 cycles for 2,048 of MFMA,
// DESIGN.md).  Here a 128-byte k-tile is cut into FOUR phases, each multiplying one quadrant
// (64 rows x 32 queries) of the wave's 128 x 64 tile: 4-12 ds_read_b128 + one half-tile of LDS-DMA
// (2 instructions per wave) + 16 MFMAs (8 for fp8), between two raw s_barriers.  Waves 4-7 run
// ONE BARRIER BEHIND waves 0-3 -- each SIMD hosts one wave of either group -- so in every
// barrier interval one group multiplies while the other reads and stages: the matrix pipe
// always has a wave to serve (cdna_hip_programming.md 5.5 T3/T4/T5).
//
// Data movement.  LDS = 8 slots of 16 KiB, a slot = one HALF-TILE = 128 rows x 128 bytes:
//   B0 / B1: queries  wc * 64 + {0..31} / {32..63}   of the four wave columns wc
//   A0 / A1: rows     wr * 128 + {0..63} / {64..127} of the two wave rows wr
// i.e. exactly what phase 0 (B0, A0), phase 1 (B1) and phase 2 (A1) read; phase 3 reads nothing
// (B0 is still in registers).  The stream of half-tiles B0 A0 B1 A1 | B0 A0 B1 A1 | ... runs
// SEVEN half-tiles ahead of the phase that consumes it -- across k-tiles and across output tiles:
// the kernel is persistent (one workgroup per CU walks its list of output tiles) and the stream
// simply continues into the next tile, so its first loads are in flight before this tile's
// epilogue starts.  Slot of half-tile H is H mod 8.
//   RAW: phase P issues half-tile P + 7, then waits vmcnt(10) -- all but the 5 youngest
//        half-tiles, i.e. everything up to P + 2, which is what phase P + 1 reads -- BEFORE its first
//        barrier; the reads come one phase later, after both groups have passed a barrier behind
//        that wait (LDS-DMA is ordered for a reader only by the issuers' vmcnt + a barrier).
//   WAR: half-tile P + 7 overwrites P - 1, whose last read was two phases ago -- except B0 of the
//        current k-tile (read in phase 0, overwritten in phase 1): phase 0 therefore issues its
//        four B reads first and retires them with lgkmcnt(8) before its first barrier.
// The DMA is `buffer_load_dwordx4 ... offen lds`: a per-tile descriptor (scalar) + a per-lane byte
// offset fixed for the whole kernel + the k offset as scalar soffset -- no address arithmetic per
// issue, and rows past n are dropped by the descriptor's range check (never read).  With no next
// tile the descriptor has zero records: the instructions still issue (every wave keeps the same
// vmcnt bookkeeping) and move nothing.  The XOR bank swizzle sits on the source side, as in
// gemm_tiled.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_tiled.h"

namespace svs {

typedef float f32x2_t __attribute__((ext_vector_type(2)));   // (also in fp8.h)

constexpr int PG_TILE = 256;              // rows and queries per output tile
constexpr int PG_SLOT = 1024;             // u32x4 per slot (16 KiB)
constexpr int PG_LDS_BYTES = 8 * PG_SLOT * 16;
constexpr int PG_THREADS = 512;

// PG_DMA_IN_MMA = 1: a phase's two LDS-DMA pieces are issued between its MFMAs (after MFMA pair
// PG_DMA_AT0 of the first k half and PG_DMA_AT1 of the second) instead of with its fragment reads.
// The stream then runs half a phase later: at a phase's counted wait the youngest half-tile in
// flight is P + 6, and "everything up to P + 2" is all but the FOUR youngest half-tiles: vmcnt(8).
// Measured on real data the two placements are within 2 % of each other (configs[2]: 2.42-2.50 ms
// with the reads, 2.44-2.62 between the MFMAs; configs[4]: 7.17 vs 7.19): the default issues them
// with the reads, svs_index_set_variant(8) selects the other (tools/cfg_time.py).
#ifndef PG_DMA_IN_MMA
#define PG_DMA_IN_MMA 0
#endif
#ifndef PG_DMA_STAGGER
#define PG_DMA_STAGGER 0
#endif
#ifndef PG_DMA_AT0
#define PG_DMA_AT0 1
#endif
#ifndef PG_DMA_AT1
#define PG_DMA_AT1 1
#endif
// The counted wait of a phase: everything but the youngest 5 half-tiles (4 when the pieces are issued between the
// MFMAs) must have landed.  An A half-tile is 2 pieces per wave, a B half-tile PB (2 at 256-query tiles, 1 at
// 128-query tiles); the half-tiles alternate B A B A, so the count depends on what the YOUNGEST one is:
//   youngest = A: A B A B A = 6 + 2 PB  (4 youngest: 4 + 2 PB),   youngest = B: B A B A B = 4 + 3 PB  (4 + 2 PB)
#ifdef PG_VMCNT_FORCE   // (tools: DMA-only ablations with more pieces in flight than the ring allows -- results are wrong)
#define PG_VMCNT_OF(kind) PG_VMCNT_FORCE
#else
#define PG_VMCNT_OF(kind) (kDmaInMma ? 4 + 2 * PB : (((kind) & 1) ? 6 + 2 * PB : 4 + 3 * PB))
#endif
#define PG_VMCNT PG_VMCNT_OF(0)   // the smaller of the two: what a wait that must hold in every phase uses

#ifdef PG_CLOCKS   // tools/gemm_phased_bench.hip only: shader cycles and 100 MHz ticks a workgroup spent in the kernel
__device__ unsigned long long* pg_clock_buf;
#define PG_CLOCK_STAMP(slot)                                                                   \
  do {                                                                                         \
    if (threadIdx.x == 0) {                                                                    \
      pg_clock_buf[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime();                    \
      pg_clock_buf[blockIdx.x * 8 + 2 + (slot)] = __builtin_amdgcn_s_memrealtime();            \
    }                                                                                          \
  } while (0)
#define PG_LOOP_CLOCK(var) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); } while (0)
#else
#define PG_CLOCK_STAMP(slot) do { } while (0)
#define PG_LOOP_CLOCK(var) do { } while (0)
#endif

#ifdef PG_TRACE   // tools/gemm_phased_bench.hip only: s_memtime stamps of 16 phases of one workgroup's third tile,
// kept in LDS behind the ring and copied out at the end.  Slots per phase: 0 = after the phase's loads and
// waits (before barrier 1), 1 = fragments in (after barrier 1), 2 = MFMAs issued (before barrier 2),
// 3 = after barrier 2.  (Each stamp waits lgkmcnt(0): the one before barrier 1 also waits for the
// phase's own fragment reads, which the untraced kernel leaves in flight across the barrier.)
__device__ unsigned long long* pg_trace_buf;
constexpr int PG_TRACE_KT0 = 4, PG_TRACE_KTS = 4, PG_TRACE_TILE = 2, PG_TRACE_BLOCK = 37;
#define PG_STAMP(PH, SLOT)                                                                                  \
  do {                                                                                                      \
    if (tracing && kt >= PG_TRACE_KT0 && kt < PG_TRACE_KT0 + PG_TRACE_KTS) {                                \
      unsigned long long t_;                                                                                \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
      if (lane == 0)                                                                                        \
        *(volatile unsigned long long*)((char*)pg_lds + PG_LDS_TOTAL + ((wave * PG_TRACE_KTS * 4 + (kt - PG_TRACE_KT0) * 4 + (PH)) * 4 + (SLOT)) * 8) = t_; \
    }                                                                                                       \
  } while (0)
#else
#define PG_STAMP(PH, SLOT) do { } while (0)
#endif

// output tile `id` -> (row tile, query tile): the gy query tiles of a row tile get ids that are
// equal mod 8 and adjacent in time, so they run on ONE XCD together (workgroups are dealt
// round-robin over the XCDs) and the corpus tile comes from HBM once, from that L2 afterwards.
__device__ __forceinline__ void pg_tile_of(int id, int gx, int gy, int* bx, int* by) {
  const int full = (gx >> 3) << 3;
  if (id < full * gy) {
    const int grp = id / (8 * gy), within = id - grp * 8 * gy;
    *bx = grp * 8 + (within & 7);
    *by = within >> 3;
  } else {
    const int rem = gx - full, t = id - full * gy;
    *bx = full + t % rem;
    *by = t / rem;
  }
}

// a buffer descriptor whose words are explicitly wave-uniform (scalar registers)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pg_rsrc(const void* p, int bytes) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int N, class F>
__device__ __forceinline__ void pg_static_for(F&& f) {   // f(integral_constant<0>) ... f(integral_constant<N - 1>)
  if constexpr (N > 0) {
    pg_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// Fused epilogue, second sweep: the lanes whose score is not below the threshold (a NaN passes) store
// (score bits, base + IMM) at their LDS address `ak` and step it; every lane leaves with exec all ones.
template <int IMM>
__device__ __forceinline__ void pg_park_if(unsigned& ak, float v, float thr, uint32_t base) {
  uint32_t code;
  asm volatile(
      "v_add_u32_e32 %1, %5, %4\n\t"
      "v_cmpx_nlt_f32_e32 vcc, %2, %3\n\t"
      "ds_write2_b32 %0, %2, %1 offset1:1\n\t"
      "v_add_u32_e32 %0, 8, %0\n\t"
      "s_mov_b64 exec, -1"
      : "+v"(ak), "=&v"(code)
      : "v"(v), "v"(thr), "v"(base), "n"(IMM)
      : "vcc", "memory");
}

// LDS byte address of a piece's destination = the wave's base + a literal, formed where it is used: the sixteen
// destinations of a wave are loop invariants, and hoisted they cost sixteen scalar registers the kernel does
// not have (-> v_readlane restores of spilled SGPRs inside the k loop)
template <int IMM>
__device__ __forceinline__ unsigned pg_lds_dest(unsigned wave_base) {
  unsigned a;
  asm volatile("s_add_u32 %0, %1, %2" : "=s"(a) : "s"(wave_base), "n"(IMM) : "scc");
  return a;
}

__device__ __forceinline__ int pg_voff(int vbase, int uniform_bytes) {
  int vo;
  asm volatile("v_add_u32_e32 %0, %1, %2" : "=v"(vo) : "s"(uniform_bytes), "v"(vbase));
  return vo;
}

struct PgTile {
  __amdgpu_buffer_rsrc_t a, b;   // corpus rows / query rows of the tile
  int64_t row0;
  int q0;
};

// Fragment reads: inline asm with base + immediate offset.  Left to itself hipcc hoists all 48
// fragment addresses of the two k-tile bodies out of the loop as separate VGPRs and spills
// accumulators to pay for them (and every reload of a spilled address waits vmcnt(0), which
// drains the LDS-DMA pipeline).
#define PG_DS_READ(dst, base, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(imm))
// adr[h]: the lane's byte address in slot 0; slots 4-7 lie 64 KiB further on, past the reach of a ds_read's
// immediate, and get their base from one v_add per k half made where it is used (kept as four more registers
// for the whole kernel they were what pushed the fp8 form to 256 VGPRs and into scratch)
template <int SLOT>
__device__ __forceinline__ unsigned pg_slot_base(unsigned adr0) {
  if constexpr (SLOT < 4) return adr0;
  unsigned a;
  asm volatile("v_add_u32_e32 %0, 0x10000, %1" : "=v"(a) : "v"(adr0));
  return a;
}
template <int SLOT, int MT0 = 0>   // (MT0 = 2: timing experiment 53, only the second half of the fragments is read)
__device__ __forceinline__ void pg_read_a(u32x4 (&fa)[4][2], const unsigned (&adr)[2]) {
  const unsigned b0 = pg_slot_base<SLOT>(adr[0]), b1 = pg_slot_base<SLOT>(adr[1]);
#pragma unroll
  for (int mt = MT0; mt < 4; ++mt) {
    PG_DS_READ(fa[mt][0], b0, (SLOT & 3) * 16384 + mt * 2048);
    PG_DS_READ(fa[mt][1], b1, (SLOT & 3) * 16384 + mt * 2048);
  }
}
template <int SLOT, int NTQ>
__device__ __forceinline__ void pg_read_b(u32x4 (&fb)[NTQ][2], const unsigned (&adr)[2]) {
  const unsigned b0 = pg_slot_base<SLOT>(adr[0]), b1 = pg_slot_base<SLOT>(adr[1]);
#pragma unroll
  for (int nt = 0; nt < NTQ; ++nt) {
    PG_DS_READ(fb[nt][0], b0, (SLOT & 3) * 16384 + nt * 2048);
    PG_DS_READ(fb[nt][1], b1, (SLOT & 3) * 16384 + nt * 2048);
  }
}
#undef PG_DS_READ
// "the fragments are in": the wait names every register the asm reads filled, so that nothing
// that uses them can be scheduled in front of it (cdna_hip_programming.md 5.7, form (ii))
__device__ __forceinline__ void pg_landed_a(u32x4 (&fa)[4][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[2][0]), "+v"(fa[2][1]),
                 "+v"(fa[3][0]), "+v"(fa[3][1])::"memory");
}
__device__ __forceinline__ void pg_landed_b(u32x4 (&fb)[2][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1])::"memory");
}
__device__ __forceinline__ void pg_landed_b(u32x4 (&fb)[1][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][0]), "+v"(fb[0][1])::"memory");
}

// ---- side region of the LDS, behind the 128 KiB ring (fused top-k and fp8 scales) ------------
// Per output tile the epilogue needs 256 thresholds (fused top-k), and for fp8 256 row scales and
// 256 query scales.  An ordinary global load in the epilogue would make hipcc wait vmcnt(0) -- it
// cannot count past LDS-DMA in flight -- i.e. drain the seven half-tiles prefetched for the next
// tile, and then sit out the load's own latency; four returning global atomics per lane (one per
// query column) did the same, one round trip each: together a quarter of the kernel's time.
// So: the side data arrives by LDS-DMA too (issued before the tile's k loop, landed long before its
// epilogue), and the epilogue PARKS its surviving candidates in LDS -- each wave in its own eighth
// of the lot, slots from a wave-uniform counter and the lanes' ranks in the compare mask: no
// atomics, no waits; they are flushed to the global candidate lists during the NEXT tile's k loop, in two steps
// eight phases apart: returning global atomics issued by inline asm (hipcc inserts no wait for
// them), their slots consumed after the loop's own counted waits have long covered them.
// Two copies of everything, indexed by the tile's parity.
#ifdef PG_TRACE
constexpr int PG_PARK = 512;      // (trace builds keep their stamps behind the side region)
#else
constexpr int PG_PARK = 1536;     // candidates a workgroup can park per tile (~420 expected at k = 100, prefix n / 64): 192 per wave.  (1024 through round 3: a
                                  // corpus stored topic by topic puts one query's 128 survivors into ONE wave's 128 rows, plus ~50 of the others': every such
                                  // wave overflowed into the one-atomic-per-candidate path -- tools/clustered_corpus_time.py)
#endif
constexpr int PG_SIDE = PG_LDS_BYTES;                   // byte offsets
constexpr int PG_SIDE_THR = PG_SIDE;                    // f32 [2][256]
constexpr int PG_SIDE_RS = PG_SIDE_THR + 2048;          // f32 [2][256] row scales (fp8)
constexpr int PG_SIDE_QS = PG_SIDE_RS + 2048;           // f32 [2][256] query scales (fp8)
constexpr int PG_SIDE_CNT = PG_SIDE_QS + 2048;          // u32 [2][8]: candidates parked by each wave
constexpr int PG_SIDE_KEY = PG_SIDE_CNT + 64;           // u32 [2][PG_PARK][2]: (score bits, code); wave w owns entries [w * PG_PARK / 8, ..)
constexpr int PG_LDS_TOTAL = PG_SIDE_KEY + 2 * PG_PARK * 8;  // 161,856 bytes of the 163,840
static_assert(PG_LDS_TOTAL <= 160 * 1024 && PG_PARK % PG_THREADS == 0 && PG_PARK % 8 == 0, "the parking lot must fit the LDS, in whole rounds of the flush loops");
// a parked candidate's code: row inside the tile (8 bits) | query inside the tile << 8
constexpr int PG_AGG_MIN = 96;     // candidates parked by a wave from which its flush adds per RUN of a query (flush_group): a shuffled corpus parks ~52 +- 7
constexpr int PG_FEW_DENSE = 8;    // ... or a wave on the compare-and-branch path (it expected <= PG_SPARSE_MAX survivors and counts no lanes) that parked this many
constexpr int PG_LANE_DENSE = 6;   // ... or survivors of ONE lane (the sweeps count them): few queries per topic put few candidates into a wave, all of one query
constexpr int PG_SPARSE_MAX = 16;  // survivors per wave and tile up to which the epilogue compares-and-branches per group of four registers
                                   // (~150 cycles per group that holds one: at 16 still below the sweeps' fixed 10-11 k cycles per tile)
constexpr int PG_FLUSH_KT = 2, PG_MIN_KT = 6;   // the flush brackets k-tiles 2 and 3 of the next tile

// LDS accesses of the epilogue / flush: inline asm, so that hipcc neither orders them against the
// LDS-DMA in flight (with a vmcnt(0)) nor counts them
__device__ __forceinline__ float pg_lds_read_f32(unsigned adr) {
  float v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(adr) : "memory");
  return v;
}
__device__ __forceinline__ f32x4_t pg_lds_read_f32x4(unsigned adr) {
  f32x4_t v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(adr) : "memory");
  return v;
}
// the same reads issued without a wait, and the waits that name what they fill (one LDS round trip for a batch of
// reads instead of one per read: the fp8 epilogue made 16 of them in a row, ~2 k cycles of a 7 k-cycle epilogue)
__device__ __forceinline__ void pg_lds_issue_f32(float& dst, unsigned adr) { asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(adr) : "memory"); }
__device__ __forceinline__ void pg_lds_issue_f32x4(f32x4_t& dst, unsigned adr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(adr) : "memory"); }
template <int N>
__device__ __forceinline__ void pg_lds_landed(float (&a)[N]) {
  static_assert(N == 2 || N == 4, "");
  if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::"memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1])::"memory");
}
__device__ __forceinline__ void pg_lds_landed(f32x4_t (&a)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])::"memory");
}
__device__ __forceinline__ uint32_t pg_lds_read_u32(unsigned adr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(adr) : "memory");
  return v;
}
__device__ __forceinline__ uint64_t pg_lds_read_u64(unsigned adr) {
  uint64_t v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(adr) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t pg_lds_add_rtn(unsigned adr, uint32_t x) {
  uint32_t v;
  asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(adr), "v"(x) : "memory");
  return v;
}
__device__ __forceinline__ void pg_lds_write_u64(unsigned adr, uint64_t x) { asm volatile("ds_write_b64 %0, %1" ::"v"(adr), "v"(x) : "memory"); }
__device__ __forceinline__ void pg_lds_write_u32(unsigned adr, uint32_t x) { asm volatile("ds_write_b32 %0, %1" ::"v"(adr), "v"(x) : "memory"); }

// EXP != 0: timing-only ablations for tools/gemm_phased_bench.hip (results are wrong): 1 no LDS-DMA
// in the main loop, 2 no fragment reads, 3 no MFMAs, 7 no stagger (all waves in the same phase),
// 14 no epilogue.
// QT: queries per output tile.  256: four wave columns of 64 queries (two 16-query MFMA columns per quadrant).
// 128 (panels of 65 .. 128 queries -- what a coalescer forms on a reduced-precision index): four wave columns of
// 32 queries, ONE 16-query MFMA column per quadrant; a B half-tile is then 64 queries = one LDS-DMA piece per
// wave (8 KiB of its 16 KiB slot), and the counted waits below follow from the pieces per half-tile.
template <bool FUSE, int EB, int EXP = 0, int QT = 256>
__global__ __launch_bounds__(PG_THREADS) void gemm_phased_kernel(
    const uint8_t* __restrict__ M, const uint8_t* __restrict__ Q, float* __restrict__ scores,
    int64_t n, int ldb, int64_t sstride, int nq, int gx, int gy, uint32_t* __restrict__ fstate_words, int fstate_stride,
    uint64_t* __restrict__ fcand, uint32_t fcap, const float* __restrict__ fthr, int fthr_stride,
    const float* __restrict__ rscale, const float* __restrict__ qscale, const TgPairs pairs = TgPairs{}) {
  static_assert(EB == 1 || EB == 2, "f16 or fp8 operands");
  static_assert(QT == 256 || QT == 128, "256- or 128-query tiles");
  extern __shared__ u32x4 pg_lds[];
  PG_CLOCK_STAMP(0);
  constexpr int NTQ = QT / 128;           // 16-query MFMA columns per quadrant
  constexpr int MT = 8, NT = 2 * NTQ;
  constexpr int PB = QT / 128;            // LDS-DMA pieces per wave in a B half-tile (an A half-tile: 2)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r16 = lane & 15, g = lane >> 4;
  const int KT = ldb / TG_BKB;                       // k-tiles per output tile (even, >= PG_MIN_KT: checked by the host)
  const int G = gridDim.x, total = gx * gy;
  // (integer division runs on the vector ALU; whatever hangs on its result -- the loop bounds, the tile
  //  descriptors -- must be told that it is wave-uniform, or it is kept in VGPRs: see tile_desc)
  // kMap (timing experiments 50 / 51, gy == 4 and 256 workgroups only): other workgroup -> tile maps.  The default
  // (pg_tile_of) gives every XCD, per step, 8 row tiles x the 4 query tiles; 50: 32 row tiles x ONE query tile (its
  // 768 KiB of queries stay in that XCD's L2; the corpus tile is fetched by four XCDs at the same step: Infinity
  // Cache); 51: 16 row tiles x 2 query tiles.
  constexpr int kMap = (EXP == 50 || EXP == 52) ? 1 : (EXP == 51 ? 2 : 0);
  const bool alt_map = kMap != 0 && gy == 4 && G == 256;
  int my_tiles_v = (total - (int)blockIdx.x + G - 1) / G;
  if (alt_map) {
    const int x = (int)blockIdx.x & 7, c = (int)blockIdx.x >> 3;
    if (kMap == 1) { const int rset = x >> 2, lim = (gx - rset + 1) / 2; my_tiles_v = lim > c ? (lim - c + 31) / 32 : 0; }
    else { const int rset = x >> 1, lim = (gx - rset + 3) / 4, cc = c >> 1; my_tiles_v = lim > cc ? (lim - cc + 15) / 16 : 0; }
  }
  const int my_tiles = __builtin_amdgcn_readfirstlane(my_tiles_v);

  // ---- per-lane constants of the staging side: byte offset of the lane's 16 bytes inside a tile
  // ONE register: the 16 bytes of row li0 = 8 wave + lane / 8 of a tile.  The eight (operand, half, instruction)
  // pieces a wave stages differ from it by whole rows, the same for every lane:
  //   A half h, instruction jj: row (li >> 6) * 128 + h * 64 + (li & 63) with li = li0 + 64 jj  =  li0 + jj * 128 + h * 64
  //   B half h, instruction jj: row (li >> 5) * 64 + h * 32 + (li & 31)                         =  li0 + (wave >> 2) * 32 + jj * 128 + h * 32
  // (the swizzle depends on bits 1-3 of li only).  The row part must stay in the VECTOR offset -- the descriptor's
  // range check does not see soffset -- so each piece adds its scalar to the base just before it is issued
  // (pg_voff: one v_add, volatile so that hipcc does not hoist the eight sums back into eight live registers).
  const unsigned wave_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)pg_lds + (unsigned)wave * 1024u;   // LDS byte address of the wave's 1 KiB in slot 0
  int vbase, vbase_tm = 0;
  {
    const int li0 = 8 * wave + (lane >> 3), pc = lane & 7;
    vbase = li0 * ldb + (pc ^ tg_swz(li0)) * 16;
    if constexpr (EXP == 40 || EXP == 45) vbase_tm = li0 * TG_BKB + (pc ^ tg_swz(li0)) * 16;
    if constexpr (EXP == 41 || EXP == 46) vbase_tm = wave * 8 * ldb + (lane >> 3) * TG_BKB + (pc ^ tg_swz(li0)) * 16;   // (group `wave` of the half-tile; the piece's rows are whole groups)
  }
  // ---- ... and of the reading side: LDS byte address of the lane's 16 bytes in slot 0 / slot 4
  // (pg_read_a / pg_read_b)
  unsigned adrA[2], adrB[2];                         // [k half h]
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    adrA[h] = (unsigned)(((wr * 64 + r16) * 8 + ((4 * h + g) ^ tg_swz(r16))) * 16);
    adrB[h] = (unsigned)(((wc * (16 * NTQ) + r16) * 8 + ((4 * h + g) ^ tg_swz(r16))) * 16);
  }

  // Row tiles are taken in a SCATTERED order (fused form): row tile bx of the schedule is row tile (bx * perm_s) mod gx of
  // the corpus, perm_s a prime that does not divide gx (1,009 row tiles = 258 k rows apart at 1M rows).  In schedule order the 256 workgroups sit on 64-256 NEIGHBOURING row
  // tiles at any moment; in a corpus stored topic by topic those belong to one or two topics, and every workgroup's
  // candidates of the moment go to the lists of the same few queries (their counters, their next free cache lines).
  // Scattered, the tiles in flight are a sample of the whole corpus, as they are for a shuffled one.  (The query tiles of
  // a row tile keep their place: same XCD, adjacent in time.)
  int perm_s = 1;
  if constexpr (FUSE && kMap == 0) {
    // (a prime that does not divide gx is coprime to it; the first of a fixed list that leaves at least two strides: no loop)
    const int sgx = __builtin_amdgcn_readfirstlane(gx);
    int c = 1;
    if (sgx > 2 * 3 && sgx % 3) c = 3;
    if (sgx > 2 * 7 && sgx % 7) c = 7;
    if (sgx > 2 * 31 && sgx % 31) c = 31;
    if (sgx > 2 * 127 && sgx % 127) c = 127;
    if (sgx > 2 * 1009 && sgx % 1009) c = 1009;
    if (sgx > 2 * 7919 && sgx % 7919) c = 7919;
    perm_s = __builtin_amdgcn_readfirstlane(c);
  }
  auto tile_desc = [&](int j) __attribute__((always_inline)) {
    PgTile t;
    if (j < my_tiles) {
      int bx, by;
      pg_tile_of((int)blockIdx.x + j * G, gx, gy, &bx, &by);
      if constexpr (FUSE && kMap == 0) bx = (int)(((long long)bx * perm_s) % gx);
      if (alt_map) {
        const int x = (int)blockIdx.x & 7, c = (int)blockIdx.x >> 3;
        if (kMap == 1) { by = x & 3; bx = (j * 32 + c) * 2 + (x >> 2); }
        else { by = 2 * (x & 1) + (c & 1); bx = (j * 16 + (c >> 1)) * 4 + (x >> 1); }
      }
      // (the divisions above run on the vector ALU: without these two, hipcc keeps the descriptors
      //  below in VGPRs and wraps EVERY LDS-DMA instruction in a waterfall loop -- four
      //  v_readfirstlane, two compares, a saveexec and a branch per piece, in among the MFMAs)
      bx = __builtin_amdgcn_readfirstlane(bx);
      by = __builtin_amdgcn_readfirstlane(by);
      t.row0 = (int64_t)bx * PG_TILE;
      t.q0 = by * QT;
      const int64_t live = n - t.row0 < PG_TILE ? n - t.row0 : PG_TILE;
      t.a = pg_rsrc(M + t.row0 * ldb, (int)(live * ldb));
      t.b = pg_rsrc(Q + (int64_t)t.q0 * ldb, QT * ldb);
    } else {   // no such tile: zero records, every load through it is dropped
      t.row0 = 0;
      t.q0 = 0;
      t.a = pg_rsrc(M, 0);
      t.b = pg_rsrc(Q, 0);
    }
    return t;
  };
  PgTile cur = tile_desc(0), nxt = tile_desc(1);

  // kind: 0 = B0, 1 = A0, 2 = B1, 3 = A1.  NEXT: the half-tile belongs to the next output tile
  // (a compile-time fact: only the last two k-tiles of a tile stage across the seam).
  auto stage_piece = [&](auto KIND, auto SLOT, auto NEXT, int kt, auto JJ) __attribute__((always_inline)) {
    constexpr int kind = decltype(KIND)::value, slot = decltype(SLOT)::value, jj = decltype(JJ)::value;
    constexpr bool next = decltype(NEXT)::value != 0;
    constexpr bool tm = (EXP == 40 || EXP == 45) && (kind & 1);   // (timing-only: the corpus addressed as if tile-major -- a k-tile of a row tile one contiguous 32 KiB block)
    constexpr bool il = (EXP == 41 || EXP == 46) && (kind & 1);   // (timing-only: as if 8-row-interleaved -- every 1 KiB piece contiguous, a group of 8 rows k-tile by k-tile)
    const int soff = (EXP == 21 || EXP == 25) ? 0 : (tm ? kt * (PG_TILE * TG_BKB) : (il ? kt * 1024 : kt * TG_BKB));   // (ablation 21: always the first k-tile: L2 hits)
    const __amdgpu_buffer_rsrc_t rs = (kind & 1) ? (next ? nxt.a : cur.a) : (next ? nxt.b : cur.b);
    // B, QT = 128: half-tile row li = li0 (0 .. 63) holds query (li >> 4) * 32 + h * 16 + (li & 15) = li0 + (wave >> 1) * 16 + h * 16
    const int rows = (kind & 1) ? jj * 128 + (kind >> 1) * 64
                                : (QT == 256 ? wr * 32 + jj * 128 + (kind >> 1) * 32 : (wave >> 1) * 16 + (kind >> 1) * 16);
    const int vo = tm ? pg_voff(vbase_tm, rows * TG_BKB) : pg_voff(il ? vbase_tm : vbase, rows * ldb);
    const unsigned dst = pg_lds_dest<slot * PG_SLOT * 16 + jj * 8192>(wave_lds);
    if constexpr ((EXP == 20 || EXP == 40 || EXP == 45 || EXP == 52 || EXP == 41 || EXP == 46 || EXP == 62 || EXP == 63) && (kind & 1))   // (corpus rows nontemporal: the single-query-tile form, see launch_tiled_eb)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, vo, soff, 0, 2);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, vo, soff, 0, 0);
  };
  auto stage = [&](auto KIND, auto SLOT, auto NEXT, int kt) __attribute__((always_inline)) {
    stage_piece(KIND, SLOT, NEXT, kt, std::integral_constant<int, 0>{});
    if constexpr ((decltype(KIND)::value & 1) || PB == 2) stage_piece(KIND, SLOT, NEXT, kt, std::integral_constant<int, 1>{});
  };
  // side data of tile `t` into copy `par`: every wave issues the same instructions (waves 4-7
  // repeat waves 0-3: same bytes to the same place), 256 bytes each
  auto stage_side = [&](const PgTile& t, int par) __attribute__((always_inline)) {
    // (formed from an opaque copy of the thread id, per tile: computed once per kernel, `col * 4` and
    //  `col * fthr_stride * 4` stay live across the k loop, and in the fp8 form -- 256 registers -- they were
    //  spilled: each reload waits vmcnt(0), i.e. drains the LDS-DMA ring, twice per output tile)
    int tid_s = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_s));
    const int col = tid_s & 255;   // = (wave & 3) * 64 + lane
    char* side = (char*)pg_lds;
    if constexpr (FUSE) {
      const int live = nq - t.q0 < QT ? nq - t.q0 : QT;
      // (stride 0: one threshold for every query -- pair mode -- is one 4-byte record read by every lane)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(fthr + (int64_t)t.q0 * fthr_stride), 0,
                                                                          fthr_stride ? live * fthr_stride * 4 : 4, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(side + PG_SIDE_THR + par * 1024 + (wave & 3) * 256),
                                               4, col * fthr_stride * 4, 0, 0, 0);
    }
    if constexpr (EB == 1) {
      const int64_t lrows = n - t.row0 < PG_TILE ? n - t.row0 : PG_TILE;
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(rscale + t.row0), 0, (int)lrows * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(qscale + t.q0), 0, QT * 4, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (__attribute__((address_space(3))) void*)(side + PG_SIDE_RS + par * 1024 + (wave & 3) * 256), 4, col * 4, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void*)(side + PG_SIDE_QS + par * 1024 + (wave & 3) * 256), 4, col * 4, 0, 0, 0);
    }
  };

  f32x4_t acc[MT][NT];
  u32x4 fa[4][2], fb0[NTQ][2], fb1[NTQ][2];
  if constexpr (EXP == 2 || EXP == 5 || EXP == 25 || EXP == 26 || EXP == 45 || EXP == 46) {   // (ablation without fragment reads: defined operands)
    const u32x4 c = {0x3c003c00u + (uint32_t)lane, 0x3c003c00u, 0x38003800u, 0x3c003c00u};
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i][0] = fa[i][1] = c;
#pragma unroll
    for (int i = 0; i < NTQ; ++i) fb0[i][0] = fb0[i][1] = fb1[i][0] = fb1[i][1] = c;
  }
  // one quadrant: rows 64 i .. 64 i + 63, queries 32 j .. 32 j + 31 of the wave tile
  // s0 / s1: the phase's two LDS-DMA pieces when they are issued BETWEEN the MFMAs (kDmaInMma)
  auto mma = [&](auto I, auto J, const u32x4 (&fb)[NTQ][2], auto&& s0, auto&& s1) __attribute__((always_inline)) {
    constexpr int i = decltype(I)::value, j = decltype(J)::value;
    // (kOrder, timing experiments 60-63: the order of a quadrant's MFMAs.  0: corpus fragment outer, query fragment inner;
    //  1: the same as a snake -- ONE operand changes from each MFMA to the next; 2: query fragment outer, snake over the corpus
    //  fragments)
    constexpr int kOrder = (EXP == 60 || EXP == 62) ? 1 : ((EXP == 61 || EXP == 63) ? 2 : 0);
    auto step_of = [](int s, int* mt, int* nt) __attribute__((always_inline)) {
      if (kOrder == 2) { *nt = s / 4; *mt = (*nt & 1) ? 3 - s % 4 : s % 4; }
      else { *mt = s / NTQ; *nt = (kOrder == 1 && (*mt & 1)) ? NTQ - 1 - s % NTQ : s % NTQ; }
    };
    if constexpr (EB == 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < 4 * NTQ; ++s) {
          int mt, nt;
          step_of(s, &mt, &nt);
          acc[i * 4 + mt][j * NTQ + nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
              __builtin_bit_cast(h8, fa[mt][h]), __builtin_bit_cast(h8, fb[nt][h]), acc[i * 4 + mt][j * NTQ + nt], 0, 0, 0);
          if (s % NTQ != NTQ - 1) continue;
          const int grp = s / NTQ;   // (0 .. 3: after every NTQ-th MFMA)
#if PG_DMA_STAGGER
          // the four waves of a group share the CU's vector-memory front end (64 B/clk: 16 cycles per
          // piece): wave wc issues after MFMA pair wc, so no piece queues behind another wave's
          if (wc == grp) {
            __builtin_amdgcn_sched_barrier(0);
            if (h == 0) s0(); else s1();
            __builtin_amdgcn_sched_barrier(0);
          }
#else
          if (h == 0 && grp == PG_DMA_AT0) { __builtin_amdgcn_sched_barrier(0); s0(); __builtin_amdgcn_sched_barrier(0); }
          if (h == 1 && grp == PG_DMA_AT1) { __builtin_amdgcn_sched_barrier(0); s1(); __builtin_amdgcn_sched_barrier(0); }
#endif
        }
    } else {
      i32x8 x[4], y[NTQ];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const u32x4 al = fa[mt][0], ah = fa[mt][1];
        x[mt] = (i32x8){(int)al.x, (int)al.y, (int)al.z, (int)al.w, (int)ah.x, (int)ah.y, (int)ah.z, (int)ah.w};
      }
#pragma unroll
      for (int nt = 0; nt < NTQ; ++nt) {
        const u32x4 bl = fb[nt][0], bh = fb[nt][1];
        y[nt] = (i32x8){(int)bl.x, (int)bl.y, (int)bl.z, (int)bl.w, (int)bh.x, (int)bh.y, (int)bh.z, (int)bh.w};
      }
#pragma unroll
      for (int s = 0; s < 4 * NTQ; ++s) {
        int mt, nt;
        step_of(s, &mt, &nt);
        acc[i * 4 + mt][j * NTQ + nt] =
            __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x[mt], y[nt], acc[i * 4 + mt][j * NTQ + nt], 0, 0, 0, 0, 0, 0);
        if (s == NTQ - 1) { __builtin_amdgcn_sched_barrier(0); s0(); __builtin_amdgcn_sched_barrier(0); }
        if (s == 3 * NTQ - 1) { __builtin_amdgcn_sched_barrier(0); s1(); __builtin_amdgcn_sched_barrier(0); }
      }
    }
  };
#define PG_C(v) std::integral_constant<int, (v)>{}
  constexpr bool kStage = EXP != 1 && EXP != 26, kRead = EXP != 2 && EXP != 5 && EXP != 25 && EXP != 26 && EXP != 45 && EXP != 46, kMma = EXP != 3 && EXP != 5 && EXP != 25 && EXP != 26 && EXP != 45 && EXP != 46;   // (45: tile-major corpus, DMA + barriers only)   // (26: barriers only)   // (EXP 5: LDS-DMA and barriers only)
  constexpr bool kDmaInMma = (PG_DMA_IN_MMA != 0) != (EXP == 30);   // (EXP 30: the other placement, A/B)
  // (the between-the-MFMAs placement issues BOTH pieces of a half-tile unconditionally; at 128-query tiles a query half-tile is
  //  one piece per wave -- stage() skips the second there -- so that placement, and with it variants 8 / 9 of
  //  svs_index_set_variant, exists at 256-query tiles only: the 128-query route always takes EXP 20 / 0)
  static_assert(!(kDmaInMma && QT == 128), "LDS-DMA pieces between the MFMAs: 256-query tiles only");
  // what follows the loads of a phase: [retire the B reads] wait for the NEXT phase's data, barrier,
  // fragments in, 16 MFMAs at raised priority (keeps hipcc from moving them over the barriers), barrier
#define PG_SYNC_AND_MMA(PH, I, J, FB, LGKM8, LAND, KIND, SLOT, NEXT, KT_)    \
  do {                                                                      \
    if constexpr (kStage && !kDmaInMma) stage(PG_C(KIND), PG_C(SLOT), PG_C(NEXT), KT_); \
    if (LGKM8 && kRead && !kDmaInMma) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(EXP == 53 ? 4 : 8) : "memory");  \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PG_VMCNT_OF(KIND)) : "memory");   \
    PG_STAMP(PH, 0);                                                        \
    __builtin_amdgcn_s_barrier();                                           \
    if constexpr (kRead) { LAND; }                                          \
    PG_STAMP(PH, 1);                                                        \
    __builtin_amdgcn_sched_barrier(0);                                      \
    __builtin_amdgcn_s_setprio(1);                                          \
    if constexpr (kMma)                                                     \
      mma(PG_C(I), PG_C(J), FB,                                             \
          [&]() { if constexpr (kStage && kDmaInMma) stage_piece(PG_C(KIND), PG_C(SLOT), PG_C(NEXT), KT_, PG_C(0)); }, \
          [&]() { if constexpr (kStage && kDmaInMma) stage_piece(PG_C(KIND), PG_C(SLOT), PG_C(NEXT), KT_, PG_C(1)); }); \
    else if constexpr (kStage && kDmaInMma) stage(PG_C(KIND), PG_C(SLOT), PG_C(NEXT), KT_); \
    __builtin_amdgcn_s_setprio(0);                                          \
    __builtin_amdgcn_sched_barrier(0);                                      \
    PG_STAMP(PH, 2);                                                        \
    __builtin_amdgcn_s_barrier();                                           \
    PG_STAMP(PH, 3);                                                        \
  } while (0)

  // k-tile `kt` of the current output tile; PAR = parity of the k-tile in the stream (its slots
  // are 4 PAR + kind, the other parity's are the ones being refilled).  TAIL: 0 = anywhere but the
  // tile's last two k-tiles, 1 / 2 = the last but one / the last: their prefetch crosses into the
  // next output tile (its k-tiles 0 and 1), through the next tile's descriptors.
#ifdef PG_TRACE
  bool tracing = false;
#endif
  auto ktile = [&](auto PAR, auto TAIL, int kt) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value, tail = decltype(TAIL)::value;
    constexpr int mine = 4 * par, other = 4 * (par ^ 1);
    // half-tile staged in phase 0 is of k-tile kt + 1, in phases 1-3 of k-tile kt + 2
    constexpr bool n1 = tail == 2, n2 = tail != 0;
    const int k1 = tail == 2 ? 0 : kt + 1, k2 = tail == 0 ? kt + 2 : tail - 1;
    // phase 0: B0, A0 in; quadrant (0, 0); stage A1 of the next k-tile
    if constexpr (kRead) {
      pg_read_b<mine + 0>(fb0, adrB);
      __builtin_amdgcn_sched_barrier(0);
      pg_read_a<mine + 1, EXP == 53 ? 2 : 0>(fa, adrA);   // (53: what phase 0 would cost with four of its twelve reads elsewhere -- results wrong)
    }
    PG_SYNC_AND_MMA(0, 0, 0, fb0, true, (pg_landed_b(fb0), pg_landed_a(fa)), 3, other + 3, n1, k1);
    // phase 1: B1 in; quadrant (0, 1); stage B0 of the k-tile after next (over this one's B0)
    if constexpr (kRead) pg_read_b<mine + 2>(fb1, adrB);
    PG_SYNC_AND_MMA(1, 0, 1, fb1, false, pg_landed_b(fb1), 0, mine + 0, n2, k2);
    // phase 2: A1 in; quadrant (1, 1); stage A0
    if constexpr (kRead) pg_read_a<mine + 3>(fa, adrA);
    PG_SYNC_AND_MMA(2, 1, 1, fb1, false, pg_landed_a(fa), 1, mine + 1, n2, k2);
    // phase 3: nothing to read; quadrant (1, 0); stage B1
    PG_SYNC_AND_MMA(3, 1, 0, fb0, false, (void)0, 2, mine + 2, n2, k2);
  };

  // ---- flush of the candidates parked by the PREVIOUS tile (copy pp, queries pq0 ..): step A
  // issues one returning global atomic per candidate (slot in the query's list), step B -- four
  // k-tiles = 32 counted vmcnt waits later -- stores the keys
  constexpr int PG_FSLOTS = PG_PARK / PG_THREADS;   // parked entries a thread flushes per tile
  static_assert(PG_FSLOTS == 3 || PG_FSLOTS == 1, "flush_b's waits name the slot registers one by one");
  uint32_t fslot[3] = {0u, 0u, 0u};
  int last_wcount = PG_SPARSE_MAX + 1;   // survivors this wave parked in its previous tile (wave-uniform; the first tile takes the sweeps)
  int f_pp = 0, f_q0 = 0;     // wave-uniform
  int64_t f_row0 = 0;
  // (tid / lane copies behind an empty asm: what the flush and the epilogue derive from them is recomputed per
  //  tile -- a handful of ALU instructions -- instead of being computed once, kept live across the k loop and,
  //  the register file being full there, spilled: every reload of a spilled VGPR waits vmcnt(0), i.e. for the
  //  whole LDS-DMA ring)
  // One atomic per RUN of a query, not per candidate: the sweeps park lane by lane, so the 32 rows a lane holds for one
  // query sit in consecutive entries, and a corpus stored topic by topic fills whole runs with ONE query (every row of a
  // tile a neighbour of the same few queries).  All 256 workgroups are then on that topic's tiles at the same moment, and
  // one atomic per candidate put ~400 of them per tile on the same 32 counters: their latency, no longer covered by two
  // k-tiles, came out at the counted waits (1M x 1536 f16 in 32 / 256 topics: 3.6 / 4.5 ms of score stage against 2.9
  // shuffled; the materialised path, which has no flush, 3.2 whatever the order).  Each round of 64 entries takes the
  // query of its first live entry, counts the lanes that share it (a ballot) and lets the first of them add that count;
  // the others add for themselves as before.  flush_b recomputes the same grouping from the same LDS words instead of
  // keeping it in registers across the two k-tiles in between.  Only waves that parked PG_AGG_MIN candidates or more do any of
  // this: the ordinary tile (a shuffled corpus parks ~52 per wave) keeps round 3's flush, instruction for instruction -- with
  // the grouping on every tile configs[2]'s kernel was 0-6 % slower by box (same-box A/B, tools/ab_before_after.sh).
  // (a second run per round -- a lane that holds two of the topic's queries parks them alternately -- was measured too:
  //  no further gain, 3.4-3.6 ms either way)
  auto flush_group = [&](bool valid, int q, int* first, unsigned long long* same) __attribute__((always_inline)) -> bool {
    const unsigned long long vm = __ballot(valid);
    if (vm == 0) return false;   // (wave-uniform)
    *first = (int)__builtin_ctzll(vm);
    const int q0 = __builtin_amdgcn_readlane(q, *first);
    *same = __ballot(valid && q == q0);
    return true;
  };
  auto flush_a = [&]() __attribute__((always_inline)) {
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane_f = tid & 63;
#pragma unroll
    for (int r = 0; r < PG_PARK / PG_THREADS; ++r) {
      const int e = tid + r * PG_THREADS, w = e / (PG_PARK / 8), i = e % (PG_PARK / 8);
      const uint32_t nwf = pg_lds_read_u32(PG_SIDE_CNT + (f_pp * 8 + w) * 4), nw = nwf & 0x7fffffffu;   // (bit 31: flush by runs, set by the epilogue)
      // (ONE asm site per round for both forms: the atomic's result register is written when the atomic RETURNS, long
      //  after the asm statement -- with a site in each branch hipcc joined the two results through a copy made right
      //  behind the asm, i.e. of a register the atomic had not written yet)
      bool issue;
      uint32_t add = 1u;
      int q = 0;
      if (__builtin_amdgcn_readfirstlane((int)nwf) >= 0) {   // (the ordinary tile: one atomic per candidate, no grouping in the k loop)
        issue = (uint32_t)i < nw;
        if (issue) {
          q = (int)(pg_lds_read_u32(PG_SIDE_KEY + (f_pp * PG_PARK + e) * 8 + 4) >> 8);
          // (a padded query of the last query tile can park a candidate only through a NaN score -- its threshold is
          //  +inf -- and has no list: dropped here and in flush_b)
          issue = f_q0 + q < nq;
        }
      } else {
        q = (int)(pg_lds_read_u32(PG_SIDE_KEY + (f_pp * PG_PARK + e) * 8 + 4) >> 8);
        const bool valid = (uint32_t)i < nw && f_q0 + q < nq;
        int first = 0;
        unsigned long long same = 0;
        const bool any = flush_group(valid, q, &first, &same);
        const bool in_run = ((same >> lane_f) & 1ull) != 0;
        issue = any && valid && (!in_run || lane_f == first);
        add = in_run ? (uint32_t)__builtin_popcountll(same) : 1u;
      }
      if (issue) {
        uint32_t* p = fstate_words + (int64_t)(f_q0 + q) * fstate_stride;
        asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(fslot[r]) : "v"(p), "v"(add) : "memory");
      }
    }
  };
  // drain == false: the caller guarantees that at least PG_VMCNT vector-memory instructions were issued after
  // flush_a's atomics (see the invariant at the call site in the k loop); drain == true: wait for everything
  auto flush_b = [&](bool drain) __attribute__((always_inline)) {
    if (!drain) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(fslot[0]), "+v"(fslot[1]), "+v"(fslot[2]) : "n"(PG_VMCNT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(fslot[0]), "+v"(fslot[1]), "+v"(fslot[2])::"memory");
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane_f = tid & 63;
#pragma unroll
    for (int r = 0; r < PG_PARK / PG_THREADS; ++r) {
      const int e = tid + r * PG_THREADS, w = e / (PG_PARK / 8), i = e % (PG_PARK / 8);
      const uint32_t nwf = pg_lds_read_u32(PG_SIDE_CNT + (f_pp * 8 + w) * 4), nw = nwf & 0x7fffffffu;
      if (__builtin_amdgcn_readfirstlane((int)nwf) >= 0) {   // (as flush_a decided, from the same word)
        if ((uint32_t)i < nw && fslot[r] < fcap) {
          const uint64_t sc = pg_lds_read_u64(PG_SIDE_KEY + (f_pp * PG_PARK + e) * 8);   // (score bits, code)
          const uint32_t code = (uint32_t)(sc >> 32);
          if (f_q0 + (int)(code >> 8) >= nq) continue;
          fcand[(int64_t)(f_q0 + (int)(code >> 8)) * fcap + fslot[r]] =
              ((uint64_t)score_key(__builtin_bit_cast(float, (uint32_t)sc)) << 32) | (uint32_t)(pairs.row_base + f_row0 + (code & 255u));
        }
        continue;
      }
      const uint64_t sc = pg_lds_read_u64(PG_SIDE_KEY + (f_pp * PG_PARK + e) * 8);   // (score bits, code)
      const uint32_t code = (uint32_t)(sc >> 32);
      const int q = (int)(code >> 8);
      const bool valid = (uint32_t)i < nw && f_q0 + q < nq;
      int first;
      unsigned long long same;
      if (!flush_group(valid, q, &first, &same)) continue;
      // (the run's base came back to its first lane; a lane's place in the run is its rank inside the mask)
      const uint32_t run_base = (uint32_t)__builtin_amdgcn_readlane((int)fslot[r], first);
      const bool in_run = ((same >> lane_f) & 1ull) != 0;
      const uint32_t slot = in_run ? run_base + __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u)) : fslot[r];
      if (valid && slot < fcap)
        fcand[(int64_t)(f_q0 + q) * fcap + slot] =
            ((uint64_t)score_key(__builtin_bit_cast(float, (uint32_t)sc)) << 32) | (uint32_t)(pairs.row_base + f_row0 + (code & 255u));
    }
  };

  // ---- epilogue of one tile: scales (fp8), then scores out (materialised) or candidates parked
  auto epilogue = [&](const PgTile& t, int par) __attribute__((always_inline)) {
    const int wrow = wr * 128, wq = wc * (QT / 4);
    int lane_e = (int)threadIdx.x;   // (the copy is made opaque BEFORE the mask: `threadIdx.x & 63` itself was kept live -- and spilled in the fp8 form)
    asm volatile("" : "+v"(lane_e));
    lane_e &= 63;
    const int lane = lane_e, r16 = lane_e & 15, g = lane_e >> 4;   // (shadow the kernel's: see flush_a)
    if constexpr (!FUSE) {
      // (rows / queries past the ends are not stored; fp8 scales are applied by tg_epilogue's own loads: this
      //  path is the prefix pass and set_variant(6), not the hot one)
      tg_epilogue<false, EB, MT, NT, PG_TILE>(acc, t.row0, t.q0, wrow, wq, lane, n, nq, scores, sstride, fstate_words, fstate_stride,
                                              fcand, fcap, fthr, fthr_stride, rscale, qscale);
    } else {
      // One pass over the wave's 32 x 4 score registers.  Per register: one compare (its lane mask
      // lands in scalar registers), one scalar test; only if some lane survives (a third of the
      // time at k = 100 over 1M rows) do those lanes compute their slot -- wave counter + rank inside
      // the mask -- and write (key, query) into the wave's eighth of the parking lot.  A wave that
      // overfills its eighth sends the rest straight to the global candidate lists (below); only a
      // query whose GLOBAL list overflows (32,768 candidates: rows ordered by similarity to it, or a
      // query made of NaNs -- every score survives) is marked by select_final and re-run by the host
      // through the materialised path.
      const int lr0 = wrow + 4 * g;                                        // the lane's first row inside the tile
      const int lim = (int)(n - t.row0 < PG_TILE ? n - t.row0 : PG_TILE);  // live rows of this tile
      if constexpr (EB == 1) {   // per-row and per-query dequantisation scales, applied in place (as tg_epilogue: v * (rs * qs))
        // two scores per instruction (v_pk_mul_f32: each half is an ordinary IEEE f32 product, so the bits are those of
        // the scalar form): 128 instead of 256 multiplies per lane
        float q1[NT];
        f32x4_t rsv[MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) pg_lds_issue_f32(q1[j], PG_SIDE_QS + par * 1024 + (wq + j * 16 + r16) * 4);
#pragma unroll
        for (int i = 0; i < MT; ++i) pg_lds_issue_f32x4(rsv[i], PG_SIDE_RS + par * 1024 + (lr0 + i * 16) * 4);
        pg_lds_landed(q1);
        pg_lds_landed(rsv);
        f32x2_t qs[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) qs[j] = (f32x2_t){q1[j], q1[j]};
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const f32x4_t rs = rsv[i];
          const f32x2_t rs01 = {rs[0], rs[1]}, rs23 = {rs[2], rs[3]};
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            // (in place by construction: left to hipcc the 128 products go to new registers, the accumulator
            //  quads stay live until their last element is read, and ~50 values spill)
            f32x2_t lo = {acc[i][j][0], acc[i][j][1]}, hi = {acc[i][j][2], acc[i][j][3]};
            const f32x2_t s01 = rs01 * qs[j], s23 = rs23 * qs[j];
            asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(lo) : "v"(s01));
            asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(hi) : "v"(s23));
            acc[i][j] = (f32x4_t){lo[0], lo[1], hi[0], hi[1]};
          }
        }
      }
      constexpr int WCAP = PG_PARK / 8;
      const unsigned kadr = PG_SIDE_KEY + (par * PG_PARK + wave * WCAP) * 8;
      int wcount = 0;   // wave-uniform
      auto pass = [&](auto FULL, auto QALL) __attribute__((always_inline)) {
        constexpr bool full = decltype(FULL)::value, qall = decltype(QALL)::value;   // whole tile inside the corpus / the batch
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int ql = wq + j * 16 + r16;
          const float thr = pg_lds_read_f32(PG_SIDE_THR + par * 1024 + ql * 4);
          // pair mode (svs_index_top_pairs, TgPairs in gemm_tiled.h): the queries ARE corpus rows; a score counts only
          // for a row above the query's own, and only for queries this chunk is responsible for
          // (32-bit and tile-local: rows of this tile count from local row plo + 1 on; opaque, or hipcc forms all 128
          //  64-bit comparisons of a lane ahead of time and spills them)
          int plo = -1;
          bool mine = true;
          if (pairs.on) {
            const long long qrow = pairs.query_row0 + t.q0 + ql, dlt = qrow - (pairs.row_base + t.row0);
            plo = dlt < -1 ? -1 : (dlt > PG_TILE ? PG_TILE : (int)dlt);
            mine = qrow >= pairs.first_query;
          }
          asm volatile("" : "+v"(plo));
          const unsigned long long qokm = __ballot(t.q0 + ql < nq && mine);
          // (opaque: the 128 codes of a lane do not depend on the tile, and hipcc otherwise computes them all at
          //  kernel entry and spills them -- ~500 registers' worth of scratch traffic behind vmcnt(0) waits)
          uint32_t cbase = (uint32_t)(lr0 | (ql << 8));
          asm volatile("" : "+v"(cbase));
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            // the four compares of a tile first (their lane masks land in scalar registers: a
            // compare -> scalar test -> branch chain per register costs ~60 cycles of latency each)
            float v[4];
            unsigned long long mask[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[i][j][r];
              // lanes whose score is not below the threshold (unordered-or-greater-equal: a NaN passes)
              mask[r] = __builtin_amdgcn_fcmpf(v[r], thr, 11 /* FCMP_UGE */);
              if constexpr (!qall) mask[r] &= qokm;
              if constexpr (!full)
                mask[r] &= __ballot(lr0 + i * 16 + r < lim && lr0 + i * 16 + r > plo);
            }
            // ONE scalar test per group of four registers (the no-survivor path then runs 4 compares, 3 s_or and
            // a not-taken branch per group); with thresholds from a long prefix (configs[4]: 42 survivors per tile)
            // nine groups in ten are empty
            if (__builtin_expect((mask[0] | mask[1] | mask[2] | mask[3]) != 0, 0)) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (mask[r] != 0) {
                  const bool hit = ((mask[r] >> lane) & 1ull) != 0;
                  const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask[r], 0u));
                  if (hit) {
                    const int slot = wcount + rank;
                    if (slot < WCAP) {   // (score bits, code): the order-preserving key is made at flush time
                      pg_lds_write_u64(kadr + slot * 8, ((uint64_t)(cbase + (uint32_t)(i * 16 + r)) << 32) | __builtin_bit_cast(uint32_t, v[r]));
                    } else {
                      // the wave's eighth is full: this candidate goes straight to the query's global list (a
                      // returning atomic and a store, and hipcc's vmcnt(0) behind the atomic drains the DMA
                      // ring -- the price of a tile full of one query's neighbours, paid only there)
                      uint32_t* hp = fstate_words + (int64_t)(t.q0 + ql) * fstate_stride;
                      const uint32_t gs = __hip_atomic_fetch_add(hp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                      if (gs < fcap)
                        fcand[(int64_t)(t.q0 + ql) * fcap + gs] = ((uint64_t)score_key(v[r]) << 32) | (uint32_t)(pairs.row_base + t.row0 + lr0 + i * 16 + r);
                      else if (fthr_stride)
                        // The query's list is full: select_final will hand it back for a re-run whatever else arrives, so
                        // its threshold goes to +inf -- the tiles that stage their thresholds after this store (every
                        // workgroup's next one) park nothing for it.  Without this a corpus sorted by similarity to the
                        // queries ran EVERY tile through this one-atomic-per-score path: 14 ms instead of 0.9 per 256
                        // queries over 1M rows (tools/sorted_corpus_time.py).  (A stale read elsewhere is harmless: the
                        // only difference is how many doomed candidates are still appended.)
                        *(volatile float*)(fthr + (int64_t)(t.q0 + ql) * fthr_stride) = __builtin_inff();
                    }
                  }
                  wcount += __builtin_popcountll(mask[r]);
                }
              }
            }
          }
        }
      };
      // Interior tiles (all but the last row tile and the last query tile): the same result WITHOUT a
      // branch per register.  With thresholds from an n / 64 prefix ~420 candidates survive per tile:
      // a third of the registers held one, and the compare -> mask -> branch -> rank path above cost
      // 21 k cycles per tile (3.9 k with no survivor).  Here: (1) every lane counts its own survivors
      // (compare + add-with-carry per register), (2) one wave scan gives each lane the first of ITS
      // slots in the wave's eighth, (3) a second sweep parks them: v_cmpx makes the survivors the
      // active lanes, they store (score, code) and step their address, exec is restored -- five
      // instructions per register, the same whatever survives.
      bool w_dense = false, w_few = false;   // (wave-uniform)
      auto sweep = [&]() __attribute__((always_inline)) -> bool {   // true: more survivors than the wave's eighth holds, nothing parked
        float thr[NT];
        uint32_t base[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) pg_lds_issue_f32(thr[j], PG_SIDE_THR + par * 1024 + (wq + j * 16 + r16) * 4);
        pg_lds_landed(thr);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int ql = wq + j * 16 + r16;
          // (the last query tile of a batch that is not a whole number of tiles: its padded queries -- zero vectors
          //  -- take no part: nothing reaches +inf but a NaN, and the flush drops those)
          if (t.q0 + ql >= nq) thr[j] = __builtin_inff();
          base[j] = (uint32_t)(lr0 | (ql << 8));
        }
        auto val = [&](int i, int j, int r) { return acc[i][j][r]; };
        uint32_t cnt = 0;
        {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) cnt += !(val(i, j, r) < thr[j]) ? 1u : 0u;   // (a NaN passes)
        }
        // (a lane -- 32 rows x 4 queries -- with PG_LANE_DENSE survivors or more has met one of its queries' own topic: the
        //  wave's flush goes by runs.  A shuffled corpus gives a lane 0.8 on average: one wave in a hundred says so.)
        w_dense = __ballot(cnt >= (uint32_t)PG_LANE_DENSE) != 0;
        // inclusive scan over the wave: four row shifts, then the two row broadcasts
        uint32_t x = cnt;
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
        int total = __builtin_amdgcn_readlane((int)x, 63);
        // more than the wave's eighth holds (rows ordered by topic: a tile full of one query's neighbours): the
        // register-by-register path (the caller's) parks what fits and sends the rest to the global lists
        if (__builtin_expect(total > WCAP, 0)) return true;
        wcount = total;
        if (total != 0) {
          unsigned ak = kadr + (x - cnt) * 8;   // the lane's first slot
          // (the code is computed INSIDE the asm, from a literal: handed in as an operand, hipcc computes all 128
          //  codes ahead, spills them, and reloads each behind a vmcnt(0) that drains the LDS-DMA ring)
          pg_static_for<NT>([&](auto J) {
            pg_static_for<MT>([&](auto I) {
              pg_static_for<4>([&](auto R) {
                constexpr int j = decltype(J)::value, i = decltype(I)::value, r = decltype(R)::value;
                pg_park_if<i * 16 + r>(ak, val(i, j, r), thr[j], base[j]);
              });
            });
          });
        }
        return false;
      };
      // pair mode: tiles wholly above the diagonal are ordinary tiles, tiles wholly on or below it hold nothing,
      // the few that straddle it (or hold queries of the previous chunk) take the masked path
      const long long prow0 = pairs.row_base + t.row0, pq0 = pairs.query_row0 + t.q0;
      const bool pair_plain = !pairs.on || (prow0 > pq0 + QT - 1 && pq0 >= pairs.first_query);
      const bool pair_empty = pairs.on && prow0 + PG_TILE - 1 <= pq0;
      // Interior tiles take the two branch-free sweeps when the wave's last tile held many survivors (configs[2]:
      // ~50 per wave and tile, a third of the registers hold one) and the grouped compare-and-branch path when it
      // held few (configs[4], thresholds from a 156 k-row prefix: ~5 per wave and tile -- the sweeps' 7 vector
      // instructions per register against 1 + a scalar test per four).  Either is exact whatever survives.
      // (one call site per form of pass(): each is ~10 k lines of ISA)
      bool general = !pair_empty && !(lim == PG_TILE && pair_plain && EXP != 31);   // (EXP 31: the branchy path everywhere, A/B)
      bool grouped_full = false;
      if (!pair_empty && !general) {
        // (a partly filled last QUERY tile is an interior tile too: the sweeps give its padded queries +inf
        //  thresholds; only the grouped path needs the per-lane query mask, i.e. the general form of pass())
        const bool few = last_wcount <= PG_SPARSE_MAX;
        w_few = few;
        if (few) {
          if (t.q0 + QT <= nq) grouped_full = true;
          else general = true;
        } else if (sweep()) general = true;   // (the sweeps found more than the wave's eighth holds: register by register)
      }
      // (measured and rejected: all 32 groups' compares first, their any-survivor bits folded into one scalar word per
      //  query column, THEN 32 bit tests -- no compare -> scalar test -> branch chain per group: 8.0 k cycles per tile
      //  against 7.3 k for this form, profiles/r3_gemm_phased_bench_fp8_grouped_bits.txt)
      if (grouped_full) pass(std::true_type{}, std::true_type{});
      if (general) pass(std::false_type{}, std::false_type{});
      last_wcount = EXP == 32 ? PG_SPARSE_MAX + 1 : wcount;   // (EXP 32: the sweeps on every interior tile, A/B)
      if (lane == 0)
        pg_lds_write_u32(PG_SIDE_CNT + (par * 8 + wave) * 4, (uint32_t)(wcount < WCAP ? wcount : WCAP) | ((w_dense || wcount >= PG_AGG_MIN || (w_few && wcount >= PG_FEW_DENSE)) ? 0x80000000u : 0u));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // parked entries written before this wave's next barrier
    }
  };

  // ---- prologue: half-tiles 0 .. 6 (k-tile 0 whole, B0 A0 B1 of k-tile 1), wait for the first
  // two, one barrier for everybody, then waves 4-7 drop one barrier behind
  stage(PG_C(0), PG_C(0), PG_C(0), 0);
  stage(PG_C(1), PG_C(1), PG_C(0), 0);
  stage(PG_C(2), PG_C(2), PG_C(0), 0);
  stage(PG_C(3), PG_C(3), PG_C(0), 0);
  stage(PG_C(0), PG_C(4), PG_C(0), 1);
  stage(PG_C(1), PG_C(5), PG_C(0), 1);
  stage(PG_C(2), PG_C(6), PG_C(0), 1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PG_VMCNT) : "memory");
  __builtin_amdgcn_s_barrier();
  const bool behind = EXP == 7 ? false : wr == 1;
  if (behind) __builtin_amdgcn_s_barrier();

  // (KT is even -- the host checks it -- so a tile always starts on slot parity 0.  A run-time
  // parity, `if (par) ktile<1> else ktile<0>`, made hipcc spill ~190 registers: it gave up
  // keeping the accumulators in place across the join of the two bodies.)
  unsigned long long loop_t0 = 0, loop_t1 = 0, loop_cycles = 0;   // (PG_CLOCKS builds: cycles inside the k loops)
  unsigned long long epi_t0 = 0, epi_t1 = 0, epi_cycles = 0;      // (... and inside the epilogues)
  (void)loop_cycles; (void)epi_cycles;
  for (int T = 0; T < my_tiles; ++T) {
#ifdef PG_TRACE
    tracing = blockIdx.x == PG_TRACE_BLOCK && T == PG_TRACE_TILE;
#endif
    const int tp = T & 1;
    if constexpr (FUSE) stage_side(cur, tp);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    PG_LOOP_CLOCK(loop_t0);
    for (int kt = 0; kt < KT - 2; kt += 2) {
      // The previous tile's candidates: atomics out before this pair of k-tiles (flush_a), slots used after it
      // (flush_b); the returned slots live in registers only across this straight-line stretch.
      // INVARIANT (what the counted wait in flush_b rests on).  flush_a's returning atomics are issued by inline asm,
      // so hipcc inserts no wait for them; they sit in the wave's in-order vmcnt queue like any load.  Between
      // flush_a and flush_b the wave issues the LDS-DMA pieces of two k-tiles: PG_FLUSH_YOUNGER = 16 vector-memory
      // instructions at 256-query tiles (12 at 128-query tiles), all YOUNGER than the atomics.  `s_waitcnt vmcnt(PG_VMCNT)`
      // retires everything but the PG_VMCNT youngest, so it covers the atomics iff PG_FLUSH_YOUNGER >= PG_VMCNT.
      // A build that issues no DMA in the loop (ablations 1 and 26: kStage false) or forces a larger count
      // (PG_VMCNT_FORCE) does not have that cover and must wait vmcnt(0) instead: round 2's "no LDS-DMA in loop"
      // ablation waited the counted form, the slots landed after hipcc had reused fslot[]'s registers, and the
      // candidate store went through a corrupted address -- the GPU memory fault of gpurun_out/pb8.log.
      // The flush needs loop iteration kt == PG_FLUSH_KT to exist in THIS loop (not in the two tail k-tiles):
      // PG_FLUSH_KT + 2 <= PG_MIN_KT - 2, and the host (phased_ok) refuses rows shorter than PG_MIN_KT k-tiles.
      constexpr int PG_FLUSH_YOUNGER = 2 * (2 * 2 + 2 * PB);   // two k-tiles: 2 A half-tiles of 2 pieces + 2 B half-tiles of PB each
      constexpr bool kFlushCovered = kStage && PG_FLUSH_YOUNGER >= PG_VMCNT;
#ifndef PG_VMCNT_FORCE
      static_assert(EXP != 0 && EXP != 20 && EXP != 30 && EXP != 31 && EXP != 32 ? true : kFlushCovered,
                    "shipped forms: the counted wait in flush_b must cover flush_a's atomics");
#endif
      static_assert(PG_FLUSH_KT % 2 == 0 && PG_FLUSH_KT + 2 <= PG_MIN_KT - 2, "the flush iteration must lie inside the main k loop");
      const bool flush = FUSE && T > 0 && kt == PG_FLUSH_KT;
      if (flush) flush_a();
      ktile(PG_C(0), PG_C(0), kt);
      ktile(PG_C(1), PG_C(0), kt + 1);
      if (flush) flush_b(!kFlushCovered);
    }
    ktile(PG_C(0), PG_C(1), KT - 2);
    ktile(PG_C(1), PG_C(2), KT - 1);
    PG_LOOP_CLOCK(loop_t1);
    loop_cycles += loop_t1 - loop_t0;
    if constexpr (EXP == 14) {   // (ablation: no epilogue; the accumulators stay live)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else {
      // Both groups run their epilogues TOGETHER: waves 0-3 wait one barrier for waves 4-7's last
      // quadrant, and afterwards waves 4-7 drop one barrier behind again.  (Left staggered, each
      // group's epilogue ran while the other stood at a barrier: twice the epilogue per tile.)
      if (!behind && EXP != 7) __builtin_amdgcn_s_barrier();
      PG_LOOP_CLOCK(epi_t0);
      epilogue(cur, tp);
      PG_LOOP_CLOCK(epi_t1);
      epi_cycles += epi_t1 - epi_t0;
      if (behind) __builtin_amdgcn_s_barrier();
    }
    f_pp = tp;
    f_q0 = cur.q0;
    f_row0 = cur.row0;
    cur = nxt;
    nxt = tile_desc(T + 2);
  }
  // waves 0-3 owe the barrier waves 4-7 took after the last epilogue (EXP 14: at the start);
  // nothing may still be landing in LDS
  if (!behind && EXP != 7) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (FUSE && EXP != 14) {   // the last tile's candidates: flushed in place
    __builtin_amdgcn_s_barrier();      // every wave has parked
    flush_a();
    flush_b(true);
  }
  PG_CLOCK_STAMP(1);
#ifdef PG_CLOCKS
  if (threadIdx.x == 0) pg_clock_buf[blockIdx.x * 8 + 4] = loop_cycles;
  if (threadIdx.x == 0) pg_clock_buf[blockIdx.x * 8 + 5] = epi_cycles;
#endif
#ifdef PG_TRACE
  if (blockIdx.x == PG_TRACE_BLOCK) {
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * PG_TRACE_KTS * 4 * 4; i += PG_THREADS)
      pg_trace_buf[i] = *(volatile unsigned long long*)((char*)pg_lds + PG_LDS_TOTAL + i * 8);
  }
#endif
#undef PG_SYNC_AND_MMA
#undef PG_STAMP
#undef PG_C
}

}  // namespace svs
