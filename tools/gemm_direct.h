// EXPERIMENT (round 4; tools only, not part of libsvs_amd): a panel kernel for up to 128 queries over an f16 (EB = 2) or
// e4m3 (EB = 1) corpus whose CORPUS operand never touches the LDS -- VERDICT r3 item 2: "take the corpus operand off the
// LDS-DMA path where one query tile owns every corpus byte".  It is correct (tools/gemm_direct_bench checks scores and fused
// candidates against gemm_tiled_kernel) and it is NOT faster than the shipped gemm_phased_kernel<., ., 20, 128>:
// 0.59-0.65 ms against 0.58-0.64 at 1M x 1536 f16 x 128 queries, 0.34-0.36 against 0.34-0.36 at fp8
// (profiles/r4_gemm_direct_bench_*.txt).  What the harness rows say:
//   * its memory side alone (loads, counted waits, barriers, query pieces; nothing multiplied) streams the corpus at
//     6.3-6.5 TB/s, 0.47-0.49 ms -- so the 5.2 TB/s of LDS-DMA row pieces is indeed not a property of the DRAM;
//   * its on-chip side alone (no corpus loads) takes 0.33-0.35 ms: less than the memory side;
//   * together they take 0.59-0.65 ms: ~0.75 of the SUM, not the maximum -- and so did every form tried: corpus blocks
//     transposed through a wave-private LDS buffer (384 KiB of LDS traffic per slab and CU: 0.61), in registers at the
//     slab's start (0.65), in registers in the shadow of the previous slab's MFMAs (0.59), two or three slabs in flight,
//     loads issued at the slab's end or at its top (EXP 8), four waves of 64 rows (one per SIMD: 0.85-0.9, nothing to run
//     while a wave waits) or eight of 32, nontemporal or plain loads.  Delaying the loads of the bare stream by the length
//     of a slab's MFMA work costs it the same third (EXP 7: 0.49 -> 0.71 ms).  With the matrix pipe busy the chip runs at
//     1.7-2.05 GHz instead of the stream's 2.1-2.4, and the same 4.7-5.2 TB/s come out of two unrelated loop structures:
//     what bounds the 65-128-query panels is the chip under MFMA load, not the staging path.  Hence no third GEMM kernel in
//     the library.
//
// The design, for the record:
//   * a wave owns 32 private rows of the workgroup's 256 (eight waves, two per SIMD: one multiplies while the other
//     waits for its loads); per 256-byte k-slab it issues 8 nontemporal `buffer_load_dwordx4`, each covering 4 rows x
//     256 contiguous bytes, PF slabs ahead of their use (16-24 KiB in flight per wave);
//   * the MFMA wants lane = row l & 15, k-group l >> 4: 16 rows x 64 B per operand register, the load shape that
//     streams at 5.4 TB/s here.  So piece m of a 16-row block is loaded with lane (x = l >> 4, y = (l >> 2) & 3,
//     z = l & 3) -> row 4 y + m, 16-byte chunk 4 x + z -- every QUAD of lanes 64 contiguous bytes, every row 256: 6.65 TB/s
//     for the bare stream -- and the four pieces are transposed IN REGISTERS between the piece index m and the lane's
//     position z in its quad (two rounds of quad-permute DPP moves + selects): register t of lane (x, y, z') ends up with
//     row 4 y + z' = l & 15, chunk 4 x + t -- an MFMA operand whose k-group l >> 4 holds chunk 4 (l >> 4) + t in step t;
//   * only the QUERIES are shared: their k-slabs (128 queries x 256 B = 32 KiB) stream L2 -> LDS by LDS-DMA through
//     a three-slot ring, two slabs ahead, ONE workgroup barrier per slab; the ring runs on across row tiles (the
//     kernel is persistent).  Query fragments are read in the same k map (chunk 4 g + t for k-group g in step t; fp8:
//     chunks 4 g + 2 u, + 1 feed one 16x16x128 step) from an image swizzled for that map (see rd0).
// The dot products are those of gemm_tiled_kernel summed in another k order (a slab's sixteen chunks are dealt to the
// four MFMA steps round-robin instead of in runs of four): scores agree to rounding (max |diff| 7e-8 on unit vectors),
// not bit for bit.
//
// Fused top-k epilogue (FUSE): per tile a wave holds 32 rows x 128 queries = 4096 scores in 64 accumulator
// registers.  Survivors of the thresholds are PARKED in a wave-private LDS list (score bits, row << 7 | query) by
// ballot + mbcnt -- no atomics -- and flushed to the per-query global lists 64 at a time during the NEXT tile's first
// slabs: the returning atomic is issued in one slab iteration and its slot consumed in the following one.
//
// What hipcc had to be kept from doing (each cost wrong results or hundreds of spilled registers on the way):
//   - inline-asm MFMAs with "+a" accumulators: it moves accumulators between the register files at the loop header
//     and, not knowing the asm is an MFMA, reads them before the last one has landed (wrong scores in the last
//     accumulator only); "a"-class operands also make it split a wave's 256 registers 128 / 128 between the files;
//   - computing the 64 per-lane candidate codes / list addresses of the epilogue once per kernel and keeping them live
//     across the slab loop (~100 registers): they hang on a per-tile opaque copy of the lane id;
//   - scheduling an MFMA above the hand-counted `s_waitcnt lgkmcnt` of its query fragment: the fragment is an in/out
//     operand of the wait.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../svs_amd/csrc/gemm_tiled.h"

namespace svs {

constexpr int GD_WAVES = 8;                     // two per SIMD: one multiplies while the other waits for its loads or its LDS round trip
constexpr int GD_RB = 2;                        // 16-row MFMA blocks per wave
constexpr int GD_WROWS = 16 * GD_RB;            // rows per wave
constexpr int GD_ROWS = GD_WAVES * GD_WROWS;    // rows per workgroup tile (256)
constexpr int GD_QT = 128;                      // queries per tile = the whole panel
constexpr int GD_NJ = GD_QT / 16;               // 16-query MFMA blocks
constexpr int GD_SLAB = 256;                    // bytes of k per row per slab
constexpr int GD_BSLOT = GD_QT * GD_SLAB;       // one query slab in the LDS (32 KiB)
constexpr int GD_NSLOT = 3;
constexpr int GD_LDS_B = GD_NSLOT * GD_BSLOT;   // 96 KiB
constexpr int GD_PARK = 256;                    // parked candidates per wave and tile
constexpr int GD_LDS_TOTAL = GD_LDS_B + GD_WAVES * GD_PARK * 8;
constexpr int GD_NA = 4 * GD_RB;                // corpus loads per wave and slab (4 rows x 256 B each)
constexpr int GD_NB = GD_QT / 4 / GD_WAVES;     // query pieces per wave and slab (4 queries x 256 B each)

// preconditions of the kernel (host side): whole pairs of slabs per row, descriptors inside 32 bits
__host__ __device__ inline bool gd_shape_ok(int64_t ldb, int nq) { return nq >= 1 && nq <= GD_QT && ldb % (2 * GD_SLAB) == 0 && ldb >= 4 * GD_SLAB && ldb <= (1 << 20); }
// corpus slabs in flight per wave: 3 where the slabs of a row divide by it (d = 1536, 3072, ... in f16 / fp8), else 2
__host__ __device__ inline int gd_prefetch(int64_t ldb) { return (ldb / GD_SLAB) % 3 == 0 ? 3 : 2; }

#ifdef GD_CLOCKS   // tools/gemm_direct_bench only: shader cycles and 100 MHz ticks a workgroup spent in the kernel
__device__ unsigned long long* gd_clock_buf;
#define GD_CLOCK_STAMP(slot)                                                                   \
  do {                                                                                         \
    if (threadIdx.x == 0) {                                                                    \
      gd_clock_buf[blockIdx.x * 4 + (slot)] = __builtin_amdgcn_s_memtime();                    \
      gd_clock_buf[blockIdx.x * 4 + 2 + (slot)] = __builtin_amdgcn_s_memrealtime();            \
    }                                                                                          \
  } while (0)
#else
#define GD_CLOCK_STAMP(slot) do { } while (0)
#endif

// a buffer descriptor whose words are explicitly wave-uniform (scalar registers)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gd_rsrc(const void* p, int bytes) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int N, class F>
__device__ __forceinline__ void pg_static_for_gd(F&& f) {   // f(integral_constant<0>) ... f(integral_constant<N - 1>)
  if constexpr (N > 0) {
    pg_static_for_gd<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// Q: the staged queries, [q_rows][ldb] bytes (q_rows >= nq rows exist; rows nq .. are zero; rows past q_rows are never read:
// the descriptor's range check answers 0 for them).
// EXP (tools/gemm_direct_bench only; results wrong): 1 = loads, waits and barriers only (the memory side alone: each slab's
// registers are written to the LDS once, nothing is transposed or multiplied); 2 = no corpus loads (the on-chip side alone);
// 3 = corpus loads without the nontemporal hint.
// PF: corpus slabs a wave keeps in flight ahead of the one it multiplies (its prefetch register sets).  At 6.5 TB/s a load
// takes ~5 us to come back (128 KiB in flight per CU / 25 GB/s per CU, Little's law) and a slab ~1.9 us to multiply: two
// slabs ahead leave every slab waiting ~1.5 us for its registers (measured: 3.4 us per slab); three cover it.
template <bool FUSE, int EB, int EXP = 0, int PF = 2>
__global__ __launch_bounds__(GD_WAVES * 64) void gemm_direct_kernel(
    const uint8_t* __restrict__ M, const uint8_t* __restrict__ Q, float* __restrict__ scores,
    int64_t n, int ldb, int64_t sstride, int nq, int q_rows, int n_tiles,
    uint32_t* __restrict__ fstate_words, int fstate_stride, uint64_t* __restrict__ fcand, uint32_t fcap,
    const float* __restrict__ fthr, int fthr_stride, const float* __restrict__ rscale, const float* __restrict__ qscale) {
  extern __shared__ u32x4 gd_lds[];
  GD_CLOCK_STAMP(0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int KS = __builtin_amdgcn_readfirstlane(ldb / GD_SLAB);          // slabs per row (even)
  const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * KS;                                        // slab iterations of this workgroup

  // ---- LDS addresses (bytes) ------------------------------------------------------------------------------
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)gd_lds;
  uint32_t* park = (uint32_t*)((uint8_t*)gd_lds + GD_LDS_B) + wave * (GD_PARK * 2);
  // Query image: row = query, slot c' = chunk ^ swz(query & 15), swz(r) = (r & 3) << 2 | r >> 2.  A fragment read takes chunk
  // 4 g + t of query r16: slot ((4 g) ^ swz(r16)) ^ t -- for every 16-lane group of ds_read_b128 ({0-3, 12-15, 20-27}, ...)
  // sixteen different slots of the 256-byte bank row: conflict-free (the plain (row & 15) XOR is not, for this k map).
  const unsigned swz16 = (unsigned)(((r16 & 3) << 2) | (r16 >> 2));
  const unsigned rd0 = (unsigned)(r16 * GD_SLAB) + ((((unsigned)(4 * g)) ^ swz16) << 4);

  // ---- descriptors -----------------------------------------------------------------------------------------
  // corpus: one per tile, based at the WAVE's first row (rows past n are dropped by the range check: they read 0)
  auto rows_rsrc = [&](int tile_i) __attribute__((always_inline)) {
    const int64_t t = (int64_t)blockIdx.x + (int64_t)tile_i * gridDim.x;
    const int64_t row0 = t * GD_ROWS + wave * GD_WROWS;
    int64_t live = tile_i < my_tiles ? n - row0 : 0;
    live = live < 0 ? 0 : (live > GD_WROWS ? GD_WROWS : live);
    return gd_rsrc(M + (row0 < n ? row0 : 0) * (int64_t)ldb, (int)(live * ldb));
  };
  // lane (x = l >> 4, y = (l >> 2) & 3, z = l & 3) of piece m reads row 4 y + m (m in the scalar offset), chunk 4 x + z
  const int a_voff = (4 * ((lane >> 2) & 3)) * ldb + ((4 * (lane >> 4) + (lane & 3)) << 4);
  __amdgpu_buffer_rsrc_t rs_cur = rows_rsrc(0), rs_nxt = rows_rsrc(1);
  const __amdgpu_buffer_rsrc_t rs_q = gd_rsrc(Q, (q_rows < GD_QT ? q_rows : GD_QT) * ldb);
  // queries: wave w stages queries 16 w .. 16 w + 15 of every slab, 4 pieces of 4 queries x 256 B; the LDS image is
  // lane-linear (row l >> 4, slot l & 15), so the swizzle sits on the SOURCE chunk: slot c holds chunk c ^ swz(query & 15)
  int b_voff[GD_NB];
#pragma unroll
  for (int i = 0; i < GD_NB; ++i) {
    const int ql = wave * (4 * GD_NB) + 4 * i + g, qr = ql & 15;
    b_voff[i] = ql * ldb + ((r16 ^ (((qr & 3) << 2) | (qr >> 2))) << 4);
  }

  // ---- per-lane query constants: thresholds and (fp8) query scales of queries 16 j + r16 --------------------
  float thr[GD_NJ], qs[GD_NJ];
#pragma unroll
  for (int j = 0; j < GD_NJ; ++j) {
    const int q = 16 * j + r16;
    thr[j] = (FUSE && q < nq) ? fthr[(int64_t)q * fthr_stride] : __builtin_inff();   // padded queries: nothing survives
    qs[j] = (EB == 1 && q < nq) ? qscale[q] : 0.f;
  }

  u32x4 P[PF][GD_NA];                      // corpus prefetch: PF slabs x (GD_NA pieces of 4 rows x 256 B)
  if constexpr (EXP == 2) {
#pragma unroll
    for (int f = 0; f < PF; ++f)
#pragma unroll
      for (int m = 0; m < GD_NA; ++m) P[f][m] = (u32x4){(unsigned)lane, 2u, 3u, 4u};
  }
  f32x4_t acc[GD_RB][GD_NJ];
#pragma unroll
  for (int i = 0; i < GD_RB; ++i)
#pragma unroll
    for (int j = 0; j < GD_NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  auto issue_b = [&](int s, bool live, int slot) __attribute__((always_inline)) {   // query slab s (of any tile: they repeat) into ring slot `slot`
#pragma unroll
    for (int i = 0; i < GD_NB; ++i) {
      const unsigned dst = lds0 + (unsigned)slot * GD_BSLOT + (unsigned)(wave * (4 * GD_NB) + 4 * i) * GD_SLAB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, b_voff[i],
                                               live ? s * GD_SLAB : 0x7ffff000, 0, 0);   // (past the end: out of range, nothing moves)
    }
  };
  auto issue_a = [&](auto PAR, int s, bool nxt) __attribute__((always_inline)) {   // corpus slab s of the current tile (nxt: of the tile after it) into prefetch set PAR
    constexpr int par = decltype(PAR)::value;
    const __amdgpu_buffer_rsrc_t rs = nxt ? rs_nxt : rs_cur;
#pragma unroll
    for (int m = 0; m < GD_NA; ++m)       // piece m: block m / 4, rows (m & 3), 4 + (m & 3), ... of it
      if constexpr (EXP != 2)
        P[par][m] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, a_voff, s * GD_SLAB + ((m >> 2) * 16 + (m & 3)) * ldb, EXP == 3 ? 0 : 2));   // aux 2 = nt
  };

  // ---- fused epilogue state -----------------------------------------------------------------------------------
  int parked = 0;            // entries in the wave's parking list (wave-uniform)
  int flushed = 0;           // ... of which already handed to flush_a
  uint32_t fl_bits = 0, fl_code = 0, fl_slot = 0;   // one entry per lane between flush_a and flush_b
  int fl_n = 0;              // lanes holding one
  int64_t park_row0 = 0;     // global row of the parked tile's first row (this wave's)
  auto flush_a = [&]() __attribute__((always_inline)) {     // next 64 parked entries: one returning atomic each on the query's list length
    fl_n = parked - flushed;
    fl_n = fl_n > 64 ? 64 : fl_n;
    if (lane < fl_n) {
      fl_bits = park[2 * (flushed + lane)];
      fl_code = park[2 * (flushed + lane) + 1];
      fl_slot = atomicAdd(fstate_words + (int64_t)(fl_code & 127u) * fstate_stride, 1u);
    }
    flushed += fl_n;
  };
  auto flush_b = [&]() __attribute__((always_inline)) {
    if (lane < fl_n && fl_slot < fcap)
      fcand[(int64_t)(fl_code & 127u) * fcap + fl_slot] = ((uint64_t)fl_bits << 32) | (uint32_t)(park_row0 + (fl_code >> 7));
    fl_n = 0;
  };
  auto flush_all = [&]() __attribute__((always_inline)) {   // (end of the kernel, or a tile whose survivors do not fit the list)
    flush_b();
    while (flushed < parked) { flush_a(); flush_b(); }
    parked = flushed = 0;
  };

  auto epilogue = [&](int tile_i) __attribute__((always_inline)) {
    // (an opaque copy of the lane id, made per tile: formed from `lane` itself, the 64 per-lane codes row << 7 | query, the
    //  list addresses and the row offsets of the sweep are loop invariants that hipcc computes once per KERNEL and keeps in
    //  registers across the slab loop -- ~100 registers, i.e. the spills of the first build)
    int le = lane;
    asm volatile("" : "+v"(le));
    const int g = le >> 4, r16 = le & 15;
    const int64_t t = (int64_t)blockIdx.x + (int64_t)tile_i * gridDim.x;
    const int64_t row0 = t * GD_ROWS + wave * GD_WROWS;                  // this wave's first row
    const int live = (int)(n - row0 < GD_WROWS ? (n - row0 < 0 ? 0 : n - row0) : GD_WROWS);
    if constexpr (EB == 1) {   // dequantisation: v * (row scale * query scale), the tiled kernel's order
#pragma unroll
      for (int i = 0; i < GD_RB; ++i) {
        f32x4_t rs;
        const int64_t ob = row0 + 16 * i + 4 * g;
        if (ob + 3 < n) rs = *(const f32x4_t*)(rscale + ob);
        else
          for (int r = 0; r < 4; ++r) rs[r] = ob + r < n ? rscale[ob + r] : 0.f;
#pragma unroll
        for (int j = 0; j < GD_NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] *= rs[r] * qs[j];
      }
    }
    if constexpr (!FUSE) {
#pragma unroll
      for (int j = 0; j < GD_NJ; ++j) {
        const int q = 16 * j + r16;
        if (q < nq) {
          float* o = scores + (int64_t)q * sstride;
#pragma unroll
          for (int i = 0; i < GD_RB; ++i) {
            const int64_t ob = row0 + 16 * i + 4 * g;
            if (ob + 3 < n) *(f32x4_t*)(o + ob) = acc[i][j];
            else
              for (int r = 0; r < 4; ++r)
                if (ob + r < n) o[ob + r] = acc[i][j][r];
          }
        }
      }
    } else {
      // the previous tile's list must be gone before this one parks (it normally is: flushed during this tile's slabs)
      if (flushed < parked || fl_n) flush_all();
      parked = flushed = 0;
      park_row0 = row0;
#pragma unroll
      for (int i = 0; i < GD_RB; ++i)
#pragma unroll
        for (int j = 0; j < GD_NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[i][j][r];
            const int lr = 16 * i + 4 * g + r;
            const bool hit = !(v < thr[j]) && lr < live;                 // (a NaN score passes: it ranks largest, keys.h)
            const unsigned long long m = __ballot(hit);
            if (m) {
              const int at = parked + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
              if (hit) {
                if (at < GD_PARK) {
                  park[2 * at] = score_key(v);
                  park[2 * at + 1] = ((uint32_t)lr << 7) | (uint32_t)(16 * j + r16);
                } else {
                  // more survivors than the list holds (rows ordered by topic: a tile full of one query's neighbours):
                  // the rest goes straight to the global lists, one returning atomic per survivor (slow, rare, exact)
                  const int q = 16 * j + r16;
                  fuse_offer(fstate_words + (int64_t)q * fstate_stride, fcand + (int64_t)q * fcap, fcap, thr[j], v, (uint32_t)(row0 + lr));
                }
              }
              parked += __builtin_popcountll(m);
              parked = parked > GD_PARK ? GD_PARK : parked;
            }
          }
    }
#pragma unroll
    for (int i = 0; i < GD_RB; ++i)
#pragma unroll
      for (int j = 0; j < GD_NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };

  // ---- one slab: 32 rows x 128 queries x 256 B of k, per wave ------------------------------------------------------
  // The query-fragment reads are inline asm with hand-counted waits: left to hipcc every fragment read is followed by
  // `s_waitcnt lgkmcnt(0)` and two MFMAs (its waitcnt pass cannot count reads behind LDS-DMA), i.e. the LDS latency is
  // paid 32 times per slab.  Here a ring of GD_BRING fragment steps runs ahead of the MFMAs that use them; the fragment
  // is an in/out operand of its wait, so the MFMA that uses it cannot be scheduled above the wait.
  const unsigned rdt[4] = {rd0, rd0 ^ 16u, rd0 ^ 32u, rd0 ^ 48u};                         // chunk 4 g + t of the lane's query
#define GD_DS_READ(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
#define GD_DS_WRITE(addr, val) asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(val) : "memory")
#define GD_LGKM1(N, v) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N))
#define GD_LGKM2(N, v, w) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(v), "+v"(w) : "n"(N))
  constexpr int GD_BRING = EB == 2 ? 4 : 2;   // query-fragment steps in flight ahead of the MFMAs (a step = 2 MFMAs: 32 cycles f16, 64 fp8)
  // 4 x 4 transposition of a 16-row block between the piece index and the lane's place in its quad, in registers, dword by
  // dword: round 1 swaps (piece bit 0, lane bit 0), round 2 (piece bit 1, lane bit 1); a swap is one quad-permute DPP move of
  // each partner and one select each.  After it register t of lane l holds row l & 15, chunk 4 (l >> 4) + t of the slab.
  const bool odd0 = (lane & 1) != 0, odd1 = (lane & 2) != 0;
  // unit U of the 32 that transpose a slab's two blocks: block U / 16, dword (U / 4) % 4, then round 1 pair 0, round 1 pair 1,
  // round 2 pair 0, round 2 pair 1 (the rounds of one dword in this order: round 2 reads what round 1 wrote)
  auto xpose_unit = [&](u32x4 (&R)[GD_NA], auto UU) __attribute__((always_inline)) {
    constexpr int U = decltype(UU)::value;
    constexpr int o = 4 * (U / 16), w = (U / 4) % 4, k = U % 4;
    if constexpr (k < 2) {          // pairs (0, 1), (2, 3): lanes z <-> z ^ 1
      const unsigned a = R[o + 2 * k][w], b = R[o + 2 * k + 1][w];
      const unsigned as = (unsigned)__builtin_amdgcn_mov_dpp((int)a, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
      const unsigned bs = (unsigned)__builtin_amdgcn_mov_dpp((int)b, 0xB1, 0xF, 0xF, true);
      R[o + 2 * k][w] = odd0 ? bs : a;
      R[o + 2 * k + 1][w] = odd0 ? b : as;
    } else {                        // pairs (0, 2), (1, 3): lanes z <-> z ^ 2
      const unsigned a = R[o + k - 2][w], b = R[o + k][w];
      const unsigned as = (unsigned)__builtin_amdgcn_mov_dpp((int)a, 0x4E, 0xF, 0xF, true);   // quad_perm [2, 3, 0, 1]
      const unsigned bs = (unsigned)__builtin_amdgcn_mov_dpp((int)b, 0x4E, 0xF, 0xF, true);
      R[o + k - 2][w] = odd1 ? bs : a;
      R[o + k][w] = odd1 ? b : as;
    }
  };
  auto slab = [&](auto PAR, int s, int gs, int slot) __attribute__((always_inline)) {   // s: slab inside the tile, gs: slab iteration of this workgroup
    constexpr int par = decltype(PAR)::value;
    const int s2 = s + 2 >= KS ? s + 2 - KS : s + 2;        // the query slab fetched now, two ahead
    const int sa = s + PF >= KS ? s + PF - KS : s + PF;     // the corpus slab fetched at the end, PF ahead
    // In the wave's in-order queue, behind this slab's corpus registers (fetched at the end of iteration gs - PF) and its query
    // pieces (start of gs - 2) sit the corpus loads of PF - 1 later slabs and the query pieces of one: they may stay in flight.
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EXP == 8 ? GD_NA + GD_NB : (PF - 1) * GD_NA + GD_NB) : "memory");
    __builtin_amdgcn_s_barrier();           // every wave's pieces of this slab are in; everyone is done with the slot refilled next
    issue_b(s2, gs + 2 < total, slot == 0 ? 2 : slot - 1);   // ring slot (gs + 2) % 3
    const unsigned bslot = lds0 + (unsigned)slot * GD_BSLOT;
    unsigned vb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) vb[t] = bslot + rdt[t];
    constexpr int NSTEP = (EB == 2 ? 4 : 2) * GD_NJ;        // (k-step, query block) pairs: 32 (f16) / 16 (fp8)
    constexpr int RPS = EB == 2 ? 1 : 2;                    // fragment reads per step (fp8: both chunks of a 128-byte k-step)
    if constexpr (EXP == 1 || EXP == 7) {   // (the memory side alone; EXP 7: with the next loads issued ~4,000 cycles into the slab, as after a slab's MFMAs; keep the loads alive: one LDS store per piece into the wave's own list)
#pragma unroll
      for (int m = 0; m < GD_NA; ++m) GD_DS_WRITE((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)park + (unsigned)(lane << 4), P[par][m]);
      if constexpr (EXP == 7) asm volatile("s_sleep 63" ::: "memory");
      issue_a(PAR, sa, s + PF >= KS);
      return;
    }
    u32x4 bfr[GD_BRING][RPS];
    auto read_b = [&](auto I) __attribute__((always_inline)) {
      constexpr int i = decltype(I)::value;
      constexpr int t = i / GD_NJ, j = i % GD_NJ;
      auto& bf = bfr;           // (named outside the `if constexpr`: clang does not capture a variable that a generic
      const auto& va = vb;      //  lambda uses only inside a discarded statement)
      if constexpr (EB == 2) {
        GD_DS_READ(bf[i % GD_BRING][0], va[t], j * 16 * GD_SLAB);
      } else {
        GD_DS_READ(bf[i % GD_BRING][0], va[2 * t], j * 16 * GD_SLAB);
        GD_DS_READ(bf[i % GD_BRING][RPS - 1], va[2 * t + 1], j * 16 * GD_SLAB);
      }
    };
    constexpr bool early = EXP == 8;        // (EXP 8, PF = 3: this slab's loads go out at its TOP, into the set the last slab freed; own transposition up front)
    if constexpr (early) {
      const int se = s + PF - 1 >= KS ? s + PF - 1 - KS : s + PF - 1;
      issue_a(std::integral_constant<int, (par + PF - 1) % PF>{}, se, s + PF - 1 >= KS);
    }
    pg_static_for_gd<GD_BRING>([&](auto I) __attribute__((always_inline)) { read_b(I); });
    if constexpr (early) pg_static_for_gd<32>([&](auto U) __attribute__((always_inline)) { xpose_unit(P[par], U); });
    __builtin_amdgcn_sched_barrier(0);
    constexpr int nxt = (par + 1) % PF;     // the set the NEXT slab multiplies: transposed here, in the shadow of this slab's MFMAs
    pg_static_for_gd<NSTEP>([&](auto I) __attribute__((always_inline)) {
      constexpr int i = decltype(I)::value;
      constexpr int t = i / GD_NJ, j = i % GD_NJ;
      constexpr int ahead = NSTEP - 1 - i < GD_BRING - 1 ? NSTEP - 1 - i : GD_BRING - 1;   // fragment steps still in flight behind this one
      if constexpr (EB == 2) GD_LGKM1(ahead * RPS, bfr[i % GD_BRING][0]);
      else GD_LGKM2(ahead * RPS, bfr[i % GD_BRING][0], bfr[i % GD_BRING][1]);
      // (the MFMAs are builtins: hipcc pads their hazards itself -- an inline-asm MFMA is invisible to its hazard recogniser, and it
      //  does move accumulators between the register files around the loop, reading them too early: measured, wrong scores)
#pragma unroll
      for (int rb = 0; rb < GD_RB; ++rb) {
        if constexpr (EB == 2) {
          acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, P[par][4 * rb + t]), __builtin_bit_cast(h8, bfr[i % GD_BRING][0]),
                                                              acc[rb][j], 0, 0, 0);
        } else {
          const u32x4 al = P[par][4 * rb + 2 * t], ah = P[par][4 * rb + 2 * t + 1], bl = bfr[i % GD_BRING][0], bh = bfr[i % GD_BRING][1];
          const i32x8 x = {(int)al.x, (int)al.y, (int)al.z, (int)al.w, (int)ah.x, (int)ah.y, (int)ah.z, (int)ah.w};
          const i32x8 y = {(int)bl.x, (int)bl.y, (int)bl.z, (int)bl.w, (int)bh.x, (int)bh.y, (int)bh.z, (int)bh.w};
          acc[rb][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, y, acc[rb][j], 0, 0, 0, 0, 0, 0);
        }
      }
      if constexpr (i + GD_BRING < NSTEP) read_b(std::integral_constant<int, i + GD_BRING>{});
      // a slice of the next slab's transposition: VALU work that issues while this step's MFMAs run (4 instructions per
      // 16-cycle MFMA pair at f16).  Its first use waits for that slab's loads -- fetched two iterations ago.
      if constexpr (!early)
        pg_static_for_gd<32 / NSTEP>([&](auto V) __attribute__((always_inline)) {
          xpose_unit(P[nxt], std::integral_constant<int, i * (32 / NSTEP) + decltype(V)::value>{});
        });
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (!early) issue_a(PAR, sa, s + PF >= KS);   // the set's registers are free again: refill it PF slabs ahead
  };
#undef GD_DS_READ
#undef GD_DS_WRITE
#undef GD_LGKM1
#undef GD_LGKM2

  // ---- prologue: PF corpus slabs and two query slabs in flight (in the queue order the loop keeps) --------------------
  if constexpr (EXP == 8) {
    issue_b(0, true, 0);
    issue_a(std::integral_constant<int, 0>{}, 0, false);
    issue_b(1, true, 1);
    issue_a(std::integral_constant<int, 1>{}, 1, false);
  } else
  pg_static_for_gd<PF>([&](auto F) __attribute__((always_inline)) {
    constexpr int f = decltype(F)::value;
    if (f == PF - 2) issue_b(0, true, 0);
    if (f == PF - 1) issue_b(1, true, 1);
    issue_a(F, f, false);
  });
  if constexpr (EXP != 1 && EXP != 7 && EXP != 8)   // the first slab's registers are transposed here; every later slab's during the slab before it
    pg_static_for_gd<32>([&](auto U) __attribute__((always_inline)) { xpose_unit(P[0], U); });

  int gs = 0, slot = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    for (int s = 0; s < KS; s += PF) {
      pg_static_for_gd<PF>([&](auto F) __attribute__((always_inline)) {
        constexpr int f = decltype(F)::value;
        if constexpr (FUSE) {   // the previous tile's survivors leave during this tile's first slabs, 64 per slab
          flush_b();
          if (flushed < parked) flush_a();
        }
        slab(F, s + f, gs + f, slot);
        slot = slot == 2 ? 0 : slot + 1;
      });
      gs += PF;
    }
    epilogue(ti);
    rs_cur = rs_nxt;
    rs_nxt = rows_rsrc(ti + 2);
  }
  if constexpr (FUSE) flush_all();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the ring's last pieces: nothing may be in flight to the LDS at exit)
  GD_CLOCK_STAMP(1);
}

}  // namespace svs
