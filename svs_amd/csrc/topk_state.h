// Shared state of the streaming top-k pre-filter (written by the score kernels,
// consumed and re-zeroed by select_final_kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keys.h"

namespace svs {

// ---- streaming top-k pre-filter fused into the score kernels -----------------
// The 12 top bits of the orderable score key split ALL floats into 4096 ordered
// "levels" (1/8 octave each).  state->level_cnt[L] counts the scores seen at
// level L; once a level holds k scores, nothing below that level can be among
// the k best, so state->cut (monotone, atomicMax) rises to it.  A wave appends
// (key,row) to the candidate list iff level(score) >= cut.  Exact by
// construction: a stale (lower) cut only admits extra candidates, and every row
// at or above the FINAL cut was admitted.  After ~N/3 rows of a 1M corpus the
// pass rate is ~4e-4, so the atomics are noise next to the HBM stream; the
// final kernel (select.h) picks the exact k out of the few thousand survivors.
constexpr int SEL_LEVELS = 4096;
struct SelState {
  uint32_t cut;     // lowest level still admitted
  uint32_t n_cand;  // candidates appended (may exceed the capacity: then invalid)
  uint32_t pad0, pad1;
  uint32_t level_cnt[SEL_LEVELS];
};
constexpr int SEL_STATE_WORDS = sizeof(SelState) / 4;

struct FuseArgs {
  SelState* state;   // nullptr: plain score kernel
  uint64_t* cand;
  uint32_t k;
  uint32_t cap;
};

// v is wave-uniform.  cut_hint was read when the wave started.
__device__ __forceinline__ void fuse_offer(const FuseArgs& fa, float v, int64_t row, int lane,
                                           uint32_t cut_hint) {
  const uint32_t key = score_key(v);
  const uint32_t level = key >> 20;
  if (level < cut_hint) return;
  if (lane == 0) {
    const uint32_t cut = __hip_atomic_load(&fa.state->cut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (level >= cut) {
      const uint32_t seen = atomicAdd(&fa.state->level_cnt[level], 1u) + 1u;
      if (seen >= fa.k && level > cut) atomicMax(&fa.state->cut, level);
      const uint32_t slot = atomicAdd(&fa.state->n_cand, 1u);
      if (slot < fa.cap) fa.cand[slot] = ((uint64_t)key << 32) | (uint32_t)row;
    }
  }
}


}  // namespace svs
