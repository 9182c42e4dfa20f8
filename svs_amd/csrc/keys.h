// Order-preserving keys for the top-k stage.
//
// The reference orders results with Python's sorted(..., reverse=True) over
// (float(score), int(index)) tuples (src/svs/util.py:203): score descending,
// ties by index descending.  We pack both into ONE unsigned 64-bit key
//     key = orderable(score) << 32 | row
// so that "larger key" == "earlier in the reference's output".  Rows are unique,
// hence keys are unique and every select/sort below is deterministic no matter
// how workgroups are scheduled or how the corpus is sharded.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SVS_HD __host__ __device__ __forceinline__
#else
#define SVS_HD inline
#endif

namespace svs {

SVS_HD uint32_t f32_bits(float f) {
  union { float f; uint32_t u; } c;
  c.f = f;
  return c.u;
}
SVS_HD float bits_f32(uint32_t u) {
  union { float f; uint32_t u; } c;
  c.u = u;
  return c.f;
}

// f32 -> u32 whose unsigned order equals the float order.  -0.0 is folded onto
// +0.0 (Python compares them equal, so the tie falls through to the index) and
// any NaN maps to the maximum (np.argpartition sorts NaN last == largest).
SVS_HD uint32_t score_key(float s) {
  uint32_t u = f32_bits(s);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN
  if (u == 0x80000000u) u = 0u;                              // -0.0 -> +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
SVS_HD float key_score(uint32_t k) {
  if (k == 0xffffffffu) return bits_f32(0x7fc00000u);  // NaN
  return bits_f32((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
SVS_HD uint64_t make_key(float s, uint32_t row) { return ((uint64_t)score_key(s) << 32) | row; }

}  // namespace svs
