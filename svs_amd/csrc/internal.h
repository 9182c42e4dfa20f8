/*
 * internal.h -- entry points libsvs_amd.so exports for its OWN tests, tools and bench; NOT part of the
 * drop-in boundary (include/svs_amd.h) and not a contract: they may change or go in any round.
 */
#ifndef SVS_AMD_INTERNAL_H
#define SVS_AMD_INTERNAL_H
#include "../../include/svs_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* The NEXT coalesced pass on this handle waits (at most 5 s) until n single-query calls are queued, so that a
 * pass of a chosen size can be formed on purpose (parity tests of the coalesced route); one shot, 0 cancels. */
int32_t svs_internal_coalesce_hold(svs_index* idx, int32_t n);
/* Process-wide knobs for A/B measurements:
 *   0  fused path: threshold prefix = n / value rows (default 64; at least 16,384 rows)
 *   1  host batches of more than 1 MiB: 0 = host copies through the helper-thread pool (default), 1 = serial staging (round 3)
 *   2  helper threads of the host copy pool (default 3; 0 = none) */
int32_t svs_internal_tune(int32_t what, int64_t value);
/* multi.hip -> svs_amd.hip: carries a worker thread's error message over to the caller's thread */
int32_t svs_internal_set_error(int32_t code, const char* msg);
#ifdef __cplusplus
}
#endif
#endif
