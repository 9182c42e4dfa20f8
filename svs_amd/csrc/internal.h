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
 *   1  host batches: 0 = f16 batches pulled from pinned memory by the staging kernel, chunk by chunk (default);
 *      1 = staged DMA for every dtype (round 3)
 *   2  fused path's threshold rows: 1 = a sample spread over the whole corpus (default); 0 = the first rows (rounds 1-3) */
int32_t svs_internal_tune(int32_t what, int64_t value);
/* Seconds since the start of the calling thread's last svs_index_search(host batch) at which: [0] scratch was planned,
 * [1] the queries were in pinned memory (and their DMA enqueued), [2] every kernel was enqueued, [3] the stream had
 * drained, [4] the results were in the caller's buffers; [5] = the number of the call's queries whose fused candidate list
 * overflowed and that were re-run through the materialised path (a count, not a time). */
int32_t svs_internal_host_phases(double* out, int32_t n);
/* multi.hip -> svs_amd.hip: carries a worker thread's error message over to the caller's thread */
int32_t svs_internal_set_error(int32_t code, const char* msg);
#ifdef __cplusplus
}
#endif
#endif
