/* star_amd_async.h -- optional companion of staramd_map_batch (include/star_amd.h): the upload of the NEXT batch while the current one is on the device.
 *
 * staramd_map_batch is a blocking call, as the loop it replaces (ReadAlignChunk_mapChunk.cpp:30-32: one chunk after the other): upload, kernels, download.
 * The chunk that follows is usually parsed already (ReadAlignChunk_processChunks.cpp fills chunk k+1 while chunk k is mapped); handing it to the context
 * BEFORE the call for chunk k lets its upload (81 MB for 400 k pairs of 2x101) run beside the kernels of chunk k, on a second stream into a second set
 * of input buffers.  Nothing else changes: staramd_map_batch recognises the batch it was shown (same `bases` and `readOffset` pointers, same nReads) and skips
 * its own upload; a batch it was not shown is uploaded as always, and a prefetched batch that never arrives is simply overwritten by the next prefetch.
 *
 * The arrays of the prefetched batch must stay unchanged until the staramd_map_batch call for it returns; page-locked arrays (staramd_pinned_alloc) make the
 * copy asynchronous.  One prefetched batch per context; calls on one context are serialised by the caller, like all others.  Returns 0, or a negative
 * STARAMD_ERR_* (the batch is then uploaded by staramd_map_batch in the ordinary way: a failed prefetch is never fatal).
 */
#ifndef STAR_AMD_ASYNC_H
#define STAR_AMD_ASYNC_H
#include "star_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
int staramd_prefetch_batch(staramd_ctx *ctx, const staramd_batch *next);
/* A prefetched batch is recognised by its `bases` / `readOffset` pointers and nReads alone, so: between staramd_prefetch_batch(next) and the staramd_map_batch
 * call for `next` the caller must not refill those arrays with another batch.  A caller that gives a prefetched batch up (an error path, a phase that ends
 * early) calls staramd_prefetch_cancel before it reuses the arrays; staramd_map_batch does the same by itself whenever it returns an error, and
 * staramd_map_resident refuses to run when a prefetch has overwritten the resident batch. */
int staramd_prefetch_cancel(staramd_ctx *ctx);
/* staramd_map_batch calls of this context that found their upload done ahead (tests and the front end's report: a prefetch path that never runs is not a fast path) */
uint64_t staramd_prefetch_hits(staramd_ctx *ctx);

/* The two halves of staramd_map_batch, for a caller that keeps the device busy across the copy of the results (~100 MB per batch of 400 k pairs: 2 ms of a 47 ms call):
 *   staramd_map_begin(ctx, batch)        upload (unless staramd_prefetch_batch has done it) and every kernel of the batch enqueued; returns without waiting.
 *                                        One batch in flight per context; staramd_map_batch / staramd_map_resident refuse to run meanwhile.
 *   staramd_map_end(ctx, results, next)  waits for the kernels of the batch in flight (a work-space overflow is handled as in staramd_map_batch: grown, run again);
 *                                        starts the copy of its results on the copy stream; begins `next` (may be NULL) at once, so that its kernels run beside that
 *                                        copy; returns when `results` is complete.  STARAMD_ERR_RESULT_OVERFLOW leaves the batch in flight: call again with larger
 *                                        arrays (trCount / exCount say how large).  Any other error ends the batch in flight; an error of `next`'s begin is returned
 *                                        with `results` complete (staramd_last_error says which it was).
 * The arrays of `batch` / `next` must stay unchanged until the staramd_map_end call that finishes them returns; results as for staramd_map_batch.
 * Same kernels, same order, same buffers as staramd_map_batch: the results are identical. */
int staramd_map_begin(staramd_ctx *ctx, const staramd_batch *batch);
/* waits for the kernels of the batch in flight and nothing else (the first step of staramd_map_end): a caller that wants to hand staramd_map_end the batch that is next
 * in line looks for it AFTER this wait -- it may have arrived meanwhile */
int staramd_map_wait(staramd_ctx *ctx);
int staramd_map_end(staramd_ctx *ctx, staramd_results *results, const staramd_batch *next);
/* batches whose kernels were begun inside staramd_map_end, beside the copy of the results before them (tests and the front end's report) */
uint64_t staramd_overlapped_batches(staramd_ctx *ctx);
/* times the kernels of a batch were enqueued on this context -- a batch whose work space or result arrays were too small is counted as often as it ran (tests) */
uint64_t staramd_launch_count(staramd_ctx *ctx);
#ifdef __cplusplus
}
#endif
#endif
