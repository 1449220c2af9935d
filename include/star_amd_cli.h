/* star_amd_cli.h -- the command-line front end as a function (libstaramd_cli.so; star_amd/bin/star_amd is a three-line main around it).
 *
 * What it is the analogue of: the main() of the reference (source/STAR.cpp:40-270) for --runMode alignReads / genomeGenerate: parameters,
 * genome load, one worker per mapping resource pulling chunks of reads from one input (ReadAlignChunk::processChunks,
 * source/ReadAlignChunk_processChunks.cpp), the merge of the workers' junction tables and counters at the end (source/outputSJ.cpp:39-83).
 * Here a mapping resource is one MI355X (one staramd_ctx, include/star_amd.h): `--gpuDevices 0,1,...` gives one mapper thread per GPU, all fed
 * from the one FASTQ reader and emitting through one post-map / writer stage in input order; junction table and Stats live in the one host
 * object, so inside a process there is nothing to exchange.  With one PROCESS per GPU (torchrun; bench.py, star_amd/multi_gpu.py) the
 * `exchange` hook is where the ranks all_gather their junction tables over RCCL before a phase ends.
 */
#ifndef STAR_AMD_CLI_H
#define STAR_AMD_CLI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct staramd_cli_hooks {
    void *user;
    /* called once, when the warm-up reads (--benchWarmupReads) are mapped AND written and the pipeline is empty, before the reader
     * goes on: a barrier across ranks goes here; the timed region of the report starts when it returns */
    void (*warmup_done)(void *user);
    /* called at the end of every mapping phase with the host handle (sah_*), before sah_next_phase / sah_finish: cross-rank junction
     * table / counter exchange; return non-zero to abort */
    int  (*exchange)(void *user, void *sah_handle, int lastPhase);
} staramd_cli_hooks;

#define STARAMD_CLI_MAX_DEV 16
typedef struct staramd_cli_report {
    uint64_t reads;                /* reads (pairs) mapped in all phases                                             */
    double   wallMapping;          /* first chunk submitted -> last output byte written (all phases, index load excluded) */
    uint64_t timedReads;           /* the same two for the region after the warm-up pause (= everything without --benchWarmupReads) */
    double   timedWall;
    double   genomeLoadSeconds, indexUploadSeconds;
    int      nDevices;
    double   deviceBusy[STARAMD_CLI_MAX_DEV];     /* seconds inside staramd_map_batch per engine context, timed region */
    double   deviceMs[STARAMD_CLI_MAX_DEV];       /* HIP-event device time per device, timed region                   */
    double   stageMs[8];           /* engine stages summed over the timed batches (staramd_get_timings order)         */
    uint64_t counters[64];         /* engine counters summed over the timed batches (staramd_get_counters order)      */
    double   parseBusy, emitBusy;  /* seconds the reader / the post-map+writer stage were busy, timed region          */
    uint64_t batches;              /* timed batches                                                                   */
    double   pass1Seconds;         /* --twopassMode Basic: 1st pass + junction insertion + index re-upload            */
    double   finishSeconds;        /* after the last batch is handed to the writer: last writes, SJ.out.tab, Log.final.out (inside timedWall) */
    int      nContexts;            /* engine contexts (= mapper threads): nDevices x STARAMD_CONTEXTS_PER_GPU; deviceBusy / deviceMs are per context */
    double   convertBusy;          /* seconds the second half of the reader (text -> numeric batch, its own thread) was busy, timed region; parseBusy is the first half (input + line table) */
    double   emitParts[4];         /* what emitBusy is made of, whole run (seconds): [0] waiting for a free text-buffer set (= for the writer), [1] formatting on the threads, [2] serial tail
                                      of a batch (junction merge, hand-over); [3] the writer thread's own busy time (write calls into the output file) */
    uint64_t fastPaths[4];         /* batches that took a path with a silent fallback behind it, whole run: [0] SAM / BAM text written through a mapping of the output file (fallback: positional
                                      writes), [1] read in place from mappings of the input files (fallback: copied out of the page cache), [2] uploaded ahead of their staramd_map_batch call
                                      (staramd_prefetch_batch; fallback: uploaded by the call), [3] begun inside staramd_map_end, beside the copy of the results of the batch before (fallback: one
                                      blocking staramd_map_batch per batch) */
    double   cpuSeconds[8];        /* thread-CPU seconds per stage in the timed region (sah_cpu_seconds): input + line table, text -> numeric, mapper threads, post-map + formatting, file writes, other */
} staramd_cli_report;

/* Runs the whole job; returns the process exit code (0 ok).  hooks / report may be NULL. */
int staramd_cli_main(int argc, char **argv, const staramd_cli_hooks *hooks, staramd_cli_report *report);

#ifdef __cplusplus
}
#endif
#endif
