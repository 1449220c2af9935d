/* star_amd_host.h -- C interface of the HOST library (libstaramd_host.so): everything of STAR's alignReads around the hot path -- parameters, genomeDir
 * loader, read batcher, post-map (multMapSelect ... SAM/BAM/SJ/Log writers), the phases of 2-pass / BySJout, quantification, chimeric detection.
 * It is NOT the drop-in boundary (that is include/star_amd.h, the engine); it is what a driver needs to run whole jobs around an engine context:
 * star_amd/csrc/host/main.cpp (the command line), star_amd/capi.py (tests, bench).  All functions return 0 / a count on success and -1 on error
 * (text in sah_error) unless said otherwise.  One handle = one run; calls on a handle are made from one thread, except that in the pipelined
 * variant sah_parse_slot and sah_emit_slot* may each live on their own thread (each called in batch order).
 *
 *   h = sah_create(argc, argv)                         STAR's command line (Parameters::inputParameters, source/Parameters.cpp:311)
 *   loop over phases:                                  1 pass; 2 with --twopassMode Basic; + the 2nd stage of --outFilterType BySJout
 *     while (n = sah_next_batch(h, max, &b)) > 0:      ReadAlignChunk::processChunks                     (source/ReadAlignChunk_processChunks.cpp)
 *         staramd_map_batch(ctx, &b, &res)             ReadAlign::mapOneRead for every read              (include/star_amd.h)
 *         if (sah_merged_batch(h, &mb) > 0) staramd_map_batch(ctx, &mb, &resM)       --peOverlapNbasesMin: peMergeRA->mapOneRead()
 *         if (sah_wasp_batch(h, &res, &wb) > 0) staramd_map_batch(ctx, &wb, &resW)   --waspOutputMode:     waspRA->mapOneRead()
 *         sah_wasp_results(h, &res, &resW or NULL)
 *         sah_emit_merged(h, &res, &resM or NULL)      multMapSelect ... outputAlignments               (source/ReadAlign_oneRead.cpp:87-111)
 *     phase = sah_next_phase(h)                        0 done; 1 index changed -> staramd_update_index(ctx, sah_genome(h), sah_params(h));
 *                                                      2 junction whitelist   -> staramd_set_novel_junctions(ctx, start, end, n, 2)
 *   sah_finish(h); sah_destroy(h)                      SJ.out.tab, Log.final.out, sorted BAM, counts ... */
#ifndef STAR_AMD_HOST_H
#define STAR_AMD_HOST_H
#include "star_amd.h"
#include "star_amd_index.h"
#ifdef __cplusplus
extern "C" {
#endif

void *sah_create(int argc, char **argv, char *errbuf, int errlen);      /* NULL on error, text in errbuf */
void  sah_destroy(void *h);
const char *sah_error(void *h);
int   sah_tool_done(void *h);                                           /* 1: --runMode inputAlignmentsFromBAM, everything happened in sah_create */
/* --runMode genomeGenerate: sah_create scanned the FASTA files; build SA + SAindex into these buffers with staramd_index_build
 * (star_amd_index.h), then sah_generate_finish inserts the annotated junctions and writes the genomeDir files */
/* call BEFORE sah_create: junction insertion runs on the device through fn (= staramd_sjdb_insert); NULL = host restatement (sjdb_insert.cpp) */
void  sah_set_sjdb_device_fn(int (*fn)(int device, const staramd_sjdb_args *, staramd_sjdb_result *), int device);
/* junction insertion on the arrays RESIDENT in the engine contexts: fn runs staramd_insert_junctions on every context (include/star_amd.h);
 * used once sah_engines_ready was called; sah_index_in_engine: 1 = the last phase change left the new index in the engines already
 * (follow with staramd_update_tables instead of staramd_update_index); reading it clears it */
void  sah_set_sjdb_resident_fn(int (*fn)(void *user, const staramd_sjdb_args *, staramd_sjdb_result *), void *user);
void  sah_engines_ready(void *h);
/* chimeric detection (--chimSegmentMin > 0, --chimMultimapNmax 0, no merging of overlapping mates): the engine runs the partner loop (staramd_params::resultSelect 2)
 * instead of returning every transcript of every window.  Call before the engine contexts are created, when staramd_capabilities() has STARAMD_CAP_CHIM_SELECT;
 * returns 1 when the run's parameters now say so, 0 when they do not qualify. */
int   sah_chim_select_on_device(void *h);
int   sah_index_in_engine(void *h);
int   sah_generate_mode(void *h);
int   sah_generate_buffers(void *h, const uint8_t **G, uint64_t *nGenome, uint32_t *GstrandBit, uint32_t *saIndexNbases,
                           uint8_t **SA, uint64_t *saCap, uint8_t **SAi, uint64_t *saiCap);
int   sah_generate_finish(void *h, uint64_t nSA, uint64_t nSAbyte, uint64_t nSAibyte);
const staramd_genome *sah_genome(void *h);                              /* what staramd_create / staramd_update_index take */
const staramd_params *sah_params(void *h);
uint64_t sah_batch_reads(void *h);                                      /* --gpuBatchReads */
int   sah_device(void *h);                                              /* --gpuDevice */
int   sah_threads(void *h);                                             /* --runThreadN */
double sah_genome_load_seconds(void *h);
/* seconds of the post-map stage so far: out[0] waiting for a free text-buffer set, [1] formatting on threads, [2] serial tail of the batches, [3] the writer thread busy */
void  sah_emit_seconds(void *h, double out[4]);
/* batches so far that were [0] written through a mapping of the output file, [1] read in place from mappings of the input files */
void  sah_fast_path_counts(void *h, uint64_t out[2]);
int   sah_needs_second_batch(void *h);                                 /* 1: batches of this run can have a second batch built from them (sah_merged_slot / sah_wasp_slot) */
/* thread-CPU seconds per pipeline stage, process-wide, since the last reset: [0] input + line table, [1] text -> numeric batch, [2] mapper threads (the host side of
 * staramd_map_batch), [3] post-map + formatting, [4] output file writes, [5] everything else that was counted.  sah_cpu_add: the stage threads of the front end add their own. */
void  sah_cpu_add(int stage, uint64_t ns);
void  sah_cpu_seconds(double out[8], int reset);

/* one batch at a time */
int   sah_next_batch(void *h, uint64_t maxReads, staramd_batch *out);   /* number of reads, 0 at the end of the input of this phase */
int   sah_merged_batch(void *h, staramd_batch *out);                    /* merged mates of the current batch, 0 = none */
int   sah_wasp_batch(void *h, const staramd_results *res, staramd_batch *out);      /* allele-swapped reads built from the batch's results, 0 = none */
int   sah_wasp_results(void *h, const staramd_results *res, const staramd_results *resWasp);
int   sah_emit(void *h, const staramd_results *res);
int   sah_emit_merged(void *h, const staramd_results *res, const staramd_results *resMerged);

/* pipelined: three batch slots, so that parsing of batch k+1, mapping of batch k and post-map of batch k-1 overlap */
int   sah_parse_slot(void *h, int slot, uint64_t maxReads, staramd_batch *out);
/* sah_parse_slot in two halves, for two threads (each half called in batch order): sah_fill_slot reads the text of the next batch and finds its lines
 * (returns the number of reads, 0 at the end of the input, < 0 on error); sah_convert_slot turns it into the numeric batch while sah_fill_slot already
 * reads the next one */
int   sah_fill_slot(void *h, int slot, uint64_t maxReads);
/* Memory for the numeric arrays of the batches (what staramd_map_batch copies to the device): pass staramd_pinned_alloc / staramd_pinned_free for page-locked
 * memory.  Process-wide; call once before the first batch is parsed (before sah_create is simplest) and not again. */
void  sah_set_batch_alloc(void *(*alloc)(uint64_t bytes), void (*release)(void *p));
int   sah_convert_slot(void *h, int slot, staramd_batch *out);
int   sah_merged_slot(void *h, int slot, staramd_batch *out);
int   sah_wasp_slot(void *h, int slot, const staramd_results *res, staramd_batch *out);
int   sah_wasp_results_slot(void *h, int slot, const staramd_results *res, const staramd_results *resWasp);
int   sah_emit_slot(void *h, int slot, const staramd_results *res);
int   sah_emit_slot_merged(void *h, int slot, const staramd_results *res, const staramd_results *resMerged);

/* phases and the end of the run */
int   sah_next_phase(void *h);                                          /* 0 / 1 / 2, see above; -1 on error */
uint64_t sah_novel_junctions(void *h, const uint64_t **start, const uint64_t **end);
int   sah_in_pass1(void *h);
/* --limitSjdbInsertNsj and the length of one inserted junction sequence (2 * sjdbOverhang + 1) of this run: what a junction insertion can grow the index by at most */
uint64_t sah_limit_sjdb_insert(void *h);
uint32_t sah_sjdb_length(void *h);
int   sah_pass1_end(void *h);
int   sah_in_stage1(void *h);
int   sah_finish(void *h);

/* several ranks on one job (star_amd/multi_gpu.py): junction tables, counters and gene counts are exchanged between ranks, rank 0 writes */
uint64_t sah_sj_export(void *h, void *buf, uint64_t capRecords);
int   sah_sj_import(void *h, const void *buf, uint64_t nRecords);
void  sah_sj_clear(void *h);
void  sah_sj_select(void *h, int which);
int   sah_stats_export(void *h, uint64_t *out);
int   sah_stats_import_add(void *h, const uint64_t *in);
uint64_t sah_quant_export(void *h, uint64_t *out, uint64_t cap);
int   sah_quant_import_add(void *h, const uint64_t *in, uint64_t n);
uint64_t sah_sizeof(int which);                                         /* struct sizes, for the ctypes mirror */

#ifdef __cplusplus
}
#endif
#endif
