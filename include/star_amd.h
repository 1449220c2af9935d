/* star_amd.h -- C ABI of the MI355X seed-search-and-stitch engine.
 *
 * The reference (alexdobin/STAR 2.7.11b) has no plugin / FFI interface; the seam this library
 * replaces is the per-read member function
 *       int ReadAlign::mapOneRead()            source/ReadAlign.h:187, source/ReadAlign_mapOneRead.cpp:6-118
 * called from the read loop of ReadAlignChunk::mapChunk (source/ReadAlignChunk_mapChunk.cpp:30-32),
 * batched to one input chunk.  Inputs and outputs are exactly what that function reads and
 * leaves behind in the ReadAlign object:
 *   in : Read1[0] (numeric combined read), Lread, readLength[2], outFilterMismatchNmaxTotal,
 *        the read-only Genome (G, SA, SAi, chr tables, sjdb tables) and the Parameters subset
 *        that reaches the hot path;
 *   out: nW, trAll[iW][0..nWinTr[iW]-1] (best-first per window), trBest, maxScoreMate[], mapMarker.
 *
 * Plain C: pointers + sizes, caller-owned buffers, int return codes, no exceptions.
 * One context per GPU; calls on one context are serialised by the caller.
 */
#ifndef STAR_AMD_H
#define STAR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STARAMD_MAX_N_EXONS 20        /* MAX_N_EXONS, source/IncludeDefine.h:131 */
#define STARAMD_READ_LEN_MAX 650      /* DEF_readSeqLengthMax, source/IncludeDefine.h:140 */
#define STARAMD_SPACER_BASE 11        /* MARK_FRAG_SPACER_BASE, source/IncludeDefine.h:174 */

/* error codes (negative) */
#define STARAMD_OK 0
#define STARAMD_ERR_ARG (-1)
#define STARAMD_ERR_DEVICE (-2)            /* HIP runtime failure; see staramd_last_error() */
#define STARAMD_ERR_RESULT_OVERFLOW (-3)   /* caller's result arrays too small */
#define STARAMD_ERR_SCRATCH_OVERFLOW (-4)  /* a per-read device work-space cap was exceeded */
#define STARAMD_ERR_FATAL_READ (-5)        /* a reference exitWithError condition was hit (see status bits) */

/* ---- read-only index: what Genome::genomeLoad leaves in `class Genome`
 *      (source/Genome.h:26-56, source/Genome_genomeLoad.cpp:139-169,315-336,471-521).
 *      All pointers are HOST pointers; staramd_create copies them to HBM once. ---- */
typedef struct staramd_genome {
    const uint8_t *G;            /* nGenome bytes, codes 0..3 ACGT, 4 N, 5 padding (1 byte/base on disk)   */
    uint64_t nGenome;
    const uint8_t *SA;           /* packed suffix array, (GstrandBit+1) bits/entry (PackedArray.h:24-32)    */
    uint64_t nSA, nSAbyte;
    const uint8_t *SAi;          /* packed SAindex, (GstrandBit+3) bits/entry                               */
    uint64_t nSAi, nSAibyte;
    uint32_t GstrandBit;
    uint32_t gSAindexNbases;     /* L-mer table depth (14 for human)                                        */
    uint64_t genomeSAindexStart[17]; /* offsets of the L=1..Nbases tables, [Nbases]=nSAi                    */
    uint32_t gSAsparseD;
    uint32_t gChrBinNbits;
    const uint64_t *chrStart;    /* nChrReal+1 entries                                                      */
    const uint64_t *chrLength;   /* nChrReal                                                                */
    uint32_t nChrReal;
    const uint32_t *chrBin;      /* chrBinN entries: chromosome of every 2^gChrBinNbits block (Genome.cpp:209) */
    uint64_t chrBinN;
    /* splice-junction database (Genome_genomeLoad.cpp:471-521) */
    uint64_t sjGstart;           /* start of the inserted junction "chromosome"                             */
    uint32_t sjdbOverhang, sjdbLength;
    uint32_t sjdbN;
    const uint64_t *sjDstart, *sjAstart, *sjdbStart, *sjdbEnd;
    const uint8_t *sjdbMotif, *sjdbShiftLeft, *sjdbShiftRight, *sjdbStrand;
} staramd_genome;

/* ---- Parameters that reach the hot path (defaults: source/parametersDefault) ---- */
typedef struct staramd_params {
    uint32_t readNmates;                    /* 1 or 2 */
    /* seed search */
    uint32_t seedSearchStartLmax;           /* 50  */
    double   seedSearchStartLmaxOverLread;  /* 1.0 */
    uint32_t seedSearchLmax;                /* 0   */
    uint32_t seedMultimapNmax;              /* 10000 */
    uint32_t seedPerReadNmax;               /* 1000 */
    uint32_t seedPerWindowNmax;             /* 50 */
    uint32_t seedSplitMin;                  /* 12 */
    uint32_t seedMapMin;                    /* 5 */
    uint32_t maxNsplit;                     /* 10, hard-coded Parameters.cpp:473 */
    /* windows (as re-derived by Genome_genomeLoad.cpp:382-410) */
    uint32_t winAnchorMultimapNmax;         /* 50 */
    uint32_t winBinNbits;                   /* 16 */
    uint32_t winAnchorDistNbins;            /* 9 */
    uint32_t winFlankNbins;                 /* 4 */
    uint32_t winBinChrNbits;                /* gChrBinNbits - winBinNbits */
    uint64_t winBinN;
    uint32_t alignWindowsPerReadNmax;       /* 10000 */
    uint32_t alignTranscriptsPerWindowNmax; /* 100 */
    uint32_t alignTranscriptsPerReadNmax;   /* 10000 */
    /* stitching */
    uint64_t alignIntronMin;                /* 21 */
    uint64_t alignIntronMax;                /* 0 */
    uint64_t alignMatesGapMax;              /* 0 */
    uint32_t alignSJoverhangMin;            /* 5 */
    uint32_t alignSJDBoverhangMin;          /* 3 */
    int32_t  alignSJstitchMismatchNmax[4];  /* 0 -1 0 0 */
    uint32_t alignSplicedMateMapLmin;       /* 0 */
    double   alignSplicedMateMapLminOverLmate; /* 0.66 */
    uint8_t  alignEndsTypeExt[2][2];        /* Parameters.cpp:966-983: all 0 for Local */
    int32_t  alignEndsProtrudeNbasesMax;    /* 0 */
    uint8_t  alignEndsProtrudeConcordantPair; /* consumed by the SAM writer only */
    uint8_t  alignSoftClipAtReferenceEnds;  /* 1 = Yes */
    uint8_t  alignInsertionFlushRight;      /* 0 */
    uint8_t  outFilterIntronStrandsRemoveInconsistent; /* 1 */
    uint8_t  outFilterIntronMotifs;         /* 0 None, 1 RemoveNoncanonical, 2 RemoveNoncanonicalUnannotated */
    uint8_t  outSAMstrandFieldIntronMotif;  /* 0 */
    uint8_t  chimSegmentMinPositive;        /* P.pCh.segmentMin>0: record every transcript */
    uint8_t  outFilterBySJoutStage;         /* 0; 1 / 2 = stages of --outFilterType BySJout (2: whitelist of staramd_set_novel_junctions) */
    int32_t  scoreGap, scoreGapNoncan, scoreGapGCAG, scoreGapATAC;   /* 0 -8 -4 -8 */
    int32_t  scoreDelOpen, scoreDelBase, scoreInsOpen, scoreInsBase; /* -2 -2 -2 -2 */
    int32_t  scoreStitchSJshift;            /* 1 */
    int32_t  sjdbScore;                     /* 2 */
    double   scoreGenomicLengthLog2scale;   /* -0.25 */
    int32_t  outFilterMultimapScoreRange;   /* 1 */
    double   outFilterMismatchNoverLmax;    /* 0.3 */
    uint32_t outFilterMatchNmin;            /* 0: Lread<outFilterMatchNmin => MARKER_READ_TOO_SHORT */
    /* ours (no reference flag): which transcripts staramd_map_batch returns.
     *   0  every transcript recorded in every window (the whole trAll[][] of the reference)
     *   1  only the transcripts ReadAlign::multMapSelect can pick: maxScore + outFilterMultimapScoreRange >= trBest->maxScore
     *      (ReadAlign_multMapSelect.cpp:26-44).  That is all the default post-map path reads; it cuts the result copy ~10x.
     *      Must be 0 when chimeric detection (chimSegmentMin > 0) wants the other windows.
     * Contract of 1: trBest, status, unmappedLength and every field of the RETURNED transcripts / exons are the reference's exactly; nW / nTr
     * count what is returned; maxScoreMate[] is 0 (see staramd_read_result) and the windows that cannot reach the selection threshold are not stitched at all
     * (INTEGRATION.md "Which outputs are the reference's exactly").  A caller that needs trAll[][] or maxScoreMate[] themselves passes 0.
     *   2  chimeric detection with the partner chosen on the device (chimSegmentMinPositive 1): every window is stitched and recorded as under 0, the engine
     *      runs the partner loop of ReadAlign::chimericDetectionOld (ReadAlign_chimericDetectionOld.cpp:50-108: best window against the heads of the other
     *      windows and the other transcripts of its own) and returns what 1 returns plus the partner.  A read with a partner has STARAMD_ST_CHIM_PARTNER set;
     *      then maxScoreMate[0] / [1] = chimScoreBest / chimScoreNext of that loop and unmappedLength = index of the partner (relative to trOffset) |
     *      chimStr of the partner << 30.  Without the bit the loop found none (chimScoreBest 0).  Cuts the result copy of the chimeric configuration ~15x. */
    uint32_t resultSelect;
    uint32_t chimSegmentMin, chimSegmentReadGapMax;   /* P.pCh.segmentMin, P.pCh.segmentReadGapMax: read by resultSelect 2 only */
} staramd_params;

/* ---- one batch of reads: what ReadAlign::oneRead prepares before calling mapOneRead
 *      (source/ReadAlign_oneRead.cpp:35-78) ---- */
typedef struct staramd_batch {
    uint32_t nReads;
    const uint8_t  *bases;        /* concatenated Read1[0] of every read: mate1 | 11 | revcomp(mate2); codes 0..4, 11 */
    const uint64_t *readOffset;   /* nReads+1 offsets into bases; Lread[i] = readOffset[i+1]-readOffset[i]  */
    const uint16_t *mate1Length;  /* readLength[0]; readLength[1] = Lread - readLength[0] - 1 when paired     */
    const uint16_t *mmMaxTotal;   /* outFilterMismatchNmaxTotal per read (ReadAlign_oneRead.cpp:78)            */
} staramd_batch;

/* status bits per read */
#define STARAMD_ST_MAPPED_WINDOWS        0x0001u  /* nW>0 */
#define STARAMD_ST_READ_TOO_SHORT        0x0002u  /* MARKER_READ_TOO_SHORT                    mapOneRead.cpp:100 */
#define STARAMD_ST_NO_GOOD_PIECES        0x0004u  /* MARKER_NO_GOOD_PIECES                    :104 */
#define STARAMD_ST_ALL_PIECES_MULTI      0x0008u  /* MARKER_ALL_PIECES_EXCEED_seedMultimapNmax :108 */
#define STARAMD_ST_NO_GOOD_WINDOW        0x0010u  /* MARKER_NO_GOOD_WINDOW           stitchPieces.cpp:344 */
#define STARAMD_ST_TOO_MANY_ANCHORS      0x0020u  /* MARKER_TOO_MANY_ANCHORS_PER_WINDOW assignAlignToWindow.cpp:76 */
#define STARAMD_ST_FATAL_SEEDS_PER_READ  0x0100u  /* reference exits: storeAligns.cpp:46-51 */
#define STARAMD_ST_TR_PER_READ_LIMIT     0x0200u  /* reference logs a WARNING and stops: stitchPieces.cpp:290-294 */
#define STARAMD_ST_WINDOWS_LIMIT         0x0400u  /* alignWindowsPerReadNmax reached (silent in the reference)    */
#define STARAMD_ST_CHIM_PARTNER          0x0800u  /* resultSelect 2: a chimeric partner was chosen (maxScoreMate[] / unmappedLength carry it, see staramd_params) */
#define STARAMD_ST_SCRATCH_OVERFLOW      0x8000u  /* device work-space cap exceeded: results for this read invalid */

typedef struct staramd_read_result {
    uint32_t status;
    uint32_t nW;              /* windows that recorded >=1 transcript (ReadAlign::nW after stitchPieces) */
    uint32_t nTr;             /* total transcripts over those windows                                    */
    uint32_t trOffset;        /* first transcript of this read in staramd_results.tr                     */
    int32_t  trBest;          /* index (relative to trOffset) of trBest, -1 if none                      */
    int32_t  maxScoreMate[2]; /* resultSelect 0: exactly ReadAlign::maxScoreMate[].  resultSelect 1: always 0 (windows that cannot hold a selectable
                               * transcript are not stitched, so the running maxima do not exist).  Nothing outside the hot path reads the field
                               * (stitchWindowAligns.cpp:234,246 are its only uses in the reference). */
    uint32_t unmappedLength;  /* trBest->rLength of the unmapped classifications (mapOneRead.cpp:100-111); 0 for a read of length 0, where the
                               * reference reports what the previous read of the same thread left in splitR[1][0] */
} staramd_read_result;

/* compact Transcript (source/Transcript.h:10-81): every field the post-map code reads */
typedef struct staramd_transcript {
    uint32_t iW;              /* ordinal of the window among windows with transcripts */
    uint32_t exonOffset;      /* first exon in staramd_results.ex */
    uint16_t nExons;
    uint16_t rStart, rLength, roStart;
    uint8_t  Str, roStr;
    int8_t   iFrag;           /* -1 both mates */
    uint8_t  sjMotifStrand;
    uint32_t Chr;
    uint64_t gStart, gLength;
    int32_t  maxScore;
    uint32_t nMatch, nMM, mappedLength;
    uint32_t nGap, lGap, nDel, lDel, nIns, lIns;
    uint16_t nUnique, nAnchor;
    uint16_t intronMotifs[3];
    uint16_t pad0;
    uint32_t pad1;            /* explicit tail padding: records are compared byte for byte, always 0 */
} staramd_transcript;

typedef struct staramd_exon {
    uint64_t G;               /* EX_G */
    uint16_t R, L;            /* EX_R, EX_L */
    int32_t  sjA;             /* EX_sjA (-1 none) */
    uint8_t  iFrag;           /* EX_iFrag */
    int8_t   canonSJ;         /* junction AFTER this exon (canonSJ[iex]); undefined for the last exon */
    uint8_t  sjAnnot, sjStr;
    uint16_t shiftSJ[2];
    uint32_t pad0, pad1;      /* explicit tail padding, always 0 */
} staramd_exon;

typedef struct staramd_results {
    staramd_read_result *reads;   /* nReads entries                       */
    staramd_transcript  *tr;      uint64_t trCapacity;  uint64_t trCount;  /* count filled by the callee */
    staramd_exon        *ex;      uint64_t exCapacity;  uint64_t exCount;
    /* timing of the device work of this call, measured with HIP events on the engine's stream */
    float msSeed, msWindows, msStitch, msTotalDevice;
} staramd_results;

typedef struct staramd_ctx staramd_ctx;

/* Upload the index to HBM (once per GPU). maxBatchReads/maxBatchBases size the device work space. */
int  staramd_create(staramd_ctx **out, int device, const staramd_genome *g, const staramd_params *p,
                    uint32_t maxBatchReads, uint64_t maxBatchBases);
/* One more context on the device of `owner` that maps against the OWNER's resident index: work space, stream and events are its own, the
 * 30 GB of index are not uploaded a second time.  Two (or more) contexts of one GPU fed by two host threads keep the device busy while the
 * results of a batch are copied out and the next batch is copied in, and fill the low-occupancy tails of each other's launches (the
 * reference's counterpart: several ReadAlignChunk workers over one shared Genome, source/STAR.cpp:194-201).  Calls that change the index
 * (staramd_update_index / insert_junctions / update_tables / set_novel_junctions) go to the owner while no sharer is mapping; the sharers
 * follow.  Destroy the sharers before the owner. */
int  staramd_create_shared(staramd_ctx **out, staramd_ctx *owner, uint32_t maxBatchReads, uint64_t maxBatchBases);
/* Page-locked host memory for the caller-owned batch / result arrays ("caller-owned pinned buffers"): staramd_map_batch copies
 * from / into them by DMA; with pageable memory the runtime stages every copy through a bounce buffer of its own.  Plain memory still works. */
void *staramd_pinned_alloc(uint64_t bytes);
void  staramd_pinned_free(void *p);
/* Replace the index after sjdbInsertJunctions (two-pass); same semantics as create's upload. */
int  staramd_update_index(staramd_ctx *ctx, const staramd_genome *g, const staramd_params *p);
/* Junction insertion into the index RESIDENT in HBM (sjdbBuildIndex, source/sjdbBuildIndex.cpp:15-333): the suffix search of
 * the new junction sequences, the merge into a new packed suffix array and the SAindex of the result all happen on the device; SA / SAindex never
 * cross PCIe (staramd_update_index re-uploads ~30 GB for a human index).  `a` is what sjdbPrepare produced (struct in star_amd_index.h; its G / SA
 * / SAout / SAiOut members are ignored).  When SAout / SAiOut are not NULL the new packed arrays are also copied out (--sjdbInsertSave All).
 * Follow with staramd_update_tables, which brings the junction tables and the parameters of the new index. */
struct staramd_sjdb_args; struct staramd_sjdb_result;
int  staramd_insert_junctions(staramd_ctx *ctx, const struct staramd_sjdb_args *a, uint8_t *SAout, uint64_t saOutCapacity, uint8_t *SAiOut, uint64_t saiOutCapacity,
                              struct staramd_sjdb_result *res);
/* Does the device have the memory for staramd_insert_junctions of up to maxJunctions junctions of sjdbLength bases each (the transient work space is several
 * times the suffix array: ~110 GB for a human index)?  1 = yes, 0 = no (or the context is not usable).  The front end asks once, before it releases its host copy
 * of the suffix array: with 0 it keeps that copy and inserts through host buffers (staramd_sjdb_insert + staramd_update_index) instead
 * (sjdbInsertJunctions, source/sjdbInsertJunctions.cpp:11-102, needs the suffix array wherever it runs). */
int  staramd_insert_junctions_fits(staramd_ctx *ctx, uint64_t maxJunctions, uint32_t sjdbLength);
/* New chromosome / junction tables and parameters for an index whose big arrays (G, SA, SAindex) are already in place in HBM; g->G / SA / SAi are not read. */
int  staramd_update_tables(staramd_ctx *ctx, const staramd_genome *g, const staramd_params *p);
/* 2nd stage of --outFilterType BySJout.  Replaces the mutation of P.sjNovelStart / P.sjNovelEnd / P.sjNovelN and
 * P.outFilterBySJoutStage between the two mapping stages (source/STAR.cpp:203-220, source/outputSJ.cpp:139-161): with
 * stage == 2 a transcript is recorded only if each of its unannotated junctions (start = first intron base, end = last
 * intron base, genome coordinates) is in this list (source/stitchWindowAligns.cpp:169-177).  The list must be sorted by
 * (start, end) -- the order of the collapsed junction table it is built from.  stage 0 / 1 turn the check off again.
 * Only the two small arrays and the parameter block are uploaded; the index stays where it is. */
int  staramd_set_novel_junctions(staramd_ctx *ctx, const uint64_t *start, const uint64_t *end, uint64_t n, uint32_t stage);
/* what this engine library can do beyond the base contract, a bit mask: STARAMD_CAP_CHIM_SELECT = staramd_params::resultSelect 2 */
#define STARAMD_CAP_CHIM_SELECT 1u
uint32_t staramd_capabilities(void);
/* Map one batch: replaces the per-read loop around ReadAlign::mapOneRead.
 * STARAMD_ERR_RESULT_OVERFLOW: r->trCount / r->exCount say what the batch needs; its results stay resident, and the next call with the SAME batch
 * (same arrays, same reads) and larger result arrays copies them out without mapping anything again. */
int  staramd_map_batch(staramd_ctx *ctx, const staramd_batch *b, staramd_results *r);
/* Same, but the batch is taken from the copy already resident in HBM from the previous call with
 * identical geometry (bench: inputs resident before the timed region). */
int  staramd_map_resident(staramd_ctx *ctx, staramd_results *r);
void staramd_destroy(staramd_ctx *ctx);
const char *staramd_last_error(void);
/* algorithmic counters of the last batch: nSAi, nSAprobe, nGcmp, nSAenum, nGstitch, ... */
int  staramd_get_counters(staramd_ctx *ctx, uint64_t *out, int n);
/* HIP-event times (ms) of the stages of the last batch, on the engine's stream:
 * [0] seed search  [1] windows  [2] stitch order  [3] stitch walk (dominant kernel k_stitch_win)
 * [4] verify + replay + finish  [5] scan + gather  [6] total  [7] of [1]: the middle + last k_windows launches  [8] of [3]: the k_stitch_lane launch */
int  staramd_get_timings(staramd_ctx *ctx, float *out, int n);

#ifdef __cplusplus
}
#endif
#endif
