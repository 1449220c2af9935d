// dev.h -- device-side types shared by the gfx950 kernels of the seed-search-and-stitch engine.
//
// Data layout in HBM:
//   G     genome, 1 byte/base exactly as in genomeDir (codes 0..5), with GPAD bytes of code 5 on both
//         sides so that extension / junction-repeat scans can run off either end without branches
//   SA    packed suffix array, (GstrandBit+1) bits per entry, read as two aligned 64-bit words
//   SAi   packed L-mer prefix table, (GstrandBit+3) bits per entry
// Per batch: read bases (numeric, combined PE read), per-read seed tables (PC), window tables
// (WC/WA) and window transcripts are handed from kernel to kernel through bump-allocated pools.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/star_amd.h"

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;
typedef int32_t i32;
typedef uint16_t u16;
typedef uint8_t u8;
typedef int8_t i8;

#define GPAD 1024                 // bytes of code 5 before and after the genome on the device
#define NBREAK_MAX 96             // break points of the genomic-length score term

// read-only index + parameters, passed by value to every kernel
struct DevIndex {
    const u8 *G;                  // points at genome base 0 (GPAD bytes of 5 precede it)
    const u64 *SA, *SAi;          // packed arrays as 64-bit words
    const u32 *chrBin;
    const u64 *chrStart, *chrLength;
    const u64 *sjDstart, *sjAstart, *sjdbStart, *sjdbEnd;
    const u8 *sjdbMotif, *sjdbShiftLeft, *sjdbShiftRight, *sjdbStrand;
    u64 nGenome, nSA, sjGstart;
    u64 saiStart[17];
    u64 saMask, saiMask, strandMask, saiAbsentBit, saiNbit;
    u32 saBits, saiBits, strandBit, saiNbases, sparseD, sjdbOverhang, sjdbLength, sjdbN, nChrReal;
    // genomic-length log2 score term as integer break points
    i32 glScoreAt1; i32 glStep; u32 nBreak; u64 glBreak[NBREAK_MAX];
    staramd_params P;
};

// seed = one row of the reference's PC table (IncludeDefine.h:181-189); saEnd = saStart+nrep-1
struct DSeed { u64 saStart; u32 nrep; u16 rStart, L; u8 dir, iFrag; u8 pad[6]; };
// one row of the reference's WA table (IncludeDefine.h:197-204)
struct DWA { u64 gStart; u32 nrep; u16 L, rStart; i32 sjA; u8 anchor, iFrag; u8 pad[2]; };
// window with seeds, output of the window kernel
struct DWin { u32 read; u32 chr; u32 waOffset; u16 nWA; u8 str; u8 pad; };

// per-read bookkeeping carried between kernels
struct DRead {
    u32 status; u32 seedOffset; u32 nSeeds; u32 unmappedLength;
    u32 winOffset; u32 nWin;            // windows with seeds (window pool)
    u32 wtOffset; u32 nWt;              // window-transcript blocks (one per window that recorded transcripts)
    i32 maxScoreMate[2];
    i32 bestW;                          // ordinal of trBest's window among recorded windows, -1
    u32 nTr, nEx;                       // totals for the gather step
};

// working transcript on the device: exon rows are already in the output format
struct DTr {
    staramd_exon ex[STARAMD_MAX_N_EXONS];
    u32 nExons; i32 maxScore;
    u32 nMatch, nMM, nGap, lGap, nDel, lDel, nIns, lIns, nUnique, nAnchor;
    u32 rStart, rLength, mappedLength, roStart;
    u64 gStart, gLength;
    i32 iFrag; u16 intronMotifs[3]; u8 sjMotifStrand; u8 pad;
};
#define DTR_HDR_BYTES (sizeof(DTr) - sizeof(staramd_exon) * STARAMD_MAX_N_EXONS)

// one frame of the explicit depth-first walk of k_stitch (pushed only when a seed is included)
struct Frame { DTr tr; i32 Score; u32 tR2; u64 tG2; u32 iA; u32 state; };
// per-lane window scratch of k_windows
struct WScr { u32 coreS, coreE, extS, extE; u32 chr; u32 waBlock; u32 lrec; u16 nWA; u8 str; u8 alive; };

// block of transcripts recorded for one window, in the window-transcript pool
struct DWinTr { u32 read; u32 trOffset; u32 nTr; u32 exOffset; u32 nEx; u32 chr; u8 str; u8 pad[3]; };

enum { DC_nSAi, DC_nSAprobe, DC_nGcmp, DC_nSAenum, DC_nGstitch, DC_nSeeds, DC_nWindows, DC_nWA, DC_nNodes, DC_nLeaves,
       DC_nStitchCalls, DC_nExtendCalls, DC_nTrOut, DC_N };

// pools and cursors of one batch
struct DevBatch {
    u32 nReads;
    const u8 *bases; const u64 *readOffset; const u16 *mate1Length; const u16 *mmMaxTotal;
    DRead *reads;
    DSeed *seedPool; u32 seedCap;
    DWin *winPool; u32 winCap; DWA *waPool; u32 waCap;
    DWinTr *wtPool; u32 wtCap;
    staramd_transcript *trPool; u32 trCap; staramd_exon *exPool; u32 exCap;
    u32 *cursors;      // [0] seed pool, [1] win pool, [2] wa pool, [3] wt pool, [4] tr pool, [5] ex pool, [6] overflow flags, [8..10] work queues
    u64 *counters;     // DC_N
};

// ---- packed array access: PackedArray::operator[] (source/PackedArray.h:24-32) with aligned loads ----
__device__ __forceinline__ u64 packedGet(const u64 *a, u64 i, u32 bits, u64 mask) {
    u64 b = i * bits; u64 w = b >> 6; u32 s = (u32)(b & 63);
    u64 lo = a[w];
    u64 v = lo >> s;
    if (s + bits > 64) v |= a[w + 1] << (64 - s);
    return v & mask;
}

__device__ __forceinline__ u8 compBase(u8 c) { return c < 4 ? (u8)(3 - c) : c; }   // complementSeqNumbers, SequenceFuns.cpp:4-14
