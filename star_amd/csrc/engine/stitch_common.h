// stitch_common.h -- state and accessors shared by the cooperative stitcher (k_stitch.hip) and the scalar
// restatement kept for the shadow-validation build (stitch_scalar.h).
#pragma once
#include "dev.h"

extern __shared__ u32 ldsReads[];          // blockDim.x * ldsWords 32-bit words: 4-bit packed read of every lane

struct Hdr {                               // transcript header + walk position, kept in registers
    u64 gStart, tG2;
    u32 nExons; i32 Score;
    u32 nMatch, nMM, nGap, lGap, nDel, lDel, nIns, lIns, nUnique, nAnchor, rStart, tR2;
};
struct SFrame { Hdr h; u32 iA; u32 pad; staramd_exon eA; };      // 112 bytes

// nodes / leaves / stitch / extend call counters are diagnostics: they cost registers in a kernel that is 37 VGPRs over its budget,
// so only the profile and shadow builds keep them (bench.py --profile-sections reports them); the genome-bytes counter stays
#if defined(STARAMD_PROFILE) || defined(STARAMD_SHADOW)
#define DIAG(x) x
#else
#define DIAG(x)
#endif

struct StitchCtx {
    const DevIndex *X;
    u32 ldsByte;                           // byte offset of this lane's packed read inside ldsReads
    u32 Lread; u32 str;                    // R[] accessor: Read1[0] for + windows, Read1[2] (rev-comp) for - windows
    u32 readLength[2]; u32 mmMaxTotal;
    i32 maxScoreMate[2];
    i32 sens[2];                           // see DWinOut::sens
    GCache ca, cb;                         // donor-side and acceptor-side genome streams
    u64 nGstitch; u32 nStitchCalls, nExtendCalls, nNodes, nLeaves, nLeavesBound, nLeavesEarly;      // (all but the first: DIAG builds only)
    u64 *shadow;                           // shadow-validation build: disagreement counters
#ifdef STARAMD_PROFILE
    u64 prof[16];
    u32 sprofKind; u64 sprof[12];       // coopStitch by kind of join: [0,1] annotated (cycles, calls) [2,3] no gap / equal gap [4,5] deletion / junction [6,7] mate join [8,9] insertion / rejected early; [10] window set-up [11] item fetch + flush
#endif
    u8 *candBase; u32 candTop, candCap, nCand; bool logOn, logOvf;   // candidate log of the current window (see DWinOut)
};

__device__ __forceinline__ u8 rdNib(const StitchCtx &c, u32 j) {
#ifdef STARAMD_WAVE_EMUL
    // coopExtend evaluates 64 positions at once; the lanes behind the position that ends the scan (mate spacer, end of the piece) may index past the
    // read -- their values are masked out.  The LDS of the device answers an out-of-range read with 0; host memory does not.
    if (j >= c.Lread + 1u) return 0;
#endif
    u8 b = ((const __attribute__((address_space(3))) u8 *)ldsReads)[c.ldsByte + (j >> 1)];
    return (j & 1) ? (u8)(b >> 4) : (u8)(b & 15);
}
__device__ __forceinline__ u8 RD(const StitchCtx &c, u32 i) {        // R[i], ReadAlign_stitchPieces.cpp:321
    if (c.str == 0) return rdNib(c, i);
    return compBase(rdNib(c, c.Lread - 1 - i));
}
__device__ __forceinline__ u8 GA(StitchCtx &c, u64 pos) { c.nGstitch++; return gcGet(c.X->G, c.ca, (i64)pos); }
__device__ __forceinline__ u8 GB(StitchCtx &c, u64 pos) { c.nGstitch++; return gcGet(c.X->G, c.cb, (i64)pos); }


// ---- windows and leaves that are not walked (both stitch kernels) -----------------------------------------------------------------------
// Score bound: every read base scores at most +1 (seed, gap fill, extension), every junction at most perJ (the positive part of the junction scores), and a
// transcript of a window with n seeds has at most n - 1 junctions: a transcript over the mates `mates` of a window scores <= their lengths + perJ * (n - 1).
// multMapSelect only picks transcripts with maxScore >= trBest->maxScore - range (ReadAlign_multMapSelect.cpp:26-44), and trBest >= every recorded score `best`.
//
// What a window W that is not walked changes elsewhere: only maxScoreMate[f], f in mates(W) (stitchWindowAligns.cpp:232-247), which the walk of a LATER window W'
// reads in one place -- the clause that records a single-mate-f transcript T' although it is out of range of the best of W'.  In a window that holds one mate only,
// and in every window of a single-end read, that clause can never decide (there maxScoreMate[f] >= every leaf score of the window so far >= the window's best, so
// "in range of maxScoreMate" implies "in range of the window's best"): single-end reads have no cross-window dependence at all.  In a two-mate window W' a lower
// maxScoreMate[f] records MORE such T' (all of them out of range of W's bound, hence unselectable), and the only thing a recorded T' does to other records is to
// remove single-mate-f transcripts R whose blocks it covers (:267-285; a two-mate transcript is never covered by a single-mate one, and a T' that pushes a record off a
// full list ranks above it).  So the result is exact for what is selected as soon as, besides W's own bound, NO single-mate transcript of the read can be selected:
//     singleBar = longest mate + perJ * (most seeds of any window of the read - 1) + range  <  best.
// The same bar covers the single-mate leaves of a two-mate window that are not finalised (5.6) and the single-mate windows skipped after the two-mate ones (sweep 1).
__device__ __forceinline__ i32 pruneSingleBar(const staramd_params &P, i32 perJ, u32 len0, u32 len1, u32 maxSeedsRead) {
#ifdef STARAMD_PRUNE_SLACK_R5          // A/B builds: the slack of rounds 2-5 (as many junctions as a transcript has exon slots)
    maxSeedsRead = STARAMD_MAX_N_EXONS;
#endif
    const u32 nj = maxSeedsRead > 0 ? min(maxSeedsRead - 1u, (u32)STARAMD_MAX_N_EXONS - 1u) : 0u;
    return (i32)max(len0, len1) + perJ * (i32)nj + P.outFilterMultimapScoreRange;
}
// a window (mates, nWA seeds) of a read with mate lengths len0 / len1 need not be walked once a score `best` is recorded in the read
__device__ __forceinline__ bool pruneWindow(const staramd_params &P, i32 perJ, u32 mates, u32 nWA, u32 len0, u32 len1, i32 singleBar, i32 best) {
    const i32 bound = (i32)((mates & 1u) ? len0 : 0u) + (i32)((mates & 2u) ? len1 : 0u) + perJ * ((i32)nWA - 1);
    return bound + P.outFilterMultimapScoreRange < best && (P.readNmates != 2 || singleBar < best);
}

#ifdef STARAMD_PROFILE
#define PROF_T0() u64 prof_t0_ = __builtin_readcyclecounter()
#define PROF_ADD(c, k) (c).prof[k] += __builtin_readcyclecounter() - prof_t0_
#define PROF_MARK(c, k) { u64 prof_t1_ = __builtin_readcyclecounter(); (c).prof[k] += prof_t1_ - prof_t0_; prof_t0_ = prof_t1_; }
#define SPROF_KIND(k) (c).sprofKind = (k)
#define SPROF_ADD(c) { const u64 d_ = __builtin_readcyclecounter() - prof_t0_; const u32 k_ = (c).sprofKind; \
    if (k_ == 0) { (c).sprof[0] += d_; (c).sprof[1]++; } else if (k_ == 1) { (c).sprof[2] += d_; (c).sprof[3]++; } else if (k_ == 2) { (c).sprof[4] += d_; (c).sprof[5]++; } \
    else if (k_ == 3) { (c).sprof[6] += d_; (c).sprof[7]++; } else { (c).sprof[8] += d_; (c).sprof[9]++; } }
#else
#define SPROF_KIND(k)
#define SPROF_ADD(c)
#define PROF_T0()
#define PROF_ADD(c, k)
#define PROF_MARK(c, k)
#endif
