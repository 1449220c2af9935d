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
    u64 nGstitch; u32 nStitchCalls, nExtendCalls, nNodes, nLeaves;
    u64 *shadow;                           // shadow-validation build: disagreement counters
#ifdef STARAMD_PROFILE
    u64 prof[16];
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


#ifdef STARAMD_PROFILE
#define PROF_T0() u64 prof_t0_ = __builtin_readcyclecounter()
#define PROF_ADD(c, k) (c).prof[k] += __builtin_readcyclecounter() - prof_t0_
#define PROF_MARK(c, k) { u64 prof_t1_ = __builtin_readcyclecounter(); (c).prof[k] += prof_t1_ - prof_t0_; prof_t0_ = prof_t1_; }
#else
#define PROF_T0()
#define PROF_ADD(c, k)
#define PROF_MARK(c, k)
#endif
