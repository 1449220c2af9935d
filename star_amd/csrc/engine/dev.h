// dev.h -- device-side types shared by the gfx950 kernels of the seed-search-and-stitch engine.
//
// Data layout in HBM:
//   G     genome, 1 byte/base exactly as in genomeDir (codes 0..5), with GPAD bytes of code 5 on both
//         sides so that extension / junction-repeat scans can run off either end without branches;
//         the device reads it in aligned 8-byte words (GCache) -- 8 bases per gather
//   SA    packed suffix array, (GstrandBit+1) bits per entry, read as two aligned 64-bit words
//   SAi   packed L-mer prefix table, (GstrandBit+3) bits per entry
// Per batch: read bases (numeric, combined PE read; plus a 4-bit packed copy for LDS staging),
// per-read seed tables (PC), window tables (WC/WA) and window transcripts are handed from kernel to
// kernel through bump-allocated pools.
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
typedef int16_t i16;
typedef int8_t i8;

#define GPAD 4096                 // bytes of code 5 before and after the genome on the device
#define NBREAK_MAX 96             // break points of the genomic-length score term
#define WA_MAX 64                 // device limit for --seedPerWindowNmax (one lane per window seed in k_windows)

// read-only index + parameters, passed by value to every kernel
struct DevIndex {
    const u8 *G;                  // points at genome base 0 (GPAD bytes of 5 precede it); 8-byte aligned
    const u64 *SA, *SAi;          // packed arrays as 64-bit words
    const void *SAK; u32 sakBases, padSak;      // keys beside the suffix array (k_seed.hip: two words per entry), or null; sakBases = the prefix length the keys lie behind (saiNbases)
    const u32 *chrBin;
    const u64 *chrStart, *chrLength;
    const u64 *sjDstart, *sjAstart, *sjdbStart, *sjdbEnd;
    const u8 *sjdbMotif, *sjdbShiftLeft, *sjdbShiftRight, *sjdbStrand;
    // the four of them in one word per junction (built at upload): motif | strand << 3 | shiftLeft << 8 | shiftRight << 16 -- the stitcher wants all four at once, one load
    const u32 *sjdbInfo;
    const u64 *sjNovelStart, *sjNovelEnd; u64 sjNovelN;       // whitelist of the 2nd stage of BySJout (staramd_set_novel_junctions)
    // (start, end) -> junction: open-addressing table over the annotated junctions, built at upload (engine.hip buildSjdbHash).  Slot = two words:
    // (index + 1) << 40 | start, sjdbInfo << 40 | end; an empty slot is 0.  binarySearch2 (binarySearch2.cpp:3-43) walks ~19 dependent loads through sjdbStart for 350 k
    // junctions; the table answers with one.  Null when the coordinates / the junction count do not fit the packing: the bisection is used then.
    const u64 *sjdbHash; u32 sjdbHashMask; u32 padHash;
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
// seed stage, lane = unit (k_seed.hip): what a unit keeps of a seed until the read's units are all done; one (piece, direction, start point) of a read's schedule;
// per read: where its groups start, how many, and what the classification needs of qualitySplit
#define SEED_SLOTS 8u
struct SeedSlot { u64 i0; u32 nrep; u16 shift, L; };                                   // L bit 15: Nrep does not fit 32 bits
struct SeedUnit { u32 read, group; u16 pS, pL; u8 iFrag, istart, nstart, kind; };      // kind 0: forward + backward from start point 0; 1: forward; 2: backward
struct SeedPlan { u32 group0; u16 nGroups, nSplit, LgoodMin, handOn; u32 pad; };
struct SeedWork { SeedUnit *units; u32 unitCap; SeedSlot *slots; u32 *groupHead; u32 groupCap; SeedPlan *plan; u32 *handOn; u32 slotLimit; };      // slotLimit <= SEED_SLOTS (tests lower it)
// one row of the reference's WA table (IncludeDefine.h:197-204)
struct DWA { u64 gStart; u32 nrep; u16 L, rStart; i32 sjA; u8 anchor, iFrag; u8 pad[2]; };
// window with seeds, output of the window kernel
struct DWin { u32 read; u32 chr; u32 waOffset; u16 nWA; u8 str; u8 mates; };   // mates: bit f set = the window holds seeds of mate f

// per-read bookkeeping carried between kernels
struct DRead {
    u32 status; u32 seedOffset; u32 nSeeds; u32 unmappedLength;
    u32 winOffset; u32 nWin;            // windows with seeds (window pool)
    u32 wtOffset; u32 nWt;              // wtOffset: largest seed count among the read's windows (k_windows); nWt: windows that recorded transcripts
    i32 maxScoreMate[2];
    i32 bestW;                          // ordinal of trBest's window among recorded windows, -1
    u32 nTr, nEx;                       // totals for the gather step
    i32 pruneBest;                      // best window score seen so far among the read's window items (k_stitch_win window pruning; atomicMax)
    u32 pad0;
};

// result of stitching one window (slot = index of the window in winPool)
//   mm[f]   best score of a single-mate transcript of mate f among the leaves of this window (order independent)
//   sens[f] smallest (Score + outFilterMultimapScoreRange) among the leaves of mate f whose recording was decided by the
//           maxScoreMate clause alone (stitchWindowAligns.cpp:245-247); INT32_MAX if none.  The window was stitched with
//           the incoming maxScoreMate of minIn[]; the result is exact iff sens[f] >= the true incoming value
//   candOff32 / nCand: the window's CANDIDATE LOG (every leaf that reached the record decision, as output-format records,
//           in walk order) inside candPool, offset in 32-byte units; nCand = 0xFFFFFFFF: log not available.  Re-deciding
//           the window for another incoming maxScoreMate only needs a replay of this log, not a second walk.
struct DWinOut { u32 trOffset, nTr, exOffset, nEx; i32 mm[2]; i32 sens[2]; i32 minIn[2]; i32 headScore; u32 candOff32; u64 headGlen; u32 nCand; u32 pad; };

enum { DC_nSAi, DC_nSAprobe, DC_nGcmp, DC_nSAenum, DC_nGstitch, DC_nSeeds, DC_nWindows, DC_nWA, DC_nNodes, DC_nLeaves,
       DC_nStitchCalls, DC_nExtendCalls, DC_nTrOut, DC_nOvfWin, DC_nOvfStitch, DC_nRedoWin, DC_nReplayWin,
       DC_shadowBad, DC_shadowN, DC_shadowExtBad, DC_shadowExtN,   // shadow-validation build only (see stitch_scalar.h)
       DC_prof0, DC_prof1, DC_prof2, DC_prof3, DC_prof4, DC_prof5, DC_prof6, DC_prof7,   // -DSTARAMD_PROFILE build: shader-clock cycles per section of k_stitch_win
       DC_prof8, DC_prof9, DC_prof10, DC_prof11, DC_prof12, DC_prof13, DC_prof14, DC_prof15,
       DC_nPrunedWin,                                                  // windows not walked because no transcript of theirs could be selected (k_stitch_win)
       DC_nRewalkRead,                                                 // light reads whose two-mate windows did not clear the bar: walked again in window order
       DC_nLaneItems,                                                  // reads stitched by the lane-per-read kernel (k_stitch_lane); the others went to the cooperative one
       DC_nOwnerLookups, DC_nOwnerMisses,                              // k_windows pass B: loci that passed the covered-bins filter / of those, loci no window owns (filter false positives)
       DC_nAnchorLoci, DC_nAnchorReplayed,                             // k_windows pass A: anchor loci enumerated / replayed one by one (not owned by a window when their chunk was read)
       DC_wprof5, DC_wprof6, DC_wprof7,                                // -DSTARAMD_PROFILE build: more sections of k_windows
       DC_nSkippedLeaves, DC_nRewalkWin,                               // stitch kernels: single-mate leaves (and subtrees of them) not finalised / two-mate windows walked again in full
       DC_nLeavesBound, DC_nLeavesEarly,                               // profile / shadow builds: leaves dropped by their score bound before the extensions / by their score after them (k_stitch.hip finalizeTranscript)
       DC_sprof0, DC_sprof1, DC_sprof2, DC_sprof3, DC_sprof4, DC_sprof5, DC_sprof6, DC_sprof7, DC_sprof8, DC_sprof9, DC_sprof10, DC_sprof11,   // -DSTARAMD_PROFILE build: coopStitch by kind of join (cycles, calls) and the walk's own parts
       DC_N };

// cursors[] slots.  Every counter has a 128-byte line of its own (CS words apart): an L2 channel serves the atomics of one line one after the other (~2 ns each, measured:
// an empty launch of 524 288 lanes that each fail one ticket costs 1 ms), and the allocation cursors of a kernel -- windows, seed rows, work items -- would share a line and a queue
#define CS 32
enum { CUR_SEED = 0 * CS, CUR_WIN = 1 * CS, CUR_WA = 2 * CS, CUR_TR = 4 * CS, CUR_EX = 5 * CS, CUR_FLAGS = 6 * CS,
       CUR_TICKET_SEED = 8 * CS, CUR_TICKET_WIN = 9 * CS, CUR_OVF_WIN = 11 * CS, CUR_TICKET_WIN2 = 13 * CS, CUR_OVF_WIN2 = 12 * CS, CUR_TICKET_WIN3 = 14 * CS,
       // stitch stage: work lists of window ids and their tickets
       CUR_ST_TICKET0 = 16 * CS, CUR_ITEM = 25 * CS,                               // pass 0: all work items (reads or windows)
       CUR_ST_REDO = 19 * CS, CUR_ST_TICKET1 = 20 * CS,                            // pass 1, full re-walk (no candidate log available)
       CUR_ST_REPLAY = 21 * CS, CUR_ST_TICKETR = 22 * CS,                          // pass 1, replay of the candidate log
       CUR_ST_HEAVY = 23 * CS, CUR_ST_TICKETH = 24 * CS,                           // pass 0, items deferred by the lean-LDS launch to the full-size launch
       CUR_ST_HEAVY2 = 30 * CS, CUR_ST_TICKETH2 = 31 * CS,                         // pass 0 in three launches (lane kernel, main cooperative launch, full-depth launch): what the main launch defers
       // seed stage, lane = unit (k_seed.hip): groups and units handed out by k_seed_plan, the ticket of k_seed_units, reads handed on to k_seed_search
       CUR_SEED_GROUPS = 26 * CS, CUR_SEED_UNITS = 27 * CS, CUR_TICKET_SEED_UNITS = 28 * CS, CUR_OVF_SEED = 29 * CS,
       CUR_N = 32 * CS };
// CUR_FLAGS bits: pool overflows (the host grows the pool and re-runs the batch)
enum { OVF_SEEDPOOL = 1, OVF_WINPOOL = 4, OVF_TRPOOL = 16, OVF_HARD = 64 };

// pools and cursors of one batch
struct DevBatch {
    u32 nReads;
    const u8 *bases; const u64 *readOffset; const u16 *mate1Length; const u16 *mmMaxTotal;
    const u32 *packed; u32 packWords;   // 4-bit packed reads, packWords 32-bit words per read (k_pack_reads)
    DRead *reads;
    DSeed *seedPool; u32 seedCap;
    DWin *winPool; u32 winCap; DWA *waPool; u32 waCap;
    DWinOut *wout;                      // winCap slots
    staramd_transcript *trPool; u32 trCap; staramd_exon *exPool; u32 exCap;
    u32 *order;        // work items in stitch order: sorted by estimated work, then dealt round-robin to groups of 64 tickets
    u32 *costHist;     // 32 cost classes + 32 offsets
    // stitch work items: bit 31 set = a whole (light) read, its windows walked in order by one wavefront; else a window id
    u32 *items; u8 *itemClass;   // winCap entries; itemClass ~ log2(estimated walk size)
    u32 *ovfWin;       // reads deferred by the first pass of k_windows (table rows in LDS exhausted)
    u32 *ovfWin2;      // reads deferred by the middle pass (larger LDS table) to the pass with the table in global memory
    u32 *redoList, *replayList;        // stitch pass-1 work lists (window ids)
    u32 *heavyList;                    // pass-0 items whose windows hold more seeds than the lean launch has LDS for
    u32 *heavyList2;                   // ... and, when the lane kernel fills heavyList, what the main cooperative launch behind it hands on to the full-depth one
    u8 *candPool; u64 candWaveBytes;   // candidate logs: one private region per wavefront of k_stitch_win
    u32 *candTops;                     // bytes of its region a wavefront of the first pass-0 launch used (the second one goes on behind them)
    u32 *cursors;      // CUR_*
    u64 *counters;     // DC_N
};

// Index arrays and batch buffers are global memory, but their pointers are read from structures (or rebuilt from integers),
// so the compiler knows no address space and emits flat_load; a flat access also counts against lgkmcnt and holds up
// the waits for scalar loads and LDS.  GLOBAL(T, p) states the address space: global_load.
#define GLOBAL(T, p) ((const __attribute__((address_space(1))) T *)(p))

// ---- packed array access: PackedArray::operator[] (source/PackedArray.h:24-32) with aligned loads ----
__device__ __forceinline__ u64 packedGet(const u64 *a0, u64 i, u32 bits, u64 mask) {
    const __attribute__((address_space(1))) u64 *a = GLOBAL(u64, a0);
    u64 b = i * bits; u64 w = b >> 6; u32 s = (u32)(b & 63);
    u64 lo = a[w];
    u64 v = lo >> s;
    if (s + bits > 64) v |= a[w + 1] << (64 - s);
    return v & mask;
}

__device__ __forceinline__ u8 compBase(u8 c) { return c < 4 ? (u8)(3 - c) : c; }   // complementSeqNumbers, SequenceFuns.cpp:4-14

// ---- genome access in aligned 8-byte words: one gather serves 8 consecutive bases of a scan ----
struct GCache { i64 base; u64 word; };
__device__ __forceinline__ void gcInit(GCache &c) { c.base = (i64)0x7fffffffffffff00ll; c.word = 0; }
__device__ __forceinline__ u8 gcGet(const u8 *G, GCache &c, i64 pos) {
    i64 b = pos & ~7ll;
    if (b != c.base) { c.word = *GLOBAL(u64, G + b); c.base = b; }
    return (u8)(c.word >> ((u32)(pos & 7) * 8));
}

#define SJH_START_BITS 40u
#define SJ_INFO(motif, strand, shL, shR) ((u32)(motif) | ((u32)(strand) << 3) | ((u32)(shL) << 8) | ((u32)(shR) << 16))
#define SJ_INFO_MOTIF(i) ((i) & 7u)
#define SJ_INFO_STRAND(i) (((i) >> 3) & 3u)
#define SJ_INFO_SHL(i) (((i) >> 8) & 255u)
#define SJ_INFO_SHR(i) (((i) >> 16) & 255u)
__device__ __forceinline__ u32 sjdbHashSlot(u64 start, u32 mask) { return (u32)((start * 0x9E3779B97F4A7C15ull) >> 40) & mask; }
// one lane: index of the junction (x, y) or -1
__device__ __forceinline__ int sjdbHashFind(const u64 *tab_, u32 mask, u64 x, u64 y) {
    const __attribute__((address_space(1))) u64 *tab = GLOBAL(u64, tab_);
    u32 h = sjdbHashSlot(x, mask);
    for (u32 n = 0; n <= mask; n++, h = (h + 1u) & mask) {
        const u64 s = tab[2u * h];
        if (s == 0) return -1;
        if ((s & ((1ull << SJH_START_BITS) - 1ull)) == x && (tab[2u * h + 1u] & ((1ull << SJH_START_BITS) - 1ull)) == y) return (int)(s >> SJH_START_BITS) - 1;
    }
    return -1;
}

// LOCKSTEP(): marks a place where the lanes of a wavefront exchange data through memory and rely on executing in lock step (all lanes
// have done what precedes before any lane does what follows; LDS and L1 serve one wavefront in order).  Nothing on the device.  The
// wavefront emulator of the CPU tests (oracle/wave_emul/emu.h) runs the lanes one after the other and makes it a rendezvous.
#ifdef STARAMD_WAVE_EMUL
#define LOCKSTEP() emu_wave_sync()
#else
#define LOCKSTEP() ((void)0)
#endif

// ---- wave helpers (wave = 64 lanes on gfx950) ----
__device__ __forceinline__ u32 laneId() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u32 bcast32(u32 v, u32 srcLane) { return (u32)__shfl((int)v, (int)srcLane, 64); }
__device__ __forceinline__ u64 bcast64(u64 v, u32 srcLane) {
    u32 lo = bcast32((u32)v, srcLane), hi = bcast32((u32)(v >> 32), srcLane);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u32 first32(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ i32 firstI(i32 v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64 first64(u64 v) { return ((u64)first32((u32)(v >> 32)) << 32) | first32((u32)v); }
// The index of a wavefront inside its block, threadIdx.x >> 6, is the same for all 64 lanes -- which nothing tells the compiler.  Left as it is, the wavefront's LDS slice,
// every pointer into it, every value loaded through them and every branch on those values count as lane-dependent: k_stitch_win then keeps ~140 wave-uniform values in
// vector registers (168 VGPRs + 82 spilled instead of 106 and none) and compiles 325 wave-uniform branches to exec-mask sequences.  -DWAVE_INDEX_PLAIN restores that
// (the kernels of round 4), for A/B runs.
#ifdef WAVE_INDEX_PLAIN
#define WAVE_INDEX(x) (x)
#else
#define WAVE_INDEX(x) first32(x)
#endif
// number of set bits of a 64-lane ballot mask in the lanes strictly below / up to and including this lane
__device__ __forceinline__ u32 cntBelow(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }
__device__ __forceinline__ u32 cntUpTo(u64 m, u32 lane) { return cntBelow(m) + (u32)((m >> lane) & 1ull); }
// wave-wide max / sum through the DPP network (row shifts + row broadcasts, 6 VALU ops, no LDS round trips);
// result returned in every lane (read from lane 63 as a scalar)
__device__ __forceinline__ u32 waveMaxU32(u32 v) {
    u32 t;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:1
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:2
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:4
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:8  -> lane 15 of each row = row max
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;   // row_bcast:15 into rows 1,3
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;   // row_bcast:31 into rows 2,3
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ u32 waveSumU32(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);                      // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);                      // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);                      // row_shr:4 (banks 1-3)
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);                      // row_shr:8 (banks 2-3) -> lane 15 = row sum
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);                      // row_bcast:15
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);                      // row_bcast:31
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
// inclusive prefix sum over the lanes (the same six DPP steps; lane i gets v[0] + ... + v[i])
__device__ __forceinline__ u32 waveScanInclU32(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);                      // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);                      // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);                      // row_shr:4 (banks 1-3)
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);                      // row_shr:8 (banks 2-3) -> a scan inside every row of 16
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);                      // row_bcast:15: rows 1 and 3 add the total of the row before
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);                      // row_bcast:31: rows 2 and 3 add the total of the first half
    return v;
}
// value of lane srcLane, srcLane wave-uniform
__device__ __forceinline__ u32 laneGet32(u32 v, u32 srcLane) { return (u32)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)srcLane)); }
__device__ __forceinline__ u64 laneGet64(u64 v, u32 l) { return ((u64)laneGet32((u32)(v >> 32), l) << 32) | laneGet32((u32)v, l); }
// declare a wave-uniform value to the compiler: every dword goes through v_readfirstlane, so what is computed from it
// lives in SGPRs and branches on it are scalar branches (s_cbranch_scc) instead of exec-mask sequences
template <class T> __device__ __forceinline__ T uni(const T &v) {
    static_assert(sizeof(T) % 4 == 0, "uni: dword multiple");
    T r; const u32 *s = (const u32 *)&v; u32 *d = (u32 *)&r;
#pragma unroll
    for (u32 i = 0; i < sizeof(T) / 4; i++) d[i] = (u32)__builtin_amdgcn_readfirstlane((int)s[i]);
    return r;
}
__device__ __forceinline__ u32 firstLane(u64 m) { return (u32)__ffsll((long long)m) - 1u; }   // m != 0
