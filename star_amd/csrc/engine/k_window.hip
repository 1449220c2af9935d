// k_window.hip -- kernel 2: alignment windows and seed-to-window assignment, one wavefront per read.
//
// Replaces, per read, the first half of ReadAlign::stitchPieces (source/ReadAlign_stitchPieces.cpp:12-185):
//   pass A  anchors create / merge windows     createExtendWindowsWithAlign (ReadAlign_createExtendWindowsWithAlign.cpp:7-84)
//   flanks  every live window grows by winFlankNbins bins                   (ReadAlign_stitchPieces.cpp:96-118)
//   pass B  every seed locus is assigned to a window  assignAlignToWindow   (ReadAlign_assignAlignToWindow.cpp:6-130)
//   sjdb    loci inside the inserted junction sequences are split           (sjAlignSplit.cpp:3-15)
//
// The reference keeps a 2 x winBinN uint16 map (195 KB for human) that it memsets per read.  Here the map is never
// materialised: during pass A live windows are disjoint bin intervals, so "which window owns bin b" is an interval
// test over the read's (few) windows; after the flank extension the owner is the LAST flank writer, else the core
// owner -- exactly what the reference's write order into winBin produces.
//
// Mapping: one wavefront (64 lanes) per read.
//   * the SA interval of a seed is enumerated 64 entries at a time: lane i reads entry i of the chunk, so a
//     chunk is one or two coalesced 33-bit-packed loads per lane from consecutive words of SA;
//   * every lane converts its locus (strand flip, sjdb split) and looks its bin up in the window table, which
//     lives in LDS (all lanes read the same table row: broadcast, no bank conflicts);
//   * a ballot collects the lanes whose locus hit a window; the hits are replayed one by one in SA order
//     (the reference's order: replacement / eviction rules are order dependent), and each replay step is itself
//     wave-parallel: lane j holds row j of the window's seed list (<= 64 rows), the overlap test is one ballot,
//     the sorted insert is one shifted store;
//   * pass A replays the anchor loci in order the same way; the search for the nearest window left/right of a
//     new bin is a wave-wide max/min reduction over the table rows.
// Reads that need more windows / seed-list blocks than the compact per-wave work space are deferred to a second
// launch of the same kernel (big = 1) whose table lives in global memory with the reference's own limits.
#include "dev.h"

#define NOWIN 0xFFFFFFFFu
static_assert(sizeof(DWin) == 16, "a window record is two 8-byte words (emission stores it as such)");
static_assert(sizeof(DWA) == 24 && offsetof(DWA, iFrag) == 21, "a seed-list row is three 8-byte words, iFrag in byte 21 (emission)");
#ifndef WIN_EMIT_LANES
#define WIN_EMIT_LANES 1            // emission: windows of one or two seeds copied by a lane each (windowsBody)
#endif
#ifndef WIN_UNIQ
#define WIN_UNIQ 1                  // one-locus seeds converted and looked up together (windowsBody)
#endif
#ifdef STARAMD_PROFILE
#define WPROF_T0() u64 wprof_t0_ = __builtin_readcyclecounter()
#define WPROF_MARK(k) { u64 t1_ = __builtin_readcyclecounter(); wprof[k] += t1_ - wprof_t0_; wprof_t0_ = t1_; }
#else
#define WPROF_T0()
#define WPROF_MARK(k)
#endif

// The window table is a structure of arrays that lives in LDS (first and middle launch) or in global memory (last launch).
// The pointer type carries the address space: with plain pointers every table access compiled to a flat_load / flat_store
// (no ds_ instruction in the whole kernel, rocprofv3 SQ_INSTS_LDS ~ 0 against 1650 flat instructions per pair).
template <bool BIG> struct WPtr;
template <> struct WPtr<true> { typedef u32 *P; };
template <> struct WPtr<false> { typedef __attribute__((address_space(3))) u32 *P; };
template <bool BIG> struct WTab {   // per-wave window table
    typename WPtr<BIG>::P coreS, coreE, extS, extE, meta, blk, lrec, nwa;
};
// meta = chr << 2 | str << 1 | alive
#define WBITS 4096u                 // per-read hash bitmap of the bins covered by windows (quick reject of loci outside every window): bits in the first and last launch
template <bool BIG> struct WS {
    WTab<BIG> t; DWA *arena; typename WPtr<BIG>::P bitmap;
    u32 nW, capW, nBlocks, capBlocks, Lread;
    u32 hashMask;                   // bits of the bitmap - 1 (a power of two; the middle launch has 16x the bits: its reads cover thousands of bins,
                                    // a 4096-bit map is saturated and every locus of a 10000-fold seed would go through the serial owner lookup)
    bool overflow, tooMany, winLimit;
    bool ownMap;                    // the words at `bitmap` hold the owner map of this read (else the Bloom filter)
    u32 ownMask;                    // its slots - 1
};

// after writes to the seed lists (global memory, rows exchanged between lanes): wait for them
__device__ __forceinline__ void tabFence() { __threadfence_block(); }
// after writes to the table only: in LDS the DS instructions of one wavefront execute in order, nothing to wait for
template <bool BIG> __device__ __forceinline__ void rowFence() { if (BIG) __threadfence_block(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); }
__device__ __forceinline__ void bitOr(u32 *w, u32 v) { atomicOr(w, v); }
__device__ __forceinline__ void bitOr(__attribute__((address_space(3))) u32 *w, u32 v) { __hip_atomic_fetch_or(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// max of (key << 32 | payload) over the wave, keys distinct where non-zero: DPP max of the keys, then the payload of the
// lane that holds the maximum
__device__ __forceinline__ u64 waveMax64(u64 v) {
    u32 key = (u32)(v >> 32);
    u32 kmax = waveMaxU32(key);
    if (kmax == 0) return 0;
    u64 m = __ballot(key == kmax);
    return ((u64)kmax << 32) | laneGet32((u32)v, firstLane(m));
}
__device__ __forceinline__ u32 waveMin32(u32 v) { return ~waveMaxU32(~v); }

// seed `src` (wave-uniform) of the rows the lanes preloaded (lane i = row i of the read's seed table)
__device__ __forceinline__ DSeed seedOfLane(const DSeed &mine, u32 src) {
    DSeed r; const u32 *sw = (const u32 *)&mine; u32 *d = (u32 *)&r;
#pragma unroll
    for (u32 i = 0; i < 5; i++) d[i] = laneGet32(sw[i], src);
    d[5] = 0;
    return r;
}

// ReadAlign_createExtendWindowsWithAlign.cpp:7-84 ; all arguments wave-uniform; returns 1 on TOO_MANY_WINDOWS / overflow
// aChr = chrBin[aBin >> winBinChrNbits], looked up by the lane that enumerated the locus.  Every core bin of a window lies on the chromosome the window was
// created on (a window only grows by bins of its own chromosome, :28,:47,:66), so "is the neighbour on my chromosome" is a comparison with the neighbour's
// table row: the serial replay of the anchor loci makes no global-memory access at all.
template <bool BIG> __device__ static int createExtendWindowsWithAlign(const DevIndex &X, WS<BIG> &s, u64 a1, u32 aStr, u32 aChr, u32 lane) {
    const staramd_params &P = X.P;
    u32 aBin = (u32)(a1 >> P.winBinNbits);
    u32 lo = aBin > P.winAnchorDistNbins ? aBin - P.winAnchorDistNbins : 0;
    u64 hi = min((u64)aBin + P.winAnchorDistNbins + 1, P.winBinN);     // exclusive
    bool own = false;
    u64 candL = 0;                       // (coreE+1) << 32 | index   ; 0 = none
    u64 candR = 0;                       // (~coreS) << 32 | index    ; 0 = none (max of ~coreS = min coreS)
    for (u32 j = lane; j < s.nW; j += 64) {
        u32 m = s.t.meta[j];
        if (!(m & 1u) || ((m >> 1) & 1u) != aStr) continue;
        u32 cs = s.t.coreS[j], ce = s.t.coreE[j];
        if (aBin >= cs && aBin <= ce) own = true;
        if (aBin > 0 && ce < aBin && ce >= lo) { u64 v = ((u64)(ce + 1) << 32) | j; if (v > candL) candL = v; }
        if (cs > aBin && (u64)cs < hi) { u64 v = ((u64)(~cs) << 32) | j; if (v > candR) candR = v; }
    }
    if (__any(own)) return 0;
    candL = waveMax64(candL); candR = waveMax64(candR);
    u32 iWinL = NOWIN, iWinR = NOWIN;
    if (candL) { const u32 j = (u32)candL; if ((s.t.meta[j] >> 2) == aChr) iWinL = j; }
    if (candR) { const u32 j = (u32)candR; if ((s.t.meta[j] >> 2) == aChr) iWinR = j; }
    if (iWinL == NOWIN && iWinR == NOWIN) {
        u32 iWin = s.nW;
        if (iWin >= s.capW) { s.overflow = true; return 1; }
        if (lane == 0) {
            s.t.meta[iWin] = (aChr << 2) | (aStr << 1) | 1u;
            s.t.coreS[iWin] = aBin; s.t.coreE[iWin] = aBin; s.t.extS[iWin] = aBin; s.t.extE[iWin] = aBin;
            s.t.blk[iWin] = NOWIN; s.t.lrec[iWin] = 0; s.t.nwa[iWin] = 0;
        }
        s.nW++;
        rowFence<BIG>();
        if (s.nW >= P.alignWindowsPerReadNmax) { s.nW = P.alignWindowsPerReadNmax - 1; s.winLimit = true; return 1; }
    } else {
        u32 iWin = iWinL != NOWIN ? iWinL : iWinR;                            // left window overwrites right (:57)
        u32 binLeft = iWinL != NOWIN ? s.t.coreS[iWinL] : aBin;
        u32 binRight = iWinR != NOWIN ? s.t.coreE[iWinR] : aBin;
        if (lane == 0) {
            if (iWinL != NOWIN && iWinR != NOWIN) s.t.meta[iWinR] &= ~1u;        // kill right window (:77-80)
            s.t.coreS[iWin] = binLeft; s.t.coreE[iWin] = binRight;
        }
        rowFence<BIG>();
    }
    return 0;
}

// ReadAlign_assignAlignToWindow.cpp:6-130 ; all arguments wave-uniform; lane j holds row j of the window's list
template <bool BIG> __device__ static void assignAlignToWindow(const DevIndex &X, WS<BIG> &s, u32 iW, u64 a1, u32 aLength, u32 aNrep, u32 aFrag, u32 aRstart, bool aAnchor, i32 sjA, u32 lane) {
    const staramd_params &P = X.P;
    u32 n = s.t.nwa[iW]; u32 lrec = s.t.lrec[iW];
    if (!aAnchor && aLength < lrec) return;
    u32 b = s.t.blk[iW];
    if (b == NOWIN) {
        if (s.nBlocks >= s.capBlocks) { s.overflow = true; return; }
        b = s.nBlocks++;
        LOCKSTEP();                               // every lane has read blk[iW] (and counted the block) before lane 0 fills it in
        if (lane == 0) s.t.blk[iW] = b;
    }
    DWA *A = s.arena + (u64)b * WA_MAX;
    DWA nw; nw.gStart = a1; nw.nrep = aNrep; nw.L = (u16)aLength; nw.rStart = (u16)aRstart; nw.sjA = sjA; nw.anchor = aAnchor ? 1 : 0; nw.iFrag = (u8)aFrag; nw.pad[0] = nw.pad[1] = 0;
    bool have = lane < n;
    DWA e; e.gStart = 0; e.nrep = 0; e.L = 0; e.rStart = 0; e.sjA = 0; e.anchor = 0; e.iFrag = 0; e.pad[0] = e.pad[1] = 0;
    if (have) e = A[lane];
    {
        bool ov = have && aFrag == e.iFrag && e.sjA == sjA && a1 + e.rStart == e.gStart + aRstart
                  && ((aRstart >= e.rStart && aRstart < (u32)e.rStart + e.L) || (aRstart + aLength >= e.rStart && aRstart + aLength < (u32)e.rStart + e.L));
        u64 m = __ballot(ov);
        if (m) {
            u32 iA = (u32)__ffsll((long long)m) - 1;
            u32 Lold = laneGet32(e.L, iA);
            if (aLength > Lold) {
                u64 m2 = __ballot(have && lane != iA && aRstart < e.rStart);
                u32 iA0 = m2 ? (u32)__ffsll((long long)m2) - 1 : n;
                if (iA0 > iA) --iA0;
                if (iA0 < iA) { if (lane >= iA0 && lane < iA) A[lane + 1] = e; }
                else if (iA0 > iA) { if (lane > iA && lane <= iA0) A[lane - 1] = e; }
                if (lane == 0) A[iA0] = nw;
                tabFence();
            }
            return;
        }
    }
    if (n == P.seedPerWindowNmax) {
        lrec = waveMin32((have && e.anchor != 1) ? (u32)e.L : s.Lread + 1);
        if (lane == 0) s.t.lrec[iW] = lrec;
        if (lrec == s.Lread + 1) { s.tooMany = true; tabFence(); return; }
        if (!aAnchor && aLength < lrec) { tabFence(); return; }
        bool keep = have && (e.anchor == 1 || e.L > lrec);
        u64 km = __ballot(keep);
        u32 pos = (u32)__popcll(km & ((1ull << lane) - 1ull));
        if (keep) A[pos] = e;
        n = (u32)__popcll(km);
        if (lane == 0) s.t.nwa[iW] = n;
        tabFence();
        have = lane < n;
        if (have) e = A[lane];
    }
    if (aAnchor || aLength > lrec) {
        u64 m3 = __ballot(have && aRstart < e.rStart);
        u32 iA = m3 ? (u32)__ffsll((long long)m3) - 1 : n;
        if (have && lane >= iA) A[lane + 1] = e;
        if (lane == 0) { A[iA] = nw; s.t.nwa[iW] = n + 1; }
        tabFence();
    }
}

// sjAlignSplit.cpp:3-15
__device__ __forceinline__ bool sjAlignSplit(const DevIndex &X, u64 a1, u32 aLength, u64 &a1D, u32 &aLengthD, u64 &a1A, u32 &aLengthA, u32 &isj) {
    const u64 off = a1 - X.sjGstart;
    u64 q, sj1;
    if (off < 0x100000000ull) { const u32 q32 = (u32)off / X.sjdbLength; q = q32; sj1 = (u32)off - q32 * X.sjdbLength; }   // 32-bit divide: the inserted region is small
    else { q = off / X.sjdbLength; sj1 = off - q * X.sjdbLength; }
    if (sj1 < X.sjdbOverhang && sj1 + aLength > X.sjdbOverhang) {
        isj = (u32)q;
        aLengthD = (u32)(X.sjdbOverhang - sj1); aLengthA = aLength - aLengthD;
        a1D = GLOBAL(u64, X.sjDstart)[isj] + sj1; a1A = GLOBAL(u64, X.sjAstart)[isj];
        return true;
    }
    return false;
}

// the per-read filter of "bins covered by some window" is a Bloom filter with two hash functions: ~180 covered bins in 4096 bits pass 4 % of the foreign
// loci with one function, 0.7 % with two -- every false positive is a serial owner look-up, and a repeat seed enumerates hundreds of foreign loci
__device__ __forceinline__ u32 binHash(u32 str, u32 bin, u32 mask) { return (bin * 2u + str) & mask; }
__device__ __forceinline__ u32 binHash2(u32 str, u32 bin, u32 mask) { return (((bin * 2u + str) * 0x9E3779B1u) >> 14) & mask; }
template <class BP> __device__ __forceinline__ bool binCovered(BP bitmap, u32 str, u32 bin, u32 mask) {
    const u32 h1 = binHash(str, bin, mask), h2 = binHash2(str, bin, mask);
    return ((bitmap[h1 >> 5] >> (h1 & 31u)) & (bitmap[h2 >> 5] >> (h2 & 31u)) & 1u) != 0;
}

// ---- owner map: (strand, bin) -> window, an open-addressing hash table in the LDS words that otherwise hold the Bloom filter --------------------------
// The reference's winBin array answers "which window owns the bin of this locus" with one load; the interval-owner form above answers it with a scan of
// the window table (ownerWave), one locus at a time -- ~60 serial look-ups per read pair, and for a read in a repeat family with a thousand windows and
// 10^5 loci the whole of its time.  When the bins the windows of a read cover fit into the table (almost always: ~180 bins), they are entered once after
// the flank extension -- word = (key + 1) << 11 | flank << 10 | window, where an atomic max per key resolves exactly what the reference's write order
// into winBin does (ReadAlign_stitchPieces.cpp:96-118: a flank writer beats the core owner, the last flank writer beats the earlier ones) -- and every
// lane looks its own locus up: one or two LDS probes, 64 loci at once.
#define OWN_BITS 10u
template <class BP> __device__ __forceinline__ void ownInsert(BP tab, u32 mask, u32 key, u32 val) {
    const u32 word = ((key + 1u) << (OWN_BITS + 1u)) | val;
    u32 h = (key * 0x9E3779B1u >> 12) & mask;
    for (;;) {
        u32 expected = 0;
        if (__hip_atomic_compare_exchange_strong(&tab[h], &expected, word, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
        if ((expected >> (OWN_BITS + 1u)) == key + 1u) { __hip_atomic_fetch_max(&tab[h], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return; }
        h = (h + 1u) & mask;
    }
}
template <class BP> __device__ __forceinline__ u32 ownLookup(BP tab, u32 mask, u32 key) {
    u32 h = (key * 0x9E3779B1u >> 12) & mask;
    for (;;) {
        const u32 v = tab[h];
        if (v == 0) return NOWIN;
        if ((v >> (OWN_BITS + 1u)) == key + 1u) return v & ((1u << OWN_BITS) - 1u);
        h = (h + 1u) & mask;
    }
}

// pass-B owner of a bin, wave-parallel (lane j tests window j): last flank writer wins, else the core owner
// (ReadAlign_stitchPieces.cpp:96-118 write order)
template <bool BIG> __device__ static u32 ownerWave(const WS<BIG> &s, u32 str, u32 bin, u32 lane) {
    u32 core = NOWIN, flank = NOWIN;
    for (u32 j0 = 0; j0 < s.nW; j0 += 64) {
        u32 j = j0 + lane; bool inExt = false, inCore = false;
        if (j < s.nW) {
            u32 m = s.t.meta[j];
            if ((m & 1u) && ((m >> 1) & 1u) == str && bin >= s.t.extS[j] && bin <= s.t.extE[j]) { inExt = true; inCore = bin >= s.t.coreS[j] && bin <= s.t.coreE[j]; }
        }
        u64 mc = __ballot(inExt && inCore), mf = __ballot(inExt && !inCore);
        if (mc) core = j0 + 63u - (u32)__clzll((long long)mc);
        if (mf) flank = j0 + 63u - (u32)__clzll((long long)mf);
    }
    return flank != NOWIN ? flank : core;
}

// per-wave work space in global memory: [table rows + bitmap (big pass only)] [seed-list blocks]
__host__ __device__ inline u64 winWaveBytes(u32 capW, u32 capBlocks, u32 big) {
    u64 b = (u64)capBlocks * WA_MAX * sizeof(DWA);
    if (big) b += (u64)capW * 8 * sizeof(u32) + WBITS / 8;
    return (b + 255) & ~255ull;
}



extern __shared__ u32 ldsTab[];     // LDS launches: wavesPerBlock * (capW * 8 + hashBits / 32) words

// mode 0: every read, table in LDS (capW rows); reads that outgrow it go to list ovfWin
// mode 2: the reads of ovfWin, table still in LDS but with more rows (blocks of one wavefront); reads that outgrow that go to list ovfWin2
// mode 1: the reads of ovfWin2 (of ovfWin when no mode-2 launch ran: useMid = 0), table in global memory with the reference's own limits (BIG)
template <bool BIG> __device__ __forceinline__ void windowsBody(const DevIndex *__restrict__ Xp, DevBatch B, u8 *scratch, u32 capW, u32 capBlocks, u32 mode, u32 lightEst, u32 useMid, u32 hashBits) {
    const u32 big = BIG ? 1u : 0u;
    const bool ownMapEnable = (useMid & 2u) != 0; useMid &= 1u;      // (bit 1 of the argument: owner map on, STARAMD_WIN_OWNER_MAP)
    const DevIndex &X = *Xp;
    const staramd_params &P = X.P;
    u32 lane = threadIdx.x & 63u, waveInBlock = WAVE_INDEX(threadIdx.x >> 6);
    u32 wavesPerBlock = blockDim.x >> 6;
    u32 wave = blockIdx.x * wavesPerBlock + waveInBlock;
    WS<BIG> s;
    u8 *mine = scratch + (u64)wave * winWaveBytes(capW, capBlocks, big);
    typename WPtr<BIG>::P tab;
    if constexpr (BIG) tab = (u32 *)(mine + (u64)capBlocks * WA_MAX * sizeof(DWA));
    else tab = (typename WPtr<false>::P)ldsTab + waveInBlock * (capW * 8 + hashBits / 32);
    s.hashMask = hashBits - 1u;
    s.bitmap = tab + capW * 8;
    s.t.coreS = tab; s.t.coreE = tab + capW; s.t.extS = tab + 2 * capW; s.t.extE = tab + 3 * capW;
    s.t.meta = tab + 4 * capW; s.t.blk = tab + 5 * capW; s.t.lrec = tab + 6 * capW; s.t.nwa = tab + 7 * capW;
    s.arena = (DWA *)mine; s.capW = capW; s.capBlocks = capBlocks;
    u64 nSAenum = 0, nWindows = 0, nWAtot = 0; u32 nOvf = 0;
    u32 nOwnerLookups = 0, nOwnerMisses = 0, nAnchorLoci = 0, nAnchorReplayed = 0;
#ifdef STARAMD_PROFILE
    u64 wprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const u32 *inList = mode == 0u ? nullptr : (mode == 2u || !useMid) ? B.ovfWin : B.ovfWin2;
    const u32 nItems = mode == 0u ? B.nReads : (mode == 2u || !useMid) ? B.cursors[CUR_OVF_WIN] : B.cursors[CUR_OVF_WIN2];
    const u32 ticketSlot = mode == 0u ? CUR_TICKET_WIN : mode == 2u ? CUR_TICKET_WIN2 : CUR_TICKET_WIN3;
    for (;;) {
        u32 it = 0;
        if (lane == 0) it = atomicAdd(&B.cursors[ticketSlot], 1u);
        it = first32(it);
        if (it >= nItems) break;
        u32 ir = inList ? inList[it] : it;
        // (of the read's record only these three words are kept -- as scalars; the words the kernel changes are stored one by one at the end: the record itself would
        // sit in 16 vector registers from here to there)
        struct { u32 nSeeds, status; } rd;
        rd.nSeeds = first32(B.reads[ir].nSeeds);
        if (rd.nSeeds == 0) continue;
        rd.status = first32(B.reads[ir].status);
        const DSeed *PC = B.seedPool + first32(B.reads[ir].seedOffset);
        s.nW = 0; s.nBlocks = 0; s.tooMany = false; s.winLimit = false; s.overflow = false;
        s.Lread = (u32)(B.readOffset[ir + 1] - B.readOffset[ir]);
        WPROF_T0();
        // The seed table of the read and the first suffix-array entry of every seed are fetched up front, lane i = seed i: two round trips for the whole
        // read.  Most seeds are unique (one locus): walking the table seed by seed, as both passes do, would otherwise wait for a table row and then for a
        // random suffix-array word, ~34 dependent round trips per read pair.
        const u32 nPre = min(rd.nSeeds, 64u);
        DSeed mySeed; { u32 *z = (u32 *)&mySeed; z[0] = z[1] = z[2] = z[3] = z[4] = z[5] = 0; }
        u64 myA1 = 0;
        if (lane < nPre) { mySeed = PC[lane]; myA1 = packedGet(X.SA, mySeed.saStart, X.saBits, X.saMask); }
        // Seeds of ONE locus outside the inserted junction sequences (12 of the 17 seeds of a pair) are converted here once, lane i = seed i -- strand flip, read
        // start, chromosome of the bin -- instead of one seed per trip of the loops of pass A and pass B with one lane at work: myA1 becomes the converted locus,
        // uInfo = 1 << 31 | strand << 30 | x << 16 | rStart as pass B wants it, x = the chromosome in pass A, the window + 1 in pass B (14 bits: genomes of up to
        // 16 383 sequences; others keep the loops).  Both passes then take such a seed with two lane reads.  (WIN_UNIQ=0: A/B builds without.)
        u32 uInfo = 0;
#if WIN_UNIQ
        if (lane < nPre && mySeed.nrep == 1u && X.nChrReal < 0x3FFFu) {
            u64 a1 = myA1; u32 aStr = (u32)(a1 >> X.strandBit); a1 &= X.strandMask;
            const u32 aL = mySeed.L; u32 aR = mySeed.rStart;
            if (mySeed.dir == 1 && aStr == 0) { aStr = 1; aR = s.Lread - (aL + aR); }
            else if (mySeed.dir == 0 && aStr == 1) { aR = s.Lread - (aL + aR); a1 = X.nGenome - (aL + a1); }
            else if (mySeed.dir == 1 && aStr == 1) { aStr = 0; a1 = X.nGenome - (aL + a1); }
            if (a1 < X.sjGstart) {
                const u32 chr = GLOBAL(u32, X.chrBin)[(u32)(a1 >> P.winBinNbits) >> P.winBinChrNbits];
                if (chr < 0x3FFFu) { myA1 = a1; uInfo = 0x80000000u | (aStr << 30) | (chr << 16) | (aR & 0xFFFFu); }
            }
        }
#endif
        // ---- pass A: anchors (ReadAlign_stitchPieces.cpp:41-93)
        for (u32 iP = 0; iP < rd.nSeeds && !s.overflow; iP++) {
#if WIN_UNIQ
            const u32 xInfo = laneGet32(uInfo, iP < nPre ? iP : 0u);
            if (iP < nPre && (xInfo & 0x80000000u)) {          // a seed converted above: its one locus goes straight into the replay
                if (1u > P.winAnchorMultimapNmax) continue;
                nSAenum++; nAnchorLoci++; nAnchorReplayed++;
                createExtendWindowsWithAlign(X, s, laneGet64(myA1, iP), (xInfo >> 30) & 1u, (xInfo >> 16) & 0x3FFFu, lane);
                continue;
            }
#endif
            const DSeed sd = iP < nPre ? seedOfLane(mySeed, iP) : PC[iP];
            const u64 preA1 = laneGet64(myA1, iP < nPre ? iP : 0u);
            if (sd.nrep > P.winAnchorMultimapNmax) continue;
            u32 aDir = sd.dir, aLength = sd.L;
            bool stop = false;
            for (u32 base = 0; base < sd.nrep && !stop; base += 64) {
                u32 cnt = min(64u, sd.nrep - base);
                // lane i: locus i of the chunk
                u64 a1 = 0, a1A = 0; u32 aStr = 0; u32 kind = 0;       // kind: bit 0 = the locus (the donor half of a split one), bit 1 = the acceptor half of a split locus
                u32 aChr = 0, aChrA = 0;
                if (lane < cnt) {
                    a1 = (sd.nrep == 1u && iP < nPre) ? preA1 : packedGet(X.SA, sd.saStart + base + lane, X.saBits, X.saMask);
                    aStr = (u32)(a1 >> X.strandBit); a1 &= X.strandMask;
                    if (aDir == 1 && aStr == 0) aStr = 1;
                    else if (aDir == 0 && aStr == 1) a1 = X.nGenome - (aLength + a1);
                    else if (aDir == 1 && aStr == 1) { aStr = 0; a1 = X.nGenome - (aLength + a1); }
                    kind = 1;
                    if (a1 >= X.sjGstart) {
                        u64 a1D; u32 lD, lA, isj;
                        if (sjAlignSplit(X, a1, aLength, a1D, lD, a1A, lA, isj)) { a1 = a1D; kind = 3; } else kind = 0;
                    }
                    if (kind) aChr = GLOBAL(u32, X.chrBin)[(u32)(a1 >> P.winBinNbits) >> P.winBinChrNbits];
                    if (kind & 2u) aChrA = GLOBAL(u32, X.chrBin)[(u32)(a1A >> P.winBinNbits) >> P.winBinChrNbits];
                }
                nSAenum += cnt; nAnchorLoci += cnt;
                WPROF_MARK(6);
                // A locus whose bin lies inside the core of a live window changes nothing (:12-17 returns at once), and a bin that is owned stays owned (cores
                // only grow; a window killed by a bridge lies inside the bridging one).  So the loci owned NOW are dropped from the replay, every lane testing its
                // own locus against the table rows (all lanes read the same row: a broadcast) -- typically all but the first few loci of a repeat family
                if (cnt >= 4u && cnt * 16u >= s.nW) {
                    const u32 bD = (u32)(a1 >> P.winBinNbits), bA = (u32)(a1A >> P.winBinNbits);
                    bool ownD = false, ownA = false;
                    for (u32 j = 0; j < s.nW; j++) {
                        const u32 m = s.t.meta[j];
                        if (!(m & 1u) || ((m >> 1) & 1u) != aStr) continue;
                        const u32 cs = s.t.coreS[j], ce = s.t.coreE[j];
                        ownD |= bD >= cs && bD <= ce; ownA |= bA >= cs && bA <= ce;
                    }
                    if (ownD) kind &= ~1u;
                    if (ownA) kind &= ~2u;
                }
                WPROF_MARK(7);
                for (u64 lm = __ballot(kind != 0); lm; lm &= lm - 1) {
                    const u32 l = firstLane(lm); nAnchorReplayed++;
                    const u32 k = laneGet32(kind, l); const u32 xs = laneGet32(aStr, l);
                    if (k & 1u) { if (createExtendWindowsWithAlign(X, s, laneGet64(a1, l), xs, laneGet32(aChr, l), lane)) { stop = true; break; } }
                    if (k & 2u) { if (createExtendWindowsWithAlign(X, s, laneGet64(a1A, l), xs, laneGet32(aChrA, l), lane)) { stop = true; break; } }
                }
                WPROF_MARK(0);
            }
        }
        WPROF_MARK(0);
        // ---- flanks (:96-118): one lane per window
        if (!s.overflow) {
            for (u32 j = lane; j < s.nW; j += 64) {
                u32 m = s.t.meta[j];
                s.t.nwa[j] = 0; s.t.lrec[j] = 0; s.t.blk[j] = NOWIN;
                if (!(m & 1u)) continue;
                u32 chr = m >> 2;
                u32 wb = s.t.coreS[j];
                for (u32 ii = 0; ii < P.winFlankNbins && wb > 0 && GLOBAL(u32, X.chrBin)[(wb - 1) >> P.winBinChrNbits] == chr; ii++) wb--;
                s.t.extS[j] = wb;
                wb = s.t.coreE[j];
                for (u32 ii = 0; ii < P.winFlankNbins && (u64)wb + 1 < P.winBinN && GLOBAL(u32, X.chrBin)[(wb + 1) >> P.winBinChrNbits] == chr; ii++) wb++;
                s.t.extE[j] = wb;
            }
            for (u32 k = lane; k < hashBits / 32; k += 64) s.bitmap[k] = 0;
            // bins covered by the windows of this read: into the owner map when they fill at most 5/8 of its slots (keys of 20 bits, windows of 10), else into the Bloom filter
            u32 nCov = 0;
            for (u32 j = lane; j < s.nW; j += 64) if (s.t.meta[j] & 1u) nCov += s.t.extE[j] - s.t.extS[j] + 1u;
            nCov = waveSumU32(min(nCov, 0xFFFFFFu));
            s.ownMask = hashBits / 32u - 1u;
            s.ownMap = !BIG && ownMapEnable && s.nW <= (1u << OWN_BITS) && P.winBinN < (1u << 19) && nCov * 8u <= (hashBits / 32u) * 5u;
            rowFence<BIG>();
            for (u32 j = lane; j < s.nW; j += 64) {
                u32 m = s.t.meta[j];
                if (!(m & 1u)) continue;
                const u32 str = (m >> 1) & 1u, cs = s.t.coreS[j], ce = s.t.coreE[j];
                for (u32 b = s.t.extS[j]; b <= s.t.extE[j]; b++) {
                    if (s.ownMap) { ownInsert(s.bitmap, s.ownMask, b * 2u + str, ((b < cs || b > ce) ? (1u << OWN_BITS) : 0u) | j); continue; }
                    const u32 h1 = binHash(str, b, s.hashMask), h2 = binHash2(str, b, s.hashMask);
                    bitOr(&s.bitmap[h1 >> 5], 1u << (h1 & 31u)); bitOr(&s.bitmap[h2 >> 5], 1u << (h2 & 31u));
                }
            }
            rowFence<BIG>();
        }
        nWindows += s.nW;
        WPROF_MARK(1);
        // ---- pass B: all seeds (:129-185)
#if WIN_UNIQ
        // the converted one-locus seeds look their windows up together (owner map only: the Bloom-filter form keeps them in the loop); window + 1 takes the chromosome's place
        const bool uniqB = s.ownMap && !s.overflow;
        if (uniqB && (uInfo & 0x80000000u)) {
            const u32 w = ownLookup(s.bitmap, s.ownMask, (u32)(myA1 >> P.winBinNbits) * 2u + ((uInfo >> 30) & 1u));
            uInfo = (uInfo & 0xC000FFFFu) | ((w == NOWIN ? 0u : w + 1u) << 16);
        }
#endif
        for (u32 iP = 0; iP < rd.nSeeds && !s.overflow && !s.tooMany; iP++) {
#if WIN_UNIQ
            const u32 xInfo = laneGet32(uInfo, iP < nPre ? iP : 0u);
            if (uniqB && iP < nPre && (xInfo & 0x80000000u)) {
                nSAenum++;
                const u32 w1 = (xInfo >> 16) & 0x3FFFu;
                if (w1 == 0u) continue;
                const u32 x3 = laneGet32(((const u32 *)&mySeed)[3], iP), x4 = laneGet32(((const u32 *)&mySeed)[4], iP);       // rStart | L << 16, dir | iFrag << 8
                const u32 uL = x3 >> 16; const bool uAnchor = 1u <= P.winAnchorMultimapNmax;
                if (!uAnchor && uL < s.t.lrec[w1 - 1u]) continue;
                assignAlignToWindow(X, s, w1 - 1u, laneGet64(myA1, iP), uL, 1u, (x4 >> 8) & 0xFFu, xInfo & 0xFFFFu, uAnchor, -1, lane);
                continue;
            }
#endif
            const DSeed sd = iP < nPre ? seedOfLane(mySeed, iP) : PC[iP];
            const u64 preA1 = laneGet64(myA1, iP < nPre ? iP : 0u);
            u32 aNrep = sd.nrep, aFrag = sd.iFrag, aLength = sd.L, aDir = sd.dir;
            bool aAnchor = aNrep <= P.winAnchorMultimapNmax;
            for (u32 base = 0; base < aNrep && !s.overflow && !s.tooMany; base += 64) {
                u32 cnt = min(64u, aNrep - base);
                u64 a1 = 0, a1A = 0; u32 aRstart = 0, lD = 0, lA = 0, isj = 0; u32 wD = NOWIN, wA = NOWIN; bool split = false;
                u32 binD = 0, binA = 0, lStr = 0; bool candD = false, candA = false;
                if (lane < cnt) {
                    a1 = (aNrep == 1u && iP < nPre) ? preA1 : packedGet(X.SA, sd.saStart + base + lane, X.saBits, X.saMask);
                    u32 aStr;
#if WIN_UNIQ
                    if (iP < nPre && (xInfo & 0x80000000u)) { aStr = (xInfo >> 30) & 1u; aRstart = xInfo & 0xFFFFu; }       // (converted above; here because the read has no owner map)
                    else
#endif
                    {
                        aStr = (u32)(a1 >> X.strandBit); a1 &= X.strandMask;
                        aRstart = sd.rStart;
                        if (aDir == 1 && aStr == 0) { aStr = 1; aRstart = s.Lread - (aLength + aRstart); }
                        else if (aDir == 0 && aStr == 1) { aRstart = s.Lread - (aLength + aRstart); a1 = X.nGenome - (aLength + a1); }
                        else if (aDir == 1 && aStr == 1) { aStr = 0; a1 = X.nGenome - (aLength + a1); }
                    }
                    if (a1 >= X.sjGstart) {
                        u64 a1D;
                        if (sjAlignSplit(X, a1, aLength, a1D, lD, a1A, lA, isj)) {
                            split = true; a1 = a1D;
                            binD = (u32)(a1D >> P.winBinNbits); binA = (u32)(a1A >> P.winBinNbits); lStr = aStr;
                            if (s.ownMap) { wD = ownLookup(s.bitmap, s.ownMask, binD * 2u + aStr); wA = ownLookup(s.bitmap, s.ownMask, binA * 2u + aStr); }
                            else { candD = binCovered(s.bitmap, aStr, binD, s.hashMask); candA = binCovered(s.bitmap, aStr, binA, s.hashMask); }
                        }
                    } else {
                        lD = aLength;
                        binD = (u32)(a1 >> P.winBinNbits); lStr = aStr;
                        if (s.ownMap) wD = ownLookup(s.bitmap, s.ownMask, binD * 2u + aStr);
                        else candD = binCovered(s.bitmap, aStr, binD, s.hashMask);
                    }
                }
                WPROF_MARK(2);
                if (!s.ownMap) {
                    // loci that pass the Bloom filter are looked up one by one, all lanes testing one window each
                    for (u64 cm = __ballot(candD); cm; cm &= cm - 1) { u32 l = firstLane(cm); u32 w = ownerWave(s, laneGet32(lStr, l), laneGet32(binD, l), lane); if (lane == l) wD = w; nOwnerLookups++; if (w == NOWIN) nOwnerMisses++; }
                    for (u64 cm = __ballot(candA); cm; cm &= cm - 1) { u32 l = firstLane(cm); u32 w = ownerWave(s, laneGet32(lStr, l), laneGet32(binA, l), lane); if (lane == l) wA = w; nOwnerLookups++; if (w == NOWIN) nOwnerMisses++; }
                }
                // a seed that is no anchor and shorter than what a full window has already turned away (lrec, assignAlignToWindow.cpp:11: it only ever rises) is
                // dropped here by its own lane instead of in the replay: repeat seeds with thousands of loci leave the replay after the windows have filled up
                if (!aAnchor) {
                    if (wD != NOWIN && lD < s.t.lrec[wD]) wD = NOWIN;
                    if (wA != NOWIN && lA < s.t.lrec[wA]) wA = NOWIN;
                }
                nSAenum += cnt;
                WPROF_MARK(5);
                u64 hm = __ballot(wD != NOWIN || wA != NOWIN);
                while (hm) {
                    u32 l = (u32)__ffsll((long long)hm) - 1; hm &= hm - 1;
                    u32 xwD = laneGet32(wD, l), xwA = laneGet32(wA, l);
                    u32 xsplit = laneGet32(split ? 1u : 0u, l);
                    u32 xr = laneGet32(aRstart, l), xlD = laneGet32(lD, l);
                    i32 xsj = xsplit ? (i32)laneGet32(isj, l) : -1;
                    if (xwD != NOWIN) assignAlignToWindow(X, s, xwD, laneGet64(a1, l), xlD, aNrep, aFrag, xr, aAnchor, xsj, lane);
                    if (xwA != NOWIN && !s.tooMany && !s.overflow) assignAlignToWindow(X, s, xwA, laneGet64(a1A, l), laneGet32(lA, l), aNrep, aFrag, xr + xlD, aAnchor, xsj, lane);
                    if (s.tooMany || s.overflow) break;
                }
                WPROF_MARK(3);
            }
        }
        if (s.winLimit) rd.status |= STARAMD_ST_WINDOWS_LIMIT;
        if (s.overflow) {
            if (lane == 0) {
                if (big) { rd.status |= STARAMD_ST_SCRATCH_OVERFLOW; atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_HARD); B.reads[ir].status = rd.status; }
                else if (mode == 2u) { u32 k = atomicAdd(&B.cursors[CUR_OVF_WIN2], 1u); B.ovfWin2[k] = ir; }
                else { u32 k = atomicAdd(&B.cursors[CUR_OVF_WIN], 1u); B.ovfWin[k] = ir; nOvf++; }
            }
            continue;
        }
        if (s.tooMany) { rd.status |= STARAMD_ST_TOO_MANY_ANCHORS | STARAMD_ST_NO_GOOD_WINDOW; if (lane == 0) B.reads[ir].status = rd.status; continue; }   // nW=0 (:76-80)
        // ---- emit windows that hold seeds, in window order
        u32 nOut = 0, nWA = 0; u32 est = 0, nMax = 0;
        for (u32 j = lane; j < s.nW; j += 64) { u32 n = s.t.nwa[j]; if (n > 0) { nOut++; nWA += n; est += 1u << min(n, 20u); nMax = max(nMax, n); } }
        for (int o = 32; o > 0; o >>= 1) { nOut += (u32)__shfl_xor((int)nOut, o, 64); nWA += (u32)__shfl_xor((int)nWA, o, 64); est += (u32)__shfl_xor((int)est, o, 64); nMax = max(nMax, (u32)__shfl_xor((int)nMax, o, 64)); }
        u32 rdWinOffset = 0, rdNWin = 0;           // (what k_seed_* left there: 0, 0 -- classifyRead)
        if (nOut > 0) {
            // stitch work items: a light read (its walks are bounded by est = sum over windows of 2^seeds) is ONE item -- its
            // windows are walked in order by one wavefront, so maxScoreMate is carried exactly and nothing has to be
            // re-decided; the windows of a heavy read are separate items (k_stitch_win / k_stitch_verify / k_stitch_replay)
            const bool light = est <= lightEst;
            const u32 nIt = light ? 1u : nOut;
            u32 wo = 0, ao = 0, io = 0;
            if (lane == 0) { wo = atomicAdd(&B.cursors[CUR_WIN], nOut); ao = atomicAdd(&B.cursors[CUR_WA], nWA); io = atomicAdd(&B.cursors[CUR_ITEM], nIt); }
            wo = first32(wo); ao = first32(ao); io = first32(io);
            if (wo + nOut > B.winCap || ao + nWA > B.waCap || io + nIt > B.winCap) { if (lane == 0) atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_WINPOOL); continue; }
            rdWinOffset = wo; rdNWin = nOut;
            const u32 ioRead = io;
            if (light && lane == 0) B.items[io] = 0x80000000u | ir;
#if WIN_EMIT_LANES
            // Windows in index order get consecutive places in the pools.  One window per trip of a loop means one dependent round trip per window (its rows come from
            // the seed-list arena in global memory) -- twenty per pair, most of them for a window of one or two seeds.  So: lane = window; places by a prefix sum over the
            // lanes; a window of one or two rows is copied by its own lane (all of them in ONE round trip), the few longer lists by the whole wavefront as before.
            for (u32 j0 = 0; j0 < s.nW; j0 += 64) {
                const u32 j = j0 + lane;
                const u32 n = j < s.nW ? s.t.nwa[j] : 0u;
                const u32 inclW = waveScanInclU32(n ? 1u : 0u), inclA = waveScanInclU32(n);
                const u32 myW = wo + inclW - (n ? 1u : 0u), myA = ao + inclA - n, myI = io + inclW - (n ? 1u : 0u);
                u32 m = 0, blk = 0;
                if (n) { m = s.t.meta[j]; blk = s.t.blk[j]; }
                if (n == 1u || n == 2u) {
                    const DWA *A = s.arena + (u64)blk * WA_MAX;
                    // (rows as three 8-byte words each: a struct copy of 24 bytes went through scratch memory)
                    const u64 *src = (const u64 *)A; u64 *dst = (u64 *)&B.waPool[myA];
                    const u64 a0 = src[0], a1 = src[1], a2 = src[2], b0 = src[3 * (n - 1u)], b1 = src[3 * (n - 1u) + 1u], b2 = src[3 * (n - 1u) + 2u];
                    dst[0] = a0; dst[1] = a1; dst[2] = a2;
                    if (n == 2u) { dst[3] = b0; dst[4] = b1; dst[5] = b2; }
                    const u32 f0 = (u32)(a2 >> 40) & 0xFFu, f1 = (u32)(b2 >> 40) & 0xFFu;          // DWA::iFrag (byte 21 of a row)
                    const u8 mates = (u8)(((f0 == 0 || f1 == 0) ? 1u : 0u) | ((f0 != 0 || f1 != 0) ? 2u : 0u));
                    { u64 *dw = (u64 *)&B.winPool[myW]; dw[0] = (u64)ir | ((u64)(m >> 2) << 32); dw[1] = (u64)myA | ((u64)(n | (((m >> 1) & 1u) << 16) | ((u32)mates << 24)) << 32); }      // = DWin {read, chr, waOffset, nWA, str, mates}
                    if (!light) { B.items[myI] = myW; B.itemClass[myI] = (u8)min(n + 1u, 31u); }
                }
                for (u64 lm = __ballot(n > 2u); lm; lm &= lm - 1) {                     // the longer lists: lane = row
                    const u32 l = firstLane(lm);
                    const u32 xn = laneGet32(n, l), xm = laneGet32(m, l), xW = laneGet32(myW, l), xA = laneGet32(myA, l), xI = laneGet32(myI, l);
                    const DWA *A = s.arena + (u64)laneGet32(blk, l) * WA_MAX;
                    u8 fr = 0;
                    if (lane < xn) { const DWA row = A[lane]; fr = row.iFrag; B.waPool[xA + lane] = row; }
                    const u8 mates = (u8)((__ballot(lane < xn && fr == 0) ? 1u : 0u) | (__ballot(lane < xn && fr != 0) ? 2u : 0u));
                    if (lane == 0) {
                        DWin d; d.read = ir; d.chr = xm >> 2; d.waOffset = xA; d.nWA = (u16)xn; d.str = (u8)((xm >> 1) & 1u); d.mates = mates; B.winPool[xW] = d;
                        if (!light) { B.items[xI] = xW; B.itemClass[xI] = (u8)min(xn + 1u, 31u); }
                    }
                }
                const u32 totW = laneGet32(inclW, 63u), totA = laneGet32(inclA, 63u);
                wo += totW; ao += totA; io += light ? 0u : totW;
            }
#else
            for (u32 j = 0; j < s.nW; j++) {
                u32 n = s.t.nwa[j];
                if (n == 0) continue;
                u32 m = s.t.meta[j];
                const DWA *A = s.arena + (u64)s.t.blk[j] * WA_MAX;
                u8 fr = 0;
                if (lane < n) { const DWA row = A[lane]; fr = row.iFrag; B.waPool[ao + lane] = row; }
                const u8 mates = (u8)((__ballot(lane < n && fr == 0) ? 1u : 0u) | (__ballot(lane < n && fr != 0) ? 2u : 0u));
                if (lane == 0) {
                    DWin d; d.read = ir; d.chr = m >> 2; d.waOffset = ao; d.nWA = (u16)n; d.str = (u8)((m >> 1) & 1u); d.mates = mates; B.winPool[wo] = d;
                    if (!light) { B.items[io] = wo; B.itemClass[io] = (u8)min(n + 1u, 31u); io++; }
                }
                wo++; ao += n;
            }
#endif
            if (light && lane == 0) {
                // cost class of a light read = bits of its walk-size estimate (k_order_* sorts by it, k_stitch_lane takes the classes up to its cap)
                const u32 cls = 32u - (u32)__clz((int)est);
                B.itemClass[ioRead] = (u8)cls;
            }
            nWAtot += nWA;
        }
        if (lane == 0) { DRead *out = &B.reads[ir]; out->status = rd.status; out->winOffset = rdWinOffset; out->nWin = rdNWin; out->wtOffset = nMax; }
        WPROF_MARK(4);
    }
    if (lane == 0) {
#ifdef STARAMD_PROFILE
        if (mode == 0u) { for (int k = 0; k < 5; k++) atomicAdd((unsigned long long *)&B.counters[DC_prof8 + k], (unsigned long long)wprof[k]);
                          for (int k = 5; k < 8; k++) atomicAdd((unsigned long long *)&B.counters[DC_wprof5 + k - 5], (unsigned long long)wprof[k]); }
#endif
        atomicAdd((unsigned long long *)&B.counters[DC_nSAenum], (unsigned long long)nSAenum);
        atomicAdd((unsigned long long *)&B.counters[DC_nWindows], (unsigned long long)nWindows);
        atomicAdd((unsigned long long *)&B.counters[DC_nWA], (unsigned long long)nWAtot);
        if (nOvf) atomicAdd((unsigned long long *)&B.counters[DC_nOvfWin], (unsigned long long)nOvf);
        atomicAdd((unsigned long long *)&B.counters[DC_nOwnerLookups], (unsigned long long)nOwnerLookups); atomicAdd((unsigned long long *)&B.counters[DC_nOwnerMisses], (unsigned long long)nOwnerMisses);
        atomicAdd((unsigned long long *)&B.counters[DC_nAnchorLoci], (unsigned long long)nAnchorLoci); atomicAdd((unsigned long long *)&B.counters[DC_nAnchorReplayed], (unsigned long long)nAnchorReplayed);
    }
}

#ifndef WIN_WAVES
#define WIN_WAVES 6         // minimum waves per SIMD the register allocation of the LDS launches is held to.  The kernel waits on dependent memory round trips (SA entry ->
                            // table rows -> seed list): more resident wavefronts is what pays, 80 VGPRs with 32 spilled registers included (same box, 3.1 Gb, ms per
                            // 400 k pairs, 128-row table: 4 waves 23.6, 5 (192 rows) 22.3, 6 -> 20.9, 7 -> 21.2, 8 -> 22.8; profiles/r04_ab_session*.txt)
#endif
extern "C" __global__ void __launch_bounds__(256, WIN_WAVES) k_windows(const DevIndex *__restrict__ Xp, DevBatch B, u8 *scratch, u32 capW, u32 capBlocks, u32 mode, u32 lightEst, u32 useMid, u32 hashBits) {
    windowsBody<false>(Xp, B, scratch, capW, capBlocks, mode, lightEst, useMid, hashBits);
}
extern "C" __global__ void __launch_bounds__(256, 4) k_windows_big(const DevIndex *__restrict__ Xp, DevBatch B, u8 *scratch, u32 capW, u32 capBlocks, u32 lightEst, u32 useMid) {
    windowsBody<true>(Xp, B, scratch, capW, capBlocks, 1u, lightEst, useMid, WBITS);
}

// ---- stitch order: work items sorted by class ~ log2(estimated walk size), largest first (counting sort
// over 32 classes), then dealt round-robin over groups of 64 consecutive tickets: a wavefront takes 64 consecutive tickets
// per round, so every wavefront gets one window of each stratum instead of 64 heavy (mutually divergent) ones.
extern "C" __global__ void __launch_bounds__(256) k_order_hist(DevBatch B) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch
    __shared__ u32 h[32];
    if (threadIdx.x < 32) h[threadIdx.x] = 0;
    __syncthreads();
    u32 n = B.cursors[CUR_ITEM]; u32 slots = ((n + 63u) / 64u) * 64u;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += gridDim.x * blockDim.x) {
        B.order[i] = 0xFFFFFFFFu;
        if (i < n) atomicAdd(&h[B.itemClass[i] & 31u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 32 && h[threadIdx.x]) atomicAdd(&B.costHist[threadIdx.x], h[threadIdx.x]);
}
extern "C" __global__ void k_order_offsets(DevBatch B) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch      // 1 thread: class offsets, heaviest class first
    u32 off = 0;
    for (int c = 31; c >= 0; c--) { u32 n = B.costHist[c]; B.costHist[32 + c] = off; off += n; }
}
extern "C" __global__ void __launch_bounds__(256) k_order_scatter(DevBatch B) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch
    // per block: histogram of its chunk in LDS, ONE global reservation per class, ranks inside the block from LDS atomics
    __shared__ u32 h[32], base[32];
    u32 n = B.cursors[CUR_ITEM]; u32 G = (n + 63u) / 64u;
    u32 perBlock = (n + gridDim.x - 1) / gridDim.x;
    u32 lo = blockIdx.x * perBlock, hi = min(n, lo + perBlock);
    if (threadIdx.x < 32) h[threadIdx.x] = 0;
    __syncthreads();
    for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&h[B.itemClass[i] & 31u], 1u);
    __syncthreads();
    if (threadIdx.x < 32) { u32 c = h[threadIdx.x]; base[threadIdx.x] = c ? atomicAdd(&B.costHist[32 + threadIdx.x], c) : 0; h[threadIdx.x] = 0; }
    __syncthreads();
    for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        u32 cls = B.itemClass[i] & 31u;
        u32 pos = base[cls] + atomicAdd(&h[cls], 1u);
        B.order[(pos % G) * 64u + pos / G] = B.items[i];
    }
}
