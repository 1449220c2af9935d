// k_window.hip -- kernel 2: alignment windows and seed-to-window assignment.
//
// Replaces, per read, the first half of ReadAlign::stitchPieces (source/ReadAlign_stitchPieces.cpp:12-185):
//   pass A  anchors create / merge windows     createExtendWindowsWithAlign (ReadAlign_createExtendWindowsWithAlign.cpp:7-84)
//   flanks  every live window grows by winFlankNbins bins                   (ReadAlign_stitchPieces.cpp:96-118)
//   pass B  every seed locus is assigned to a window  assignAlignToWindow   (ReadAlign_assignAlignToWindow.cpp:6-130)
//   sjdb    loci inside the inserted junction sequences are split           (sjAlignSplit.cpp:3-15)
//
// The reference keeps a 2 x winBinN uint16 map (195 KB for human) that it memsets per read.  Here
// the map is never materialised: during pass A live windows are disjoint bin intervals, so
// "which window owns bin b" is an interval test over the read's (few) windows; after the flank
// extension the owner is the LAST flank writer, else the core owner -- exactly what the
// reference's write order into winBin produces.
// Mapping: one lane = one read; the SA intervals of a seed are contiguous so neighbouring
// iterations of a lane hit the same 64-byte line of the packed SA.
#include "dev.h"


struct WinState {
    const DevIndex *X;
    WScr *W; u32 nW; u32 capW;
    DWA *arena; u32 nBlocks; u32 capBlocks; u32 blockSize;
    bool tooManyAnchors, windowsLimit, overflow;
    u32 Lread;
};

#define NOWIN 0xFFFFFFFFu

// pass-A owner: live windows are disjoint intervals [coreS,coreE]
__device__ static u32 ownerA(const WinState &s, u32 str, u32 bin) {
    for (u32 i = 0; i < s.nW; i++) { const WScr &w = s.W[i]; if (w.alive && w.str == str && bin >= w.coreS && bin <= w.coreE) return i; }
    return NOWIN;
}
// pass-B owner: last flank writer wins, else the core owner (ReadAlign_stitchPieces.cpp:96-118 write order)
__device__ static u32 ownerB(const WinState &s, u32 str, u32 bin) {
    u32 core = NOWIN, flank = NOWIN;
    for (u32 i = 0; i < s.nW; i++) {
        const WScr &w = s.W[i];
        if (!w.alive || w.str != str || bin < w.extS || bin > w.extE) continue;
        if (bin >= w.coreS && bin <= w.coreE) core = i; else flank = i;
    }
    return flank != NOWIN ? flank : core;
}

// ReadAlign_createExtendWindowsWithAlign.cpp:7-84 ; returns 1 on TOO_MANY_WINDOWS
__device__ static int createExtendWindowsWithAlign(WinState &s, u64 a1, u32 aStr) {
    const DevIndex &X = *s.X; const staramd_params &P = X.P;
    u32 aBin = (u32)(a1 >> P.winBinNbits);
    if (ownerA(s, aStr, aBin) != NOWIN) return 0;
    u32 aChr = X.chrBin[aBin >> P.winBinChrNbits];
    // nearest window on the left whose last bin lies in [aBin-dist, aBin-1]
    u32 iWinL = NOWIN, iWinR = NOWIN;
    if (aBin > 0) {
        u32 lo = aBin > P.winAnchorDistNbins ? aBin - P.winAnchorDistNbins : 0;
        u32 best = 0; bool found = false;
        for (u32 i = 0; i < s.nW; i++) {
            const WScr &w = s.W[i];
            if (w.alive && w.str == aStr && w.coreE < aBin && w.coreE >= lo && (!found || w.coreE > best)) { found = true; best = w.coreE; iWinL = i; }
        }
        if (found && X.chrBin[best >> P.winBinChrNbits] != aChr) iWinL = NOWIN;
    }
    if ((u64)aBin + 1 < P.winBinN) {
        u64 hi = min((u64)aBin + P.winAnchorDistNbins + 1, P.winBinN);     // exclusive
        u32 best = 0; bool found = false;
        for (u32 i = 0; i < s.nW; i++) {
            const WScr &w = s.W[i];
            if (w.alive && w.str == aStr && w.coreS > aBin && (u64)w.coreS < hi && (!found || w.coreS < best)) { found = true; best = w.coreS; iWinR = i; }
        }
        if (found && X.chrBin[best >> P.winBinChrNbits] != aChr) iWinR = NOWIN;
    }
    if (iWinL == NOWIN && iWinR == NOWIN) {
        u32 iWin = s.nW;
        if (iWin >= s.capW) { s.overflow = true; return 1; }
        WScr &w = s.W[iWin];
        w.chr = aChr; w.str = (u8)aStr; w.coreS = w.coreE = aBin; w.extS = w.extE = aBin; w.alive = 1; w.nWA = 0; w.lrec = 0; w.waBlock = NOWIN;
        s.nW++;
        if (s.nW >= P.alignWindowsPerReadNmax) { s.nW = P.alignWindowsPerReadNmax - 1; s.windowsLimit = true; return 1; }
    } else {
        u32 iWin = iWinL != NOWIN ? iWinL : iWinR;                            // left window overwrites right (:57)
        u32 binLeft = iWinL != NOWIN ? s.W[iWinL].coreS : aBin;
        u32 binRight = iWinR != NOWIN ? s.W[iWinR].coreE : aBin;
        if (iWinL != NOWIN && iWinR != NOWIN) s.W[iWinR].alive = 0;           // kill right window (:77-80)
        s.W[iWin].coreS = binLeft; s.W[iWin].coreE = binRight;
    }
    return 0;
}

// ReadAlign_assignAlignToWindow.cpp:6-130
__device__ static void assignAlignToWindow(WinState &s, u64 a1, u32 aLength, u32 aStr, u32 aNrep, u32 aFrag, u32 aRstart, bool aAnchor, i32 sjA) {
    const DevIndex &X = *s.X; const staramd_params &P = X.P;
    u32 iW = ownerB(s, aStr, (u32)(a1 >> P.winBinNbits));
    if (iW == NOWIN) return;
    WScr &w = s.W[iW];
    if (!aAnchor && aLength < w.lrec) return;
    if (w.waBlock == NOWIN) {
        if (s.nBlocks >= s.capBlocks) { s.overflow = true; return; }
        w.waBlock = s.nBlocks++;
    }
    DWA *A = s.arena + (u64)w.waBlock * s.blockSize;
    u32 n = w.nWA;
    {
        u32 iA;
        for (iA = 0; iA < n; iA++) {
            const DWA &o = A[iA];
            if (aFrag == o.iFrag && o.sjA == sjA && a1 + o.rStart == o.gStart + aRstart
                && ((aRstart >= o.rStart && aRstart < (u32)o.rStart + o.L) || (aRstart + aLength >= o.rStart && aRstart + aLength < (u32)o.rStart + o.L))) break;
        }
        if (iA < n) {
            if (aLength > A[iA].L) {
                u32 iA0;
                for (iA0 = 0; iA0 < n; iA0++) if (iA0 != iA && aRstart < A[iA0].rStart) break;
                if (iA0 > iA) --iA0;
                if (iA0 < iA) { for (u32 i = iA; i > iA0; i--) A[i] = A[i - 1]; }
                else if (iA0 > iA) { for (u32 i = iA; i < iA0; i++) A[i] = A[i + 1]; }
                DWA e; e.gStart = a1; e.nrep = aNrep; e.L = (u16)aLength; e.rStart = (u16)aRstart; e.sjA = sjA; e.anchor = aAnchor ? 1 : 0; e.iFrag = (u8)aFrag; e.pad[0] = e.pad[1] = 0;
                A[iA0] = e;
            }
            return;
        }
    }
    if (n == P.seedPerWindowNmax) {
        w.lrec = s.Lread + 1;
        for (u32 iA = 0; iA < n; iA++) if (A[iA].anchor != 1) w.lrec = min(w.lrec, (u32)A[iA].L);
        if (w.lrec == s.Lread + 1) { s.tooManyAnchors = true; return; }
        if (!aAnchor && aLength < w.lrec) return;
        u32 iA1 = 0;
        for (u32 iA = 0; iA < n; iA++) if (A[iA].anchor == 1 || A[iA].L > w.lrec) { A[iA1] = A[iA]; iA1++; }
        n = iA1; w.nWA = (u16)n;
    }
    if (aAnchor || aLength > w.lrec) {
        u32 iA;
        for (iA = 0; iA < n; iA++) if (aRstart < A[iA].rStart) break;
        for (u32 i = n; i > iA; i--) A[i] = A[i - 1];
        DWA e; e.gStart = a1; e.nrep = aNrep; e.L = (u16)aLength; e.rStart = (u16)aRstart; e.sjA = sjA; e.anchor = aAnchor ? 1 : 0; e.iFrag = (u8)aFrag; e.pad[0] = e.pad[1] = 0;
        A[iA] = e;
        w.nWA = (u16)(n + 1);
    }
}

// sjAlignSplit.cpp:3-15
__device__ __forceinline__ bool sjAlignSplit(const DevIndex &X, u64 a1, u32 aLength, u64 &a1D, u32 &aLengthD, u64 &a1A, u32 &aLengthA, u32 &isj) {
    u64 sj1 = (a1 - X.sjGstart) % X.sjdbLength;
    if (sj1 < X.sjdbOverhang && sj1 + aLength > X.sjdbOverhang) {
        isj = (u32)((a1 - X.sjGstart) / X.sjdbLength);
        aLengthD = (u32)(X.sjdbOverhang - sj1); aLengthA = aLength - aLengthD;
        a1D = X.sjDstart[isj] + sj1; a1A = X.sjAstart[isj];
        return true;
    }
    return false;
}

extern "C" __global__ void __launch_bounds__(256) k_windows(DevIndex X, DevBatch B, u8 *scratch, u32 capW, u32 capBlocks) {
    u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    const staramd_params &P = X.P;
    u32 blockSize = P.seedPerWindowNmax;
    u64 perLane = (u64)capW * sizeof(WScr) + (u64)capBlocks * blockSize * sizeof(DWA);
    WinState s; s.X = &X;
    s.W = (WScr *)(scratch + (u64)lane * perLane); s.capW = capW;
    s.arena = (DWA *)(scratch + (u64)lane * perLane + (u64)capW * sizeof(WScr)); s.capBlocks = capBlocks; s.blockSize = blockSize;
    u64 nSAenum = 0, nWindows = 0, nWAtot = 0;
    for (;;) {
        u32 ir = atomicAdd(&B.cursors[9], 1u);
        if (ir >= B.nReads) break;
        DRead rd = B.reads[ir];
        if (rd.nSeeds == 0) continue;
        const DSeed *PC = B.seedPool + rd.seedOffset;
        s.nW = 0; s.nBlocks = 0; s.tooManyAnchors = false; s.windowsLimit = false; s.overflow = false;
        s.Lread = (u32)(B.readOffset[ir + 1] - B.readOffset[ir]);
        // ---- pass A: anchors (ReadAlign_stitchPieces.cpp:41-93)
        for (u32 iP = 0; iP < rd.nSeeds && !s.overflow; iP++) {
            const DSeed sd = PC[iP];
            if (sd.nrep > P.winAnchorMultimapNmax) continue;
            u32 aDir = sd.dir, aLength = sd.L;
            for (u64 iSA = sd.saStart; iSA < sd.saStart + sd.nrep; iSA++) {
                nSAenum++;
                u64 a1 = packedGet(X.SA, iSA, X.saBits, X.saMask);
                u32 aStr = (u32)(a1 >> X.strandBit); a1 &= X.strandMask;
                if (aDir == 1 && aStr == 0) aStr = 1;
                else if (aDir == 0 && aStr == 1) a1 = X.nGenome - (aLength + a1);
                else if (aDir == 1 && aStr == 1) { aStr = 0; a1 = X.nGenome - (aLength + a1); }
                if (a1 >= X.sjGstart) {
                    u64 a1D, a1A; u32 lD, lA, isj;
                    if (sjAlignSplit(X, a1, aLength, a1D, lD, a1A, lA, isj)) {
                        if (createExtendWindowsWithAlign(s, a1D, aStr)) break;
                        if (createExtendWindowsWithAlign(s, a1A, aStr)) break;
                    }
                } else if (createExtendWindowsWithAlign(s, a1, aStr)) break;
            }
        }
        // ---- flanks (:96-118)
        for (u32 i = 0; i < s.nW; i++) {
            WScr &w = s.W[i];
            w.nWA = 0; w.lrec = 0; w.waBlock = NOWIN;
            if (!w.alive) continue;
            u32 wb = w.coreS;
            for (u32 ii = 0; ii < P.winFlankNbins && wb > 0 && X.chrBin[(wb - 1) >> P.winBinChrNbits] == w.chr; ii++) wb--;
            w.extS = wb;
            wb = w.coreE;
            for (u32 ii = 0; ii < P.winFlankNbins && (u64)wb + 1 < P.winBinN && X.chrBin[(wb + 1) >> P.winBinChrNbits] == w.chr; ii++) wb++;
            w.extE = wb;
        }
        nWindows += s.nW;
        // ---- pass B: all seeds (:129-185)
        for (u32 iP = 0; iP < rd.nSeeds && !s.overflow && !s.tooManyAnchors; iP++) {
            const DSeed sd = PC[iP];
            u32 aNrep = sd.nrep, aFrag = sd.iFrag, aLength = sd.L, aDir = sd.dir;
            bool aAnchor = aNrep <= P.winAnchorMultimapNmax;
            for (u64 iSA = sd.saStart; iSA < sd.saStart + sd.nrep && !s.tooManyAnchors; iSA++) {
                nSAenum++;
                u64 a1 = packedGet(X.SA, iSA, X.saBits, X.saMask);
                u32 aStr = (u32)(a1 >> X.strandBit); a1 &= X.strandMask;
                u32 aRstart = sd.rStart;
                if (aDir == 1 && aStr == 0) { aStr = 1; aRstart = s.Lread - (aLength + aRstart); }
                else if (aDir == 0 && aStr == 1) { aRstart = s.Lread - (aLength + aRstart); a1 = X.nGenome - (aLength + a1); }
                else if (aDir == 1 && aStr == 1) { aStr = 0; a1 = X.nGenome - (aLength + a1); }
                if (a1 >= X.sjGstart) {
                    u64 a1D, a1A; u32 lD, lA, isj;
                    if (sjAlignSplit(X, a1, aLength, a1D, lD, a1A, lA, isj)) {
                        assignAlignToWindow(s, a1D, lD, aStr, aNrep, aFrag, aRstart, aAnchor, (i32)isj);
                        if (!s.tooManyAnchors) assignAlignToWindow(s, a1A, lA, aStr, aNrep, aFrag, aRstart + lD, aAnchor, (i32)isj);
                    }
                } else assignAlignToWindow(s, a1, aLength, aStr, aNrep, aFrag, aRstart, aAnchor, -1);
            }
        }
        if (s.windowsLimit) rd.status |= STARAMD_ST_WINDOWS_LIMIT;
        if (s.overflow) { rd.status |= STARAMD_ST_SCRATCH_OVERFLOW; atomicOr(&B.cursors[6], 2u); B.reads[ir] = rd; continue; }
        if (s.tooManyAnchors) { rd.status |= STARAMD_ST_TOO_MANY_ANCHORS | STARAMD_ST_NO_GOOD_WINDOW; B.reads[ir] = rd; continue; }   // nW=0 (:76-80)
        // ---- emit windows that hold seeds, in window order
        u32 nOut = 0, nWA = 0;
        for (u32 i = 0; i < s.nW; i++) if (s.W[i].nWA > 0) { nOut++; nWA += s.W[i].nWA; }
        if (nOut > 0) {
            u32 wo = atomicAdd(&B.cursors[1], nOut), ao = atomicAdd(&B.cursors[2], nWA);
            if (wo + nOut > B.winCap || ao + nWA > B.waCap) { rd.status |= STARAMD_ST_SCRATCH_OVERFLOW; atomicOr(&B.cursors[6], 4u); }
            else {
                rd.winOffset = wo; rd.nWin = nOut;
                for (u32 i = 0; i < s.nW; i++) {
                    const WScr &w = s.W[i];
                    if (w.nWA == 0) continue;
                    DWin d; d.read = ir; d.chr = w.chr; d.waOffset = ao; d.nWA = w.nWA; d.str = w.str; d.pad = 0;
                    B.winPool[wo++] = d;
                    const DWA *A = s.arena + (u64)w.waBlock * blockSize;
                    for (u32 k = 0; k < w.nWA; k++) B.waPool[ao + k] = A[k];
                    ao += w.nWA;
                }
                nWAtot += nWA;
            }
        }
        B.reads[ir] = rd;
    }
    atomicAdd((unsigned long long *)&B.counters[DC_nSAenum], (unsigned long long)nSAenum);
    atomicAdd((unsigned long long *)&B.counters[DC_nWindows], (unsigned long long)nWindows);
    atomicAdd((unsigned long long *)&B.counters[DC_nWA], (unsigned long long)nWAtot);
}
