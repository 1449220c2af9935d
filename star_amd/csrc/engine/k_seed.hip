// k_seed.hip -- kernel 1: maximal-mappable-prefix seed search.
//
// Replaces, per read, the seed phase of ReadAlign::mapOneRead (source/ReadAlign_mapOneRead.cpp:17-93):
// qualitySplit (SequenceFuns.cpp:411-444), maxMappableLength2strands
// (ReadAlign_maxMappableLength2strands.cpp:5-115), maxMappableLength / findMultRange /
// compareSeqToGenome (SuffixArrayFuns.cpp:10-207) and storeAligns (ReadAlign_storeAligns.cpp:10-160).
//
// Mapping: one lane = one read, 64 independent reads per wavefront, persistent lanes pulling
// read ids from a device-wide ticket counter.  The kernel is a chain of dependent random gathers
// (SAindex entry -> packed SA word -> genome bytes), so throughput is set by the number of
// independent chains in flight; every lane carries one.  The seed schedule inside a read is kept
// strictly in the reference's order because storeAligns' de-duplication is order dependent
// (first (rStart,Length) in schedule order wins).
#include "dev.h"

struct SeedCnt { u32 nSAi, nSAprobe, nGcmp; };     // per lane and kernel: a lane compares < 2^32 bases

// 8 bytes starting at an arbitrary address, little endian, through aligned 8-byte loads (the arrays are padded)
__device__ __forceinline__ u64 load8(const u8 *p) {
    const u64 a = (u64)p; const __attribute__((address_space(1))) u64 *q = GLOBAL(u64, a & ~7ull); const u32 sh = (u32)(a & 7ull) * 8u;
    const u64 w0 = q[0];
    if (sh == 0) return w0;
    return (w0 >> sh) | (q[1] << (64u - sh));
}
// bytes p[0], p[-1], ..., p[-7] as a little-endian sequence (a scan that walks backwards)
__device__ __forceinline__ u64 load8rev(const u8 *p) { return __builtin_bswap64(load8(p - 7)); }
// complementSeqNumbers on 8 codes at once: c < 4 -> 3 - c (== 3 ^ c), others unchanged
__device__ __forceinline__ u64 comp8(u64 x) {
    const u64 ge4 = ((x + 0x7C7C7C7C7C7C7C7Cull) & 0x8080808080808080ull) >> 7;       // 1 in bytes >= 4 (codes <= 11: no carry between bytes)
    return x ^ (0x0303030303030303ull & ~(ge4 * 0xFFull));
}

// SuffixArrayFuns.cpp:10-104 -- the four read/genome direction variants folded into one loop, 8 bases per step:
// read and genome are both 1 byte/base, so a step is two 8-byte words, one XOR and a count-trailing-zeros
__device__ static u32 compareSeqToGenome(const DevIndex &X, const u8 *R, u32 S, u32 N, u32 L, u64 iSA, bool dirR, bool &compRes, SeedCnt &cn) {
    cn.nSAprobe++;
    u64 SAstr = packedGet(X.SA, iSA, X.saBits, X.saMask);
    bool dirG = (SAstr >> X.strandBit) == 0;
    SAstr &= X.strandMask;
    bool useComp = dirR != dirG;
    const u8 *g = dirG ? X.G + SAstr + L : X.G + (X.nGenome - 1 - SAstr) - L;
    const u8 *s = dirR ? R + S + L : R + S - L;
    const u32 n = N - L;
    for (u32 ii = 0; ii < n; ii += 8) {
        u64 s8 = dirR ? load8(s + ii) : load8rev(s - ii);
        const u64 g8 = dirG ? load8(g + ii) : load8rev(g - ii);
        if (useComp) s8 = comp8(s8);
        u64 d = s8 ^ g8;
        const u32 rem = n - ii;
        if (rem < 8) d &= (1ull << (rem * 8)) - 1ull;                   // bases behind the end of the piece do not count
        if (d) {
            const u32 k = (u32)__builtin_ctzll(d) >> 3;
            const u8 sc = (u8)(s8 >> (k * 8)), gc = (u8)(g8 >> (k * 8));
            cn.nGcmp += ii + k + 1;
            compRes = dirG ? (sc > gc) : !(sc > gc || gc > 3);
            return ii + k + L;
        }
    }
    cn.nGcmp += n;
    return N;
}

__device__ __forceinline__ u64 medianUint2(u64 a, u64 b) { return a / 2 + b / 2 + (a % 2 + b % 2) / 2; }

// SuffixArrayFuns.cpp:106-131.  IDX: suffix-array indices inside one search are offsets from `base` (the lower end of the interval the SAindex
// look-up returned).  An interval is almost always shorter than 2^32 entries; then the seven indices of a search are 32-bit values (IDX = u32),
// which halves the registers they take in a kernel that is held to 64.  The same code with IDX = u64 serves the rest (a look-up whose upper
// neighbour is absent searches up to the end of a suffix array of 6.3 * 10^9 entries).
template <class IDX> __device__ static IDX findMultRange(const DevIndex &X, const u8 *R, u64 base, IDX i3, u32 L3, IDX i1, u32 L1, IDX i1a, u32 L1a, IDX i1b, u32 L1b, bool dirR, u32 S, SeedCnt &cn) {
    bool compRes;
    if (L1 < L3) { L1b = L1; i1b = i1; i1a = i3; }
    else if (L1a < L1) { L1b = L1a; i1b = i1a; i1a = i1; }
    while (((u64)i1b + 1 < (u64)i1a) | ((u64)i1b > (u64)i1a + 1)) {
        IDX i1c = (IDX)medianUint2(i1a, i1b);
        u32 L1c = compareSeqToGenome(X, R, S, L3, L1b, base + i1c, dirR, compRes, cn);
        if (L1c == L3) i1a = i1c; else { i1b = i1c; L1b = L1c; }
    }
    return i1a;
}

// SuffixArrayFuns.cpp:133-207
template <class IDX> __device__ static u64 maxMappableLengthT(const DevIndex &X, const u8 *R, u32 S, u32 N, u64 i1in, u64 i2in, bool dirR, u32 &L, u64 &ind0, u64 &ind1, SeedCnt &cn) {
    bool compRes;
    const u64 base = i1in;
    u32 L1, L2, L3, L1a, L1b, L2a, L2b; IDX i1 = 0, i2 = (IDX)(i2in - i1in), i3, i1a, i1b, i2a, i2b;
    L1 = compareSeqToGenome(X, R, S, N, L, base + i1, dirR, compRes, cn);
    L2 = compareSeqToGenome(X, R, S, N, L, base + i2, dirR, compRes, cn);
    L = min(L1, L2);
    L1a = L1; L1b = L1; i1a = i1; i1b = i1; L2a = L2; L2b = L2; i2a = i2; i2b = i2;
    i3 = i1; L3 = L1;
    while ((u64)i1 + 1 < (u64)i2) {
        i3 = (IDX)medianUint2(i1, i2);
        L3 = compareSeqToGenome(X, R, S, N, L, base + i3, dirR, compRes, cn);
        if (L3 == N) break;
        if (compRes) { if (L3 > L1) { L1b = L1a; L1a = L1; i1b = i1a; i1a = i1; } i1 = i3; L1 = L3; }
        else { if (L3 > L2) { L2b = L2a; L2a = L2; i2b = i2a; i2a = i2; } i2 = i3; L2 = L3; }
        L = min(L1, L2);
    }
    if (L3 < N) { if (L1 > L2) { i3 = i1; L3 = L1; } else { i3 = i2; L3 = L2; } }
    i1 = findMultRange<IDX>(X, R, base, i3, L3, i1, L1, i1a, L1a, i1b, L1b, dirR, S, cn);
    i2 = findMultRange<IDX>(X, R, base, i3, L3, i2, L2, i2a, L2a, i2b, L2b, dirR, S, cn);
    L = L3; ind0 = base + i1; ind1 = base + i2;
    return (u64)i2 - (u64)i1 + 1;
}
__device__ __forceinline__ u64 maxMappableLength(const DevIndex &X, const u8 *R, u32 S, u32 N, u64 i1, u64 i2, bool dirR, u32 &L, u64 &ind0, u64 &ind1, SeedCnt &cn) {
    if (i2 - i1 < 0xFFFFFFFFull) return maxMappableLengthT<u32>(X, R, S, N, i1, i2, dirR, L, ind0, ind1, cn);
    return maxMappableLengthT<u64>(X, R, S, N, i1, i2, dirR, L, ind0, ind1, cn);
}

struct SeedState {
    DSeed *PC; u32 nP; u32 cap;
    u32 nA; u32 multNmin, multNminL;        // nA: only ever compared with 0 (ReadAlign_mapOneRead.cpp:106), saturating
    bool fatal;
};

// ReadAlign_storeAligns.cpp:10-160 (OPTIM_STOREaligns_SIMPLE branch)
__device__ static void storeAligns(const DevIndex &X, SeedState &st, u32 iDir, u32 Shift, u64 Nrep, u32 L, u64 ind0, u32 iFrag) {
    if (Nrep > X.P.seedMultimapNmax) { if (Nrep < st.multNmin || st.multNmin == 0) { st.multNmin = (u32)min(Nrep, (u64)0xFFFFFFFFu); st.multNminL = L; } return; }
    st.nA |= 1u;
    u32 rStart = iDir == 0 ? Shift : Shift + 1 - L;
    int iP;
    for (iP = (int)st.nP - 1; iP >= 0; iP--) {
        if (st.PC[iP].rStart <= rStart) {
            if (st.PC[iP].rStart == rStart && st.PC[iP].L < L) continue;
            if (st.PC[iP].rStart == rStart && st.PC[iP].L == L) return;
            break;
        }
    }
    iP++;
    if (st.nP + 1 > X.P.seedPerReadNmax || st.nP + 1 > st.cap) { st.fatal = true; return; }    // reference: exitWithError :46-51
    for (int ii = (int)st.nP - 1; ii >= iP; ii--) st.PC[ii + 1] = st.PC[ii];
    st.nP++;
    DSeed s; s.saStart = ind0; s.nrep = (u32)Nrep; s.rStart = (u16)rStart; s.L = (u16)L; s.dir = (u8)iDir; s.iFrag = (u8)iFrag;
    for (int k = 0; k < 6; k++) s.pad[k] = 0;
    st.PC[iP] = s;
    if (Nrep != 1) { if (Nrep < st.multNmin || st.multNmin == 0) { st.multNmin = (u32)Nrep; st.multNminL = L; } }
}

// ReadAlign_maxMappableLength2strands.cpp:12-109: one start offset iDist of the sparse-SA loop
__device__ __forceinline__ void searchOneDist(const DevIndex &X, const u8 *R, u32 pieceStartIn, u32 pieceLengthIn, bool dirR, u32 iDist, u64 &Nrep, u64 &i0, u32 &maxL, SeedCnt &cn) {
    u32 pieceStart; u32 pieceLength = pieceLengthIn - iDist;
    u32 Lmax = min(X.saiNbases, pieceLength);
    u64 ind1 = 0;
    // L-mer prefix (ReadAlign_maxMappableLength2strands.cpp:23-37): 2 bits per base, first base most significant;
    // the bases of a piece are all 0..3, so 8 of them are packed from one 8-byte word with shifts and masks
    if (dirR) pieceStart = pieceStartIn + iDist; else pieceStart = pieceStartIn - iDist;
    for (u32 ii = 0; ii < Lmax; ii += 8) {
        const u64 raw = dirR ? load8(R + pieceStart + ii) : load8rev(R + pieceStart - ii);     // byte k = code of the (ii+k)-th base of the scan
        const u32 nb = min(8u, Lmax - ii);
        const u64 used = nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
        if ((raw & used & 0xFCFCFCFCFCFCFCFCull) == 0) {
            u64 x = dirR ? raw : (raw ^ 0x0303030303030303ull);                                   // 3 - base == 3 ^ base
            u64 z = __builtin_bswap64(x & 0x0303030303030303ull); // first base in the top byte (bytes behind the prefix may hold N / spacer codes)
            z = (z | (z >> 6)) & 0x000F000F000F000Full;
            z = (z | (z >> 12)) & 0x000000FF000000FFull;
            z = (z | (z >> 24)) & 0xFFFFull;                    // 8 bases -> 16 bits, first base most significant
            ind1 = (ind1 << (2 * nb)) | (z >> (2 * (8 - nb)));
        } else {
            // a code above 3 inside the prefix: only with --seedSearchLmax, whose backward search is given Shift + 1 bases (:from ReadAlign_mapOneRead.cpp:81-86)
            // and so runs over the start of its piece into an N, the mate spacer or the other mate.  The reference adds the code as it is
            // (index*4 + code, index*4 + (3 - code) in unsigned 64-bit arithmetic): the carries and borrows are part of its result
            for (u32 k = 0; k < nb; k++) { const u64 cde = (raw >> (8 * k)) & 0xFFull; ind1 = (ind1 << 2) + (dirR ? cde : 3ull - cde); }
        }
    }
    u32 Lind = Lmax; u64 iSA1 = 0, iSA2;
    while (Lind > 0) {
        const u64 flat = X.saiStart[Lind - 1] + ind1;                                    // (wraps like the reference's flat packed-array index)
        if (flat >= X.saiStart[X.saiNbases]) { --Lind; ind1 >>= 2; continue; }          // outside the table: the reference reads whatever lies there; treated as absent
        iSA1 = packedGet(X.SAi, flat, X.saiBits, X.saiMask); cn.nSAi++;
        if ((iSA1 & X.saiAbsentBit) == 0) break;
        --Lind; ind1 >>= 2;
    }
    if (Lind == 0) { Nrep = 0; i0 = 0; maxL = 0; return; }   // base absent from the genome (reference: out-of-bounds)
    bool iSA2good = true;
    if (X.saiStart[Lind - 1] + ind1 + 1 < X.saiStart[Lind]) {
        iSA2 = packedGet(X.SAi, X.saiStart[Lind - 1] + ind1 + 1, X.saiBits, X.saiMask); cn.nSAi++;
        if ((iSA2 & X.saiAbsentBit) == 0) iSA2 = (iSA2 & ~X.saiNbit) - 1;
        else { iSA2 = X.nSA - 1; iSA2good = false; }
    } else { iSA2 = X.nSA - 1; iSA2good = false; }
    bool iSA1noN = (iSA1 & X.saiNbit) == 0;
    u64 i1;
    if (Lind < X.saiNbases && iSA1noN && iSA2good) { i0 = iSA1; i1 = iSA2; Nrep = i1 - i0 + 1; maxL = Lind; }
    else if (iSA1 == iSA2 && iSA1noN && iSA2good) {
        i0 = i1 = iSA1; Nrep = 1; bool cr;
        maxL = compareSeqToGenome(X, R, pieceStart, pieceLength, Lind, iSA1, dirR, cr, cn);
    } else {
        maxL = (iSA2good && iSA1noN) ? Lind : 0;
        Nrep = maxMappableLength(X, R, pieceStart, pieceLength, iSA1 & ~X.saiNbit, iSA2, dirR, maxL, i0, i1, cn);
    }
}

// ReadAlign_maxMappableLength2strands.cpp:5-115.  The reference keeps (Nrep, ind0, maxL) of every start offset of a sparse suffix
// array in small tables and stores the best ones afterwards; indexed local tables would live in scratch memory here, so with
// genomeSAsparseD > 1 the offsets are simply searched twice (first for the best length, then to store) -- with the default
// full suffix array there is one offset and one search.
__device__ static void maxMappableLength2strands(const DevIndex &X, const u8 *R, SeedState &st, u32 pieceStartIn, u32 pieceLengthIn, u32 iDir, u32 &maxLbest, u32 iFrag, SeedCnt &cn) {
    const bool dirR = iDir == 0;
    const u32 nD = min(pieceLengthIn, X.sparseD);
    maxLbest = 0;
    for (u32 it = 0; it < 2 * nD; it++) {
        const u32 phase = it >= nD ? 1u : 0u, iDist = phase ? it - nD : it;
        u64 Nrep, i0; u32 maxL;
        searchOneDist(X, R, pieceStartIn, pieceLengthIn, dirR, iDist, Nrep, i0, maxL, cn);
        if (phase == 0) {
            if (maxL + iDist > maxLbest) maxLbest = maxL + iDist;
            if (nD > 1) continue;
        }
        if (maxL + iDist == maxLbest && Nrep > 0)
            storeAligns(X, st, iDir, dirR ? pieceStartIn + iDist : pieceStartIn - iDist, Nrep, maxL, i0, iFrag);
        if (nD == 1) break;
    }
}

#ifndef SEED_WAVES
#define SEED_WAVES 8        // minimum waves per SIMD the register allocation is held to (8: 64 VGPRs + spills, 10 % faster than 4 at 1 Gb: more gather chains in flight)
#endif
extern "C" __global__ void __launch_bounds__(256, SEED_WAVES) k_seed_search(const DevIndex *__restrict__ Xp, DevBatch B, DSeed *scratch, u32 scratchPerLane) {
    const DevIndex &X = *Xp;
    u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    SeedState st; st.PC = scratch + (u64)lane * scratchPerLane; st.cap = scratchPerLane;
    SeedCnt cn = {0, 0, 0}; u64 nSeedsTot = 0;
    const staramd_params &P = X.P;
    for (;;) {
        u32 ir = atomicAdd(&B.cursors[CUR_TICKET_SEED], 1u);
        if (ir >= B.nReads) break;
        const u8 *R = B.bases + B.readOffset[ir];
        u32 Lread = (u32)(B.readOffset[ir + 1] - B.readOffset[ir]);
        st.nP = 0; st.nA = 0; st.multNmin = 0; st.multNminL = 0; st.fatal = false;
        // qualitySplit (SequenceFuns.cpp:411-444) and the loop over its pieces (ReadAlign_mapOneRead.cpp:40-93) fused: a piece is searched as soon as its
        // end is found -- the pieces are processed in the order the reference stores them, and a table of pieces indexed at run time would live in
        // scratch memory (it did: 80 bytes per lane, re-read for every seed)
        u32 Nsplit = 0, LgoodMin = 0;
        const u32 seedSearchStartLmax = min(P.seedSearchStartLmax, (u32)(u64)(P.seedSearchStartLmaxOverLread * (double)(u64)(Lread - 1)));
        {
            u32 iR = 0, iFrag = 0;
            while ((iR < Lread) & (Nsplit < P.maxNsplit)) {
                while (iR < Lread && R[iR] > 3) { if (R[iR] == STARAMD_SPACER_BASE) iFrag++; iR++; }
                if (iR == Lread) break;
                const u32 pS = iR;
                for (;;) {                                                  // end of the run of good bases, 8 bases per step
                    const u64 bad = load8(R + iR) & 0xFCFCFCFCFCFCFCFCull;
                    const u32 k = bad ? ((u32)__builtin_ctzll(bad) >> 3) : 8u;
                    iR += k;
                    if (iR >= Lread) { iR = Lread; break; }
                    if (k < 8u) break;
                }
                const u32 pL = iR - pS;
                if (pL > LgoodMin) LgoodMin = pL;
                if (pL < P.seedSplitMin) continue;
                Nsplit++;
                const u32 Nstart = (P.seedSearchStartLmax > 0 && seedSearchStartLmax < pL) ? pL / seedSearchStartLmax + 1 : 1;
                const u32 Lstart = pL / Nstart;
                bool flagDirMap = true;
                for (u32 iDir = 0; iDir < 2; iDir++) {
                    for (u32 istart = 0; istart < Nstart; istart++) {
                        u32 Lm;
                        if (flagDirMap || istart > 0) {
                            u32 Lmapped = 0;
                            while (istart * Lstart + Lmapped + P.seedMapMin < pL) {
                                u32 Shift = iDir == 0 ? (pS + istart * Lstart + Lmapped) : (pS + pL - istart * Lstart - 1 - Lmapped);
                                u32 seedLength = pL - Lmapped - istart * Lstart;
                                maxMappableLength2strands(X, R, st, Shift, seedLength, iDir, Lm, iFrag, cn);
                                if (iDir == 0 && istart == 0 && Lmapped == 0 && Shift + Lm == pL) flagDirMap = false;
                                Lmapped += Lm;
                                if (Lm == 0) break;
                            }
                        }
                        if (P.seedSearchLmax > 0) {
                            u32 Shift = iDir == 0 ? (pS + istart * Lstart) : (pS + pL - istart * Lstart - 1);
                            u32 seedLength = min(P.seedSearchLmax, iDir == 0 ? (pS + pL - Shift) : (Shift + 1));
                            maxMappableLength2strands(X, R, st, Shift, seedLength, iDir, Lm, iFrag, cn);
                        }
                    }
                }
            }
        }
        // classification, ReadAlign_mapOneRead.cpp:100-115
        DRead rd;
        rd.status = 0; rd.seedOffset = 0; rd.nSeeds = 0; rd.unmappedLength = 0; rd.winOffset = 0; rd.nWin = 0; rd.wtOffset = 0; rd.nWt = 0; rd.pruneBest = 0; rd.pad0 = 0;
        rd.maxScoreMate[0] = rd.maxScoreMate[1] = 0; rd.bestW = -1; rd.nTr = 0; rd.nEx = 0;
        nSeedsTot += st.nP;
        if (st.fatal) rd.status |= STARAMD_ST_FATAL_SEEDS_PER_READ;
        else if (Lread < P.outFilterMatchNmin) { rd.status |= STARAMD_ST_READ_TOO_SHORT; rd.unmappedLength = 0; }
        else if (Nsplit == 0) { rd.status |= STARAMD_ST_NO_GOOD_PIECES; rd.unmappedLength = LgoodMin; }
        else if (st.nA == 0) { rd.status |= STARAMD_ST_ALL_PIECES_MULTI; rd.unmappedLength = st.multNminL; }
        else {
            u32 off = atomicAdd(&B.cursors[CUR_SEED], st.nP);
            if (off + st.nP > B.seedCap) { atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_SEEDPOOL); }
            else {
                rd.seedOffset = off; rd.nSeeds = st.nP;
                for (u32 k = 0; k < st.nP; k++) B.seedPool[off + k] = st.PC[k];
            }
        }
        B.reads[ir] = rd;
    }
    atomicAdd((unsigned long long *)&B.counters[DC_nSAi], (unsigned long long)cn.nSAi);
    atomicAdd((unsigned long long *)&B.counters[DC_nSAprobe], (unsigned long long)cn.nSAprobe);
    atomicAdd((unsigned long long *)&B.counters[DC_nGcmp], (unsigned long long)cn.nGcmp);
    atomicAdd((unsigned long long *)&B.counters[DC_nSeeds], (unsigned long long)nSeedsTot);
}
