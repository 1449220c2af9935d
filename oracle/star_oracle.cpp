// star_oracle.cpp -- CPU restatement of STAR's per-read hot path (TEST INFRASTRUCTURE ONLY).
//
// This file is the parity oracle of the project: a plain, scalar, single-threaded restatement of
//   ReadAlign::mapOneRead -> maxMappableLength2strands -> SuffixArrayFuns  (seed search)
//   ReadAlign::storeAligns                                                  (seed table PC)
//   ReadAlign::stitchPieces -> createExtendWindowsWithAlign / assignAlignToWindow / sjAlignSplit
//   stitchWindowAligns -> stitchAlignToTranscript / extendAlign / binarySearch2 / blocksOverlap
// of alexdobin/STAR 2.7.11b.  Every function cites the reference file:line it follows.  It is
// pinned against the reference itself (oracle/_ref/STAR built by oracle/Makefile.ref; see
// tests/test_oracle_vs_reference.py and tests/golden/).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library; the product (star_amd/) never does.
//
// It deliberately keeps the reference's data shapes (PC rows, WA rows, uint16 winBin map, the
// recursive include/exclude stitcher) so that it can be audited against the reference by eye;
// the HIP engine uses different layouts and a different decomposition.
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../include/star_amd.h"

typedef uint64_t u64;
typedef int64_t i64;

namespace {

enum { C_nSAi, C_nSAprobe, C_nGcmp, C_nSAenum, C_nGstitchReread, C_nGstitchSpan, C_nSeeds, C_nWindows,
       C_nWA, C_nNodes, C_nLeaves, C_nStitchCalls, C_nExtendCalls, C_nTrOut, C_maxNodesWin, C_maxWAWin, C_N };

// PC row: ReadAlign_storeAligns.cpp:138-144, IncludeDefine.h:181-189
struct Seed { u64 rStart, L, dir, nrep, saStart, saEnd, iFrag; };
// WA row: IncludeDefine.h:197-204
struct WAlign { u64 L, rStart, gStart, nrep, anchor, iFrag, sjA; };
// WC row: IncludeDefine.h:191-195
struct Win { u64 str, chr, gStart, gEnd; };

// Transcript: source/Transcript.h:10-81 (fields used on this path)
struct Tr {
    u64 ex[STARAMD_MAX_N_EXONS][5];           // EX_R, EX_G, EX_L, EX_iFrag, EX_sjA
    u64 shiftSJ[STARAMD_MAX_N_EXONS][2];
    int canonSJ[STARAMD_MAX_N_EXONS];
    uint8_t sjAnnot[STARAMD_MAX_N_EXONS], sjStr[STARAMD_MAX_N_EXONS];
    u64 intronMotifs[3];
    uint8_t sjMotifStrand;
    u64 nExons;
    int iFrag;
    u64 rStart, roStart, rLength, gStart, gLength;
    u64 Chr, Str, roStr;
    u64 nMatch, nMM, mappedLength, extendL;
    int maxScore;
    u64 nGap, lGap, nDel, nIns, lDel, lIns, nUnique, nAnchor;
    void reset() {                            // Transcript::reset, Transcript.cpp:13-29
        extendL = 0; rStart = 0; roStart = 0; rLength = 0; gStart = 0; gLength = 0;
        maxScore = 0; nMatch = 0; nMM = 0; nGap = 0; lGap = 0; lDel = 0; lIns = 0; nDel = 0; nIns = 0;
        nUnique = nAnchor = 0;
    }
    void add(const Tr &t) {                   // Transcript::add, Transcript.cpp:31-39
        maxScore += t.maxScore; nMatch += t.nMatch; nMM += t.nMM; nGap += t.nGap; lGap += t.lGap;
        lDel += t.lDel; nDel += t.nDel; lIns += t.lIns; nIns += t.nIns; nUnique += t.nUnique;
    }
};
enum { EX_R = 0, EX_G = 1, EX_L = 2, EX_iFrag = 3, EX_sjA = 4 };

struct Oracle {
    staramd_genome g;
    staramd_params P;
    std::vector<char> G1;         // genome with 200 bytes of code 5 either side (Genome_genomeLoad.cpp:27,306,320-323)
    char *G;
    u64 saMask, saiMask, GstrandMask, SAiMarkAbsentMaskC, SAiMarkNmask, SAiMarkNmaskC;
    u64 cnt[C_N];
    std::vector<u64> sjNovelStart, sjNovelEnd;   // P.sjNovelStart/End (2nd stage of BySJout)

    // per-read state (class ReadAlign)
    std::vector<char> R0, R1, R2; // Read1[0..2]
    char *Read1[3];
    u64 Lread, readLength[2], mmMaxTotal;
    u64 splitR[3][16]; u64 Nsplit;
    std::vector<Seed> PC; u64 nA, nUM[2], multNmin, multNminL;
    bool fatalSeeds;
    std::vector<uint16_t> winBin[2];
    std::vector<Win> WC; std::vector<std::vector<WAlign> > WA; std::vector<u64> WALrec;
    u64 nW; bool tooManyAnchors; bool windowsLimit;
    int maxScoreMate[2];
    std::vector<std::vector<Tr> > trAll;      // per window, best first
    u64 gSpanMin, gSpanMax;

    // PackedArray::operator[], PackedArray.h:24-32
    inline u64 SAat(u64 i) {
        u64 b = i * (g.GstrandBit + 1); u64 a; memcpy(&a, g.SA + b / 8, 8);
        return (a >> (b % 8)) & saMask;
    }
    inline u64 SAiAt(u64 i) {
        u64 b = i * (g.GstrandBit + 3); u64 a; memcpy(&a, g.SAi + b / 8, 8);
        cnt[C_nSAi]++;
        return (a >> (b % 8)) & saiMask;
    }
    inline char Gs(u64 pos) {                 // genome access of the stitch phase (counted)
        cnt[C_nGstitchReread]++;
        if ((i64)pos >= 0) { if (pos < gSpanMin) gSpanMin = pos; if (pos > gSpanMax) gSpanMax = pos; }
        return G[(i64)pos];
    }

    void init(const staramd_genome *gi, const staramd_params *pi) {
        g = *gi; P = *pi;
        G1.assign(g.nGenome + 400, 5);
        memcpy(G1.data() + 200, g.G, g.nGenome);
        G = G1.data() + 200;
        saMask = (g.GstrandBit + 1 >= 64) ? ~0ull : ((1ull << (g.GstrandBit + 1)) - 1);
        saiMask = (1ull << (g.GstrandBit + 3)) - 1;
        GstrandMask = ~(1ull << g.GstrandBit);                  // Genome_genomeLoad.cpp:157
        SAiMarkNmaskC = 1ull << (g.GstrandBit + 1);             // :161-166
        SAiMarkNmask = ~SAiMarkNmaskC;
        SAiMarkAbsentMaskC = 1ull << (g.GstrandBit + 2);
        winBin[0].assign(P.winBinN + 1, 0xFFFF);
        winBin[1].assign(P.winBinN + 1, 0xFFFF);
        memset(cnt, 0, sizeof(cnt));
    }

    // ------------------------------------------------------------------ SequenceFuns.cpp:411-444
    u64 qualitySplit(const char *r, u64 L, u64 maxNsplit, u64 minLsplit) {
        u64 iR = 0, iS = 0, iR1, LgoodMin = 0, iFrag = 0;
        while ((iR < L) & (iS < maxNsplit)) {
            while (iR < L && r[iR] > 3) { if (r[iR] == STARAMD_SPACER_BASE) iFrag++; iR++; }
            if (iR == L) break;
            iR1 = iR;
            while (iR < L && r[iR] <= 3) iR++;
            if ((iR - iR1) > LgoodMin) LgoodMin = iR - iR1;
            if ((iR - iR1) < minLsplit) continue;
            splitR[0][iS] = iR1; splitR[1][iS] = iR - iR1; splitR[2][iS] = iFrag; iS++;
        }
        if (iS == 0) splitR[1][0] = LgoodMin;
        return iS;
    }

    // ------------------------------------------------------------------ SuffixArrayFuns.cpp:10-104
    u64 compareSeqToGenome(u64 S, u64 N, u64 L, u64 iSA, bool dirR, bool &compRes) {
        cnt[C_nSAprobe]++;
        u64 SAstr = SAat(iSA);
        bool dirG = (SAstr >> g.GstrandBit) == 0;
        SAstr &= GstrandMask;
        i64 ii; u64 n = N - L;
        if (dirR && dirG) {
            const char *s = Read1[0] + S + L; const char *gg = G + SAstr + L;
            for (ii = 0; (u64)ii < n; ii++) if (s[ii] != gg[ii]) { cnt[C_nGcmp] += ii + 1; compRes = s[ii] > gg[ii]; return ii + L; }
            cnt[C_nGcmp] += n; return N;
        } else if (dirR && !dirG) {
            const char *s = Read1[1] + S + L; const char *gg = G + (g.nGenome - 1 - SAstr) - L;
            for (ii = 0; (u64)ii < n; ii++) if (s[ii] != gg[-ii]) { cnt[C_nGcmp] += ii + 1; compRes = !(s[ii] > gg[-ii] || gg[-ii] > 3); return ii + L; }
            cnt[C_nGcmp] += n; return N;
        } else if (!dirR && dirG) {
            const char *s = Read1[1] + S - L; const char *gg = G + SAstr + L;
            for (ii = 0; (u64)ii < n; ii++) if (s[-ii] != gg[ii]) { cnt[C_nGcmp] += ii + 1; compRes = s[-ii] > gg[ii]; return ii + L; }
            cnt[C_nGcmp] += n; return N;
        } else {
            const char *s = Read1[0] + S - L; const char *gg = G + (g.nGenome - 1 - SAstr) - L;
            for (ii = 0; (u64)ii < n; ii++) if (s[-ii] != gg[-ii]) { cnt[C_nGcmp] += ii + 1; compRes = !(s[-ii] > gg[-ii] || gg[-ii] > 3); return ii + L; }
            cnt[C_nGcmp] += n; return N;
        }
    }
    static inline u64 medianUint2(u64 a, u64 b) { return a / 2 + b / 2 + (a % 2 + b % 2) / 2; }   // :4-8

    // SuffixArrayFuns.cpp:106-131
    u64 findMultRange(u64 i3, u64 L3, u64 i1, u64 L1, u64 i1a, u64 L1a, u64 i1b, u64 L1b, bool dirR, u64 S) {
        bool compRes;
        if (L1 < L3) { L1b = L1; i1b = i1; i1a = i3; }
        else if (L1a < L1) { L1b = L1a; i1b = i1a; i1a = i1; }
        while ((i1b + 1 < i1a) | (i1b > i1a + 1)) {
            u64 i1c = medianUint2(i1a, i1b);
            u64 L1c = compareSeqToGenome(S, L3, L1b, i1c, dirR, compRes);
            if (L1c == L3) i1a = i1c; else { i1b = i1c; L1b = L1c; }
        }
        return i1a;
    }
    // SuffixArrayFuns.cpp:133-207
    u64 maxMappableLength(u64 S, u64 N, u64 i1, u64 i2, bool dirR, u64 &L, u64 *indStartEnd) {
        bool compRes;
        u64 L1, L2, i3, L3, L1a, L1b, L2a, L2b, i1a, i1b, i2a, i2b;
        L1 = compareSeqToGenome(S, N, L, i1, dirR, compRes);
        L2 = compareSeqToGenome(S, N, L, i2, dirR, compRes);
        L = std::min(L1, L2);
        L1a = L1; L1b = L1; i1a = i1; i1b = i1; L2a = L2; L2b = L2; i2a = i2; i2b = i2;
        i3 = i1; L3 = L1;
        while (i1 + 1 < i2) {
            i3 = medianUint2(i1, i2);
            L3 = compareSeqToGenome(S, N, L, i3, dirR, compRes);
            if (L3 == N) break;
            if (compRes) { if (L3 > L1) { L1b = L1a; L1a = L1; i1b = i1a; i1a = i1; } i1 = i3; L1 = L3; }
            else { if (L3 > L2) { L2b = L2a; L2a = L2; i2b = i2a; i2a = i2; } i2 = i3; L2 = L3; }
            L = std::min(L1, L2);
        }
        if (L3 < N) { if (L1 > L2) { i3 = i1; L3 = L1; } else { i3 = i2; L3 = L2; } }
        i1 = findMultRange(i3, L3, i1, L1, i1a, L1a, i1b, L1b, dirR, S);
        i2 = findMultRange(i3, L3, i2, L2, i2a, L2a, i2b, L2b, dirR, S);
        L = L3; indStartEnd[0] = i1; indStartEnd[1] = i2;
        return i2 - i1 + 1;
    }

    // ------------------------------------------------------------------ ReadAlign_storeAligns.cpp:10-160
    void storeAligns(u64 iDir, u64 Shift, u64 Nrep, u64 L, const u64 *ind, u64 iFrag) {
        if (Nrep > P.seedMultimapNmax) { if (Nrep < multNmin || multNmin == 0) { multNmin = Nrep; multNminL = L; } return; }
        nUM[Nrep == 1 ? 0 : 1] += Nrep; nA += Nrep;
        u64 rStart = iDir == 0 ? Shift : Shift + 1 - L;
        int iP; int nP = (int)PC.size();
        for (iP = nP - 1; iP >= 0; iP--) {
            if (PC[iP].rStart <= rStart) {
                if (PC[iP].rStart == rStart && PC[iP].L < L) continue;
                if (PC[iP].rStart == rStart && PC[iP].L == L) return;
                break;
            }
        }
        iP++;
        Seed s = { rStart, L, iDir, Nrep, ind[0], ind[1], iFrag };
        PC.insert(PC.begin() + iP, s);
        if (PC.size() > P.seedPerReadNmax) fatalSeeds = true;      // reference: exitWithError :46-51
        if (Nrep != 1) { if (Nrep < multNmin || multNmin == 0) { multNmin = Nrep; multNminL = L; } }   // :155
    }

    // ------------------------------------------------------------------ ReadAlign_maxMappableLength2strands.cpp:5-115
    u64 maxMappableLength2strands(u64 pieceStartIn, u64 pieceLengthIn, u64 iDir, u64 iSA1, u64 iSA2, u64 &maxLbest, u64 iFrag) {
        u64 Nrep = 0, indStartEnd[2] = {0, 0}, maxL;
        u64 D = g.gSAsparseD;
        std::vector<u64> NrepAll(D), maxLall(D); std::vector<u64> indAll(2 * D);
        maxLbest = 0;
        bool dirR = iDir == 0;
        u64 nD = std::min<u64>(pieceLengthIn, D);
        for (u64 iDist = 0; iDist < nD; iDist++) {
            u64 pieceStart; u64 pieceLength = pieceLengthIn - iDist;
            u64 Lmax = std::min<u64>(g.gSAindexNbases, pieceLength);
            u64 ind1 = 0;
            if (dirR) { pieceStart = pieceStartIn + iDist; for (u64 ii = 0; ii < Lmax; ii++) { ind1 <<= 2; ind1 += (u64)Read1[0][pieceStart + ii]; } }
            else { pieceStart = pieceStartIn - iDist; for (u64 ii = 0; ii < Lmax; ii++) { ind1 <<= 2; ind1 += 3 - (u64)Read1[0][pieceStart - ii]; } }
            u64 Lind = Lmax;
            while (Lind > 0) {
                iSA1 = SAiAt(g.genomeSAindexStart[Lind - 1] + ind1);
                if ((iSA1 & SAiMarkAbsentMaskC) == 0) break;
                --Lind; ind1 >>= 2;
            }
            if (Lind == 0) { NrepAll[iDist] = 0; indAll[2 * iDist] = indAll[2 * iDist + 1] = 0; maxLall[iDist] = 0; continue; }   // base absent from the genome (reference: out-of-bounds read)
            bool iSA2good = true;
            if (g.genomeSAindexStart[Lind - 1] + ind1 + 1 < g.genomeSAindexStart[Lind]) {
                iSA2 = SAiAt(g.genomeSAindexStart[Lind - 1] + ind1 + 1);
                if ((iSA2 & SAiMarkAbsentMaskC) == 0) iSA2 = (iSA2 & SAiMarkNmask) - 1;
                else { iSA2 = g.nSA - 1; iSA2good = false; }
            } else { iSA2 = g.nSA - 1; iSA2good = false; }
            bool iSA1noN = (iSA1 & SAiMarkNmaskC) == 0;
            if (Lind < g.gSAindexNbases && iSA1noN && iSA2good) {
                indStartEnd[0] = iSA1; indStartEnd[1] = iSA2; Nrep = iSA2 - iSA1 + 1; maxL = Lind;
            } else if (iSA1 == iSA2 && iSA1noN && iSA2good) {
                indStartEnd[0] = indStartEnd[1] = iSA1; Nrep = 1; bool cr;
                maxL = compareSeqToGenome(pieceStart, pieceLength, Lind, iSA1, dirR, cr);
            } else {
                maxL = (iSA2good && iSA1noN) ? Lind : 0;
                Nrep = maxMappableLength(pieceStart, pieceLength, iSA1 & SAiMarkNmask, iSA2, dirR, maxL, indStartEnd);
            }
            if (maxL + iDist > maxLbest) maxLbest = maxL + iDist;
            NrepAll[iDist] = Nrep; indAll[2 * iDist] = indStartEnd[0]; indAll[2 * iDist + 1] = indStartEnd[1]; maxLall[iDist] = maxL;
        }
        for (u64 iDist = 0; iDist < nD; iDist++)
            if (maxLall[iDist] + iDist == maxLbest && NrepAll[iDist] > 0)
                storeAligns(iDir, dirR ? pieceStartIn + iDist : pieceStartIn - iDist, NrepAll[iDist], maxLall[iDist], &indAll[2 * iDist], iFrag);
        return Nrep;
    }

    // ------------------------------------------------------------------ sjAlignSplit.cpp:3-15
    bool sjAlignSplit(u64 a1, u64 aLength, u64 &a1D, u64 &aLengthD, u64 &a1A, u64 &aLengthA, u64 &isj) {
        u64 sj1 = (a1 - g.sjGstart) % g.sjdbLength;
        if (sj1 < g.sjdbOverhang && sj1 + aLength > g.sjdbOverhang) {
            isj = (a1 - g.sjGstart) / g.sjdbLength;
            aLengthD = g.sjdbOverhang - sj1; aLengthA = aLength - aLengthD;
            a1D = g.sjDstart[isj] + sj1; a1A = g.sjAstart[isj];
            return true;
        }
        return false;
    }

    // ------------------------------------------------------------------ ReadAlign_createExtendWindowsWithAlign.cpp:7-84
    int createExtendWindowsWithAlign(u64 a1, u64 aStr) {
        u64 aBin = a1 >> P.winBinNbits, iBinLeft = aBin, iBinRight = aBin;
        uint16_t *wB = winBin[aStr].data();
        u64 iBin = (u64)-1, iWin = (u64)-1, iWinRight = (u64)-1;
        if (wB[aBin] == 0xFFFF) {
            bool flagMergeLeft = false;
            if (aBin > 0) {
                for (iBin = aBin - 1; iBin >= (aBin > P.winAnchorDistNbins ? aBin - P.winAnchorDistNbins : 0); --iBin) {
                    if (wB[iBin] < 0xFFFF) { flagMergeLeft = true; break; }
                    if (iBin == 0) break;
                }
                flagMergeLeft = flagMergeLeft && (g.chrBin[iBin >> P.winBinChrNbits] == g.chrBin[aBin >> P.winBinChrNbits]);
                if (flagMergeLeft) {
                    iWin = wB[iBin]; iBinLeft = WC[iWin].gStart;
                    for (u64 ii = iBin + 1; ii <= aBin; ii++) wB[ii] = (uint16_t)iWin;
                }
            }
            bool flagMergeRight = false;
            if (aBin + 1 < P.winBinN) {
                for (iBin = aBin + 1; iBin < std::min<u64>(aBin + P.winAnchorDistNbins + 1, P.winBinN); ++iBin)
                    if (wB[iBin] < 0xFFFF) { flagMergeRight = true; break; }
                flagMergeRight = flagMergeRight && (g.chrBin[iBin >> P.winBinChrNbits] == g.chrBin[aBin >> P.winBinChrNbits]);
                if (flagMergeRight) {
                    while (wB[iBin] == wB[iBin + 1]) ++iBin;
                    iBinRight = iBin; iWinRight = wB[iBin];
                    if (!flagMergeLeft) iWin = wB[iBin];
                    for (u64 ii = aBin; ii <= iBin; ii++) wB[ii] = (uint16_t)iWin;
                }
            }
            if (!flagMergeLeft && !flagMergeRight) {
                wB[aBin] = (uint16_t)(iWin = nW);
                if (WC.size() <= iWin) WC.resize(iWin + 1);
                WC[iWin].chr = g.chrBin[aBin >> P.winBinChrNbits]; WC[iWin].str = aStr;
                WC[iWin].gEnd = WC[iWin].gStart = aBin;
                ++nW;
                if (nW >= P.alignWindowsPerReadNmax) { nW = P.alignWindowsPerReadNmax - 1; windowsLimit = true; return 1; }
            } else {
                WC[iWin].gStart = iBinLeft; WC[iWin].gEnd = iBinRight;
                if (flagMergeLeft && flagMergeRight) { WC[iWinRight].gStart = 1; WC[iWinRight].gEnd = 0; }
            }
        }
        return 0;
    }

    // ------------------------------------------------------------------ ReadAlign_assignAlignToWindow.cpp:6-130
    void assignAlignToWindow(u64 a1, u64 aLength, u64 aStr, u64 aNrep, u64 aFrag, u64 aRstart, bool aAnchor, u64 sjA) {
        u64 iW = winBin[aStr][a1 >> P.winBinNbits];
        if (iW == 0xFFFF || (!aAnchor && aLength < WALrec[iW])) return;
        std::vector<WAlign> &W = WA[iW];
        {
            u64 iA;
            for (iA = 0; iA < W.size(); iA++) {
                if (aFrag == W[iA].iFrag && W[iA].sjA == sjA && a1 + W[iA].rStart == W[iA].gStart + aRstart
                    && ((aRstart >= W[iA].rStart && aRstart < W[iA].rStart + W[iA].L)
                        || (aRstart + aLength >= W[iA].rStart && aRstart + aLength < W[iA].rStart + W[iA].L))) break;
            }
            if (iA < W.size()) {
                if (aLength > W[iA].L) {
                    u64 iA0;
                    for (iA0 = 0; iA0 < W.size(); iA0++) if (iA0 != iA && aRstart < W[iA0].rStart) break;
                    if (iA0 > iA) --iA0;
                    if (iA0 < iA) { for (u64 i = iA; i > iA0; i--) W[i] = W[i - 1]; }
                    else if (iA0 > iA) { for (u64 i = iA; i < iA0; i++) W[i] = W[i + 1]; }
                    WAlign w = { aLength, aRstart, a1, aNrep, (u64)(aAnchor ? 1 : 0), aFrag, sjA };
                    W[iA0] = w;
                }
                return;
            }
        }
        if (W.size() == P.seedPerWindowNmax) {
            WALrec[iW] = Lread + 1;
            for (u64 iA = 0; iA < W.size(); iA++) if (W[iA].anchor != 1) WALrec[iW] = std::min(WALrec[iW], W[iA].L);
            if (WALrec[iW] == Lread + 1) { tooManyAnchors = true; nW = 0; return; }
            if (!aAnchor && aLength < WALrec[iW]) return;
            u64 iA1 = 0;
            for (u64 iA = 0; iA < W.size(); iA++) if (W[iA].anchor == 1 || W[iA].L > WALrec[iW]) { W[iA1] = W[iA]; iA1++; }
            W.resize(iA1);
        }
        if (aAnchor || aLength > WALrec[iW]) {
            u64 iA;
            for (iA = 0; iA < W.size(); iA++) if (aRstart < W[iA].rStart) break;
            WAlign w = { aLength, aRstart, a1, aNrep, (u64)(aAnchor ? 1 : 0), aFrag, sjA };
            W.insert(W.begin() + iA, w);
        }
    }

    // ------------------------------------------------------------------ binarySearch2.cpp:3-43
    static int binarySearch2(u64 x, u64 y, const u64 *X, const u64 *Y, int N) {
        if (N == 0 || x > X[N - 1] || x < X[0]) return -1;
        int i1 = 0, i2 = N - 1, i3 = N / 2;
        while (i2 > i1 + 1) { i3 = (i1 + i2) / 2; if (X[i3] > x) i2 = i3; else i1 = i3; }
        if (x == X[i1]) i3 = i1; else if (x == X[i2]) i3 = i2; else return -1;
        for (int jj = i3; jj >= 0; jj--) { if (x != X[jj]) break; else if (y == Y[jj]) return jj; }
        for (int jj = i3; jj < N; jj++) { if (x != X[jj]) return -1; else if (y == Y[jj]) return jj; }
        return -2;
    }

    // ------------------------------------------------------------------ blocksOverlap.cpp:3-40
    static u64 blocksOverlap(const Tr &t1, const Tr &t2) {
        u64 i1 = 0, i2 = 0, nOverlap = 0;
        while (i1 < t1.nExons && i2 < t2.nExons) {
            u64 rs1 = t1.ex[i1][EX_R], rs2 = t2.ex[i2][EX_R];
            u64 re1 = rs1 + t1.ex[i1][EX_L], re2 = rs2 + t2.ex[i2][EX_L];
            u64 gs1 = t1.ex[i1][EX_G], gs2 = t2.ex[i2][EX_G];
            if (rs1 >= re2) i2++;
            else if (rs2 >= re1) i1++;
            else if (gs1 - rs1 != gs2 - rs2) { if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
            else { nOverlap += std::min(re1, re2) - std::max(rs1, rs2); if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
        }
        return nOverlap;
    }

    // ------------------------------------------------------------------ extendAlign.cpp:6-93
    bool extendAlign(const char *Rr, u64 rStart, u64 gStart, int dR, int dG, u64 L, u64 Lprev, u64 nMMprev, u64 nMMmax,
                     double pMMmax, bool extendToEnd, Tr *trA) {
        cnt[C_nExtendCalls]++;
        int iS, iG; int Score = 0, nMatch = 0, nMM = 0;
        trA->maxScore = 0;
        const char *Rp = Rr + rStart;
        if (extendToEnd) {
            int iExt;
            for (iExt = 0; iExt < (int)L; iExt++) {
                iS = dR * iExt; iG = dG * iExt;
                if ((gStart + iG) == (u64)(-1) || Gs(gStart + iG) == 5) {
                    trA->extendL = 0; trA->maxScore = -999999999; trA->nMatch = 0; trA->nMM = nMMmax + 1; return true;
                }
                if (Rp[iS] == STARAMD_SPACER_BASE) break;
                if (Rp[iS] > 3 || G[(i64)(gStart + iG)] > 3) continue;
                if (G[(i64)(gStart + iG)] == Rp[iS]) { nMatch++; Score += 1; } else { nMM++; Score -= 1; }
            }
            if (iExt > 0) { trA->extendL = iExt; trA->maxScore = Score; trA->nMatch = nMatch; trA->nMM = nMM; return true; }
            return false;
        }
        for (int i = 0; i < (int)L; i++) {
            iS = dR * i; iG = dG * i;
            if ((gStart + iG) == (u64)(-1)) break;
            char gc = Gs(gStart + iG);
            if (gc == 5 || Rp[iS] == STARAMD_SPACER_BASE) break;
            if (Rp[iS] > 3 || gc > 3) continue;
            if (gc == Rp[iS]) {
                nMatch++; Score += 1;
                if (Score > trA->maxScore) {
                    if (nMM + nMMprev <= std::min(pMMmax * double(Lprev + i + 1), double(nMMmax))) {
                        trA->extendL = i + 1; trA->maxScore = Score; trA->nMatch = nMatch; trA->nMM = nMM;
                    }
                }
            } else {
                if (nMM + nMMprev >= std::min(pMMmax * double(Lprev + L), double(nMMmax))) break;
                nMM++; Score -= 1;
            }
        }
        return trA->extendL > 0;
    }

    // ------------------------------------------------------------------ stitchAlignToTranscript.cpp:9-415
    int stitchAlignToTranscript(u64 rAend, u64 gAend, u64 rBstart, u64 gBstart, u64 L, u64 iFragB, u64 sjAB, const char *R, Tr *trA) {
        cnt[C_nStitchCalls]++;
        if (trA->nExons >= STARAMD_MAX_N_EXONS) return -1000010;
        int Score = 0;
        u64 ne = trA->nExons;
        if (sjAB != (u64)-1 && trA->ex[ne - 1][EX_sjA] == sjAB && trA->ex[ne - 1][EX_iFrag] == iFragB && rBstart == rAend + 1 && gAend + 1 < gBstart) {
            if (g.sjdbMotif[sjAB] == 0 && (L <= g.sjdbShiftRight[sjAB] || trA->ex[ne - 1][EX_L] <= g.sjdbShiftLeft[sjAB])) return -1000006;
            trA->ex[ne][EX_L] = L; trA->ex[ne][EX_R] = rBstart; trA->ex[ne][EX_G] = gBstart;
            trA->canonSJ[ne - 1] = g.sjdbMotif[sjAB];
            trA->shiftSJ[ne - 1][0] = g.sjdbShiftLeft[sjAB]; trA->shiftSJ[ne - 1][1] = g.sjdbShiftRight[sjAB];
            trA->sjAnnot[ne - 1] = 1; trA->sjStr[ne - 1] = g.sjdbStrand[sjAB];
            trA->nExons++; trA->nMatch += L;
            Score += (int)L; Score += P.sjdbScore;
        } else {
            trA->sjAnnot[ne - 1] = 0; trA->sjStr[ne - 1] = 0;
            if (trA->ex[ne - 1][EX_iFrag] == iFragB) {
                u64 gBend = gBstart + L - 1, rBend = rBstart + L - 1;
                if (rBend <= rAend) return -1000001;
                if (gBend <= gAend) return -1000002;
                if (rBstart <= rAend) { gBstart += rAend - rBstart + 1; rBstart = rAend + 1; L = rBend - rBstart + 1; }
                for (u64 ii = rBstart; ii <= rBend; ii++) Score += 1;
                int gGap = (int)(gBstart - gAend - 1);
                int rGap = (int)(rBstart - rAend - 1);
                u64 nMatch = L, nMM = 0, Del = 0, Ins = 0, nIns = 0, nDel = 0;
                int jR = 0, jCan = 999;
                u64 gBstart1 = gBstart - rGap - 1;
                if (gGap == 0 && rGap == 0) {
                } else if (gGap > 0 && rGap > 0 && rGap == gGap) {
                    for (int ii = 1; ii <= rGap; ii++) {
                        char gc = Gs(gAend + ii);
                        if (gc < 4 && R[rAend + ii] < 4) { if (R[rAend + ii] == gc) { Score += 1; nMatch++; } else { Score -= 1; nMM++; } }
                    }
                } else if (gGap > rGap) {
                    nDel = 1; Del = gGap - rGap;
                    if (Del > P.alignIntronMax && P.alignIntronMax > 0) return -1000003;
                    int Score1 = 0, jR1 = 1;
                    do {
                        jR1--;
                        if (R[rAend + jR1] != Gs(gBstart1 + jR1) && G[(i64)(gBstart1 + jR1)] < 4 && R[rAend + jR1] == Gs(gAend + jR1)) Score1 -= 1;
                    } while (Score1 + P.scoreStitchSJshift >= 0 && int(trA->ex[ne - 1][EX_L]) + jR1 > 1);
                    int maxScore2 = -999999; Score1 = 0; int jPen = 0;
                    do {
                        char ra = R[rAend + jR1], gA = Gs(gAend + jR1), gB = Gs(gBstart1 + jR1);
                        if (ra == gA && ra != gB) Score1 += 1;
                        if (ra != gA && ra == gB) Score1 -= 1;
                        int jCan1 = -1, jPen1 = 0, Score2 = Score1;
                        if (Del >= P.alignIntronMin) {
                            char d1 = Gs(gAend + jR1 + 1), d2 = Gs(gAend + jR1 + 2), a1 = Gs(gBstart1 + jR1 - 1), a2 = gB;
                            if (d1 == 2 && d2 == 3 && a1 == 0 && a2 == 2) jCan1 = 1;
                            else if (d1 == 1 && d2 == 3 && a1 == 0 && a2 == 1) jCan1 = 2;
                            else if (d1 == 2 && d2 == 1 && a1 == 0 && a2 == 2) { jCan1 = 3; jPen1 = P.scoreGapGCAG; }
                            else if (d1 == 1 && d2 == 3 && a1 == 2 && a2 == 1) { jCan1 = 4; jPen1 = P.scoreGapGCAG; }
                            else if (d1 == 0 && d2 == 3 && a1 == 0 && a2 == 1) { jCan1 = 5; jPen1 = P.scoreGapATAC; }
                            else if (d1 == 2 && d2 == 3 && a1 == 0 && a2 == 3) { jCan1 = 6; jPen1 = P.scoreGapATAC; }
                            else { jCan1 = 0; jPen1 = P.scoreGapNoncan; }
                            Score2 += jPen1;
                        }
                        if (maxScore2 < Score2) { maxScore2 = Score2; jR = jR1; jCan = jCan1; jPen = jPen1; }
                        jR1++;
                    } while (jR1 < int(rBend) - int(rAend));
                    u64 jjL = 0, jjR = 0;
                    while (gAend + jR >= jjL && Gs(gAend - jjL + jR) == Gs(gBstart1 - jjL + jR) && G[(i64)(gAend - jjL + jR)] < 4 && jjL <= 255) jjL++;
                    while (gAend + jjR + jR + 1 < g.nGenome && Gs(gAend + jjR + jR + 1) == Gs(gBstart1 + jjR + jR + 1) && G[(i64)(gAend + jjR + jR + 1)] < 4 && jjR <= 255) jjR++;
                    if (jCan <= 0) {
                        jR -= (int)jjL;
                        if (int(trA->ex[ne - 1][EX_L]) + jR < 1) return -1000005;
                        jjR += jjL; jjL = 0;
                    }
                    for (int ii = std::min(1, jR + 1); ii <= std::max(rGap, jR); ii++) {
                        u64 g1 = (ii <= jR) ? (gAend + ii) : (gBstart1 + ii);
                        char gc = Gs(g1);
                        if (gc < 4 && R[rAend + ii] < 4) {
                            if (R[rAend + ii] == gc) { if (ii >= 1 && ii <= rGap) { Score += 1; nMatch++; } }
                            else { Score -= 1; nMM++; if (ii < 1 || ii > rGap) { Score -= 1; nMatch--; } }
                        }
                    }
                    if (g.sjdbN > 0) {
                        u64 jS = gAend + jR + 1, jE = gBstart1 + jR;
                        int sjdbInd = binarySearch2(jS, jE, g.sjdbStart, g.sjdbEnd, (int)g.sjdbN);
                        if (sjdbInd < 0) {
                            if (Del >= P.alignIntronMin) Score += P.scoreGap + jPen;
                            else { Score += (int)Del * P.scoreDelBase + P.scoreDelOpen; jCan = -1; trA->sjAnnot[ne - 1] = 0; }
                        } else {
                            jCan = g.sjdbMotif[sjdbInd];
                            if (g.sjdbMotif[sjdbInd] == 0) {
                                if (L <= g.sjdbShiftLeft[sjdbInd] || trA->ex[ne - 1][EX_L] <= g.sjdbShiftLeft[sjdbInd]) return -1000006;
                                jR += (int)g.sjdbShiftLeft[sjdbInd];
                                if (rAend + jR >= rBend) return -1000006;
                                jjL = g.sjdbShiftLeft[sjdbInd]; jjR = g.sjdbShiftRight[sjdbInd];
                            }
                            trA->sjAnnot[ne - 1] = 1; trA->sjStr[ne - 1] = g.sjdbStrand[sjdbInd];
                            Score += P.sjdbScore;
                        }
                    } else {
                        if (Del >= P.alignIntronMin) Score += P.scoreGap + jPen;
                        else { Score += (int)Del * P.scoreDelBase + P.scoreDelOpen; jCan = -1; trA->sjAnnot[ne - 1] = 0; }
                    }
                    trA->shiftSJ[ne - 1][0] = jjL; trA->shiftSJ[ne - 1][1] = jjR; trA->canonSJ[ne - 1] = jCan;
                    if (trA->sjAnnot[ne - 1] == 0) trA->sjStr[ne - 1] = (jCan > 0) ? (uint8_t)(2 - jCan % 2) : 0;
                } else if (rGap > gGap) {
                    Ins = rGap - gGap; nIns = 1;
                    if (gGap == 0) jR = 0;
                    else if (gGap < 0) { jR = 0; for (int ii = 0; ii < -gGap; ii++) Score -= 1; }
                    else {
                        int Score1 = 0, maxScore1 = 0;
                        for (int jR1 = 1; jR1 <= gGap; jR1++) {
                            char gc = Gs(gAend + jR1);
                            if (gc < 4) { Score1 += (R[rAend + jR1] == gc) ? 1 : -1; Score1 += (R[rAend + Ins + jR1] == gc) ? -1 : +1; }
                            if (Score1 > maxScore1 || (Score1 == maxScore1 && P.alignInsertionFlushRight)) { maxScore1 = Score1; jR = jR1; }
                        }
                        for (int ii = 1; ii <= gGap; ii++) {
                            u64 r1 = rAend + ii + (ii <= jR ? 0 : Ins);
                            char gc = Gs(gAend + ii);
                            if (gc < 4 && R[r1] < 4) { if (R[r1] == gc) { Score += 1; nMatch++; } else { Score -= 1; nMM++; } }
                        }
                    }
                    if (P.alignInsertionFlushRight) {
                        for (; jR < (int)rBend - (int)rAend - (int)Ins; jR++) if (R[rAend + jR + 1] != Gs(gAend + jR + 1) || G[(i64)(gAend + jR + 1)] == 4) break;
                        if (jR == (int)rBend - (int)rAend - (int)Ins) return -1000009;
                    }
                    Score += (int)Ins * P.scoreInsBase + P.scoreInsOpen;
                    jCan = -2;
                }
                if ((trA->nMM + nMM) <= mmMaxTotal && (jCan < 0 || (jCan < 7 && nMM <= (u64)P.alignSJstitchMismatchNmax[(jCan + 1) / 2]))) {
                    trA->nMM += nMM; trA->nMatch += nMatch;
                    if (Del >= P.alignIntronMin) { trA->nGap += nDel; trA->lGap += Del; } else { trA->nDel += nDel; trA->lDel += Del; }
                    if (Del == 0 && Ins == 0) trA->ex[ne - 1][EX_L] += rBend - rAend;
                    else if (Del > 0) {
                        trA->ex[ne - 1][EX_L] += jR;
                        trA->ex[ne][EX_L] = rBend - rAend - jR; trA->ex[ne][EX_R] = rAend + jR + 1; trA->ex[ne][EX_G] = gBstart1 + jR + 1;
                        trA->nExons++;
                    } else if (Ins > 0) {
                        trA->nIns += nIns; trA->lIns += Ins;
                        trA->ex[ne - 1][EX_L] += jR;
                        trA->ex[ne][EX_L] = rBend - rAend - jR - Ins; trA->ex[ne][EX_R] = rAend + jR + Ins + 1; trA->ex[ne][EX_G] = gAend + 1 + jR;
                        trA->canonSJ[ne - 1] = -2; trA->sjAnnot[ne - 1] = 0;
                        trA->nExons++;
                    }
                } else return -1000007;
            } else if (gBstart + trA->ex[0][EX_R] + P.alignEndsProtrudeNbasesMax >= trA->ex[0][EX_G] || trA->ex[0][EX_G] < trA->ex[0][EX_R]) {
                if (P.alignMatesGapMax > 0 && gBstart > trA->ex[ne - 1][EX_G] + trA->ex[ne - 1][EX_L] + P.alignMatesGapMax) return -1000004;
                for (u64 ii = rBstart; ii < rBstart + L; ii++) Score += 1;
                Tr trExtend; memset(&trExtend, 0, sizeof(trExtend));
                trExtend.reset();
                if (extendAlign(R, rAend + 1, gAend + 1, 1, 1, STARAMD_READ_LEN_MAX, trA->nMatch, trA->nMM, mmMaxTotal, P.outFilterMismatchNoverLmax,
                                P.alignEndsTypeExt[trA->ex[ne - 1][EX_iFrag]][1], &trExtend)) {
                    trA->add(trExtend); Score += trExtend.maxScore; trA->ex[ne - 1][EX_L] += trExtend.extendL;
                }
                trA->ex[ne][EX_R] = rBstart; trA->ex[ne][EX_G] = gBstart; trA->ex[ne][EX_L] = L; trA->nMatch += L;
                trExtend.reset();
                u64 extlen = P.alignEndsTypeExt[iFragB][1] ? STARAMD_READ_LEN_MAX : gBstart - trA->ex[0][EX_G] + trA->ex[0][EX_R];
                if (extendAlign(R, rBstart - 1, gBstart - 1, -1, -1, extlen, trA->nMatch, trA->nMM, mmMaxTotal, P.outFilterMismatchNoverLmax,
                                P.alignEndsTypeExt[iFragB][1], &trExtend)) {
                    trA->add(trExtend); Score += trExtend.maxScore;
                    trA->ex[ne][EX_R] -= trExtend.extendL; trA->ex[ne][EX_G] -= trExtend.extendL; trA->ex[ne][EX_L] += trExtend.extendL;
                }
                trA->canonSJ[ne - 1] = -3; trA->sjAnnot[ne - 1] = 0;
                trA->nExons++;
            } else return -1000008;
        }
        trA->ex[trA->nExons - 1][EX_iFrag] = iFragB; trA->ex[trA->nExons - 1][EX_sjA] = sjAB;
        return Score;
    }

    // ------------------------------------------------------------------ stitchWindowAligns.cpp:8-353
    void stitchWindowAligns(u64 iA, u64 nA, int Score, u64 tR2, u64 tG2, Tr trA, const std::vector<WAlign> &W, const char *R, std::vector<Tr> &wTr) {
        cnt[C_nNodes]++;
        if (iA >= nA && tR2 == 0) return;
        if (iA >= nA) {
            cnt[C_nLeaves]++;
            Tr trAstep1; memset(&trAstep1, 0, sizeof(trAstep1));
            int vOrder[2];
            if (trA.roStr == 0) { vOrder[0] = 0; vOrder[1] = 1; } else { vOrder[0] = 1; vOrder[1] = 0; }   // EXTEND_ORDER==1
            for (int iOrd = 0; iOrd < 2; iOrd++) {
                if (vOrder[iOrd] == 0) {
                    if (trA.rStart > 0) {
                        trAstep1.reset();
                        u64 imate = trA.ex[0][EX_iFrag];
                        if (extendAlign(R, trA.rStart - 1, trA.gStart - 1, -1, -1, trA.rStart, tR2 - trA.rStart + 1, trA.nMM, mmMaxTotal,
                                        P.outFilterMismatchNoverLmax, P.alignEndsTypeExt[imate][(int)(trA.Str != imate)], &trAstep1)) {
                            trA.add(trAstep1); Score += trAstep1.maxScore;
                            trA.ex[0][EX_R] = trA.rStart = trA.rStart - trAstep1.extendL;
                            trA.ex[0][EX_G] = trA.gStart = trA.gStart - trAstep1.extendL;
                            trA.ex[0][EX_L] += trAstep1.extendL;
                        }
                    }
                } else {
                    if (tR2 < Lread) {
                        trAstep1.reset();
                        u64 imate = trA.ex[trA.nExons - 1][EX_iFrag];
                        if (extendAlign(R, tR2 + 1, tG2 + 1, +1, +1, Lread - tR2 - 1, tR2 - trA.rStart + 1, trA.nMM, mmMaxTotal,
                                        P.outFilterMismatchNoverLmax, P.alignEndsTypeExt[imate][(int)(imate == trA.Str)], &trAstep1)) {
                            trA.add(trAstep1); Score += trAstep1.maxScore;
                            tR2 += trAstep1.extendL; tG2 += trAstep1.extendL;
                            trA.ex[trA.nExons - 1][EX_L] += trAstep1.extendL;
                        }
                    }
                }
            }
            u64 ne = trA.nExons;
            if (!P.alignSoftClipAtReferenceEnds &&
                ((trA.ex[ne - 1][EX_G] + Lread - trA.ex[ne - 1][EX_R]) > (g.chrStart[trA.Chr] + g.chrLength[trA.Chr]) ||
                 trA.ex[0][EX_G] < (g.chrStart[trA.Chr] + trA.ex[0][EX_R]))) return;
            trA.rLength = 0;
            for (u64 isj = 0; isj < ne; isj++) trA.rLength += trA.ex[isj][EX_L];
            trA.gLength = tG2 + 1 - trA.gStart;
            for (u64 isj = 0; isj + 1 < ne; isj++) {
                if (trA.canonSJ[isj] >= 0) {
                    if (trA.sjAnnot[isj] == 1) {
                        if ((trA.ex[isj][EX_L] < P.alignSJDBoverhangMin && (isj == 0 || trA.canonSJ[isj - 1] == -3 || (trA.sjAnnot[isj - 1] == 0 && trA.canonSJ[isj - 1] >= 0)))
                            || (trA.ex[isj + 1][EX_L] < P.alignSJDBoverhangMin && (isj == ne - 2 || trA.canonSJ[isj + 1] == -3 || (trA.sjAnnot[isj + 1] == 0 && trA.canonSJ[isj + 1] >= 0)))) return;
                    } else {
                        if (trA.ex[isj][EX_L] < P.alignSJoverhangMin + trA.shiftSJ[isj][0] || trA.ex[isj + 1][EX_L] < P.alignSJoverhangMin + trA.shiftSJ[isj][1]) return;
                    }
                }
            }
            if (ne > 1 && trA.sjAnnot[ne - 2] == 1 && trA.ex[ne - 1][EX_L] < P.alignSJDBoverhangMin) return;
            u64 sjN = 0;
            trA.intronMotifs[0] = 0; trA.intronMotifs[1] = 0; trA.intronMotifs[2] = 0;
            for (u64 iex = 0; iex + 1 < ne; iex++) if (trA.canonSJ[iex] >= 0) { sjN++; trA.intronMotifs[trA.sjStr[iex]]++; }
            if (trA.intronMotifs[1] > 0 && trA.intronMotifs[2] == 0) trA.sjMotifStrand = 1;
            else if (trA.intronMotifs[1] == 0 && trA.intronMotifs[2] > 0) trA.sjMotifStrand = 2;
            else trA.sjMotifStrand = 0;
            if (trA.intronMotifs[1] > 0 && trA.intronMotifs[2] > 0 && P.outFilterIntronStrandsRemoveInconsistent) return;
            if (sjN > 0 && trA.sjMotifStrand == 0 && P.outSAMstrandFieldIntronMotif) return;
            if (P.outFilterIntronMotifs == 1) { for (u64 iex = 0; iex + 1 < ne; iex++) if (trA.canonSJ[iex] == 0) return; }
            else if (P.outFilterIntronMotifs == 2) { for (u64 iex = 0; iex + 1 < ne; iex++) if (trA.canonSJ[iex] == 0 && trA.sjAnnot[iex] == 0) return; }
            {
                u64 nsj = 0, exl = 0;
                for (u64 iex = 0; iex < ne; iex++) {
                    exl += trA.ex[iex][EX_L];
                    if (iex == ne - 1 || trA.canonSJ[iex] == -3) {
                        if (nsj > 0 && (exl < P.alignSplicedMateMapLmin || exl < (u64)(P.alignSplicedMateMapLminOverLmate * readLength[trA.ex[iex][EX_iFrag]]))) return;
                        exl = 0; nsj = 0;
                    } else if (trA.canonSJ[iex] >= 0) nsj++;
                }
            }
            if (P.outFilterBySJoutStage == 2) {                 // :169-177: unannotated junctions must be in the filtered novel set
                for (u64 iex = 0; iex + 1 < ne; iex++) {
                    if (trA.canonSJ[iex] >= 0 && trA.sjAnnot[iex] == 0) {
                        u64 jS = trA.ex[iex][EX_G] + trA.ex[iex][EX_L], jE = trA.ex[iex + 1][EX_G] - 1;
                        if (binarySearch2(jS, jE, sjNovelStart.data(), sjNovelEnd.data(), (int)sjNovelStart.size()) < 0) return;
                    }
                }
            }
            if (trA.ex[0][EX_iFrag] != trA.ex[ne - 1][EX_iFrag]) {
                if (trA.ex[ne - 1][EX_G] + trA.ex[ne - 1][EX_L] <= trA.ex[0][EX_G]) return;
                u64 iexM2 = ne;
                for (u64 iex = 0; iex + 1 < ne; iex++) if (trA.canonSJ[iex] == -3) { iexM2 = iex + 1; break; }
                if (trA.ex[iexM2 - 1][EX_G] + trA.ex[iexM2 - 1][EX_L] > trA.ex[iexM2][EX_G]) {
                    if (trA.ex[0][EX_G] > trA.ex[iexM2][EX_G] + trA.ex[0][EX_R] + P.alignEndsProtrudeNbasesMax) return;
                    if (trA.ex[iexM2 - 1][EX_G] + trA.ex[iexM2 - 1][EX_L] > trA.ex[ne - 1][EX_G] + Lread - trA.ex[ne - 1][EX_R] + P.alignEndsProtrudeNbasesMax) return;
                    u64 iex1 = 1, iex2 = iexM2 + 1;
                    for (; iex1 < iexM2; iex1++) if (trA.ex[iex1][EX_G] >= trA.ex[iex2 - 1][EX_G] + trA.ex[iex2 - 1][EX_L]) break;
                    while (iex1 < iexM2 && iex2 < ne) {
                        if (trA.canonSJ[iex1 - 1] < 0) { iex1++; continue; }
                        if (trA.canonSJ[iex2 - 1] < 0) { iex2++; continue; }
                        if ((trA.ex[iex1][EX_G] != trA.ex[iex2][EX_G]) || ((trA.ex[iex1 - 1][EX_G] + trA.ex[iex1 - 1][EX_L]) != (trA.ex[iex2 - 1][EX_G] + trA.ex[iex2 - 1][EX_L]))) return;
                        iex1++; iex2++;
                    }
                }
            }
            if (P.scoreGenomicLengthLog2scale != 0) {
                Score += int(std::ceil(std::log2((double)(trA.ex[ne - 1][EX_G] + trA.ex[ne - 1][EX_L] - trA.ex[0][EX_G])) * P.scoreGenomicLengthLog2scale - 0.5));
                Score = std::max(0, Score);
            }
            trA.roStart = (trA.roStr == 0) ? trA.rStart : Lread - trA.rStart - trA.rLength;
            trA.maxScore = Score;
            if (trA.ex[0][EX_iFrag] == trA.ex[ne - 1][EX_iFrag]) { trA.iFrag = (int)trA.ex[0][EX_iFrag]; maxScoreMate[trA.iFrag] = std::max(maxScoreMate[trA.iFrag], Score); }
            else trA.iFrag = -1;
            // variationAdjust returns 0 without a VCF (Transcript_variationAdjust.cpp:8-11)
            if (Score + P.outFilterMultimapScoreRange >= wTr[0].maxScore
                || (trA.iFrag >= 0 && Score + P.outFilterMultimapScoreRange >= maxScoreMate[trA.iFrag]) || P.chimSegmentMinPositive) {
                u64 iTr = 0;
                trA.mappedLength = 0;
                for (u64 iex = 0; iex < ne; iex++) trA.mappedLength += trA.ex[iex][EX_L];
                u64 &nWinTr = nWinTrCur;
                while (iTr < nWinTr) {
                    u64 nOverlap = blocksOverlap(trA, wTr[iTr]);
                    u64 uNew = trA.mappedLength - nOverlap, uOld = wTr[iTr].mappedLength - nOverlap;
                    if (uNew == 0 && Score < wTr[iTr].maxScore) break;
                    else if (uOld == 0) { Tr t = wTr[iTr]; for (u64 ii = iTr + 1; ii < nWinTr; ii++) wTr[ii - 1] = wTr[ii]; nWinTr--; wTr[nWinTr] = t; }
                    else if (uOld > 0 && (uNew > 0 || Score >= wTr[iTr].maxScore)) iTr++;
                }
                if (iTr == nWinTr) {
                    for (iTr = 0; iTr < nWinTr; iTr++) if (Score > wTr[iTr].maxScore || (Score == wTr[iTr].maxScore && trA.gLength < wTr[iTr].gLength)) break;
                    for (int ii = (int)nWinTr; ii > int(iTr); ii--) wTr[ii] = wTr[ii - 1];
                    wTr[iTr] = trA;
                    if (nWinTr < P.alignTranscriptsPerWindowNmax) nWinTr++;
                }
            }
            return;
        }
        int dScore = 0;
        Tr trAi = trA;
        if (trA.nExons > 0) {
            dScore = stitchAlignToTranscript(tR2, tG2, W[iA].rStart, W[iA].gStart, W[iA].L, W[iA].iFrag, W[iA].sjA, R, &trAi);
        } else {
            trAi.ex[0][EX_R] = trAi.rStart = W[iA].rStart; trAi.ex[0][EX_G] = trAi.gStart = W[iA].gStart;
            trAi.ex[0][EX_L] = W[iA].L; trAi.ex[0][EX_iFrag] = W[iA].iFrag; trAi.ex[0][EX_sjA] = W[iA].sjA;
            trAi.nExons = 1;
            dScore += (int)W[iA].L;
            trAi.nMatch = W[iA].L;
        }
        if (dScore > -1000000) {
            if (W[iA].nrep == 1) trAi.nUnique++;
            if (W[iA].anchor > 0) trAi.nAnchor++;
            stitchWindowAligns(iA + 1, nA, Score + dScore, W[iA].rStart + W[iA].L - 1, W[iA].gStart + W[iA].L - 1, trAi, W, R, wTr);
        }
        // WA_Anchor is never 2 (dead last-anchor protection): always explore exclusion
        stitchWindowAligns(iA + 1, nA, Score, tR2, tG2, trA, W, R, wTr);
    }
    u64 nWinTrCur;

    // ------------------------------------------------------------------ ReadAlign_stitchPieces.cpp:12-350
    // returns status bits; fills trAll
    uint32_t stitchPieces(int &trBestW) {
        uint32_t status = 0;
        std::fill(winBin[0].begin(), winBin[0].end(), 0xFFFF);
        std::fill(winBin[1].begin(), winBin[1].end(), 0xFFFF);
        nW = 0; WC.clear();
        for (u64 iP = 0; iP < PC.size(); iP++) {
            if (PC[iP].nrep <= P.winAnchorMultimapNmax) {
                u64 aDir = PC[iP].dir, aLength = PC[iP].L;
                for (u64 iSA = PC[iP].saStart; iSA <= PC[iP].saEnd; iSA++) {
                    cnt[C_nSAenum]++;
                    u64 a1 = SAat(iSA); u64 aStr = a1 >> g.GstrandBit; a1 &= GstrandMask;
                    if (aDir == 1 && aStr == 0) aStr = 1;
                    else if (aDir == 0 && aStr == 1) a1 = g.nGenome - (aLength + a1);
                    else if (aDir == 1 && aStr == 1) { aStr = 0; a1 = g.nGenome - (aLength + a1); }
                    if (a1 >= g.sjGstart) {
                        u64 a1D, aLengthD, a1A, aLengthA, sj1;
                        if (sjAlignSplit(a1, aLength, a1D, aLengthD, a1A, aLengthA, sj1)) {
                            if (createExtendWindowsWithAlign(a1D, aStr)) break;
                            if (createExtendWindowsWithAlign(a1A, aStr)) break;
                        }
                    } else if (createExtendWindowsWithAlign(a1, aStr)) break;
                }
            }
        }
        WC.resize(nW);
        for (u64 iWin = 0; iWin < nW; iWin++) {
            if (WC[iWin].gStart <= WC[iWin].gEnd) {
                u64 wb = WC[iWin].gStart;
                for (u64 ii = 0; ii < P.winFlankNbins && wb > 0 && g.chrBin[(wb - 1) >> P.winBinChrNbits] == WC[iWin].chr; ii++) { wb--; winBin[WC[iWin].str][wb] = (uint16_t)iWin; }
                WC[iWin].gStart = wb;
                wb = WC[iWin].gEnd;
                for (u64 ii = 0; ii < P.winFlankNbins && wb + 1 < P.winBinN && g.chrBin[(wb + 1) >> P.winBinChrNbits] == WC[iWin].chr; ii++) { wb++; winBin[WC[iWin].str][wb] = (uint16_t)iWin; }
                WC[iWin].gEnd = wb;
            }
        }
        WA.assign(nW, std::vector<WAlign>()); WALrec.assign(nW, 0);
        cnt[C_nWindows] += nW;
        u64 nWpassB = nW;
        for (u64 iP = 0; iP < PC.size(); iP++) {
            u64 aNrep = PC[iP].nrep, aFrag = PC[iP].iFrag, aLength = PC[iP].L, aDir = PC[iP].dir;
            bool aAnchor = aNrep <= P.winAnchorMultimapNmax;
            for (u64 iSA = PC[iP].saStart; iSA <= PC[iP].saEnd; iSA++) {
                cnt[C_nSAenum]++;
                u64 a1 = SAat(iSA); u64 aStr = a1 >> g.GstrandBit; a1 &= GstrandMask;
                u64 aRstart = PC[iP].rStart;
                if (aDir == 1 && aStr == 0) { aStr = 1; aRstart = Lread - (aLength + aRstart); }
                else if (aDir == 0 && aStr == 1) { aRstart = Lread - (aLength + aRstart); a1 = g.nGenome - (aLength + a1); }
                else if (aDir == 1 && aStr == 1) { aStr = 0; a1 = g.nGenome - (aLength + a1); }
                if (a1 >= g.sjGstart) {
                    u64 a1D, aLengthD, a1A, aLengthA, isj1;
                    if (sjAlignSplit(a1, aLength, a1D, aLengthD, a1A, aLengthA, isj1)) {
                        assignAlignToWindow(a1D, aLengthD, aStr, aNrep, aFrag, aRstart, aAnchor, isj1);
                        assignAlignToWindow(a1A, aLengthA, aStr, aNrep, aFrag, aRstart + aLengthD, aAnchor, isj1);
                    } else continue;
                } else assignAlignToWindow(a1, aLength, aStr, aNrep, aFrag, aRstart, aAnchor, (u64)-1);
            }
        }
        (void)nWpassB;
        if (tooManyAnchors) status |= STARAMD_ST_TOO_MANY_ANCHORS;   // nW was set to 0 (assignAlignToWindow.cpp:76-80)
        if (windowsLimit) status |= STARAMD_ST_WINDOWS_LIMIT;
        trAll.clear();
        trBestW = -1; int bestScore = 0; u64 bestGlen = 0;  // trBest=trInit: maxScore 0, gLength 0
        u64 trNtotal = 0;
        for (u64 iW = 0; iW < nW; iW++) {
            if (WA[iW].empty()) continue;
            cnt[C_nWA] += WA[iW].size();
            Tr trA; memset(&trA, 0, sizeof(trA));
            trA.Chr = WC[iW].chr; trA.Str = WC[iW].str; trA.roStr = trA.Str; trA.maxScore = 0;
            if (trNtotal + P.alignTranscriptsPerWindowNmax >= P.alignTranscriptsPerReadNmax) { status |= STARAMD_ST_TR_PER_READ_LIMIT; break; }
            std::vector<Tr> wTr(P.alignTranscriptsPerWindowNmax + 1, trA);
            nWinTrCur = 0;
            gSpanMin = ~0ull; gSpanMax = 0;
            u64 nodes0 = cnt[C_nNodes];
            stitchWindowAligns(0, WA[iW].size(), 0, 0, 0, trA, WA[iW], Read1[trA.roStr == 0 ? 0 : 2], wTr);
            if (cnt[C_nNodes] - nodes0 > cnt[C_maxNodesWin]) { cnt[C_maxNodesWin] = cnt[C_nNodes] - nodes0; cnt[C_maxWAWin] = WA[iW].size(); }   // profile of the heaviest window (work-distribution studies)
            if (gSpanMax >= gSpanMin) cnt[C_nGstitchSpan] += gSpanMax - gSpanMin + 1;
            if (nWinTrCur == 0) continue;
            if (wTr[0].maxScore > bestScore || (wTr[0].maxScore == bestScore && wTr[0].gLength < bestGlen)) {
                trBestW = (int)trAll.size(); bestScore = wTr[0].maxScore; bestGlen = wTr[0].gLength;
            }
            wTr.resize(nWinTrCur);
            trAll.push_back(wTr);
            trNtotal += nWinTrCur;
        }
        if (bestScore == 0) { status |= STARAMD_ST_NO_GOOD_WINDOW; trAll.clear(); trBestW = -1; }
        return status;
    }

    // ------------------------------------------------------------------ ReadAlign_mapOneRead.cpp:6-118
    void mapOneRead(const uint8_t *read, u64 L, u64 len1, u64 mmMax, staramd_read_result &rr, std::vector<staramd_transcript> &otr, std::vector<staramd_exon> &oex) {
        Lread = L; readLength[0] = len1; readLength[1] = P.readNmates == 2 ? L - len1 - 1 : 0; mmMaxTotal = mmMax;
        R0.assign(L + 1, 0); R1.assign(L + 1, 0); R2.assign(L + 1, 0);
        for (u64 i = 0; i < L; i++) { char c = (char)read[i]; R0[i] = c; R1[i] = c < 4 ? 3 - c : c; }   // complementSeqNumbers, SequenceFuns.cpp:4-14
        for (u64 i = 0; i < L; i++) R2[L - 1 - i] = R1[i];                                              // ReadAlign_oneRead.cpp:69-72
        Read1[0] = R0.data(); Read1[1] = R1.data(); Read1[2] = R2.data();
        PC.clear(); nA = 0; nUM[0] = nUM[1] = 0; multNmin = 0; multNminL = 0; fatalSeeds = false;
        tooManyAnchors = false; windowsLimit = false; maxScoreMate[0] = maxScoreMate[1] = 0;
        trAll.clear();
        memset(&rr, 0, sizeof(rr)); rr.trBest = -1; rr.trOffset = (uint32_t)otr.size();
        // ReadAlign_mapOneRead.cpp:17-21: a read of length 0 (everything clipped) is not split at all, and splitR[1][0] -- which :105 then reports as
        // trBest->rLength of MARKER_NO_GOOD_PIECES -- keeps whatever the LAST read handled by this thread left there: a value that depends on the
        // thread's history (and on the thread count).  Nothing reads it; the boundary (include/star_amd.h: unmappedLength) defines it as 0 here.
        if (Lread == 0) splitR[1][0] = 0;
        Nsplit = Lread > 0 ? qualitySplit(Read1[0], Lread, P.maxNsplit, P.seedSplitMin) : 0;
        u64 seedSearchStartLmax = std::min<u64>(P.seedSearchStartLmax, (u64)(P.seedSearchStartLmaxOverLread * (Lread - 1)));
        for (u64 ip = 0; ip < Nsplit; ip++) {
            u64 Nstart = P.seedSearchStartLmax > 0 && seedSearchStartLmax < splitR[1][ip] ? splitR[1][ip] / seedSearchStartLmax + 1 : 1;
            u64 Lstart = splitR[1][ip] / Nstart;
            bool flagDirMap = true;
            for (u64 iDir = 0; iDir < 2; iDir++) {
                u64 Lmapped, Lm;
                for (u64 istart = 0; istart < Nstart; istart++) {
                    if (flagDirMap || istart > 0) {
                        Lmapped = 0;
                        while (istart * Lstart + Lmapped + P.seedMapMin < splitR[1][ip]) {
                            u64 Shift = iDir == 0 ? (splitR[0][ip] + istart * Lstart + Lmapped) : (splitR[0][ip] + splitR[1][ip] - istart * Lstart - 1 - Lmapped);
                            u64 seedLength = splitR[1][ip] - Lmapped - istart * Lstart;
                            maxMappableLength2strands(Shift, seedLength, iDir, 0, g.nSA - 1, Lm, splitR[2][ip]);
                            if (iDir == 0 && istart == 0 && Lmapped == 0 && Shift + Lm == splitR[1][ip]) flagDirMap = false;
                            Lmapped += Lm;
                            if (Lm == 0) break;   // cannot happen for a base present in the genome (reference would spin)
                        }
                    }
                    if (P.seedSearchLmax > 0) {
                        u64 Shift = iDir == 0 ? (splitR[0][ip] + istart * Lstart) : (splitR[0][ip] + splitR[1][ip] - istart * Lstart - 1);
                        u64 seedLength = std::min<u64>(P.seedSearchLmax, iDir == 0 ? (splitR[0][ip] + splitR[1][ip] - Shift) : (Shift + 1));
                        maxMappableLength2strands(Shift, seedLength, iDir, 0, g.nSA - 1, Lm, splitR[2][ip]);
                    }
                }
            }
        }
        cnt[C_nSeeds] += PC.size();
        if (fatalSeeds) { rr.status |= STARAMD_ST_FATAL_SEEDS_PER_READ; return; }
        if (Lread < P.outFilterMatchNmin) { rr.status |= STARAMD_ST_READ_TOO_SHORT; rr.unmappedLength = 0; }
        else if (Nsplit == 0) { rr.status |= STARAMD_ST_NO_GOOD_PIECES; rr.unmappedLength = (uint32_t)splitR[1][0]; }
        else if (nA == 0) { rr.status |= STARAMD_ST_ALL_PIECES_MULTI; rr.unmappedLength = (uint32_t)multNminL; }
        else {
            int bw;
            rr.status |= stitchPieces(bw);
            // staramd_params::resultSelect == 1: keep only what multMapSelect can pick (ReadAlign_multMapSelect.cpp:26-44)
            const bool sel = P.resultSelect != 0 && bw >= 0;
            const int selMin = sel ? trAll[bw][0].maxScore - P.outFilterMultimapScoreRange : 0;
            u64 nWout = 0;
            for (u64 iW = 0; iW < trAll.size(); iW++) {
                if (sel && trAll[iW][0].maxScore < selMin) continue;     // transcripts of a window are sorted best first
                if ((int)iW == bw) rr.trBest = (int32_t)(otr.size() - rr.trOffset);
                for (u64 it = 0; it < trAll[iW].size(); it++) {
                    const Tr &t = trAll[iW][it];
                    if (sel && t.maxScore < selMin) break;
                    staramd_transcript o; memset(&o, 0, sizeof(o));
                    o.iW = (uint32_t)nWout; o.exonOffset = (uint32_t)oex.size(); o.nExons = (uint16_t)t.nExons;
                    o.rStart = (uint16_t)t.rStart; o.rLength = (uint16_t)t.rLength; o.roStart = (uint16_t)t.roStart;
                    o.Str = (uint8_t)trAll[iW][0].Str; o.roStr = (uint8_t)trAll[iW][0].roStr; o.iFrag = (int8_t)t.iFrag; o.sjMotifStrand = t.sjMotifStrand;
                    o.Chr = (uint32_t)trAll[iW][0].Chr; o.gStart = t.gStart; o.gLength = t.gLength; o.maxScore = t.maxScore;
                    o.nMatch = (uint32_t)t.nMatch; o.nMM = (uint32_t)t.nMM; o.mappedLength = (uint32_t)t.mappedLength;
                    o.nGap = (uint32_t)t.nGap; o.lGap = (uint32_t)t.lGap; o.nDel = (uint32_t)t.nDel; o.lDel = (uint32_t)t.lDel; o.nIns = (uint32_t)t.nIns; o.lIns = (uint32_t)t.lIns;
                    o.nUnique = (uint16_t)t.nUnique; o.nAnchor = (uint16_t)t.nAnchor;
                    for (int k = 0; k < 3; k++) o.intronMotifs[k] = (uint16_t)t.intronMotifs[k];
                    for (u64 ie = 0; ie < t.nExons; ie++) {
                        staramd_exon e; memset(&e, 0, sizeof(e));
                        e.G = t.ex[ie][EX_G]; e.R = (uint16_t)t.ex[ie][EX_R]; e.L = (uint16_t)t.ex[ie][EX_L];
                        e.sjA = (int32_t)(i64)t.ex[ie][EX_sjA]; e.iFrag = (uint8_t)t.ex[ie][EX_iFrag];
                        if (ie + 1 < t.nExons) {
                            e.canonSJ = (int8_t)t.canonSJ[ie]; e.sjAnnot = t.sjAnnot[ie]; e.sjStr = t.sjStr[ie];
                            bool sh = t.canonSJ[ie] >= 0;   // shiftSJ is only defined for junctions (not for indels / mate gap)
                            e.shiftSJ[0] = sh ? (uint16_t)t.shiftSJ[ie][0] : 0; e.shiftSJ[1] = sh ? (uint16_t)t.shiftSJ[ie][1] : 0;
                        }
                        oex.push_back(e);
                    }
                    otr.push_back(o);
                }
                nWout++;
            }
            rr.nW = (uint32_t)nWout;
            if (rr.nW > 0) rr.status |= STARAMD_ST_MAPPED_WINDOWS;
            rr.nTr = (uint32_t)(otr.size() - rr.trOffset);
            cnt[C_nTrOut] += rr.nTr;
        }
        rr.maxScoreMate[0] = maxScoreMate[0]; rr.maxScoreMate[1] = maxScoreMate[1];
        if (P.resultSelect) rr.maxScoreMate[0] = rr.maxScoreMate[1] = 0;      // the ABI's rule (include/star_amd.h staramd_read_result): 0 under resultSelect 1
    }
};

} // namespace

extern "C" {

void *oracle_create(const staramd_genome *g, const staramd_params *p) {
    Oracle *o = new Oracle(); o->init(g, p); return o;
}
void oracle_destroy(void *h) { delete (Oracle *)h; }
// P.sjNovelStart/End + P.outFilterBySJoutStage between the stages of --outFilterType BySJout (outputSJ.cpp:139-161)
int oracle_set_novel_junctions(void *h, const uint64_t *start, const uint64_t *end, uint64_t n, uint32_t stage) {
    Oracle *o = (Oracle *)h;
    o->sjNovelStart.assign(start, start + n); o->sjNovelEnd.assign(end, end + n);
    o->P.outFilterBySJoutStage = (uint8_t)stage;
    return 0;
}

int oracle_map_batch(void *h, const staramd_batch *b, staramd_results *r) {
    Oracle *o = (Oracle *)h;
    std::vector<staramd_transcript> otr; std::vector<staramd_exon> oex;
    for (uint32_t i = 0; i < b->nReads; i++) {
        u64 off = b->readOffset[i], L = b->readOffset[i + 1] - off;
        o->mapOneRead(b->bases + off, L, b->mate1Length[i], b->mmMaxTotal[i], r->reads[i], otr, oex);
    }
    r->trCount = otr.size(); r->exCount = oex.size();
    if (otr.size() > r->trCapacity || oex.size() > r->exCapacity) return STARAMD_ERR_RESULT_OVERFLOW;
    if (!otr.empty()) memcpy(r->tr, otr.data(), otr.size() * sizeof(staramd_transcript));
    if (!oex.empty()) memcpy(r->ex, oex.data(), oex.size() * sizeof(staramd_exon));
    return STARAMD_OK;
}

// seed table (PC) of one read, for stage-level diffs against the HIP seed kernel:
// out rows = {rStart, L, dir, nrep, saStart, saEnd, iFrag}; returns number of seeds
int oracle_seed_table(void *h, const uint8_t *read, uint64_t L, uint64_t len1, uint64_t *out, int maxRows) {
    Oracle *o = (Oracle *)h;
    staramd_read_result rr; std::vector<staramd_transcript> t; std::vector<staramd_exon> e;
    staramd_params saved = o->P;
    o->P.winAnchorMultimapNmax = 0;     // no anchors -> no windows: only the seed phase runs to completion
    o->mapOneRead(read, L, len1, 10, rr, t, e);
    o->P = saved;
    int n = (int)std::min<size_t>(o->PC.size(), (size_t)maxRows);
    for (int i = 0; i < n; i++) { const Seed &s = o->PC[i]; u64 row[7] = { s.rStart, s.L, s.dir, s.nrep, s.saStart, s.saEnd, s.iFrag }; memcpy(out + 7 * i, row, sizeof(row)); }
    return (int)o->PC.size();
}

void oracle_get_counters(void *h, uint64_t *out, int n) {
    Oracle *o = (Oracle *)h;
    for (int i = 0; i < n && i < C_N; i++) out[i] = o->cnt[i];
}
void oracle_reset_counters(void *h) { memset(((Oracle *)h)->cnt, 0, sizeof(((Oracle *)h)->cnt)); }
int oracle_n_counters() { return C_N; }

}
