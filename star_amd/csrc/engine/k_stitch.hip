// k_stitch.hip -- kernel 3: stitch the seeds of every window into transcripts, extend, score, filter, rank.
//
// Replaces, per read, the second half of ReadAlign::stitchPieces (source/ReadAlign_stitchPieces.cpp:260-348)
// and everything below it:
//   stitchWindowAligns      source/stitchWindowAligns.cpp:8-353     include/exclude recursion + leaf filters + ranked insert
//   stitchAlignToTranscript source/stitchAlignToTranscript.cpp:9-415 gap fill, junction / indel placement, sjdb scoring
//   extendAlign             source/extendAlign.cpp:6-93
//   binarySearch2           source/binarySearch2.cpp:3-43
//   blocksOverlap           source/blocksOverlap.cpp:3-40
//
// The recursion is run as an explicit depth-first walk: a frame is pushed only when a seed is
// INCLUDED (the exclude branch is a tail call: same transcript, next seed), so the stack is at most
// nWA+1 deep and a transcript is copied once per include instead of twice per node.  Transcripts
// are 32-byte exon rows (already the output format) plus an 80-byte header; only the used exons
// are copied.  Leaves are finalised in exactly the reference's order (include before exclude),
// because de-duplication, the evolving window-best score and maxScoreMate[] depend on it, and
// windows of one read are walked in window order for the same reason.
// Mapping: one lane = one read, persistent lanes with a ticket counter.
#include "dev.h"


struct StitchCtx {
    const DevIndex *X;
    const u8 *R0; u32 Lread; u32 str;          // read accessor: Read1[0] for + windows, Read1[2] for - windows
    u32 readLength[2]; u32 mmMaxTotal;
    i32 maxScoreMate[2];
    u64 nGstitch, nStitchCalls, nExtendCalls, nNodes, nLeaves;
};

__device__ __forceinline__ u8 RD(const StitchCtx &c, u32 i) {        // R[i], ReadAlign_stitchPieces.cpp:321
    return c.str == 0 ? c.R0[i] : compBase(c.R0[c.Lread - 1 - i]);
}
__device__ __forceinline__ u8 GN(StitchCtx &c, u64 pos) { c.nGstitch++; return c.X->G[(i64)pos]; }

__device__ static void copyTr(DTr *dst, const DTr *src) {
    const u64 *s = (const u64 *)src; u64 *d = (u64 *)dst;
    u32 nw = (u32)((sizeof(staramd_exon) * src->nExons) / 8);
    for (u32 i = 0; i < nw; i++) d[i] = s[i];
    const u64 *sh = (const u64 *)((const u8 *)src + sizeof(staramd_exon) * STARAMD_MAX_N_EXONS);
    u64 *dh = (u64 *)((u8 *)dst + sizeof(staramd_exon) * STARAMD_MAX_N_EXONS);
    for (u32 i = 0; i < DTR_HDR_BYTES / 8; i++) dh[i] = sh[i];
}

struct ExtRes { i32 maxScore; u32 extendL, nMatch, nMM; };

// extendAlign.cpp:6-93
__device__ static bool extendAlign(StitchCtx &c, u32 rStart, u64 gStart, int dR, int dG, u32 L, u32 Lprev, u32 nMMprev, u32 nMMmax, double pMMmax, bool extendToEnd, ExtRes &e) {
    c.nExtendCalls++;
    int Score = 0, nMatch = 0, nMM = 0;
    e.maxScore = 0; e.extendL = 0; e.nMatch = 0; e.nMM = 0;
    if (extendToEnd) {
        int iExt;
        for (iExt = 0; iExt < (int)L; iExt++) {
            int iS = dR * iExt, iG = dG * iExt;
            u8 gc = 5;
            if ((gStart + (i64)iG) == (u64)(-1) || (gc = GN(c, gStart + (i64)iG)) == 5) { e.extendL = 0; e.maxScore = -999999999; e.nMatch = 0; e.nMM = nMMmax + 1; return true; }
            u8 rc = RD(c, (u32)((int)rStart + iS));
            if (rc == STARAMD_SPACER_BASE) break;
            if (rc > 3 || gc > 3) continue;
            if (gc == rc) { nMatch++; Score += 1; } else { nMM++; Score -= 1; }
        }
        if (iExt > 0) { e.extendL = (u32)iExt; e.maxScore = Score; e.nMatch = (u32)nMatch; e.nMM = (u32)nMM; return true; }
        return false;
    }
    for (int i = 0; i < (int)L; i++) {
        int iS = dR * i, iG = dG * i;
        if ((gStart + (i64)iG) == (u64)(-1)) break;
        u8 gc = GN(c, gStart + (i64)iG);
        u8 rc = RD(c, (u32)((int)rStart + iS));
        if (gc == 5 || rc == STARAMD_SPACER_BASE) break;
        if (rc > 3 || gc > 3) continue;
        if (gc == rc) {
            nMatch++; Score += 1;
            if (Score > e.maxScore) {
                if ((double)(u32)(nMM + (int)nMMprev) <= fmin(pMMmax * (double)(u64)(Lprev + i + 1), (double)nMMmax)) {
                    e.extendL = (u32)(i + 1); e.maxScore = Score; e.nMatch = (u32)nMatch; e.nMM = (u32)nMM;
                }
            }
        } else {
            if ((double)(u32)(nMM + (int)nMMprev) >= fmin(pMMmax * (double)(u64)(Lprev + L), (double)nMMmax)) break;
            nMM++; Score -= 1;
        }
    }
    return e.extendL > 0;
}

// binarySearch2.cpp:3-43
__device__ static int binarySearch2(u64 x, u64 y, const u64 *Xs, const u64 *Ys, int N) {
    if (N == 0 || x > Xs[N - 1] || x < Xs[0]) return -1;
    int i1 = 0, i2 = N - 1, i3 = N / 2;
    while (i2 > i1 + 1) { i3 = (i1 + i2) / 2; if (Xs[i3] > x) i2 = i3; else i1 = i3; }
    if (x == Xs[i1]) i3 = i1; else if (x == Xs[i2]) i3 = i2; else return -1;
    for (int jj = i3; jj >= 0; jj--) { if (x != Xs[jj]) break; else if (y == Ys[jj]) return jj; }
    for (int jj = i3; jj < N; jj++) { if (x != Xs[jj]) return -1; else if (y == Ys[jj]) return jj; }
    return -2;
}

// blocksOverlap.cpp:3-40
__device__ static u32 blocksOverlap(const DTr &t1, const DTr &t2) {
    u32 i1 = 0, i2 = 0, nOverlap = 0;
    while (i1 < t1.nExons && i2 < t2.nExons) {
        u64 rs1 = t1.ex[i1].R, rs2 = t2.ex[i2].R;
        u64 re1 = rs1 + t1.ex[i1].L, re2 = rs2 + t2.ex[i2].L;
        u64 gs1 = t1.ex[i1].G, gs2 = t2.ex[i2].G;
        if (rs1 >= re2) i2++;
        else if (rs2 >= re1) i1++;
        else if (gs1 - rs1 != gs2 - rs2) { if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
        else { nOverlap += (u32)(min(re1, re2) - max(rs1, rs2)); if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
    }
    return nOverlap;
}

__device__ __forceinline__ void addExt(DTr *t, const ExtRes &e) {      // Transcript::add of an extension (Transcript.cpp:31-39)
    t->maxScore += e.maxScore; t->nMatch += e.nMatch; t->nMM += e.nMM;
}

// stitchAlignToTranscript.cpp:9-415
__device__ static int stitchAlignToTranscript(StitchCtx &c, u32 rAend, u64 gAend, u32 rBstart, u64 gBstart, u32 L, u32 iFragB, i32 sjAB, DTr *trA) {
    const DevIndex &X = *c.X; const staramd_params &P = X.P;
    c.nStitchCalls++;
    if (trA->nExons >= STARAMD_MAX_N_EXONS) return -1000010;
    int Score = 0;
    u32 ne = trA->nExons;
    staramd_exon &eA = trA->ex[ne - 1];
    if (sjAB != -1 && eA.sjA == sjAB && eA.iFrag == iFragB && rBstart == rAend + 1 && gAend + 1 < gBstart) {
        if (X.sjdbMotif[sjAB] == 0 && (L <= X.sjdbShiftRight[sjAB] || eA.L <= X.sjdbShiftLeft[sjAB])) return -1000006;
        staramd_exon &eN = trA->ex[ne];
        eN.L = (u16)L; eN.R = (u16)rBstart; eN.G = gBstart;
        eA.canonSJ = (i8)X.sjdbMotif[sjAB]; eA.shiftSJ[0] = X.sjdbShiftLeft[sjAB]; eA.shiftSJ[1] = X.sjdbShiftRight[sjAB];
        eA.sjAnnot = 1; eA.sjStr = X.sjdbStrand[sjAB];
        trA->nExons++; trA->nMatch += L;
        Score += (int)L; Score += P.sjdbScore;
    } else {
        eA.sjAnnot = 0; eA.sjStr = 0;
        if (eA.iFrag == iFragB) {
            u64 gBend = gBstart + L - 1; u32 rBend = rBstart + L - 1;
            if (rBend <= rAend) return -1000001;
            if (gBend <= gAend) return -1000002;
            if (rBstart <= rAend) { gBstart += rAend - rBstart + 1; rBstart = rAend + 1; L = rBend - rBstart + 1; }
            Score += (int)(rBend - rBstart + 1);
            int gGap = (int)(gBstart - gAend - 1);
            int rGap = (int)(rBstart - rAend - 1);
            u32 nMatch = L, nMM = 0; u64 Del = 0; u32 Ins = 0, nIns = 0, nDel = 0;
            int jR = 0, jCan = 999;
            u64 gBstart1 = gBstart - (u64)(i64)rGap - 1;
            if (gGap == 0 && rGap == 0) {
            } else if (gGap > 0 && rGap > 0 && rGap == gGap) {
                for (int ii = 1; ii <= rGap; ii++) {
                    u8 gc = GN(c, gAend + ii), rc = RD(c, rAend + ii);
                    if (gc < 4 && rc < 4) { if (rc == gc) { Score += 1; nMatch++; } else { Score -= 1; nMM++; } }
                }
            } else if (gGap > rGap) {
                nDel = 1; Del = (u64)(i64)(gGap - rGap);
                if (Del > P.alignIntronMax && P.alignIntronMax > 0) return -1000003;
                int Score1 = 0, jR1 = 1;
                do {
                    jR1--;
                    u8 rc = RD(c, (u32)((int)rAend + jR1)), gB = GN(c, gBstart1 + (i64)jR1);
                    if (rc != gB && gB < 4 && rc == GN(c, gAend + (i64)jR1)) Score1 -= 1;
                } while (Score1 + P.scoreStitchSJshift >= 0 && (int)eA.L + jR1 > 1);
                int maxScore2 = -999999; Score1 = 0; int jPen = 0;
                do {
                    u8 ra = RD(c, (u32)((int)rAend + jR1)), gA = GN(c, gAend + (i64)jR1), gB = GN(c, gBstart1 + (i64)jR1);
                    if (ra == gA && ra != gB) Score1 += 1;
                    if (ra != gA && ra == gB) Score1 -= 1;
                    int jCan1 = -1, jPen1 = 0, Score2 = Score1;
                    if (Del >= P.alignIntronMin) {
                        u8 d1 = GN(c, gAend + (i64)jR1 + 1), d2 = GN(c, gAend + (i64)jR1 + 2), a1 = GN(c, gBstart1 + (i64)jR1 - 1), a2 = gB;
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
                } while (jR1 < (int)rBend - (int)rAend);
                u32 jjL = 0, jjR = 0;
                while (gAend + (i64)jR >= jjL && GN(c, gAend - jjL + (i64)jR) == GN(c, gBstart1 - jjL + (i64)jR) && X.G[(i64)(gAend - jjL + (i64)jR)] < 4 && jjL <= 255) jjL++;
                while (gAend + jjR + (i64)jR + 1 < X.nGenome && GN(c, gAend + jjR + (i64)jR + 1) == GN(c, gBstart1 + jjR + (i64)jR + 1) && X.G[(i64)(gAend + jjR + (i64)jR + 1)] < 4 && jjR <= 255) jjR++;
                if (jCan <= 0) {
                    jR -= (int)jjL;
                    if ((int)eA.L + jR < 1) return -1000005;
                    jjR += jjL; jjL = 0;
                }
                for (int ii = min(1, jR + 1); ii <= max(rGap, jR); ii++) {
                    u64 g1 = (ii <= jR) ? (gAend + (i64)ii) : (gBstart1 + (i64)ii);
                    u8 gc = GN(c, g1), rc = RD(c, (u32)((int)rAend + ii));
                    if (gc < 4 && rc < 4) {
                        if (rc == gc) { if (ii >= 1 && ii <= rGap) { Score += 1; nMatch++; } }
                        else { Score -= 1; nMM++; if (ii < 1 || ii > rGap) { Score -= 1; nMatch--; } }
                    }
                }
                if (X.sjdbN > 0) {
                    u64 jS = gAend + (i64)jR + 1, jE = gBstart1 + (i64)jR;
                    int sjdbInd = binarySearch2(jS, jE, X.sjdbStart, X.sjdbEnd, (int)X.sjdbN);
                    if (sjdbInd < 0) {
                        if (Del >= P.alignIntronMin) Score += P.scoreGap + jPen;
                        else { Score += (int)Del * P.scoreDelBase + P.scoreDelOpen; jCan = -1; eA.sjAnnot = 0; }
                    } else {
                        jCan = X.sjdbMotif[sjdbInd];
                        if (X.sjdbMotif[sjdbInd] == 0) {
                            if (L <= X.sjdbShiftLeft[sjdbInd] || eA.L <= X.sjdbShiftLeft[sjdbInd]) return -1000006;
                            jR += (int)X.sjdbShiftLeft[sjdbInd];
                            if ((u64)rAend + (i64)jR >= rBend) return -1000006;
                            jjL = X.sjdbShiftLeft[sjdbInd]; jjR = X.sjdbShiftRight[sjdbInd];
                        }
                        eA.sjAnnot = 1; eA.sjStr = X.sjdbStrand[sjdbInd];
                        Score += P.sjdbScore;
                    }
                } else {
                    if (Del >= P.alignIntronMin) Score += P.scoreGap + jPen;
                    else { Score += (int)Del * P.scoreDelBase + P.scoreDelOpen; jCan = -1; eA.sjAnnot = 0; }
                }
                eA.shiftSJ[0] = (u16)jjL; eA.shiftSJ[1] = (u16)jjR; eA.canonSJ = (i8)jCan;
                if (eA.sjAnnot == 0) eA.sjStr = (jCan > 0) ? (u8)(2 - jCan % 2) : 0;
            } else if (rGap > gGap) {
                Ins = (u32)(rGap - gGap); nIns = 1;
                if (gGap == 0) jR = 0;
                else if (gGap < 0) { jR = 0; Score -= -gGap; }
                else {
                    int Score1 = 0, maxScore1 = 0;
                    for (int jR1 = 1; jR1 <= gGap; jR1++) {
                        u8 gc = GN(c, gAend + jR1);
                        if (gc < 4) { Score1 += (RD(c, rAend + jR1) == gc) ? 1 : -1; Score1 += (RD(c, rAend + Ins + jR1) == gc) ? -1 : +1; }
                        if (Score1 > maxScore1 || (Score1 == maxScore1 && P.alignInsertionFlushRight)) { maxScore1 = Score1; jR = jR1; }
                    }
                    for (int ii = 1; ii <= gGap; ii++) {
                        u32 r1 = rAend + ii + (ii <= jR ? 0 : Ins);
                        u8 gc = GN(c, gAend + ii), rc = RD(c, r1);
                        if (gc < 4 && rc < 4) { if (rc == gc) { Score += 1; nMatch++; } else { Score -= 1; nMM++; } }
                    }
                }
                if (P.alignInsertionFlushRight) {
                    for (; jR < (int)rBend - (int)rAend - (int)Ins; jR++) { u8 gc = GN(c, gAend + (i64)jR + 1); if (RD(c, (u32)((int)rAend + jR + 1)) != gc || gc == 4) break; }
                    if (jR == (int)rBend - (int)rAend - (int)Ins) return -1000009;
                }
                Score += (int)Ins * P.scoreInsBase + P.scoreInsOpen;
                jCan = -2;
            }
            if ((trA->nMM + nMM) <= c.mmMaxTotal && (jCan < 0 || (jCan < 7 && (u64)nMM <= (u64)(i64)P.alignSJstitchMismatchNmax[(jCan + 1) / 2]))) {
                trA->nMM += nMM; trA->nMatch += nMatch;
                if (Del >= P.alignIntronMin) { trA->nGap += nDel; trA->lGap += (u32)Del; } else { trA->nDel += nDel; trA->lDel += (u32)Del; }
                if (Del == 0 && Ins == 0) eA.L = (u16)(eA.L + (rBend - rAend));
                else if (Del > 0) {
                    eA.L = (u16)((int)eA.L + jR);
                    staramd_exon &eN = trA->ex[ne];
                    eN.L = (u16)((int)(rBend - rAend) - jR); eN.R = (u16)((int)rAend + jR + 1); eN.G = gBstart1 + (i64)jR + 1;
                    trA->nExons++;
                } else if (Ins > 0) {
                    trA->nIns += nIns; trA->lIns += Ins;
                    eA.L = (u16)((int)eA.L + jR);
                    staramd_exon &eN = trA->ex[ne];
                    eN.L = (u16)((int)(rBend - rAend) - jR - (int)Ins); eN.R = (u16)((int)rAend + jR + (int)Ins + 1); eN.G = gAend + 1 + (i64)jR;
                    eA.canonSJ = -2; eA.sjAnnot = 0;
                    trA->nExons++;
                }
            } else return -1000007;
        } else if (gBstart + trA->ex[0].R + (i64)P.alignEndsProtrudeNbasesMax >= trA->ex[0].G || trA->ex[0].G < trA->ex[0].R) {
            if (P.alignMatesGapMax > 0 && gBstart > eA.G + eA.L + P.alignMatesGapMax) return -1000004;
            Score += (int)L;
            ExtRes e;
            if (extendAlign(c, rAend + 1, gAend + 1, 1, 1, STARAMD_READ_LEN_MAX, trA->nMatch, trA->nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax, P.alignEndsTypeExt[eA.iFrag][1] != 0, e)) {
                addExt(trA, e); Score += e.maxScore; eA.L = (u16)(eA.L + e.extendL);
            }
            staramd_exon &eN = trA->ex[ne];
            eN.R = (u16)rBstart; eN.G = gBstart; eN.L = (u16)L; trA->nMatch += L;
            u32 extlen = P.alignEndsTypeExt[iFragB][1] ? STARAMD_READ_LEN_MAX : (u32)(gBstart - trA->ex[0].G + trA->ex[0].R);
            if (extendAlign(c, rBstart - 1, gBstart - 1, -1, -1, extlen, trA->nMatch, trA->nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax, P.alignEndsTypeExt[iFragB][1] != 0, e)) {
                addExt(trA, e); Score += e.maxScore;
                eN.R = (u16)(eN.R - e.extendL); eN.G -= e.extendL; eN.L = (u16)(eN.L + e.extendL);
            }
            eA.canonSJ = -3; eA.sjAnnot = 0;
            trA->nExons++;
        } else return -1000008;
    }
    trA->ex[trA->nExons - 1].iFrag = (u8)iFragB; trA->ex[trA->nExons - 1].sjA = sjAB;
    return Score;
}

struct WinRec { DTr *T; u16 *idx; u32 nWinTr; };

// leaf of the recursion: stitchWindowAligns.cpp:16-307.  trA is modified in place (the frame is popped afterwards).
__device__ static void finalizeTranscript(StitchCtx &c, DTr &trA, int Score, u32 tR2, u64 tG2, u32 chr, WinRec &wr) {
    const DevIndex &X = *c.X; const staramd_params &P = X.P;
    c.nLeaves++;
    u32 Lread = c.Lread; u32 Str = c.str;
    ExtRes e;
    int vOrder0 = (Str == 0) ? 0 : 1;                  // EXTEND_ORDER==1, roStr==Str
    for (int iOrd = 0; iOrd < 2; iOrd++) {
        int which = iOrd == 0 ? vOrder0 : 1 - vOrder0;
        if (which == 0) {
            if (trA.rStart > 0) {
                u32 imate = trA.ex[0].iFrag;
                if (extendAlign(c, trA.rStart - 1, trA.gStart - 1, -1, -1, trA.rStart, tR2 - trA.rStart + 1, trA.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax,
                                P.alignEndsTypeExt[imate][(int)(Str != imate)] != 0, e)) {
                    addExt(&trA, e); Score += e.maxScore;
                    trA.rStart -= e.extendL; trA.ex[0].R = (u16)trA.rStart;
                    trA.gStart -= e.extendL; trA.ex[0].G = trA.gStart;
                    trA.ex[0].L = (u16)(trA.ex[0].L + e.extendL);
                }
            }
        } else {
            if (tR2 < Lread) {
                u32 imate = trA.ex[trA.nExons - 1].iFrag;
                if (extendAlign(c, tR2 + 1, tG2 + 1, +1, +1, Lread - tR2 - 1, tR2 - trA.rStart + 1, trA.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax,
                                P.alignEndsTypeExt[imate][(int)(imate == Str)] != 0, e)) {
                    addExt(&trA, e); Score += e.maxScore;
                    tR2 += e.extendL; tG2 += e.extendL;
                    trA.ex[trA.nExons - 1].L = (u16)(trA.ex[trA.nExons - 1].L + e.extendL);
                }
            }
        }
    }
    u32 ne = trA.nExons;
    staramd_exon *ex = trA.ex;
    if (!P.alignSoftClipAtReferenceEnds &&
        ((ex[ne - 1].G + Lread - ex[ne - 1].R) > (X.chrStart[chr] + X.chrLength[chr]) || ex[0].G < (X.chrStart[chr] + ex[0].R))) return;
    trA.rLength = 0;
    for (u32 i = 0; i < ne; i++) trA.rLength += ex[i].L;
    trA.gLength = tG2 + 1 - trA.gStart;
    for (u32 isj = 0; isj + 1 < ne; isj++) {
        if (ex[isj].canonSJ >= 0) {
            if (ex[isj].sjAnnot == 1) {
                if ((ex[isj].L < P.alignSJDBoverhangMin && (isj == 0 || ex[isj - 1].canonSJ == -3 || (ex[isj - 1].sjAnnot == 0 && ex[isj - 1].canonSJ >= 0)))
                    || (ex[isj + 1].L < P.alignSJDBoverhangMin && (isj == ne - 2 || ex[isj + 1].canonSJ == -3 || (ex[isj + 1].sjAnnot == 0 && ex[isj + 1].canonSJ >= 0)))) return;
            } else {
                if (ex[isj].L < P.alignSJoverhangMin + ex[isj].shiftSJ[0] || ex[isj + 1].L < P.alignSJoverhangMin + ex[isj].shiftSJ[1]) return;
            }
        }
    }
    if (ne > 1 && ex[ne - 2].sjAnnot == 1 && ex[ne - 1].L < P.alignSJDBoverhangMin) return;
    u32 sjN = 0;
    trA.intronMotifs[0] = trA.intronMotifs[1] = trA.intronMotifs[2] = 0;
    for (u32 i = 0; i + 1 < ne; i++) if (ex[i].canonSJ >= 0) { sjN++; trA.intronMotifs[ex[i].sjStr]++; }
    if (trA.intronMotifs[1] > 0 && trA.intronMotifs[2] == 0) trA.sjMotifStrand = 1;
    else if (trA.intronMotifs[1] == 0 && trA.intronMotifs[2] > 0) trA.sjMotifStrand = 2;
    else trA.sjMotifStrand = 0;
    if (trA.intronMotifs[1] > 0 && trA.intronMotifs[2] > 0 && P.outFilterIntronStrandsRemoveInconsistent) return;
    if (sjN > 0 && trA.sjMotifStrand == 0 && P.outSAMstrandFieldIntronMotif) return;
    if (P.outFilterIntronMotifs == 1) { for (u32 i = 0; i + 1 < ne; i++) if (ex[i].canonSJ == 0) return; }
    else if (P.outFilterIntronMotifs == 2) { for (u32 i = 0; i + 1 < ne; i++) if (ex[i].canonSJ == 0 && ex[i].sjAnnot == 0) return; }
    {
        u32 nsj = 0, exl = 0;
        for (u32 i = 0; i < ne; i++) {
            exl += ex[i].L;
            if (i == ne - 1 || ex[i].canonSJ == -3) {
                if (nsj > 0 && (exl < P.alignSplicedMateMapLmin || (u64)exl < (u64)(P.alignSplicedMateMapLminOverLmate * (double)(u64)c.readLength[ex[i].iFrag]))) return;
                exl = 0; nsj = 0;
            } else if (ex[i].canonSJ >= 0) nsj++;
        }
    }
    if (ex[0].iFrag != ex[ne - 1].iFrag) {
        if (ex[ne - 1].G + ex[ne - 1].L <= ex[0].G) return;
        u32 iexM2 = ne;
        for (u32 i = 0; i + 1 < ne; i++) if (ex[i].canonSJ == -3) { iexM2 = i + 1; break; }
        if (ex[iexM2 - 1].G + ex[iexM2 - 1].L > ex[iexM2].G) {
            if (ex[0].G > ex[iexM2].G + ex[0].R + (i64)P.alignEndsProtrudeNbasesMax) return;
            if (ex[iexM2 - 1].G + ex[iexM2 - 1].L > ex[ne - 1].G + Lread - ex[ne - 1].R + (i64)P.alignEndsProtrudeNbasesMax) return;
            u32 iex1 = 1, iex2 = iexM2 + 1;
            for (; iex1 < iexM2; iex1++) if (ex[iex1].G >= ex[iex2 - 1].G + ex[iex2 - 1].L) break;
            while (iex1 < iexM2 && iex2 < ne) {
                if (ex[iex1 - 1].canonSJ < 0) { iex1++; continue; }
                if (ex[iex2 - 1].canonSJ < 0) { iex2++; continue; }
                if ((ex[iex1].G != ex[iex2].G) || ((ex[iex1 - 1].G + ex[iex1 - 1].L) != (ex[iex2 - 1].G + ex[iex2 - 1].L))) return;
                iex1++; iex2++;
            }
        }
    }
    if (X.glStep != 0) {             // scoreGenomicLengthLog2scale != 0 (:221-225) as integer break points
        u64 gl = ex[ne - 1].G + ex[ne - 1].L - ex[0].G;
        i32 term = X.glScoreAt1;
        for (u32 k = 0; k < X.nBreak; k++) if (gl >= X.glBreak[k]) term += X.glStep;
        Score += term;
        Score = max(0, Score);
    }
    trA.roStart = (Str == 0) ? trA.rStart : Lread - trA.rStart - trA.rLength;
    trA.maxScore = Score;
    if (ex[0].iFrag == ex[ne - 1].iFrag) { trA.iFrag = ex[0].iFrag; c.maxScoreMate[trA.iFrag] = max(c.maxScoreMate[trA.iFrag], Score); }
    else trA.iFrag = -1;
    DTr *T = wr.T; u16 *idx = wr.idx;
    if (Score + P.outFilterMultimapScoreRange >= T[idx[0]].maxScore
        || (trA.iFrag >= 0 && Score + P.outFilterMultimapScoreRange >= c.maxScoreMate[trA.iFrag]) || P.chimSegmentMinPositive) {
        u32 iTr = 0;
        trA.mappedLength = 0;
        for (u32 i = 0; i < ne; i++) trA.mappedLength += ex[i].L;
        u32 &nWinTr = wr.nWinTr;
        while (iTr < nWinTr) {
            DTr &o = T[idx[iTr]];
            u32 nOverlap = blocksOverlap(trA, o);
            u32 uNew = trA.mappedLength - nOverlap, uOld = o.mappedLength - nOverlap;
            if (uNew == 0 && Score < o.maxScore) break;
            else if (uOld == 0) { u16 p = idx[iTr]; for (u32 ii = iTr + 1; ii < nWinTr; ii++) idx[ii - 1] = idx[ii]; nWinTr--; idx[nWinTr] = p; }
            else if (uOld > 0 && (uNew > 0 || Score >= o.maxScore)) iTr++;
        }
        if (iTr == nWinTr) {
            for (iTr = 0; iTr < nWinTr; iTr++) { DTr &o = T[idx[iTr]]; if (Score > o.maxScore || (Score == o.maxScore && trA.gLength < o.gLength)) break; }
            u16 p = idx[nWinTr];
            for (int ii = (int)nWinTr; ii > (int)iTr; ii--) idx[ii] = idx[ii - 1];
            idx[iTr] = p;
            copyTr(&T[p], &trA);
            if (nWinTr < P.alignTranscriptsPerWindowNmax) nWinTr++;
        }
    }
}

extern "C" __global__ void __launch_bounds__(256) k_stitch(DevIndex X, DevBatch B, u8 *scratch, u32 capDepth, u32 capTr) {
    u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    const staramd_params &P = X.P;
    u64 perLane = (u64)capDepth * sizeof(Frame) + (u64)capTr * sizeof(DTr) + (u64)capTr * sizeof(u16);
    perLane = (perLane + 15) & ~15ull;
    Frame *stack = (Frame *)(scratch + (u64)lane * perLane);
    WinRec wr; wr.T = (DTr *)((u8 *)stack + (u64)capDepth * sizeof(Frame)); wr.idx = (u16 *)((u8 *)wr.T + (u64)capTr * sizeof(DTr));
    StitchCtx c; c.X = &X; c.nGstitch = c.nStitchCalls = c.nExtendCalls = c.nNodes = c.nLeaves = 0;
    u64 nTrOut = 0;
    for (;;) {
        u32 ir = atomicAdd(&B.cursors[10], 1u);
        if (ir >= B.nReads) break;
        DRead rd = B.reads[ir];
        if (rd.nWin == 0) {
            if (rd.nSeeds > 0 && !(rd.status & (STARAMD_ST_SCRATCH_OVERFLOW | STARAMD_ST_NO_GOOD_WINDOW))) { rd.status |= STARAMD_ST_NO_GOOD_WINDOW; B.reads[ir] = rd; }
            continue;
        }
        c.R0 = B.bases + B.readOffset[ir]; c.Lread = (u32)(B.readOffset[ir + 1] - B.readOffset[ir]);
        c.readLength[0] = B.mate1Length[ir]; c.readLength[1] = P.readNmates == 2 ? c.Lread - c.readLength[0] - 1 : 0;
        c.mmMaxTotal = B.mmMaxTotal[ir];
        c.maxScoreMate[0] = c.maxScoreMate[1] = 0;
        u32 wtBase = atomicAdd(&B.cursors[3], rd.nWin);
        if (wtBase + rd.nWin > B.wtCap) { rd.status |= STARAMD_ST_SCRATCH_OVERFLOW; atomicOr(&B.cursors[6], 8u); B.reads[ir] = rd; continue; }
        u32 nWt = 0, trNtotal = 0, exTotal = 0; int bestScore = 0; u64 bestGlen = 0; i32 bestW = -1; bool overflow = false;
        for (u32 iw = 0; iw < rd.nWin; iw++) {
            const DWin win = B.winPool[rd.winOffset + iw];
            const DWA *WA = B.waPool + win.waOffset;
            u32 nA = win.nWA;
            if (trNtotal + P.alignTranscriptsPerWindowNmax >= P.alignTranscriptsPerReadNmax) { rd.status |= STARAMD_ST_TR_PER_READ_LIMIT; break; }
            if (nA + 1 > capDepth || P.alignTranscriptsPerWindowNmax + 1 > capTr) { overflow = true; break; }
            c.str = win.str;
            // wTr[0] = trA (maxScore 0, gLength 0): ReadAlign_stitchPieces.cpp:295
            for (u32 k = 0; k <= P.alignTranscriptsPerWindowNmax; k++) wr.idx[k] = (u16)k;
            wr.nWinTr = 0; wr.T[0].maxScore = 0; wr.T[0].gLength = 0; wr.T[0].nExons = 0; wr.T[0].mappedLength = 0;
            // root frame
            Frame *f = &stack[0];
            {
                u64 *z = (u64 *)&f->tr; for (u32 k = 0; k < sizeof(DTr) / 8; k++) z[k] = 0;
                f->Score = 0; f->tR2 = 0; f->tG2 = 0; f->iA = 0; f->state = 0;
            }
            int sp = 0;
            while (sp >= 0) {
                f = &stack[sp];
                c.nNodes++;
                if (f->iA >= nA) {                       // stitchWindowAligns.cpp:14-16
                    if (f->tR2 != 0) finalizeTranscript(c, f->tr, f->Score, f->tR2, f->tG2, win.chr, wr);
                    sp--;
                    continue;
                }
                if (f->state == 0) {
                    f->state = 1;
                    u32 iA = f->iA;
                    const DWA a = WA[iA];
                    Frame *n = &stack[sp + 1];
                    copyTr(&n->tr, &f->tr);
                    int dScore = 0;
                    if (f->tr.nExons > 0) {
                        dScore = stitchAlignToTranscript(c, f->tR2, f->tG2, a.rStart, a.gStart, a.L, a.iFrag, a.sjA, &n->tr);
                    } else {                             // first align of the transcript (:318-334)
                        staramd_exon &e0 = n->tr.ex[0];
                        e0.R = a.rStart; e0.G = a.gStart; e0.L = a.L; e0.iFrag = a.iFrag; e0.sjA = a.sjA;
                        e0.canonSJ = 0; e0.sjAnnot = 0; e0.sjStr = 0; e0.shiftSJ[0] = e0.shiftSJ[1] = 0; e0.pad0 = 0;
                        n->tr.rStart = a.rStart; n->tr.gStart = a.gStart; n->tr.nExons = 1;
                        dScore = a.L; n->tr.nMatch = a.L;
                    }
                    if (dScore > -1000000) {
                        if (a.nrep == 1) n->tr.nUnique++;
                        if (a.anchor > 0) n->tr.nAnchor++;
                        n->Score = f->Score + dScore; n->tR2 = (u32)a.rStart + a.L - 1; n->tG2 = a.gStart + a.L - 1; n->iA = iA + 1; n->state = 0;
                        sp++;
                        continue;
                    }
                }
                // exclude branch (:348-351): same transcript, next seed -- tail call
                f->iA++; f->state = 0;
            }
            if (wr.nWinTr == 0) continue;
            // ---- record the window's transcripts (trAll[iW1][0..nWinTr-1]) into the pools
            u32 nTr = wr.nWinTr, nEx = 0;
            for (u32 k = 0; k < nTr; k++) nEx += wr.T[wr.idx[k]].nExons;
            u32 to = atomicAdd(&B.cursors[4], nTr), eo = atomicAdd(&B.cursors[5], nEx);
            if (to + nTr > B.trCap || eo + nEx > B.exCap) { overflow = true; atomicOr(&B.cursors[6], 16u); break; }
            const DTr &h = wr.T[wr.idx[0]];
            if (h.maxScore > bestScore || (h.maxScore == bestScore && h.gLength < bestGlen)) { bestW = (i32)nWt; bestScore = h.maxScore; bestGlen = h.gLength; }
            DWinTr d; d.read = ir; d.trOffset = to; d.nTr = nTr; d.exOffset = eo; d.nEx = nEx; d.chr = win.chr; d.str = win.str; d.pad[0] = d.pad[1] = d.pad[2] = 0;
            B.wtPool[wtBase + nWt] = d;
            u32 eoff = 0;
            for (u32 k = 0; k < nTr; k++) {
                const DTr &t = wr.T[wr.idx[k]];
                staramd_transcript o;
                o.iW = nWt; o.exonOffset = eoff;          // relative to the block; k_gather rebases it
                o.nExons = (u16)t.nExons; o.rStart = (u16)t.rStart; o.rLength = (u16)t.rLength; o.roStart = (u16)t.roStart;
                o.Str = win.str; o.roStr = win.str; o.iFrag = (i8)t.iFrag; o.sjMotifStrand = t.sjMotifStrand; o.Chr = win.chr;
                o.gStart = t.gStart; o.gLength = t.gLength; o.maxScore = t.maxScore; o.nMatch = t.nMatch; o.nMM = t.nMM; o.mappedLength = t.mappedLength;
                o.nGap = t.nGap; o.lGap = t.lGap; o.nDel = t.nDel; o.lDel = t.lDel; o.nIns = t.nIns; o.lIns = t.lIns;
                o.nUnique = (u16)t.nUnique; o.nAnchor = (u16)t.nAnchor;
                o.intronMotifs[0] = t.intronMotifs[0]; o.intronMotifs[1] = t.intronMotifs[1]; o.intronMotifs[2] = t.intronMotifs[2]; o.pad0 = 0;
                B.trPool[to + k] = o;
                for (u32 ie = 0; ie < t.nExons; ie++) {
                    staramd_exon e = t.ex[ie];
                    if (ie + 1 == t.nExons) { e.canonSJ = 0; e.sjAnnot = 0; e.sjStr = 0; e.shiftSJ[0] = e.shiftSJ[1] = 0; }
                    else if (e.canonSJ < 0) { e.shiftSJ[0] = e.shiftSJ[1] = 0; }
                    e.pad0 = 0;
                    B.exPool[eo + eoff + ie] = e;
                }
                eoff += t.nExons;
            }
            nWt++; trNtotal += nTr; exTotal += nEx;
        }
        if (overflow) { rd.status |= STARAMD_ST_SCRATCH_OVERFLOW; atomicOr(&B.cursors[6], 32u); B.reads[ir] = rd; continue; }
        if (bestScore == 0) { rd.status |= STARAMD_ST_NO_GOOD_WINDOW; nWt = 0; trNtotal = 0; exTotal = 0; bestW = -1; }   // :344-348
        rd.wtOffset = wtBase; rd.nWt = nWt; rd.nTr = trNtotal; rd.nEx = exTotal; rd.bestW = bestW;
        rd.maxScoreMate[0] = c.maxScoreMate[0]; rd.maxScoreMate[1] = c.maxScoreMate[1];
        B.reads[ir] = rd;
        nTrOut += trNtotal;
    }
    atomicAdd((unsigned long long *)&B.counters[DC_nGstitch], (unsigned long long)c.nGstitch);
    atomicAdd((unsigned long long *)&B.counters[DC_nStitchCalls], (unsigned long long)c.nStitchCalls);
    atomicAdd((unsigned long long *)&B.counters[DC_nExtendCalls], (unsigned long long)c.nExtendCalls);
    atomicAdd((unsigned long long *)&B.counters[DC_nNodes], (unsigned long long)c.nNodes);
    atomicAdd((unsigned long long *)&B.counters[DC_nLeaves], (unsigned long long)c.nLeaves);
    atomicAdd((unsigned long long *)&B.counters[DC_nTrOut], (unsigned long long)nTrOut);
}
