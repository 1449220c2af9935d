// lane_routines_ref.h -- TEST INFRASTRUCTURE (oracle/): the one-lane routines of round 4, a base-by-base restatement of extendAlign.cpp:6-93, binarySearch2.cpp:3-43 and
// stitchAlignToTranscript.cpp:9-415 that the rewritten star_amd/csrc/engine/stitch_scalar.h is checked against (oracle/lane_routines_check.cpp, tests/test_lane_routines.py).
// Nothing under star_amd/ includes this file.


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
            if ((gStart + (i64)iG) == (u64)(-1) || (gc = GA(c, gStart + (i64)iG)) == 5) { e.extendL = 0; e.maxScore = -999999999; e.nMatch = 0; e.nMM = nMMmax + 1; return true; }
            u8 rc = RD(c, (u32)((int)rStart + iS));
            if (rc == STARAMD_SPACER_BASE) break;
            if (rc > 3 || gc > 3) continue;
            if (gc == rc) { nMatch++; Score += 1; } else { nMM++; Score -= 1; }
        }
        if (iExt > 0) { e.extendL = (u32)iExt; e.maxScore = Score; e.nMatch = (u32)nMatch; e.nMM = (u32)nMM; return true; }
        return false;
    }
    const double thrBreak = fmin(pMMmax * (double)(u64)(Lprev + L), (double)nMMmax);
    for (int i = 0; i < (int)L; i++) {
        int iS = dR * i, iG = dG * i;
        if ((gStart + (i64)iG) == (u64)(-1)) break;
        u8 gc = GA(c, gStart + (i64)iG);
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
            if ((double)(u32)(nMM + (int)nMMprev) >= thrBreak) break;
            nMM++; Score -= 1;
        }
    }
    return e.extendL > 0;
}

// binarySearch2.cpp:3-43
__device__ static int binarySearch2(u64 x, u64 y, const u64 *Xs_, const u64 *Ys_, int N) {
    const __attribute__((address_space(1))) u64 *Xs = GLOBAL(u64, Xs_), *Ys = GLOBAL(u64, Ys_);
    if (N == 0 || x > Xs[N - 1] || x < Xs[0]) return -1;
    int i1 = 0, i2 = N - 1, i3 = N / 2;
    while (i2 > i1 + 1) { i3 = (i1 + i2) / 2; if (Xs[i3] > x) i2 = i3; else i1 = i3; }
    if (x == Xs[i1]) i3 = i1; else if (x == Xs[i2]) i3 = i2; else return -1;
    for (int jj = i3; jj >= 0; jj--) { if (x != Xs[jj]) break; else if (y == Ys[jj]) return jj; }
    for (int jj = i3; jj < N; jj++) { if (x != Xs[jj]) return -1; else if (y == Ys[jj]) return jj; }
    return -2;
}

// stitchAlignToTranscript.cpp:9-415.  h / eA are working copies: the caller commits them (and eN when *added)
// only when the returned score is > -1000000, so a failed stitch leaves the transcript untouched.
// ex0R / ex0G = start of the first exon (the mate-pair branch looks at it).
__device__ static int stitchAlignToTranscript(StitchCtx &c, u32 rAend, u64 gAend, u32 rBstart, u64 gBstart, u32 L, u32 iFragB, i32 sjAB,
                                              Hdr &h, staramd_exon &eA, staramd_exon &eN, bool &added, u32 ex0R, u64 ex0G) {
    const DevIndex &X = *c.X; const staramd_params &P = X.P;
    c.nStitchCalls++;
    added = false;
    if (h.nExons >= STARAMD_MAX_N_EXONS) return -1000010;
    int Score = 0;
    if (sjAB != -1 && eA.sjA == sjAB && eA.iFrag == iFragB && rBstart == rAend + 1 && gAend + 1 < gBstart) {
        if (GLOBAL(u8, X.sjdbMotif)[sjAB] == 0 && (L <= GLOBAL(u8, X.sjdbShiftRight)[sjAB] || eA.L <= GLOBAL(u8, X.sjdbShiftLeft)[sjAB])) return -1000006;
        eN.L = (u16)L; eN.R = (u16)rBstart; eN.G = gBstart;
        eA.canonSJ = (i8)GLOBAL(u8, X.sjdbMotif)[sjAB]; eA.shiftSJ[0] = GLOBAL(u8, X.sjdbShiftLeft)[sjAB]; eA.shiftSJ[1] = GLOBAL(u8, X.sjdbShiftRight)[sjAB];
        eA.sjAnnot = 1; eA.sjStr = GLOBAL(u8, X.sjdbStrand)[sjAB];
        added = true; h.nMatch += L;
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
                    u8 gc = GA(c, gAend + ii), rc = RD(c, rAend + ii);
                    if (gc < 4 && rc < 4) { if (rc == gc) { Score += 1; nMatch++; } else { Score -= 1; nMM++; } }
                }
            } else if (gGap > rGap) {
                nDel = 1; Del = (u64)(i64)(gGap - rGap);
                if (Del > P.alignIntronMax && P.alignIntronMax > 0) return -1000003;
                int Score1 = 0, jR1 = 1;
                do {
                    jR1--;
                    u8 rc = RD(c, (u32)((int)rAend + jR1)), gB = GB(c, gBstart1 + (i64)jR1);
                    if (rc != gB && gB < 4 && rc == GA(c, gAend + (i64)jR1)) Score1 -= 1;
                } while (Score1 + P.scoreStitchSJshift >= 0 && (int)eA.L + jR1 > 1);
                int maxScore2 = -999999; Score1 = 0; int jPen = 0;
                const bool isIntron = Del >= P.alignIntronMin;
                do {
                    u8 ra = RD(c, (u32)((int)rAend + jR1)), gA = GA(c, gAend + (i64)jR1), gB = GB(c, gBstart1 + (i64)jR1);
                    if (ra == gA && ra != gB) Score1 += 1;
                    if (ra != gA && ra == gB) Score1 -= 1;
                    int jCan1 = -1, jPen1 = 0, Score2 = Score1;
                    if (isIntron) {
                        u8 d1 = GA(c, gAend + (i64)jR1 + 1), d2 = GA(c, gAend + (i64)jR1 + 2), a1 = GB(c, gBstart1 + (i64)jR1 - 1), a2 = gB;
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
                for (;;) {
                    if (!(gAend + (i64)jR >= jjL)) break;
                    u8 x = GA(c, gAend - jjL + (i64)jR);
                    if (!(x == GB(c, gBstart1 - jjL + (i64)jR) && x < 4 && jjL <= 255)) break;
                    jjL++;
                }
                for (;;) {
                    if (!(gAend + jjR + (i64)jR + 1 < X.nGenome)) break;
                    u8 x = GA(c, gAend + jjR + (i64)jR + 1);
                    if (!(x == GB(c, gBstart1 + jjR + (i64)jR + 1) && x < 4 && jjR <= 255)) break;
                    jjR++;
                }
                if (jCan <= 0) {
                    jR -= (int)jjL;
                    if ((int)eA.L + jR < 1) return -1000005;
                    jjR += jjL; jjL = 0;
                }
                for (int ii = min(1, jR + 1); ii <= max(rGap, jR); ii++) {
                    u8 gc = (ii <= jR) ? GA(c, gAend + (i64)ii) : GB(c, gBstart1 + (i64)ii);
                    u8 rc = RD(c, (u32)((int)rAend + ii));
                    if (gc < 4 && rc < 4) {
                        if (rc == gc) { if (ii >= 1 && ii <= rGap) { Score += 1; nMatch++; } }
                        else { Score -= 1; nMM++; if (ii < 1 || ii > rGap) { Score -= 1; nMatch--; } }
                    }
                }
                int sjdbInd = -1;
                if (X.sjdbN > 0) sjdbInd = X.sjdbHash ? sjdbHashFind(X.sjdbHash, X.sjdbHashMask, gAend + (i64)jR + 1, gBstart1 + (i64)jR)
                                                      : binarySearch2(gAend + (i64)jR + 1, gBstart1 + (i64)jR, X.sjdbStart, X.sjdbEnd, (int)X.sjdbN);
                if (sjdbInd < 0) {
                    if (isIntron) Score += P.scoreGap + jPen;
                    else { Score += (int)Del * P.scoreDelBase + P.scoreDelOpen; jCan = -1; eA.sjAnnot = 0; }
                } else {
                    jCan = GLOBAL(u8, X.sjdbMotif)[sjdbInd];
                    if (GLOBAL(u8, X.sjdbMotif)[sjdbInd] == 0) {
                        if (L <= GLOBAL(u8, X.sjdbShiftLeft)[sjdbInd] || eA.L <= GLOBAL(u8, X.sjdbShiftLeft)[sjdbInd]) return -1000006;
                        jR += (int)GLOBAL(u8, X.sjdbShiftLeft)[sjdbInd];
                        if ((u64)rAend + (i64)jR >= rBend) return -1000006;
                        jjL = GLOBAL(u8, X.sjdbShiftLeft)[sjdbInd]; jjR = GLOBAL(u8, X.sjdbShiftRight)[sjdbInd];
                    }
                    eA.sjAnnot = 1; eA.sjStr = GLOBAL(u8, X.sjdbStrand)[sjdbInd];
                    Score += P.sjdbScore;
                }
                eA.shiftSJ[0] = (u16)jjL; eA.shiftSJ[1] = (u16)jjR; eA.canonSJ = (i8)jCan;
                if (eA.sjAnnot == 0) eA.sjStr = (jCan > 0) ? (u8)(2 - jCan % 2) : 0;
            } else if (rGap > gGap) {
                Ins = (u32)(rGap - gGap); nIns = 1;
                if (gGap == 0) jR = 0;
                else if (gGap < 0) { jR = 0; Score -= -gGap; }
                else {
                    int Score1 = 0, maxScore1 = 0;
                    const int tieStep = P.alignInsertionFlushRight ? 0 : 1;
                    for (int jR1 = 1; jR1 <= gGap; jR1++) {
                        u8 gc = GA(c, gAend + jR1);
                        if (gc < 4) { Score1 += (RD(c, rAend + jR1) == gc) ? 1 : -1; Score1 += (RD(c, rAend + Ins + jR1) == gc) ? -1 : +1; }
                        if (Score1 >= maxScore1 + tieStep) { maxScore1 = Score1; jR = jR1; }     // flush right: an equal score moves the insertion right (:273)
                    }
                    for (int ii = 1; ii <= gGap; ii++) {
                        u32 r1 = rAend + ii + (ii <= jR ? 0 : Ins);
                        u8 gc = GA(c, gAend + ii), rc = RD(c, r1);
                        if (gc < 4 && rc < 4) { if (rc == gc) { Score += 1; nMatch++; } else { Score -= 1; nMM++; } }
                    }
                }
                if (P.alignInsertionFlushRight) {
                    for (; jR < (int)rBend - (int)rAend - (int)Ins; jR++) { u8 gc = GA(c, gAend + (i64)jR + 1); if (RD(c, (u32)((int)rAend + jR + 1)) != gc || gc == 4) break; }
                    if (jR == (int)rBend - (int)rAend - (int)Ins) return -1000009;
                }
                Score += (int)Ins * P.scoreInsBase + P.scoreInsOpen;
                jCan = -2;
            }
            if ((h.nMM + nMM) <= c.mmMaxTotal && (jCan < 0 || (jCan < 7 && (u64)nMM <= (u64)(i64)P.alignSJstitchMismatchNmax[(jCan + 1) / 2]))) {
                h.nMM += nMM; h.nMatch += nMatch;
                if (Del >= P.alignIntronMin) { h.nGap += nDel; h.lGap += (u32)Del; } else { h.nDel += nDel; h.lDel += (u32)Del; }
                if (Del == 0 && Ins == 0) eA.L = (u16)(eA.L + (rBend - rAend));
                else if (Del > 0) {
                    eA.L = (u16)((int)eA.L + jR);
                    eN.L = (u16)((int)(rBend - rAend) - jR); eN.R = (u16)((int)rAend + jR + 1); eN.G = gBstart1 + (i64)jR + 1;
                    added = true;
                } else if (Ins > 0) {
                    h.nIns += nIns; h.lIns += Ins;
                    eA.L = (u16)((int)eA.L + jR);
                    eN.L = (u16)((int)(rBend - rAend) - jR - (int)Ins); eN.R = (u16)((int)rAend + jR + (int)Ins + 1); eN.G = gAend + 1 + (i64)jR;
                    eA.canonSJ = -2; eA.sjAnnot = 0;
                    added = true;
                }
            } else return -1000007;
        } else if (gBstart + ex0R + (i64)P.alignEndsProtrudeNbasesMax >= ex0G || ex0G < ex0R) {
            if (P.alignMatesGapMax > 0 && gBstart > eA.G + eA.L + P.alignMatesGapMax) return -1000004;
            Score += (int)L;
            ExtRes e;
            if (extendAlign(c, rAend + 1, gAend + 1, 1, 1, STARAMD_READ_LEN_MAX, h.nMatch, h.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax, P.alignEndsTypeExt[eA.iFrag][1] != 0, e)) {
                h.nMatch += e.nMatch; h.nMM += e.nMM; Score += e.maxScore; eA.L = (u16)(eA.L + e.extendL);
            }
            eN.R = (u16)rBstart; eN.G = gBstart; eN.L = (u16)L; h.nMatch += L;
            // the first exon may be the one being extended above (nExons==1): its start does not move, only its length
            u32 extlen = P.alignEndsTypeExt[iFragB][1] ? STARAMD_READ_LEN_MAX : (u32)(gBstart - ex0G + ex0R);
            if (extendAlign(c, rBstart - 1, gBstart - 1, -1, -1, extlen, h.nMatch, h.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax, P.alignEndsTypeExt[iFragB][1] != 0, e)) {
                h.nMatch += e.nMatch; h.nMM += e.nMM; Score += e.maxScore;
                eN.R = (u16)(eN.R - e.extendL); eN.G -= e.extendL; eN.L = (u16)(eN.L + e.extendL);
            }
            eA.canonSJ = -3; eA.sjAnnot = 0;
            added = true;
        } else return -1000008;
    }
    // the last exon carries the mate / sjdb index of the last seed (:413-414)
    if (added) { eN.iFrag = (u8)iFragB; eN.sjA = sjAB; eN.canonSJ = 0; eN.sjAnnot = 0; eN.sjStr = 0; eN.shiftSJ[0] = eN.shiftSJ[1] = 0; eN.pad0 = 0; eN.pad1 = 0; }
    else { eA.iFrag = (u8)iFragB; eA.sjA = sjAB; }
    return Score;
}

