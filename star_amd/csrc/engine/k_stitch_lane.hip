// k_stitch_lane.hip -- kernel 3a: stitch, ONE LANE PER READ, for the reads of the lowest cost classes (class <= 3: ~5 % of the reads of a 2x101 batch; the cooperative kernel takes the rest).
//
// Replaces, per read, the second half of ReadAlign::stitchPieces (source/ReadAlign_stitchPieces.cpp:260-348) and below it
//   stitchWindowAligns      source/stitchWindowAligns.cpp:8-353     include/exclude recursion + leaf filters + ranked insert
//   stitchAlignToTranscript source/stitchAlignToTranscript.cpp:9-415 (stitch_scalar.h)
//   extendAlign             source/extendAlign.cpp:6-93             (stitch_scalar.h)
//   binarySearch2           source/binarySearch2.cpp:3-43           (stitch_scalar.h)
//   blocksOverlap           source/blocksOverlap.cpp:3-40
//
// Why a second mapping next to the wave-cooperative walk of k_stitch.hip.  The cooperative kernel gives a window to a
// whole wavefront: its per-base loops use all 64 lanes, but the control flow between them -- the include/exclude walk, the leaf filters,
// the record decision, de-duplication, bookkeeping -- is wave-uniform code that 64 lanes execute redundantly: 22 k VALU + 16 k SALU
// wave-instructions per read pair for ~10 leaves (rocprofv3, round 2), VALU pipes 40 % busy at < 1 % of the HBM roofline.  A window of a
// typical read holds 1-4 seeds and its extensions / junction scans cover a few dozen bases: there is not enough data parallelism INSIDE a
// window to feed 64 lanes, but there are 400 000 independent reads in a batch.  Here every lane walks its own read with the scalar
// restatement: 64 independent walks per wavefront, nothing executed redundantly, divergence instead of idle lanes.
//   * work item = a LIGHT read (k_windows: sum over windows of 2^seeds <= lightEst) whose windows hold at most LANE_SEEDS seeds: its
//     windows are walked in order by one lane with maxScoreMate carried (stitchWindowAligns.cpp:232-247), so the result is final -- no
//     candidate log, no verify / replay.  Everything else (windows of many seeds, heavy reads, a read whose records outgrow the lane's
//     arena) is handed to the cooperative kernel through B.heavyList, exactly as its own lean launch used to hand on heavy items;
//   * state: transcript header in registers; undo stack, exon rows and the leaf copy are per-lane arrays (private memory, which the
//     hardware lays out dword-interleaved over the lanes: lanes that access the same row touch consecutive addresses); the read is staged
//     4-bit packed in the lane's LDS slot; recorded transcripts of the current window go to a per-lane arena in HBM in the OUTPUT record
//     format, so that flushing a window is a copy;
//   * items are taken lane by lane from the cost-sorted order of k_order_* (heaviest class first, neighbours in the order = same class),
//     so that the lanes of a wavefront do walks of similar size and a lane that finishes early takes the next item.
// Window pruning and the two-mate-windows-first sweep are the ones of k_stitch_win, statement for statement.
#include "stitch_common.h"
#include "stitch_scalar.h"

#define LANE_SEEDS 8u        // seeds per window this kernel walks (undo stack / exon rows per lane are sized by it)
#define LANE_RECS 16u        // recorded transcripts per window before the item is handed to the cooperative kernel
#define LREC_HDR 96u
static_assert(sizeof(staramd_transcript) == LREC_HDR, "record header is the output transcript record");
static_assert(sizeof(staramd_exon) == 32, "exon record is 32 bytes");

struct LFrame { Hdr h; u32 iA, iLast; staramd_exon eA; };              // what an include changes: header, last exon (112 bytes)

// recorded transcripts of the current window: records (output transcript header + exon rows) bump-allocated in the lane's arena,
// rank[] = their offsets in 32-byte units, best first.  No compaction: a window that outgrows the arena goes to the cooperative kernel.
struct LaneRec { u8 *arena; u32 arenaBytes, top, nWinTr; i32 bestScore; bool overflow; u16 rank[LANE_RECS]; };

__device__ __forceinline__ const staramd_transcript *lrecT(const LaneRec &w, u32 k) { return (const staramd_transcript *)(w.arena + (u32)w.rank[k] * 32u); }

// blocksOverlap.cpp:3-40 on two exon lists
__device__ static u32 blocksOverlapLane(const staramd_exon *e1, u32 n1, const staramd_exon *e2, u32 n2) {
    u32 i1 = 0, i2 = 0, nOverlap = 0;
    while (i1 < n1 && i2 < n2) {
        const u64 rs1 = e1[i1].R, rs2 = e2[i2].R;
        const u64 re1 = rs1 + e1[i1].L, re2 = rs2 + e2[i2].L;
        const u64 gs1 = e1[i1].G, gs2 = e2[i2].G;
        if (rs1 >= re2) i2++;
        else if (rs2 >= re1) i1++;
        else if (gs1 - rs1 != gs2 - rs2) { if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
        else { nOverlap += (u32)(min(re1, re2) - max(rs1, rs2)); if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
    }
    return nOverlap;
}

// de-duplication against the recorded transcripts and ranked insert (stitchWindowAligns.cpp:267-303); t / ex = the candidate
__device__ static void recordLane(const staramd_params &P, const staramd_transcript &t, const staramd_exon *ex, LaneRec &wr) {
    const int Score = t.maxScore; const u32 ne = t.nExons;
    u32 iTr = 0;
    while (iTr < wr.nWinTr) {
        const staramd_transcript *r = lrecT(wr, iTr);
        const u32 nOverlap = blocksOverlapLane(ex, ne, (const staramd_exon *)((const u8 *)r + LREC_HDR), (u32)r->nExons);
        const u32 uNew = t.mappedLength - nOverlap, uOld = r->mappedLength - nOverlap;
        if (uNew == 0 && Score < r->maxScore) break;                                  // new one adds nothing and scores lower: dropped
        else if (uOld == 0) { for (u32 ii = iTr + 1; ii < wr.nWinTr; ii++) wr.rank[ii - 1] = wr.rank[ii]; wr.nWinTr--; }   // old one adds nothing: removed
        else iTr++;
    }
    wr.bestScore = wr.nWinTr ? lrecT(wr, 0)->maxScore : 0;                            // (records may have been removed above, the head among them)
    if (iTr != wr.nWinTr) return;
    for (iTr = 0; iTr < wr.nWinTr; iTr++) { const staramd_transcript *r = lrecT(wr, iTr); if (Score > r->maxScore || (Score == r->maxScore && t.gLength < r->gLength)) break; }
    if (iTr >= P.alignTranscriptsPerWindowNmax) return;                               // ranks behind a full list: dropped
    const u32 need = LREC_HDR + 32u * ne;
    if (wr.nWinTr >= LANE_RECS || wr.nWinTr + 1u > P.alignTranscriptsPerWindowNmax || wr.top + need > wr.arenaBytes) { wr.overflow = true; return; }
    const u32 off = wr.top; wr.top += need;
    for (u32 ii = wr.nWinTr; ii > iTr; ii--) wr.rank[ii] = wr.rank[ii - 1];
    wr.rank[iTr] = (u16)(off / 32u);
    wr.nWinTr++;
    u64 *d = (u64 *)(wr.arena + off); const u64 *sh = (const u64 *)&t;
#pragma unroll
    for (u32 i = 0; i < LREC_HDR / 8; i++) d[i] = sh[i];
    for (u32 k = 0; k < ne; k++) { const u64 *sw = (const u64 *)&ex[k]; u64 *de = d + LREC_HDR / 8 + 4u * k; de[0] = sw[0]; de[1] = sw[1]; de[2] = sw[2]; de[3] = sw[3]; }
    wr.bestScore = lrecT(wr, 0)->maxScore;
}

// leaf of the recursion: stitchWindowAligns.cpp:16-307.  ex = scratch copy of the used exons (modified here).
__device__ static void finalizeLane(StitchCtx &c, Hdr h, staramd_exon *ex, u32 chr, LaneRec &wr) {
    const DevIndex &X = *c.X; const staramd_params &P = X.P;
    c.nLeaves++;
    const u32 Lread = c.Lread, Str = c.str;
    int Score = h.Score; u32 tR2 = h.tR2; u64 tG2 = h.tG2;
    const u32 ne = h.nExons;
    // leaves that cannot leave a trace end here / behind the extensions (k_stitch.hip finalizeTranscript has the argument)
    const i32 iFragT = ex[0].iFrag == ex[ne - 1].iFrag ? (i32)ex[0].iFrag : -1;
    const i32 mateBest = iFragT < 0 ? 0x7FFFFFFF : (iFragT == 0 ? c.maxScoreMate[0] : c.maxScoreMate[1]);
    const i32 needScore = min(wr.bestScore, mateBest) - P.outFilterMultimapScoreRange;
    if (!P.chimSegmentMinPositive) {
        const u32 spacer = c.readLength[0] < Lread ? (Str == 0 ? c.readLength[0] : Lread - 1u - c.readLength[0]) : Lread;      // position of the mate spacer in R[] (none: Lread)
        const u32 availL = h.rStart > spacer ? h.rStart - spacer - 1u : h.rStart, availR = tR2 < spacer ? spacer - 1u - tR2 : Lread - 1u - tR2;
        i32 U = Score + (i32)availL + (i32)availR;
        if (X.glStep != 0) U = max(0, U + (X.glStep < 0 ? X.glScoreAt1 : X.glScoreAt1 + (i32)X.nBreak));
        if (U < needScore) return;
    }
    ExtRes e;
    const int vOrder0 = (Str == 0) ? 0 : 1;            // EXTEND_ORDER==1, roStr==Str
    for (int iOrd = 0; iOrd < 2; iOrd++) {
        const int which = iOrd == 0 ? vOrder0 : 1 - vOrder0;
        if (which == 0) {
            if (h.rStart > 0) {
                const u32 imate = ex[0].iFrag;
                if (growOnLane(c, h.rStart - 1, h.gStart - 1, -1, -1, h.rStart, tR2 - h.rStart + 1, h.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax,
                                P.alignEndsTypeExt[imate][(int)(Str != imate)] != 0, e)) {
                    h.nMatch += e.nMatch; h.nMM += e.nMM; Score += e.maxScore;
                    h.rStart -= e.extendL; h.gStart -= e.extendL;
                    ex[0].R = (u16)h.rStart; ex[0].G = h.gStart; ex[0].L = (u16)(ex[0].L + e.extendL);
                }
            }
        } else {
            if (tR2 < Lread) {
                const u32 imate = ex[ne - 1].iFrag;
                if (growOnLane(c, tR2 + 1, tG2 + 1, +1, +1, Lread - tR2 - 1, tR2 - h.rStart + 1, h.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax,
                                P.alignEndsTypeExt[imate][(int)(imate == Str)] != 0, e)) {
                    h.nMatch += e.nMatch; h.nMM += e.nMM; Score += e.maxScore;
                    tR2 += e.extendL; tG2 += e.extendL;
                    ex[ne - 1].L = (u16)(ex[ne - 1].L + e.extendL);
                }
            }
        }
    }
    const u64 gLength = tG2 + 1 - h.gStart;
    if (X.glStep != 0) {             // scoreGenomicLengthLog2scale != 0 (:221-225) as integer break points (ascending): how many are <= last exon end - first exon start = gLength
        u32 lo = 0, hi = X.nBreak;
        while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (gLength >= X.glBreak[mid]) lo = mid + 1; else hi = mid; }
        Score += X.glScoreAt1 + X.glStep * (i32)lo;
        Score = max(0, Score);
    }
    if (Score < needScore && !P.chimSegmentMinPositive) return;
    // ---- leaf filters (:83-219)
    if (!P.alignSoftClipAtReferenceEnds &&
        ((ex[ne - 1].G + Lread - ex[ne - 1].R) > (GLOBAL(u64, X.chrStart)[chr] + GLOBAL(u64, X.chrLength)[chr]) || ex[0].G < (GLOBAL(u64, X.chrStart)[chr] + ex[0].R))) return;
    u32 rLength = 0;
    for (u32 k = 0; k < ne; k++) rLength += ex[k].L;
    for (u32 k = 0; k + 1 < ne; k++) {                 // junction overhangs (:97-108)
        if (ex[k].canonSJ >= 0) {
            if (ex[k].sjAnnot == 1) {
                if ((ex[k].L < P.alignSJDBoverhangMin && (k == 0 || ex[k - 1].canonSJ == -3 || (ex[k - 1].sjAnnot == 0 && ex[k - 1].canonSJ >= 0)))
                    || (ex[k + 1].L < P.alignSJDBoverhangMin && (k == ne - 2 || ex[k + 1].canonSJ == -3 || (ex[k + 1].sjAnnot == 0 && ex[k + 1].canonSJ >= 0)))) return;
            } else {
                if (ex[k].L < P.alignSJoverhangMin + ex[k].shiftSJ[0] || ex[k + 1].L < P.alignSJoverhangMin + ex[k].shiftSJ[1]) return;
            }
        }
    }
    if (ne > 1 && ex[ne - 2].sjAnnot == 1 && ex[ne - 1].L < P.alignSJDBoverhangMin) return;
    u32 sjN = 0; u16 intronMotifs[3] = {0, 0, 0};
    for (u32 k = 0; k + 1 < ne; k++) if (ex[k].canonSJ >= 0) { sjN++; const u32 st = ex[k].sjStr; if (st == 0) intronMotifs[0]++; else if (st == 1) intronMotifs[1]++; else intronMotifs[2]++; }
    u8 sjMotifStrand;
    if (intronMotifs[1] > 0 && intronMotifs[2] == 0) sjMotifStrand = 1;
    else if (intronMotifs[1] == 0 && intronMotifs[2] > 0) sjMotifStrand = 2;
    else sjMotifStrand = 0;
    if (intronMotifs[1] > 0 && intronMotifs[2] > 0 && P.outFilterIntronStrandsRemoveInconsistent) return;
    if (sjN > 0 && sjMotifStrand == 0 && P.outSAMstrandFieldIntronMotif) return;
    if (P.outFilterIntronMotifs == 1) { for (u32 k = 0; k + 1 < ne; k++) if (ex[k].canonSJ == 0) return; }
    else if (P.outFilterIntronMotifs == 2) { for (u32 k = 0; k + 1 < ne; k++) if (ex[k].canonSJ == 0 && ex[k].sjAnnot == 0) return; }
    {   // minimum mapped length of a spliced mate (:154-167)
        u32 nsj = 0, exl = 0;
        for (u32 k = 0; k < ne; k++) {
            exl += ex[k].L;
            if (k == ne - 1 || ex[k].canonSJ == -3) {
                if (nsj > 0 && (exl < P.alignSplicedMateMapLmin || (u64)exl < (u64)(P.alignSplicedMateMapLminOverLmate * (double)(u64)(ex[k].iFrag ? c.readLength[1] : c.readLength[0])))) return;
                exl = 0; nsj = 0;
            } else if (ex[k].canonSJ >= 0) nsj++;
        }
    }
    if (P.outFilterBySJoutStage == 2) {                // unannotated junctions must be on the whitelist (:169-177)
        for (u32 k = 0; k + 1 < ne; k++)
            if (ex[k].canonSJ >= 0 && ex[k].sjAnnot == 0) {
                const u64 jS = ex[k].G + ex[k].L, jE = ex[k + 1].G - 1;
                if (junctionOnLane(jS, jE, X.sjNovelStart, X.sjNovelEnd, (int)X.sjNovelN) < 0) return;
            }
    }
    if (ex[0].iFrag != ex[ne - 1].iFrag) {             // both mates (:179-219)
        if (ex[ne - 1].G + ex[ne - 1].L <= ex[0].G) return;
        u32 iexM2 = ne;
        for (u32 k = 0; k + 1 < ne; k++) if (ex[k].canonSJ == -3) { iexM2 = k + 1; break; }
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
    if (iFragT == 0) c.maxScoreMate[0] = max(c.maxScoreMate[0], Score); else if (iFragT == 1) c.maxScoreMate[1] = max(c.maxScoreMate[1], Score);
    {
        const bool c1 = Score + P.outFilterMultimapScoreRange >= wr.bestScore || P.chimSegmentMinPositive;
        const bool c2 = iFragT >= 0 && Score + P.outFilterMultimapScoreRange >= (iFragT == 0 ? c.maxScoreMate[0] : c.maxScoreMate[1]);
        if (!(c1 || c2)) return;
    }
    // ---- the candidate as an output record (padding included: records are compared byte for byte)
    staramd_transcript o;
    o.iW = 0; o.exonOffset = 0;
    o.nExons = (u16)ne; o.rStart = (u16)h.rStart; o.rLength = (u16)rLength;
    o.roStart = (u16)((Str == 0) ? h.rStart : Lread - h.rStart - rLength);
    o.Str = (u8)Str; o.roStr = (u8)Str; o.iFrag = (i8)iFragT; o.sjMotifStrand = sjMotifStrand; o.Chr = chr;
    o.gStart = h.gStart; o.gLength = gLength; o.maxScore = Score; o.nMatch = h.nMatch; o.nMM = h.nMM; o.mappedLength = rLength;
    o.nGap = h.nGap; o.lGap = h.lGap; o.nDel = h.nDel; o.lDel = h.lDel; o.nIns = h.nIns; o.lIns = h.lIns;
    o.nUnique = (u16)h.nUnique; o.nAnchor = (u16)h.nAnchor;
    o.intronMotifs[0] = intronMotifs[0]; o.intronMotifs[1] = intronMotifs[1]; o.intronMotifs[2] = intronMotifs[2]; o.pad0 = 0; o.pad1 = 0;
    for (u32 k = 0; k < ne; k++) {
        if (k + 1 == ne) { ex[k].canonSJ = 0; ex[k].sjAnnot = 0; ex[k].sjStr = 0; ex[k].shiftSJ[0] = ex[k].shiftSJ[1] = 0; }
        else if (ex[k].canonSJ < 0) { ex[k].shiftSJ[0] = ex[k].shiftSJ[1] = 0; }
        ex[k].pad0 = 0; ex[k].pad1 = 0;
    }
    recordLane(P, o, ex, wr);
}

// next seed index > i whose bit is set in mask, nA if none
__device__ __forceinline__ u32 nextSeedLane(u32 mask, u32 i, u32 nA) {
    u32 mk = mask & ~((2u << i) - 1u);
    mk &= (1u << nA) - 1u;
    return mk ? (u32)__ffs((int)mk) - 1u : nA;
}

// depth-first walk of one window (stitchWindowAligns.cpp:8-353 called from ReadAlign_stitchPieces.cpp:321): include seed iA (if it
// stitches), then exclude it -- the undo-log walk of k_stitch.hip:stitchWindow, one lane.  Returns false when the records outgrew the lane.
// skipSingle / nSkipped: as in k_stitch.hip:stitchWindow (single-mate leaves of a two-mate window are not finalised)
__device__ static bool stitchWindowLane(StitchCtx &c, const DWin &win, const DWA *WAg, LaneRec &wr, const bool skipSingle, u32 &nSkipped) {
    const u32 nA = win.nWA;
    c.str = win.str;
    wr.nWinTr = 0; wr.top = 0; wr.overflow = false; wr.bestScore = 0;
    LFrame stack[LANE_SEEDS + 1];
    staramd_exon EX[LANE_SEEDS + 1], LEAF[LANE_SEEDS + 1];
    DWA WA[LANE_SEEDS];
    u32 compat[LANE_SEEDS];                              // seeds that CAN follow seed a (see k_stitch.hip:stitchWindow)
    for (u32 k = 0; k < nA; k++) WA[k] = WAg[k];
    for (u32 a = 0; a < nA; a++) {
        const u32 rAe = (u32)WA[a].rStart + WA[a].L - 1; const u64 gAe = WA[a].gStart + WA[a].L - 1;
        u32 mk = 0;
        for (u32 b = a + 1; b < nA; b++) {
            const u32 rBe = (u32)WA[b].rStart + WA[b].L - 1; const u64 gBe = WA[b].gStart + WA[b].L - 1;
            const bool fail = WA[b].iFrag == WA[a].iFrag && (rBe <= rAe || gBe <= gAe);
            if (!fail) mk |= 1u << b;
        }
        compat[a] = mk;
    }
    Hdr h; h.gStart = 0; h.tG2 = 0; h.nExons = 0; h.Score = 0; h.nMatch = h.nMM = h.nGap = h.lGap = h.nDel = h.lDel = h.nIns = h.lIns = 0;
    h.nUnique = h.nAnchor = 0; h.rStart = 0; h.tR2 = 0;
    u32 iA = 0, sp = 0, ex0R = 0; u64 ex0G = 0;
    u32 iLast = 0; u32 follow = ~0u;
    u32 fragFirst = 0, fragLast = 0, mate2 = 0;      // mates of the first / last seed of the working transcript; seeds of mate 2
    nSkipped = 0;
    if (skipSingle) for (u32 k = 0; k < nA; k++) if (WA[k].iFrag != 0) mate2 |= 1u << k;
    const u32 allSeeds = (1u << nA) - 1u;
    for (;;) {
        c.nNodes++;
        bool onlySingle = false;                     // the transcript holds one mate and no seed of the other mate is left: every leaf below is a single-mate one
        if (skipSingle && h.nExons > 0 && fragFirst == fragLast && iA < nA) onlySingle = (((fragFirst ? (allSeeds & ~mate2) : mate2) >> iA) == 0);
        if (iA >= nA || onlySingle) {                // leaf (stitchWindowAligns.cpp:14-16: nothing to do when tR2==0)
            if (h.tR2 != 0 && skipSingle && fragFirst == fragLast) nSkipped++;
            else if (h.tR2 != 0) {
                for (u32 k = 0; k < h.nExons; k++) LEAF[k] = EX[k];
                finalizeLane(c, h, LEAF, win.chr, wr);
                if (wr.overflow) return false;
            }
            if (sp == 0) break;
            sp--;                                    // back to the frame that included a seed: now exclude it
            h = stack[sp].h; iLast = stack[sp].iLast & 255u; fragLast = stack[sp].iLast >> 8; follow = h.nExons > 0 ? compat[iLast] : ~0u;
            iA = nextSeedLane(follow, stack[sp].iA, nA);
            if (h.nExons > 0) EX[h.nExons - 1] = stack[sp].eA;
            continue;
        }
        // ---- include branch (:311-345)
        const DWA a = WA[iA];
        Hdr hn = h; staramd_exon eA, eN; bool added = false; int dScore;
        if (h.nExons > 0) {
            eA = EX[h.nExons - 1];
            const staramd_exon eAold = eA;
            dScore = joinOnLane(c, h.tR2, h.tG2, a.rStart, a.gStart, a.L, a.iFrag, a.sjA, hn, eA, eN, added, ex0R, ex0G);
            if (dScore > -1000000) {
                stack[sp].h = h; stack[sp].iA = iA; stack[sp].iLast = iLast | (fragLast << 8); stack[sp].eA = eAold;
                EX[h.nExons - 1] = eA;
                if (added) { EX[h.nExons] = eN; hn.nExons = h.nExons + 1; }
            }
        } else {                                     // first seed of the transcript (:318-334)
            eN.R = a.rStart; eN.G = a.gStart; eN.L = a.L; eN.iFrag = a.iFrag; eN.sjA = a.sjA;
            eN.canonSJ = 0; eN.sjAnnot = 0; eN.sjStr = 0; eN.shiftSJ[0] = eN.shiftSJ[1] = 0; eN.pad0 = 0; eN.pad1 = 0;
            stack[sp].h = h; stack[sp].iA = iA; stack[sp].iLast = 0; stack[sp].eA = eN;
            EX[0] = eN;
            hn.rStart = a.rStart; hn.gStart = a.gStart; hn.nExons = 1; hn.nMatch = a.L;
            dScore = a.L;
        }
        if (dScore > -1000000) {
            if (a.nrep == 1) hn.nUnique++;
            if (a.anchor > 0) hn.nAnchor++;
            hn.Score = h.Score + dScore; hn.tR2 = (u32)a.rStart + a.L - 1; hn.tG2 = a.gStart + a.L - 1;
            if (h.nExons == 0) { ex0R = a.rStart; ex0G = a.gStart; fragFirst = a.iFrag; }
            h = hn; sp++;
            iLast = iA; fragLast = a.iFrag; follow = compat[iA];
        }
        // include succeeded: continue below it; include failed: exclude branch (:348-351) = same transcript, next seed
        iA = nextSeedLane(follow, iA, nA);
    }
    return true;
}

// copy the window's recorded transcripts (trAll[iW1][0..nWinTr-1]) into the result pools; false on pool overflow
__device__ static bool flushWindowLane(const DevBatch &B, const LaneRec &wr, DWinOut &o) {
    const u32 nTr = wr.nWinTr; u32 nEx = 0;
    o.trOffset = 0; o.nTr = 0; o.exOffset = 0; o.nEx = 0; o.headScore = 0; o.headGlen = 0;
    if (nTr == 0) return true;
    for (u32 k = 0; k < nTr; k++) nEx += lrecT(wr, k)->nExons;
    const u32 to = atomicAdd(&B.cursors[CUR_TR], nTr), eo = atomicAdd(&B.cursors[CUR_EX], nEx);
    if (to + nTr > B.trCap || eo + nEx > B.exCap) { atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_TRPOOL); return false; }
    const staramd_transcript *hd = lrecT(wr, 0);
    o.trOffset = to; o.nTr = nTr; o.exOffset = eo; o.nEx = nEx; o.headScore = hd->maxScore; o.headGlen = hd->gLength;
    u32 eoff = 0;
    for (u32 k = 0; k < nTr; k++) {
        const u64 *s = (const u64 *)lrecT(wr, k);
        const u32 ne = lrecT(wr, k)->nExons;
        u64 *dt = (u64 *)&B.trPool[to + k];
        dt[0] = (s[0] & 0xFFFFFFFFull) | ((u64)eoff << 32);      // exonOffset relative to the window block; k_gather rebases it and sets iW
#pragma unroll
        for (u32 i = 1; i < LREC_HDR / 8; i++) dt[i] = s[i];
        u64 *de = (u64 *)&B.exPool[eo + eoff];
        for (u32 i = 0; i < ne * 4; i++) de[i] = s[LREC_HDR / 8 + i];
        eoff += ne;
    }
    return true;
}

__device__ __forceinline__ void emptyWout(DWinOut &z) {
    z.trOffset = z.nTr = z.exOffset = z.nEx = 0; z.mm[0] = z.mm[1] = 0; z.sens[0] = z.sens[1] = 0x7FFFFFFF; z.minIn[0] = z.minIn[1] = 0;
    z.headScore = 0; z.candOff32 = 0; z.headGlen = 0; z.nCand = 0; z.pad = 0;
}

#ifndef LANE_WAVES
// Minimum waves per SIMD the register allocation is held to: 2 = up to 256 vector registers, which this kernel needs (202) to stay WITHOUT vector-register spills.
// Held to 3 (168 registers + ~40 spilled to scratch, beside 106 scalar registers spilled into vector-register lanes and 1.9 KB of per-lane arrays) the kernel returned
// transcripts whose exon rows were garbage, on hardware only, for ~1 read in 3 000 of a single-end set -- and whether it did depended on an unrelated edit of a branch that
// the set never takes (round 6, profiles/r06_lane_kernel_spill_corruption.txt: the same source is clean in the wavefront emulator under AddressSanitizer, clean at 2, and was
// clean at 3 until the scalar routines loaded one word instead of four bytes).  tests/test_isa_static.py holds the kernel to zero vector-register spills.  The kernel takes
// ~5 % of the reads of a 2x101 batch in 0.3 ms: its occupancy is not what the stage waits for (round 3, 1000 Mb, when it took most reads: 2 -> 43.6 ms, 3 -> 37.8, 4 -> 46.3).
#define LANE_WAVES 2
#endif
extern "C" __global__ void __launch_bounds__(256, LANE_WAVES) k_stitch_lane(const DevIndex *__restrict__ Xp, DevBatch B, u8 *laneArena, u32 laneArenaBytes, u32 ldsWords, u32 pruneEnable, u32 maxClass) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch
    const DevIndex &X = *Xp;
    const staramd_params &P = X.P;
    StitchCtx c; c.X = &X; c.nGstitch = 0; c.nStitchCalls = c.nExtendCalls = c.nNodes = c.nLeaves = c.nLeavesBound = c.nLeavesEarly = 0;
    c.ldsByte = threadIdx.x * (ldsWords * 4u);
    c.sens[0] = c.sens[1] = 0x7FFFFFFF; c.logOn = false; c.logOvf = false; c.candBase = nullptr; c.candTop = c.candCap = c.nCand = 0;
    gcInit(c.ca); gcInit(c.cb);
    LaneRec wr; wr.arena = laneArena + (u64)(blockIdx.x * blockDim.x + threadIdx.x) * laneArenaBytes; wr.arenaBytes = laneArenaBytes;
    const u32 nItemsRaw = B.cursors[CUR_ITEM];
    // items of a cost class above maxClass (class = bits of the walk-size estimate, k_windows) are left to the cooperative kernel: one lane would walk
    // them alone while the rest of the GPU waits.  k_order_scatter leaves the END of class c (heaviest first) in costHist[32 + c].
    const u32 posMin = maxClass < 31u ? B.costHist[32u + maxClass + 1u] : 0u;
    const u32 G = (nItemsRaw + 63u) / 64u;            // k_order_scatter: item of rank `pos` (heaviest class first) sits at order[(pos % G) * 64 + pos / G]
    const i32 perJ = max(0, P.sjdbScore) + max(0, max(max(P.scoreGap, P.scoreGapNoncan), max(P.scoreGapGCAG, P.scoreGapATAC)));
    const bool pruneOn = P.resultSelect == 1 && !P.chimSegmentMinPositive && X.glStep <= 0 && (pruneEnable & 3u) != 0
                         && P.scoreDelOpen <= 0 && P.scoreDelBase <= 0 && P.scoreInsOpen <= 0 && P.scoreInsBase <= 0;
    const bool sweepEnable = (pruneEnable & 2u) != 0, skipEnable = (pruneEnable & 4u) != 0;
    u32 nPruned = 0, nRewalk = 0, nLaneItems = 0, nRewalkWin = 0, nSkippedLeaves = 0;
    for (;;) {
        const u32 pos = atomicAdd(&B.cursors[CUR_ST_TICKET0], 1u);
        if (pos >= nItemsRaw) break;
        const u32 item = B.order[(pos % G) * 64u + pos / G];
        bool defer = (item & 0x80000000u) == 0 || pos < posMin;         // a window of a heavy read: cooperative kernel (candidate log, verify / replay)
        const u32 ir = item & 0x7FFFFFFFu;
        DRead rd;
        if (!defer) { rd = B.reads[ir]; defer = rd.wtOffset > LANE_SEEDS; }
        if (!defer) {
            const u32 w0 = rd.winOffset, nWin = rd.nWin;
            // ---- the read: lengths, mismatch budget, 4-bit packed copy into this lane's LDS slot
            c.Lread = (u32)(B.readOffset[ir + 1] - B.readOffset[ir]);
            c.readLength[0] = B.mate1Length[ir]; c.readLength[1] = (P.readNmates == 2 && c.readLength[0] < c.Lread) ? c.Lread - c.readLength[0] - 1 : 0;      // (a read of merged mates in a paired-end run: one piece, no second mate)
            c.mmMaxTotal = B.mmMaxTotal[ir];
            {
                const u32 *src = B.packed + (u64)ir * B.packWords;
                __attribute__((address_space(3))) u32 *dst = (__attribute__((address_space(3))) u32 *)((__attribute__((address_space(3))) u8 *)ldsReads + c.ldsByte);
                const u32 nw = (c.Lread + 7) / 8;
                for (u32 k = 0; k < nw; k++) dst[k] = GLOBAL(u32, src)[k];
            }
            // sweeps: see k_stitch_win (two-mate windows first; if their best clears the bar of every other window those are skipped unwalked)
            u32 sweep = 2;
            const i32 singleBar = pruneSingleBar(P, perJ, c.readLength[0], c.readLength[1], rd.wtOffset);
            if (pruneOn && sweepEnable && nWin > 1 && (u64)(nWin + 1u) * P.alignTranscriptsPerWindowNmax < P.alignTranscriptsPerReadNmax) {
                bool anyPair = false, anySingle = false;
                for (u32 k = 0; k < nWin; k++) { if (B.winPool[w0 + k].mates == 3u) anyPair = true; else anySingle = true; }
                if (anyPair && anySingle) sweep = 0;
            }
            i32 carry[2] = {0, 0}; i32 bestSoFar = 0;
            for (;;) {
                if (sweep == 2) { carry[0] = carry[1] = 0; bestSoFar = 0; }
                for (u32 iw = 0; iw < nWin && !defer; iw++) {
                    const u32 w = w0 + iw;
                    const DWin win = B.winPool[w];
                    if (sweep == 0 && win.mates != 3u) continue;
                    if (sweep == 1) {
                        if (win.mates == 3u) continue;                         // walked in sweep 0
                        DWinOut z; emptyWout(z); B.wout[w] = z;
                        nPruned++;
                        continue;
                    }
                    if (pruneOn && win.mates != 0 && sweep == 2) {
                        if ((u64)(nWin + 1u) * P.alignTranscriptsPerWindowNmax < P.alignTranscriptsPerReadNmax
                            && pruneWindow(P, perJ, win.mates, win.nWA, c.readLength[0], c.readLength[1], singleBar, bestSoFar)) {
                            DWinOut z; emptyWout(z); B.wout[w] = z;
                            nPruned++;
                            continue;
                        }
                    }
                    DWinOut o;
                    o.minIn[0] = carry[0]; o.minIn[1] = carry[1];
                    c.maxScoreMate[0] = carry[0]; c.maxScoreMate[1] = carry[1];
                    {   // two-mate windows: first without their single-mate leaves (see k_stitch_win)
                        bool skipSingle = pruneOn && skipEnable && win.mates == 3u && (u64)(nWin + 1u) * P.alignTranscriptsPerWindowNmax < P.alignTranscriptsPerReadNmax;
                        bool okW = true;
                        for (;;) {
                            u32 nSkipped = 0;
                            c.maxScoreMate[0] = carry[0]; c.maxScoreMate[1] = carry[1];
                            okW = stitchWindowLane(c, win, B.waPool + win.waOffset, wr, skipSingle, nSkipped);
                            if (okW && skipSingle && nSkipped) {
                                const i32 bar = singleBar;
                                if (!(bar < max(wr.bestScore, bestSoFar)) || wr.nWinTr >= P.alignTranscriptsPerWindowNmax) { skipSingle = false; nRewalkWin++; continue; }
                                nSkippedLeaves += nSkipped;
                            }
                            break;
                        }
                        if (!okW) { defer = true; break; }
                    }
                    if (!flushWindowLane(B, wr, o)) continue;
                    o.mm[0] = c.maxScoreMate[0]; o.mm[1] = c.maxScoreMate[1];
                    carry[0] = c.maxScoreMate[0]; carry[1] = c.maxScoreMate[1];
                    if (pruneOn && o.headScore > 0) bestSoFar = max(bestSoFar, o.headScore);
                    o.sens[0] = o.sens[1] = 0x7FFFFFFF;                      // the incoming maxScoreMate of a whole-read item is exact: the result is final
                    o.candOff32 = 0; o.nCand = 0; o.pad = 0;
                    B.wout[w] = o;
                }
                if (defer) break;
                if (sweep == 0) {
                    if (singleBar < bestSoFar) { sweep = 1; continue; }
                    sweep = 2; nRewalk++;
                    continue;
                }
                break;
            }
            if (!defer) nLaneItems++;
        }
        if (defer) { const u32 k = atomicAdd(&B.cursors[CUR_ST_HEAVY], 1u); B.heavyList[k] = item; }
    }
    // counters: one atomic per wavefront would need a reduction over lanes that left the loop at different times; per-lane atomics on a few
    // counters at the very end of the kernel are cheap enough (400 k lanes x 3)
    if (c.nGstitch) atomicAdd((unsigned long long *)&B.counters[DC_nGstitch], (unsigned long long)c.nGstitch);
    if (nPruned) atomicAdd((unsigned long long *)&B.counters[DC_nPrunedWin], (unsigned long long)nPruned);
    if (nRewalk) atomicAdd((unsigned long long *)&B.counters[DC_nRewalkRead], (unsigned long long)nRewalk);
    if (nLaneItems) atomicAdd((unsigned long long *)&B.counters[DC_nLaneItems], (unsigned long long)nLaneItems);
    if (nRewalkWin) atomicAdd((unsigned long long *)&B.counters[DC_nRewalkWin], (unsigned long long)nRewalkWin);
    if (nSkippedLeaves) atomicAdd((unsigned long long *)&B.counters[DC_nSkippedLeaves], (unsigned long long)nSkippedLeaves);
#if defined(STARAMD_PROFILE) || defined(STARAMD_SHADOW)
    atomicAdd((unsigned long long *)&B.counters[DC_nStitchCalls], (unsigned long long)c.nStitchCalls);
    atomicAdd((unsigned long long *)&B.counters[DC_nExtendCalls], (unsigned long long)c.nExtendCalls);
    atomicAdd((unsigned long long *)&B.counters[DC_nNodes], (unsigned long long)c.nNodes);
    atomicAdd((unsigned long long *)&B.counters[DC_nLeaves], (unsigned long long)c.nLeaves);
#endif
}
