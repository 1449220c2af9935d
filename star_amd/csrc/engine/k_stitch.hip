// k_stitch.hip -- kernel 3: stitch the seeds of every window into transcripts, extend, score, filter, rank.
//
// Replaces, per window, the second half of ReadAlign::stitchPieces (source/ReadAlign_stitchPieces.cpp:260-348)
// and everything below it:
//   stitchWindowAligns      source/stitchWindowAligns.cpp:8-353     include/exclude recursion + leaf filters + ranked insert
//   stitchAlignToTranscript source/stitchAlignToTranscript.cpp:9-415 gap fill, junction / indel placement, sjdb scoring
//   extendAlign             source/extendAlign.cpp:6-93
//   binarySearch2           source/binarySearch2.cpp:3-43
//   blocksOverlap           source/blocksOverlap.cpp:3-40
//
// Mapping: ONE WAVEFRONT PER WINDOW, all 64 lanes cooperating.
//   * The include/exclude recursion is an explicit depth-first walk over ONE working transcript with an undo log
//     (the reference copies a 1696-byte Transcript twice per node): the header lives in registers (wave-uniform),
//     a frame of the undo stack is {header, previous last exon, seed index} = 112 bytes pushed only when an include
//     succeeds, the exclude branch is a tail step.  Stack, exon rows and the record arena of the window live in the
//     wavefront's LDS slice (~13 KB), so a walk step costs LDS latency, not L2/HBM latency.
//   * Every per-base loop of the reference is a lane-parallel step, lane = read position:
//       extendAlign            match / mismatch ballots, prefix counts with v_mbcnt, break point = first set bit,
//                              best prefix = wave max of (score << 8 | 63 - lane)  (first position of the maximum)
//       gap fill               two ballots + popcounts
//       junction scan          (stitchAlignToTranscript.cpp:106-156) left scan = position of the k-th set bit of a ballot,
//                              right scan = prefix counts of the two "votes" ballots + per-lane motif penalty + wave argmax
//       repeat shifts jjL/jjR  first zero of an equality ballot
//       sjdb lookup            64-ary search over sjdbStart with coalesced probes instead of 19 dependent bisection steps
//       de-duplication         lane k runs blocksOverlap against recorded transcript k; block / remove / keep classes
//                              are ballots, the rank list is compacted and shifted with lane-parallel copies
//     Genome bytes of a scan are fetched by one coalesced load (64 consecutive bases per instruction), read bases come
//     from the 4-bit packed copy of the read staged once per window in LDS.
//   * Leaves are finalised in exactly the reference's order (include before exclude) because de-duplication and the
//     evolving window-best score depend on it.  Windows of a read are independent except for maxScoreMate[]; see
//     k_stitch_win for how that dependency is resolved exactly without serialising the windows of a read.
//   * Windows that outgrow the LDS slice (record arena) are deferred to a second launch of the same kernel whose work
//     space lives in global memory with worst-case sizes (big = 1).
#include "stitch_common.h"
#ifdef STARAMD_SHADOW
#include "stitch_scalar.h"
#else
struct ExtRes { i32 maxScore; u32 extendL, nMatch, nMM; };
#endif

#define NLANE 64u

#define GLOBAL_AS __attribute__((address_space(1)))
// the index arrays live in HBM: say so, otherwise the loads are FLAT (and every wait also drains the LDS counter)
__device__ __forceinline__ u8 gByte(const StitchCtx &c, u64 pos) { return ((const GLOBAL_AS u8 *)c.X->G)[(i64)pos]; }

// ---- extendAlign.cpp:6-93, lane = position of the scan -------------------------------------------------------------
__device__ static bool coopExtendBody(StitchCtx &c, u32 lane, u32 rStart, u64 gStart, int dR, int dG, u32 L, u32 Lprev, u32 nMMprev, u32 nMMmax, double pMMmax, bool extendToEnd, ExtRes &e);
__device__ __forceinline__ bool coopExtend(StitchCtx &c, u32 lane, u32 rStart, u64 gStart, int dR, int dG, u32 L, u32 Lprev, u32 nMMprev, u32 nMMmax, double pMMmax, bool extendToEnd, ExtRes &e) {
    PROF_T0();
    bool r = coopExtendBody(c, lane, rStart, gStart, dR, dG, L, Lprev, nMMprev, nMMmax, pMMmax, extendToEnd, e);
    PROF_ADD(c, 2);
    return r;
}
__device__ static bool coopExtendBody(StitchCtx &c, u32 lane, u32 rStart, u64 gStart, int dR, int dG, u32 L, u32 Lprev, u32 nMMprev, u32 nMMmax, double pMMmax, bool extendToEnd, ExtRes &e) {
    DIAG(c.nExtendCalls++);
    e.maxScore = 0; e.extendL = 0; e.nMatch = 0; e.nMM = 0;
    if (extendToEnd) {                      // --alignEndsType Extend*: rarely used, wave-uniform scalar loop (:18-56)
        int Score = 0, nMatch = 0, nMM = 0, iExt;
        for (iExt = 0; iExt < (int)L; iExt++) {
            int iS = dR * iExt, iG = dG * iExt;
            u8 gc = 5;
            if ((gStart + (i64)iG) == (u64)(-1) || (gc = (u8)first32(gByte(c, gStart + (i64)iG))) == 5) { e.extendL = 0; e.maxScore = -999999999; e.nMatch = 0; e.nMM = nMMmax + 1; return true; }
            u8 rc = (u8)first32(RD(c, (u32)((int)rStart + iS)));
            if (rc == STARAMD_SPACER_BASE) break;
            if (rc > 3 || gc > 3) continue;
            if (gc == rc) { nMatch++; Score += 1; } else { nMM++; Score -= 1; }
        }
        if (iExt > 0) { e.extendL = (u32)iExt; e.maxScore = Score; e.nMatch = (u32)nMatch; e.nMM = (u32)nMM; return true; }
        return false;
    }
    // The reference's loop runs `for (int i = 0; i < (int) L; i++)` (extendAlign.cpp:59) over an unsigned 64-bit L: a length that went "negative"
    // -- the second mate starting before the first exon (stitchAlignToTranscript.cpp:390: gBstart - exons[0][EX_G] + exons[0][EX_R] with
    // --alignEndsProtrude) -- means no extension at all, not four billion bases (found by the hardware fuzzer in round 3)
    if ((int)L <= 0) return false;
    const double thrBreak = fmin(pMMmax * (double)(u64)(Lprev + L), (double)nMMmax);
    int scoreBase = 0; u32 nMatchBase = 0, nMMBase = 0; int best = 0;
    for (u32 base = 0; base < L; base += NLANE) {
        u32 i = base + lane;
        bool act = i < L;
        u64 gpos = gStart + (u64)(i64)(dG * (int)i);
        u8 gc = 5, rc = STARAMD_SPACER_BASE;
        if (act && gpos != (u64)(-1)) { gc = gByte(c, gpos); rc = RD(c, (u32)((int)rStart + dR * (int)i)); }
        bool stop = !act || gc == 5 || rc == STARAMD_SPACER_BASE;            // :61-63 (and end of the scan)
        u64 sm = __ballot(stop);
        u32 Leff = sm ? firstLane(sm) : NLANE;
        bool in = lane < Leff;
        bool skip = rc > 3 || gc > 3;                                        // :65 N in read or genome: no score
        bool isM = in && !skip && gc == rc, isX = in && !skip && gc != rc;
        u64 mM = __ballot(isM), mX = __ballot(isX);
        // :78 a mismatch stops the scan when the mismatches BEFORE it already exhaust the budget
        bool brk = isX && ((double)(u32)(nMMBase + cntBelow(mX) + nMMprev) >= thrBreak);
        u64 bm = __ballot(brk);
        u32 lim = bm ? min(Leff, firstLane(bm)) : Leff;
        u64 keep = lim >= NLANE ? ~0ull : ((1ull << lim) - 1ull);
        mM &= keep; mX &= keep;
        u32 cM = cntUpTo(mM, lane), cX = cntUpTo(mX, lane);
        int score_i = scoreBase + (int)cM - (int)cX; u32 nMM_i = nMMBase + cX, nMatch_i = nMatchBase + cM;
        // :69-75 a match records a new best prefix if the score beats the recorded one and the mismatch rate allows it
        bool cand = isM && lane < lim && score_i > best
                    && (double)(u32)(nMM_i + nMMprev) <= fmin(pMMmax * (double)(u64)(Lprev + i + 1), (double)nMMmax);
        u32 key = cand ? ((((u32)score_i) << 8) | (63u - lane)) : 0u;
        u32 kmax = waveMaxU32(key);
        if (kmax) {
            u32 bl = 63u - (kmax & 255u);
            best = (int)(kmax >> 8);
            e.extendL = base + bl + 1; e.maxScore = best; e.nMatch = laneGet32(nMatch_i, bl); e.nMM = laneGet32(nMM_i, bl);
        }
        c.nGstitch += min(lim + 1u, NLANE);
        if (lim < NLANE) break;
        u32 pM = (u32)__popcll(mM), pX = (u32)__popcll(mX);
        scoreBase += (int)pM - (int)pX; nMatchBase += pM; nMMBase += pX;
    }
    return e.extendL > 0;
}

#ifdef STARAMD_SHADOW
__device__ static bool coopExtendChecked(StitchCtx &c, u32 lane, u32 rStart, u64 gStart, int dR, int dG, u32 L, u32 Lprev, u32 nMMprev, u32 nMMmax, double pMMmax, bool extendToEnd, ExtRes &e) {
    bool r = coopExtend(c, lane, rStart, gStart, dR, dG, L, Lprev, nMMprev, nMMmax, pMMmax, extendToEnd, e);
    ExtRes s; bool rs = growOnLane(c, rStart, gStart, dR, dG, L, Lprev, nMMprev, nMMmax, pMMmax, extendToEnd, s);
    bool bad = r != rs || (r && (s.maxScore != e.maxScore || s.extendL != e.extendL || s.nMatch != e.nMatch || s.nMM != e.nMM));
    if (lane == 0) { if (bad) atomicAdd((unsigned long long *)&c.shadow[2], 1ull); atomicAdd((unsigned long long *)&c.shadow[3], 1ull); }
    return r;
}
#define COOP_EXTEND coopExtendChecked
#else
#define COOP_EXTEND coopExtend
#endif

// ---- binarySearch2.cpp:3-43: index of the junction (x = start, y = end) in the sorted sjdb arrays, < 0 if absent ----
// 64-ary search: every round 64 lanes probe 64 evenly spaced starts, the count of "smaller" answers narrows the range
// 64-fold; then the run of equal starts is compared against y by all lanes at once.  (sjdb junctions are unique
// (start,end) pairs -- sjdbPrepare collapses duplicates -- so "the" match is well defined.)
__device__ static int coopSjdbFind(u32 lane, u64 x, u64 y, const u64 *Xs_, const u64 *Ys_, u32 N) {
    const GLOBAL_AS u64 *Xs = (const GLOBAL_AS u64 *)Xs_, *Ys = (const GLOBAL_AS u64 *)Ys_;
    if (N == 0 || x > Xs[N - 1] || x < Xs[0]) return -1;
    u32 lo = 0, hi = N;                                   // first index with Xs[idx] >= x lies in [lo, hi]
    while (hi - lo > NLANE) {
        u32 step = (hi - lo + NLANE - 1) / NLANE;
        u32 p = lo + lane * step;
        bool less = p < hi && Xs[p] < x;
        u32 cl = (u32)__popcll(__ballot(less));           // probes are sorted: lanes 0..cl-1 are "less"
        u32 nProbe = (hi - lo + step - 1) / step;
        u32 nlo = cl ? lo + (cl - 1) * step + 1 : lo;
        u32 nhi = cl == 0 ? lo : (cl < nProbe ? lo + cl * step : hi);
        lo = nlo; hi = nhi;
        if (hi <= lo) { hi = lo; break; }
    }
    if (hi > lo) { bool less = lo + lane < hi && Xs[lo + lane] < x; lo += (u32)__popcll(__ballot(less)); }
    // lo = first index with Xs >= x
    for (u32 base = lo; base < N; base += NLANE) {
        u32 k = base + lane;
        bool sameX = k < N && Xs[k] == x;
        bool hit = sameX && Ys[k] == y;
        u64 hm = __ballot(hit);
        if (hm) return (int)(base + firstLane(hm));
        if (__ballot(sameX) != ~0ull) break;               // run of equal starts ended inside this chunk
    }
    return -1;
}

// the same question through DevIndex::sjdbHash (dev.h): the lanes probe 64 consecutive slots of the table at once -- one round trip instead of the
// three or four dependent ones of the 64-ary search above (the junction arrays of a human annotation are 2 x 2.8 MB: every round is an L2 access)
__device__ static int coopSjdbHash(u32 lane, u64 x, u64 y, const u64 *tab_, u32 mask, u32 &info) {
    const GLOBAL_AS u64 *tab = (const GLOBAL_AS u64 *)tab_;
    const u32 h0 = sjdbHashSlot(x, mask);
    for (u32 step = 0; step <= mask; step += NLANE) {
        const u32 h = (h0 + step + lane) & mask;
        const u64 s = tab[2u * h], e = tab[2u * h + 1u];
        const bool empty = s == 0;
        const bool hit = !empty && (s & ((1ull << SJH_START_BITS) - 1ull)) == x && (e & ((1ull << SJH_START_BITS) - 1ull)) == y;
        const u64 em = __ballot(empty), hm = __ballot(hit);
        if (hm) { const u32 l = firstLane(hm); if (!em || l < firstLane(em)) { info = laneGet32((u32)(e >> SJH_START_BITS), l); return (int)(laneGet64(s, l) >> SJH_START_BITS) - 1; } return -1; }
        if (em) return -1;
    }
    return -1;
}

// ---- stitchAlignToTranscript.cpp:9-415 ------------------------------------------------------------------------------
// h / eA are working copies (wave-uniform): the caller commits them (and eN when added) only when the returned score is
// > -1000000, so a failed stitch leaves the transcript untouched.  ex0R / ex0G = start of the first exon.
__device__ static int coopStitch(StitchCtx &c, u32 lane, u32 rAend, u64 gAend, u32 rBstart, u64 gBstart, u32 L, u32 iFragB, i32 sjAB,
                                 Hdr &h, staramd_exon &eA, staramd_exon &eN, bool &added, u32 ex0R, u64 ex0G) {
    const DevIndex &X = *c.X; const staramd_params &P = X.P;
    DIAG(c.nStitchCalls++);
    added = false;
    if (h.nExons >= STARAMD_MAX_N_EXONS) return -1000010;
    int Score = 0;
    if (sjAB != -1 && eA.sjA == sjAB && eA.iFrag == iFragB && rBstart == rAend + 1 && gAend + 1 < gBstart) {
        // both seeds come from the same inserted sjdb sequence: the junction is the annotated one (:18-34)
        SPROF_KIND(0);
        const u32 sInfo = first32(GLOBAL(u32, X.sjdbInfo)[sjAB]), sMotif = SJ_INFO_MOTIF(sInfo), sShL = SJ_INFO_SHL(sInfo), sShR = SJ_INFO_SHR(sInfo);
        if (sMotif == 0 && (L <= sShR || eA.L <= sShL)) return -1000006;
        eN.L = (u16)L; eN.R = (u16)rBstart; eN.G = gBstart;
        eA.canonSJ = (i8)sMotif; eA.shiftSJ[0] = (u16)sShL; eA.shiftSJ[1] = (u16)sShR;
        eA.sjAnnot = 1; eA.sjStr = (u8)SJ_INFO_STRAND(sInfo);
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
            if (gGap == 0 && rGap == 0) { SPROF_KIND(1);
            } else if (gGap > 0 && rGap > 0 && rGap == gGap) {
                SPROF_KIND(1);
                // ---- equal gap: score the bases in between (:80-93)
                for (int base = 1; base <= rGap; base += (int)NLANE) {
                    int ii = base + (int)lane;
                    bool act = ii <= rGap;
                    u8 gc = 4, rc = 4;
                    if (act) { gc = gByte(c, gAend + ii); rc = RD(c, rAend + ii); }
                    bool ok = act && gc < 4 && rc < 4;
                    u32 pm = (u32)__popcll(__ballot(ok && rc == gc)), px = (u32)__popcll(__ballot(ok && rc != gc));
                    Score += (int)pm - (int)px; nMatch += pm; nMM += px;
                }
                c.nGstitch += (u32)rGap;
            } else if (gGap > rGap) {
                // ---- deletion or junction (:95-253)
                nDel = 1; Del = (u64)(i64)(gGap - rGap);
                if (Del > P.alignIntronMax && P.alignIntronMax > 0) return -1000003;
                SPROF_KIND(2);
                const int eAL = (int)eA.L;
                PROF_T0();
                // left scan (:104-109): walk left from the end of A while fewer than scoreStitchSJshift+1 positions
                // favour A over B, never past the start of A's exon.  jStart = where the walk stops.
                int jStart;
                {
                    const int need = P.scoreStitchSJshift + 1, lowLimit = 1 - eAL;
                    if (need <= 0) jStart = 0;
                    else {
                        int cnt = 0; jStart = lowLimit;
                        for (int base = 0;; base += (int)NLANE) {
                            int j = -(base + (int)lane);
                            bool valid = j >= lowLimit;
                            bool bad = false;
                            if (valid) {
                                u8 rc = RD(c, (u32)((int)rAend + j)), gB = gByte(c, gBstart1 + (i64)j);
                                bad = rc != gB && gB < 4 && rc == gByte(c, gAend + (i64)j);
                            }
                            u64 bmk = __ballot(bad);
                            int pc = (int)__popcll(bmk);
                            if (cnt + pc >= need) {
                                for (int k = need - cnt; k > 1; k--) bmk &= bmk - 1;
                                jStart = -(base + (int)firstLane(bmk));
                                break;
                            }
                            cnt += pc;
                            if (-(base + (int)NLANE - 1) <= lowLimit) break;      // reached the start of the exon
                        }
                    }
                }
                
                // right scan (:111-156): best junction position by votes of the bases + motif penalty
                int maxScore2 = -999999; int jPen = 0;
                const bool isIntron = Del >= P.alignIntronMin;
                const int jEnd = (int)rBend - (int)rAend;
                {
                    int s1base = 0;
                    for (int base = 0; jStart + base < jEnd; base += (int)NLANE) {
                        int j = jStart + base + (int)lane;
                        bool act = j < jEnd;
                        u8 ra = 9, gA = 8, gB = 7;
                        if (act) { ra = RD(c, (u32)((int)rAend + j)); gA = gByte(c, gAend + (i64)j); gB = gByte(c, gBstart1 + (i64)j); }
                        bool plus = act && ra == gA && ra != gB, minus = act && ra != gA && ra == gB;
                        u64 mp = __ballot(plus), mn = __ballot(minus);
                        int s1 = s1base + (int)cntUpTo(mp, lane) - (int)cntUpTo(mn, lane);
                        int jCan1 = -1, jPen1 = 0;
                        if (isIntron && act) {
                            u8 d1 = gByte(c, gAend + (i64)j + 1), d2 = gByte(c, gAend + (i64)j + 2), a1 = gByte(c, gBstart1 + (i64)j - 1), a2 = gB;
                            if (d1 == 2 && d2 == 3 && a1 == 0 && a2 == 2) jCan1 = 1;
                            else if (d1 == 1 && d2 == 3 && a1 == 0 && a2 == 1) jCan1 = 2;
                            else if (d1 == 2 && d2 == 1 && a1 == 0 && a2 == 2) { jCan1 = 3; jPen1 = P.scoreGapGCAG; }
                            else if (d1 == 1 && d2 == 3 && a1 == 2 && a2 == 1) { jCan1 = 4; jPen1 = P.scoreGapGCAG; }
                            else if (d1 == 0 && d2 == 3 && a1 == 0 && a2 == 1) { jCan1 = 5; jPen1 = P.scoreGapATAC; }
                            else if (d1 == 2 && d2 == 3 && a1 == 0 && a2 == 3) { jCan1 = 6; jPen1 = P.scoreGapATAC; }
                            else { jCan1 = 0; jPen1 = P.scoreGapNoncan; }
                        }
                        int s2 = s1 + jPen1;
                        u32 key = act ? ((((u32)(s2 + 1000000)) << 8) | (63u - lane)) : 0u;
                        u32 kmax = waveMaxU32(key);
                        int ms2 = (int)(kmax >> 8) - 1000000;
                        if (kmax && ms2 > maxScore2) {                      // strict: the first position of the maximum wins
                            u32 bl = 63u - (kmax & 255u);
                            maxScore2 = ms2; jR = jStart + base + (int)bl; jCan = (int)laneGet32((u32)jCan1, bl); jPen = (int)laneGet32((u32)jPen1, bl);
                        }
                        s1base += (int)__popcll(mp) - (int)__popcll(mn);
                    }
                    c.nGstitch += (u32)(jEnd - jStart) * (isIntron ? 5u : 2u);
                }
                
                // repeat length left / right of the junction (:159-166)
                u32 jjL = 0, jjR = 0;
                for (u32 base = 0;; base += NLANE) {
                    u32 k = base + lane;
                    bool ok = gAend + (i64)jR >= (u64)k;
                    if (ok) { u8 x = gByte(c, gAend - k + (i64)jR); ok = x == gByte(c, gBstart1 - k + (i64)jR) && x < 4 && k <= 255; }
                    u64 fm = __ballot(!ok);
                    if (fm) { jjL = base + firstLane(fm); break; }
                }
                for (u32 base = 0;; base += NLANE) {
                    u32 k = base + lane;
                    bool ok = gAend + k + (i64)jR + 1 < X.nGenome;
                    if (ok) { u8 x = gByte(c, gAend + k + (i64)jR + 1); ok = x == gByte(c, gBstart1 + k + (i64)jR + 1) && x < 4 && k <= 255; }
                    u64 fm = __ballot(!ok);
                    if (fm) { jjR = base + firstLane(fm); break; }
                }
                c.nGstitch += 2u * (jjL + jjR + 2u);
                
                if (jCan <= 0) {                                     // flush a non-canonical junction left (:168-173)
                    jR -= (int)jjL;
                    if (eAL + jR < 1) return -1000005;
                    jjR += jjL; jjL = 0;
                }
                // re-score the bases between the seeds with the junction in place (:177-195)
                {
                    const int i0 = min(1, jR + 1), i1 = max(rGap, jR);
                    for (int base = 0; i0 + base <= i1; base += (int)NLANE) {
                        int ii = i0 + base + (int)lane;
                        bool act = ii <= i1;
                        u8 gc = 4, rc = 4;
                        if (act) { gc = (ii <= jR) ? gByte(c, gAend + (i64)ii) : gByte(c, gBstart1 + (i64)ii); rc = RD(c, (u32)((int)rAend + ii)); }
                        bool ok = act && gc < 4 && rc < 4;
                        bool inGap = ii >= 1 && ii <= rGap;
                        u32 pa = (u32)__popcll(__ballot(ok && rc == gc && inGap));
                        u32 pb = (u32)__popcll(__ballot(ok && rc != gc));
                        u32 pc = (u32)__popcll(__ballot(ok && rc != gc && !inGap));
                        Score += (int)pa - (int)pb - (int)pc; nMatch += pa; nMatch -= pc; nMM += pb;
                    }
                    c.nGstitch += (u32)(i1 - i0 + 1);
                }
                
                int sjdbInd = -1; u32 jInfo = 0;          // the table answers with the junction's motif / strand / shifts in the same slot
                if (X.sjdbN > 0) {
                    if (X.sjdbHash) sjdbInd = coopSjdbHash(lane, gAend + (i64)jR + 1, gBstart1 + (i64)jR, X.sjdbHash, X.sjdbHashMask, jInfo);
                    else { sjdbInd = coopSjdbFind(lane, gAend + (i64)jR + 1, gBstart1 + (i64)jR, X.sjdbStart, X.sjdbEnd, X.sjdbN); if (sjdbInd >= 0) jInfo = first32(GLOBAL(u32, X.sjdbInfo)[sjdbInd]); }
                }
                
                if (sjdbInd < 0) {
                    if (isIntron) Score += P.scoreGap + jPen;
                    else { Score += (int)Del * P.scoreDelBase + P.scoreDelOpen; jCan = -1; eA.sjAnnot = 0; }
                } else {
                    jCan = (int)SJ_INFO_MOTIF(jInfo);
                    if (jCan == 0) {
                        const u32 sShL = SJ_INFO_SHL(jInfo);
                        if (L <= sShL || eA.L <= sShL) return -1000006;
                        jR += (int)sShL;
                        if ((u64)rAend + (i64)jR >= rBend) return -1000006;
                        jjL = sShL; jjR = SJ_INFO_SHR(jInfo);
                    }
                    eA.sjAnnot = 1; eA.sjStr = (u8)SJ_INFO_STRAND(jInfo);
                    Score += P.sjdbScore;
                }
                eA.shiftSJ[0] = (u16)jjL; eA.shiftSJ[1] = (u16)jjR; eA.canonSJ = (i8)jCan;
                if (eA.sjAnnot == 0) eA.sjStr = (jCan > 0) ? (u8)(2 - jCan % 2) : 0;
            } else if (rGap > gGap) {
                // ---- insertion (:255-305): short and rare; wave-uniform scalar loops
                Ins = (u32)(rGap - gGap); nIns = 1;
                if (gGap == 0) jR = 0;
                else if (gGap < 0) { jR = 0; Score -= -gGap; }
                else {
                    int Score1 = 0, maxScore1 = 0;
                    const int tieStep = P.alignInsertionFlushRight ? 0 : 1;
                    for (int jR1 = 1; jR1 <= gGap; jR1++) {
                        u8 gc = gByte(c, gAend + jR1);
                        if (gc < 4) { Score1 += (RD(c, rAend + jR1) == gc) ? 1 : -1; Score1 += (RD(c, rAend + Ins + jR1) == gc) ? -1 : +1; }
                        if (Score1 >= maxScore1 + tieStep) { maxScore1 = Score1; jR = jR1; }     // flush right: an equal score moves the insertion right (:273)
                    }
                    for (int ii = 1; ii <= gGap; ii++) {
                        u32 r1 = rAend + ii + (ii <= jR ? 0 : Ins);
                        u8 gc = gByte(c, gAend + ii), rc = RD(c, r1);
                        if (gc < 4 && rc < 4) { if (rc == gc) { Score += 1; nMatch++; } else { Score -= 1; nMM++; } }
                    }
                    c.nGstitch += 2u * (u32)gGap;
                }
                if (P.alignInsertionFlushRight) {
                    for (; jR < (int)rBend - (int)rAend - (int)Ins; jR++) { u8 gc = gByte(c, gAend + (i64)jR + 1); if (RD(c, (u32)((int)rAend + jR + 1)) != gc || gc == 4) break; }
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
            // ---- mate 2 after mate 1 (:352-405)
            if (P.alignMatesGapMax > 0 && gBstart > eA.G + eA.L + P.alignMatesGapMax) return -1000004;
            SPROF_KIND(3);
            Score += (int)L;
            ExtRes e;
            if (coopExtend(c, lane, rAend + 1, gAend + 1, 1, 1, STARAMD_READ_LEN_MAX, h.nMatch, h.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax, P.alignEndsTypeExt[eA.iFrag][1] != 0, e)) {
                h.nMatch += e.nMatch; h.nMM += e.nMM; Score += e.maxScore; eA.L = (u16)(eA.L + e.extendL);
            }
            eN.R = (u16)rBstart; eN.G = gBstart; eN.L = (u16)L; h.nMatch += L;
            u32 extlen = P.alignEndsTypeExt[iFragB][1] ? STARAMD_READ_LEN_MAX : (u32)(gBstart - ex0G + ex0R);
            if (coopExtend(c, lane, rBstart - 1, gBstart - 1, -1, -1, extlen, h.nMatch, h.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax, P.alignEndsTypeExt[iFragB][1] != 0, e)) {
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

// ---- address spaces -------------------------------------------------------------------------------------------------
// The walk state of a window (undo stack, exon rows, leaf copy, rank list, seed list) ALWAYS lives in the wavefront's
// LDS slice and is accessed through LDS-typed pointers (ds_read/ds_write, not flat).  Only the arena of recorded
// transcripts has two homes: LDS (BIG = false) or, for the few windows that outgrow it, global memory (BIG = true).
#define LDS __attribute__((address_space(3)))
template <bool BIG> struct AS;
template <> struct AS<false> { typedef LDS u8 *u8p; typedef LDS u64 *u64p; typedef LDS staramd_transcript *trp; typedef LDS staramd_exon *exp; };
template <> struct AS<true> { typedef u8 *u8p; typedef u64 *u64p; typedef staramd_transcript *trp; typedef staramd_exon *exp; };
// struct copies out of / into LDS (C++ copy constructors only take generic references)
template <class T> __device__ __forceinline__ T ldsGet(const LDS T *p) {
    T v; const LDS u32 *s = (const LDS u32 *)p; u32 *d = (u32 *)&v;
#pragma unroll
    for (u32 i = 0; i < sizeof(T) / 4; i++) d[i] = s[i];
    return v;
}
template <class T> __device__ __forceinline__ void ldsPut(LDS T *p, const T &v) {
    LDS u32 *d = (LDS u32 *)p; const u32 *s = (const u32 *)&v;
#pragma unroll
    for (u32 i = 0; i < sizeof(T) / 4; i++) d[i] = s[i];
}

// blocksOverlap.cpp:3-40 on two exon lists in the output record format (run by one lane)
template <class P1, class P2> __device__ static u32 blocksOverlap(P1 e1, u32 n1, P2 e2, u32 n2) {
    u32 i1 = 0, i2 = 0, nOverlap = 0;
    while (i1 < n1 && i2 < n2) {
        u64 rs1 = e1[i1].R, rs2 = e2[i2].R;
        u64 re1 = rs1 + e1[i1].L, re2 = rs2 + e2[i2].L;
        u64 gs1 = e1[i1].G, gs2 = e2[i2].G;
        if (rs1 >= re2) i2++;
        else if (rs2 >= re1) i1++;
        else if (gs1 - rs1 != gs2 - rs2) { if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
        else { nOverlap += (u32)(min(re1, re2) - max(rs1, rs2)); if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
    }
    return nOverlap;
}

// transcripts recorded for the current window: records in the OUTPUT format (staramd_transcript followed by its
// exons) bump-allocated in the window's arena; rank[] holds their offsets (32-byte units), best first.
// big = false: arena in the wavefront's LDS slice (arenaL); big = true: arena in HBM (arenaG).  The walk itself is
// instantiated once; only the few routines that touch records are specialised on the address space.
struct WinRec { LDS u8 *arenaL; u8 *arenaG; u32 arenaBytesL, arenaBytesG; bool big; u32 arenaBytes; u32 top; LDS u16 *rank; u32 nWinTr; bool overflow; i32 bestScore; };
template <bool BIG> struct ArenaSel;
template <> struct ArenaSel<false> { static __device__ __forceinline__ LDS u8 *get(const WinRec &w) { return w.arenaL; } };
template <> struct ArenaSel<true> { static __device__ __forceinline__ u8 *get(const WinRec &w) { return w.arenaG; } };
#define REC_HDR 96
static_assert(sizeof(staramd_transcript) == REC_HDR, "record header is the output transcript record");
static_assert(sizeof(staramd_exon) == 32, "exon record is 32 bytes");
template <bool BIG> __device__ __forceinline__ typename AS<BIG>::trp recAt(const WinRec &w, u32 off32) { return (typename AS<BIG>::trp)(ArenaSel<BIG>::get(w) + off32 * 32u); }
template <bool BIG> __device__ __forceinline__ typename AS<BIG>::trp recT(const WinRec &w, u32 k) { return recAt<BIG>(w, w.rank[k]); }

// slide the live records to the front of the arena (only when the bump pointer hits the end); run by lane 0
template <bool BIG> __device__ static u32 compactArena(WinRec &w) {
    u32 newTop = 0; i32 lastOrig = -1;
    for (u32 step = 0; step < w.nWinTr; step++) {
        u32 best = 0xFFFFFFFFu, bk = 0;
        for (u32 k = 0; k < w.nWinTr; k++) { u32 o = w.rank[k]; if ((i32)o > lastOrig && o < best) { best = o; bk = k; } }
        if (best == 0xFFFFFFFFu) break;
        u32 words = (REC_HDR + 32u * recAt<BIG>(w, best)->nExons) / 8;
        typename AS<BIG>::u64p s = (typename AS<BIG>::u64p)(ArenaSel<BIG>::get(w) + best * 32u), d = (typename AS<BIG>::u64p)(ArenaSel<BIG>::get(w) + newTop);
        if (d != s) for (u32 i = 0; i < words; i++) d[i] = s[i];
        w.rank[bk] = (u16)(newTop / 32u);
        lastOrig = (i32)best; newTop += words * 8;
        // records moved so far now sit at offsets < newTop <= best, i.e. never "> lastOrig" again
    }
    return newTop;
}

// record decision is taken: de-duplicate candidate (o, exon rows ex / x) against the recorded transcripts and insert it
// by rank (stitchWindowAligns.cpp:267-303).  Used by the walk (pass 0 / full re-walk) and by the replay of a candidate log.
// hdr: the 12 words of the candidate's output record header (LDS staging slot in the walk, registers in the replay)
template <bool BIG, class EXP, class HP> __device__ static void recordCandidateImpl(const staramd_params &P, u32 lane, const int Score, const u64 gLength, const u32 mappedLength, const u32 ne, HP hdr,
                                                                                   const staramd_exon &x, EXP ex, WinRec &wr) {
    // ---- de-duplication against the recorded transcripts (:267-285): lane k classifies record k
    //   BLOCK  new one adds nothing to record k and scores lower  -> the walk over the list stops, new one is dropped
    //   REMOVE record k adds nothing to the new one                -> record k is removed (if met before a BLOCK)
    {
        const u32 nW0 = wr.nWinTr;
        u32 outN = 0; bool blocked = false; u32 base = 0;
        for (; base < nW0 && !blocked; base += NLANE) {
            u32 k = base + lane; bool have = k < nW0;
            u16 rk = 0; u32 cls = 0;
            if (have) {
                rk = wr.rank[k];
                typename AS<BIG>::trp r = recAt<BIG>(wr, rk);
                u32 nOverlap = blocksOverlap(ex, ne, (typename AS<BIG>::exp)((typename AS<BIG>::u8p)r + REC_HDR), (u32)r->nExons);
                u32 uNew = mappedLength - nOverlap, uOld = r->mappedLength - nOverlap;
                if (uNew == 0 && Score < r->maxScore) cls = 1; else if (uOld == 0) cls = 2;
            }
            u64 mB = __ballot(cls == 1), mR = __ballot(cls == 2), mH = __ballot(have);
            if (mB) { u32 fb = firstLane(mB); blocked = true; mR &= (1ull << fb) - 1ull; }
            u64 keep = mH & ~mR;
            if (have && ((keep >> lane) & 1ull)) wr.rank[outN + cntBelow(keep)] = rk;
            outN += (u32)__popcll(keep);
        }
        if (blocked) {                                   // entries behind the blocking chunk keep their order
            for (; base < nW0; base += NLANE) {
                u32 k = base + lane; bool have = k < nW0;
                u16 rk = have ? wr.rank[k] : (u16)0;
                if (have) wr.rank[outN + lane] = rk;
                outN += min(NLANE, nW0 - base);
            }
            wr.nWinTr = outN;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            return;
        }
        wr.nWinTr = outN;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    // ---- ranked insert (:287-303)
    u32 iTr = wr.nWinTr;
    for (u32 base = 0; base < wr.nWinTr; base += NLANE) {
        u32 k = base + lane; bool better = false;
        if (k < wr.nWinTr) { typename AS<BIG>::trp r = recT<BIG>(wr, k); better = Score > r->maxScore || (Score == r->maxScore && gLength < r->gLength); }
        u64 bm = __ballot(better);
        if (bm) { iTr = base + firstLane(bm); break; }
    }
    if (iTr >= P.alignTranscriptsPerWindowNmax) return;          // ranks behind a full list: dropped
    u32 need = REC_HDR + 32u * ne;
    if (wr.top + need > wr.arenaBytes) {
        u32 nt = 0;
        if (lane == 0) nt = compactArena<BIG>(wr);
        wr.top = first32(nt);
        if (BIG) __threadfence_block(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // also give up when the live set alone fills 3/4 of the arena: the next leaves would compact over and over
        if (wr.top + need > wr.arenaBytes || wr.top * 4u > wr.arenaBytes * 3u) { wr.overflow = true; return; }
    }
    u32 off = wr.top; wr.top += need;
    u32 newN = min(wr.nWinTr + 1, P.alignTranscriptsPerWindowNmax);
    for (int top = (int)newN - 1; top > (int)iTr; top -= (int)NLANE) {        // shift ranks iTr..newN-2 up by one
        int k = top - (int)lane;
        u16 v = 0; bool mv = k > (int)iTr;
        if (mv) v = wr.rank[k - 1];
        LOCKSTEP();                               // every lane has read its entry before its neighbour overwrites it
        if (mv) wr.rank[k] = v;
    }
    if (lane == 0) wr.rank[iTr] = (u16)(off / 32u);
    wr.nWinTr = newN;
    if (lane == 0) {                                  // record = output transcript header + exon rows, 8-byte words
        typename AS<BIG>::u64p d = (typename AS<BIG>::u64p)(ArenaSel<BIG>::get(wr) + off);
#pragma unroll
        for (u32 i = 0; i < REC_HDR / 8; i++) d[i] = hdr[i];
    }
    if (lane < ne) {
        typename AS<BIG>::u64p d = (typename AS<BIG>::u64p)(ArenaSel<BIG>::get(wr) + off + REC_HDR + 32u * lane); const u64 *sw = (const u64 *)&x;
        d[0] = sw[0]; d[1] = sw[1]; d[2] = sw[2]; d[3] = sw[3];
    }
    if (BIG) __threadfence_block(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}

template <class EXP, class HP> __device__ static void recordCandidate(const staramd_params &P, u32 lane, const int Score, const u64 gLength, const u32 mappedLength, const u32 ne, HP hdr,
                                                                      const staramd_exon &x, EXP ex, WinRec &wr) {
    if (wr.big) { recordCandidateImpl<true>(P, lane, Score, gLength, mappedLength, ne, hdr, x, ex, wr); wr.bestScore = wr.nWinTr > 0 ? recT<true>(wr, 0)->maxScore : 0; }
    else { recordCandidateImpl<false>(P, lane, Score, gLength, mappedLength, ne, hdr, x, ex, wr); wr.bestScore = wr.nWinTr > 0 ? recT<false>(wr, 0)->maxScore : 0; }
}

// leaf of the recursion: stitchWindowAligns.cpp:16-307.  Works on a scratch copy (ex) of the used exons.
// the score below which a finished leaf of mate class iFragT (-1: both mates) leaves no trace -- see the note at finalizeTranscript
__device__ __forceinline__ i32 leafNeedScore(const StitchCtx &c, const staramd_params &P, const WinRec &wr, const i32 iFragT) {
    const i32 mateBest = iFragT < 0 ? 0x7FFFFFFF : (iFragT == 0 ? c.maxScoreMate[0] : c.maxScoreMate[1]);      // (two-mate: the clause does not apply)
    return firstI(min(wr.bestScore, mateBest) - P.outFilterMultimapScoreRange);                                // (wave-uniform: said so)
}
// (a) of that note: can the leaf of the working transcript h still reach needScore?  Its score is at most the score of the stitched part + one point per base the two
// extensions can reach (they stop at the mate spacer and at the ends of the read) + the largest value of the genomic-length term
__device__ __forceinline__ bool leafCanCount(const StitchCtx &c, const DevIndex &X, const Hdr &h, const WinRec &wr, const u32 fragFirst, const u32 fragLast) {
    const staramd_params &P = X.P;
#ifdef STARAMD_NO_LEAF_BOUND           // A/B builds (tools/build_variants.sh): without test (a)
    return true;
#endif
    if (P.chimSegmentMinPositive) return true;
    const u32 Lread = c.Lread;
    const u32 spacer = c.readLength[0] < Lread ? (c.str == 0 ? c.readLength[0] : Lread - 1u - c.readLength[0]) : Lread;       // position of the mate spacer in R[] (none: Lread)
    const u32 availL = h.rStart > spacer ? h.rStart - spacer - 1u : h.rStart, availR = h.tR2 < spacer ? spacer - 1u - h.tR2 : Lread - 1u - h.tR2;
    i32 U = h.Score + (i32)availL + (i32)availR;
    if (X.glStep != 0) U = max(0, U + (X.glStep < 0 ? X.glScoreAt1 : X.glScoreAt1 + (i32)X.nBreak));
    return U >= leafNeedScore(c, P, wr, fragFirst == fragLast ? (i32)fragFirst : -1);
}

// fragFirst / fragLast: mates of the first and the last exon (the walk keeps them).
//
// Leaves that cannot leave a trace are dropped early (ours; exact in every mode).  A finished leaf acts on the state of the read in two places only: a single-mate
// transcript raises maxScoreMate[its mate] to its score (:232-235), and a transcript is recorded when Score + range >= the window's best or >= maxScoreMate[its mate]
// (:245-247) -- all the filters in between can only drop it.  So a leaf with  Score + range < window best  and (two-mate, or  Score + range < maxScoreMate[mate] --
// which also says Score < maxScoreMate[mate]) changes nothing whatever the filters say.  (a) BEFORE the extensions Score is bounded by the score of the stitched part
// + one point per base the two extensions can reach (they stop at the mate spacer and at the ends of the read) + the largest value of the genomic-length term;
// (b) AFTER them it is known exactly.  (a) is leafCanCount above, asked by the walk before it calls this function (inside it, as an early return, the test tipped
// the register allocation into its vector-register regime: tools/isa_stats.sh); (b) is below, behind the extensions.
__device__ static void finalizeTranscript(StitchCtx &c, u32 lane, Hdr h, const LDS staramd_exon *EX, LDS staramd_exon *ex, u32 chr, WinRec &wr, const u64 glb0, const u64 glb1, LDS u64 *recSlot, const u32 fragFirst, const u32 fragLast) {
    const DevIndex &X = *c.X; const staramd_params &P = X.P;
    u32 Lread = c.Lread; u32 Str = c.str;
    int Score = h.Score; u32 tR2 = h.tR2; u64 tG2 = h.tG2;
    u32 ne = h.nExons;
    const i32 iFragT = fragFirst == fragLast ? (i32)fragFirst : -1;
    const i32 needScore = leafNeedScore(c, P, wr, iFragT);
    if (lane < ne) { staramd_exon t = ldsGet(&EX[lane]); ldsPut(&ex[lane], t); }           // the leaf works on a copy of the exon rows (the walk goes on with the originals)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    ExtRes e;
    PROF_T0();
    int vOrder0 = (Str == 0) ? 0 : 1;                  // EXTEND_ORDER==1, roStr==Str
    for (int iOrd = 0; iOrd < 2; iOrd++) {
        int which = iOrd == 0 ? vOrder0 : 1 - vOrder0;
        if (which == 0) {
            if (h.rStart > 0) {
                const u32 imate = fragFirst;
                if (COOP_EXTEND(c, lane, h.rStart - 1, h.gStart - 1, -1, -1, h.rStart, tR2 - h.rStart + 1, h.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax,
                                P.alignEndsTypeExt[imate][(int)(Str != imate)] != 0, e)) {
                    h.nMatch += e.nMatch; h.nMM += e.nMM; Score += e.maxScore;
                    h.rStart -= e.extendL; h.gStart -= e.extendL;
                    if (lane == 0) { ex[0].R = (u16)h.rStart; ex[0].G = h.gStart; ex[0].L = (u16)(ex[0].L + e.extendL); }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
            }
        } else {
            if (tR2 < Lread) {
                const u32 imate = fragLast;
                if (COOP_EXTEND(c, lane, tR2 + 1, tG2 + 1, +1, +1, Lread - tR2 - 1, tR2 - h.rStart + 1, h.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax,
                                P.alignEndsTypeExt[imate][(int)(imate == Str)] != 0, e)) {
                    h.nMatch += e.nMatch; h.nMM += e.nMM; Score += e.maxScore;
                    tR2 += e.extendL; tG2 += e.extendL;
                    if (lane == 0) ex[ne - 1].L = (u16)(ex[ne - 1].L + e.extendL);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
            }
        }
    }
    PROF_MARK(c, 13);
    u64 gLength = tG2 + 1 - h.gStart;
    if (X.glStep != 0) {             // scoreGenomicLengthLog2scale != 0 (:221-225) as integer break points: lane k holds break points k, k+64.  (last exon end - first exon start = gLength)
        const u32 nAbove = (u32)__popcll(__ballot(gLength >= glb0)) + (u32)__popcll(__ballot(gLength >= glb1));   // break points <= gLength (this lane holds points lane, lane+64)
        Score += X.glScoreAt1 + X.glStep * (i32)nAbove;
        Score = max(0, Score);
    }
#ifndef STARAMD_NO_LEAF_EARLY          // A/B builds: without test (b)
    if (Score < needScore && !P.chimSegmentMinPositive) { DIAG(c.nLeavesEarly++); return; }      // (b) of the note above: the final score is known, the filters cannot make it count
#endif
    // ---- leaf filters (:83-219), lane = exon row: every lane holds one exon of the transcript in registers, neighbours
    // come over the DPP wave shift, sums are DPP reductions, "any exon fails" is a ballot
    staramd_exon xe;
    { u64 *z = (u64 *)&xe; z[0] = z[1] = z[2] = z[3] = 0; }
    const bool isEx = lane < ne;
    if (isEx) xe = ldsGet(&ex[lane]);
    const int canon = isEx ? (int)xe.canonSJ : -9, canonPrev = __builtin_amdgcn_update_dpp(-9, canon, 0x138, 0xf, 0xf, false), canonNext = __builtin_amdgcn_update_dpp(-9, canon, 0x130, 0xf, 0xf, false);
    const int annot = isEx ? (int)xe.sjAnnot : 0, annotPrev = __builtin_amdgcn_update_dpp(0, annot, 0x138, 0xf, 0xf, false), annotNext = __builtin_amdgcn_update_dpp(0, annot, 0x130, 0xf, 0xf, false);
    const u32 exL = isEx ? (u32)xe.L : 0u, exLnext = (u32)__builtin_amdgcn_update_dpp(0, (int)exL, 0x130, 0xf, 0xf, false);
    const u32 last = ne - 1;
    const u64 ex0G = ((u64)laneGet32((u32)(xe.G >> 32), 0) << 32) | laneGet32((u32)xe.G, 0), exLG = ((u64)laneGet32((u32)(xe.G >> 32), last) << 32) | laneGet32((u32)xe.G, last);
    const u32 ex0R = laneGet32(xe.R, 0), exLR = laneGet32(xe.R, last), exLL = laneGet32(exL, last);
    const u32 ex0Frag = fragFirst, exLFrag = fragLast;
    if (!P.alignSoftClipAtReferenceEnds &&
        ((exLG + Lread - exLR) > (GLOBAL(u64, X.chrStart)[chr] + GLOBAL(u64, X.chrLength)[chr]) || ex0G < (GLOBAL(u64, X.chrStart)[chr] + ex0R))) return;
    const u32 rLength = waveSumU32(exL);
    {   // junction overhangs (:97-108)
        bool fail = false;
        if (lane < last && canon >= 0) {
            if (annot == 1) {
                fail = (exL < P.alignSJDBoverhangMin && (lane == 0 || canonPrev == -3 || (annotPrev == 0 && canonPrev >= 0)))
                       || (exLnext < P.alignSJDBoverhangMin && (lane == ne - 2 || canonNext == -3 || (annotNext == 0 && canonNext >= 0)));
            } else {
                fail = exL < P.alignSJoverhangMin + xe.shiftSJ[0] || exLnext < P.alignSJoverhangMin + xe.shiftSJ[1];
            }
        }
        if (__ballot(fail)) return;
    }
    if (ne > 1 && laneGet32((u32)annot, ne - 2) == 1 && exLL < P.alignSJDBoverhangMin) return;
    const bool isJ = lane < last && canon >= 0;
    const u32 sjN = (u32)__popcll(__ballot(isJ));
    u16 intronMotifs[3];
    intronMotifs[0] = (u16)__popcll(__ballot(isJ && xe.sjStr == 0)); intronMotifs[1] = (u16)__popcll(__ballot(isJ && xe.sjStr == 1)); intronMotifs[2] = (u16)__popcll(__ballot(isJ && xe.sjStr > 1));
    u8 sjMotifStrand;
    if (intronMotifs[1] > 0 && intronMotifs[2] == 0) sjMotifStrand = 1;
    else if (intronMotifs[1] == 0 && intronMotifs[2] > 0) sjMotifStrand = 2;
    else sjMotifStrand = 0;
    if (intronMotifs[1] > 0 && intronMotifs[2] > 0 && P.outFilterIntronStrandsRemoveInconsistent) return;
    if (sjN > 0 && sjMotifStrand == 0 && P.outSAMstrandFieldIntronMotif) return;
    if (P.outFilterIntronMotifs == 1) { if (__ballot(lane < last && canon == 0)) return; }
    else if (P.outFilterIntronMotifs == 2) { if (__ballot(lane < last && canon == 0 && annot == 0)) return; }
    {   // minimum mapped length of a spliced mate (:154-167): segments end at the mate gap (canonSJ == -3) or the last exon
        u64 mEnd = __ballot(isEx && (lane == last || canon == -3));
        u32 start = 0;
        while (mEnd) {
            u32 e = firstLane(mEnd); mEnd &= mEnd - 1;
            bool inSeg = lane >= start && lane <= e;
            u32 exl = waveSumU32(inSeg ? exL : 0u);
            u32 nsj = (u32)__popcll(__ballot(inSeg && lane != e && canon >= 0));
            u32 fragE = laneGet32(xe.iFrag, e);
            if (nsj > 0 && (exl < P.alignSplicedMateMapLmin || (u64)exl < (u64)(P.alignSplicedMateMapLminOverLmate * (double)(u64)(fragE ? c.readLength[1] : c.readLength[0])))) return;
            start = e + 1;
        }
    }
    if (P.outFilterBySJoutStage == 2) {                        // unannotated junctions must be on the whitelist (:169-177)
        for (u64 nov = __ballot(lane < last && canon >= 0 && annot == 0); nov; nov &= nov - 1) {
            const u32 l = firstLane(nov);
            const u64 jS = laneGet64(xe.G, l) + laneGet32(exL, l), jE = laneGet64(xe.G, l + 1) - 1;
            if (coopSjdbFind(lane, jS, jE, X.sjNovelStart, X.sjNovelEnd, (u32)X.sjNovelN) < 0) return;
        }
    }
    if (ex0Frag != exLFrag) {                                  // both mates (:179-219): rare inner loop kept scalar on the LDS rows
        if (exLG + exLL <= ex0G) return;
        u32 iexM2 = ne;
        { u64 mg = __ballot(lane < last && canon == -3); if (mg) iexM2 = firstLane(mg) + 1; }
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
    // (constant indices only: a dynamic index would pin the whole context in scratch memory)
    if (iFragT == 0) c.maxScoreMate[0] = max(c.maxScoreMate[0], Score); else if (iFragT == 1) c.maxScoreMate[1] = max(c.maxScoreMate[1], Score);
    PROF_MARK(c, 14);
    i32 winBest = wr.bestScore;            // wTr[0]->maxScore (trA with score 0 before any record)
    {
        bool c1 = Score + P.outFilterMultimapScoreRange >= winBest || P.chimSegmentMinPositive;
        bool c2 = iFragT >= 0 && Score + P.outFilterMultimapScoreRange >= (iFragT == 0 ? c.maxScoreMate[0] : c.maxScoreMate[1]);
        if (!(c1 || c2)) return;
        // decided by the maxScoreMate clause alone: the decision holds for any incoming maxScoreMate <= Score + range
        if (!c1) { if (iFragT == 0) c.sens[0] = min(c.sens[0], Score + P.outFilterMultimapScoreRange); else c.sens[1] = min(c.sens[1], Score + P.outFilterMultimapScoreRange); }
    }
    // ---- the candidate as an output record: header (wave-uniform) + this lane's exon row.  The header is assembled by lane 0 in a 96-byte
    // staging slot of the wavefront's LDS slice (as a local struct it sat in scratch memory: 96 B x 64 lanes = 6 KB of HBM writes per candidate)
    if (lane == 0) {
        LDS staramd_transcript *o = (LDS staramd_transcript *)recSlot;
        o->iW = 0; o->exonOffset = 0;
        o->nExons = (u16)ne; o->rStart = (u16)h.rStart; o->rLength = (u16)rLength;
        o->roStart = (u16)((Str == 0) ? h.rStart : Lread - h.rStart - rLength);
        o->Str = (u8)Str; o->roStr = (u8)Str; o->iFrag = (i8)iFragT; o->sjMotifStrand = sjMotifStrand; o->Chr = chr;
        o->gStart = h.gStart; o->gLength = gLength; o->maxScore = Score; o->nMatch = h.nMatch; o->nMM = h.nMM; o->mappedLength = rLength;
        o->nGap = h.nGap; o->lGap = h.lGap; o->nDel = h.nDel; o->lDel = h.lDel; o->nIns = h.nIns; o->lIns = h.lIns;
        o->nUnique = (u16)h.nUnique; o->nAnchor = (u16)h.nAnchor;
        o->intronMotifs[0] = intronMotifs[0]; o->intronMotifs[1] = intronMotifs[1]; o->intronMotifs[2] = intronMotifs[2]; o->pad0 = 0; o->pad1 = 0;   // padding included: records are compared byte for byte
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    staramd_exon x = xe;
    if (lane < ne) {
        if (lane + 1 == ne) { x.canonSJ = 0; x.sjAnnot = 0; x.sjStr = 0; x.shiftSJ[0] = x.shiftSJ[1] = 0; }
        else if (x.canonSJ < 0) { x.shiftSJ[0] = x.shiftSJ[1] = 0; }
        x.pad0 = 0; x.pad1 = 0;
    }
    // ---- candidate log (pass 0): lets pass 1 re-decide this window without walking it again
    if (c.logOn && !c.logOvf) {
        u32 need = REC_HDR + 32u * ne;
        if (c.candTop + need > c.candCap) c.logOvf = true;
        else {
            u64 *d = (u64 *)(c.candBase + c.candTop);
            if (lane == 0) {
#pragma unroll
                for (u32 i = 0; i < REC_HDR / 8; i++) d[i] = recSlot[i]; }
            if (lane < ne) { const u64 *sw = (const u64 *)&x; u64 *de = d + REC_HDR / 8 + 4u * lane; de[0] = sw[0]; de[1] = sw[1]; de[2] = sw[2]; de[3] = sw[3]; }
            c.candTop += need; c.nCand++;
        }
    }
    PROF_MARK(c, 15);
    { PROF_T0(); recordCandidate(P, lane, Score, gLength, rLength, ne, recSlot, x, ex, wr); PROF_ADD(c, 4); }
}

// per-window LDS work space, in bytes: undo stack, exon rows, leaf copy, rank list, seed list (+ arena in the fast path)
#define REC_HDR_BYTES 96u
// seed list rows (24 B) and compat masks (8 B): as many as the launch walks at most (capDepth - 1 seeds per window), never more than WA_MAX
__host__ __device__ inline u32 waRows(u32 capDepth) { return capDepth == 0 ? (u32)WA_MAX : (capDepth - 1u < (u32)WA_MAX ? capDepth - 1u : (u32)WA_MAX); }
__host__ __device__ inline u32 stitchStateBytes(u32 capDepth, u32 capRank, u32 arenaBytes) {
    u32 b = capDepth * (u32)sizeof(SFrame) + 2u * STARAMD_MAX_N_EXONS * 32u + ((capRank * 2u + 31u) & ~31u) + waRows(capDepth) * 32u + REC_HDR_BYTES + arenaBytes;
    return (b + 127u) & ~127u;
}

struct LaneMem { LDS SFrame *stack; LDS staramd_exon *EX, *LEAF; LDS DWA *WA; LDS u64 *compat; LDS u64 *rec; LDS u16 *rank; LDS u8 *arena; };

// next seed index > i whose bit is set in mask, nA if none
__device__ __forceinline__ u32 nextSeed(u64 mask, u32 i, u32 nA) {
    u64 mk = i >= 63u ? 0ull : (mask & ~((2ull << i) - 1ull));
    if (nA < 64u) mk &= (1ull << nA) - 1ull;
    return mk ? firstLane(mk) : nA;
}

// depth-first walk of one window (stitchWindowAligns.cpp:8-353 called from ReadAlign_stitchPieces.cpp:321):
// include seed iA (if it stitches), then exclude it.  Wave-uniform control flow.  Returns false when the arena overflowed.
// skipSingle: the leaves whose transcript holds seeds of ONE mate only are not finalised (nSkipped counts them and the subtrees that can
// only end in such leaves); the caller accepts the result when the window's final best score puts every single-mate transcript below the selection bar.
__device__ static bool stitchWindow(StitchCtx &c, u32 lane, const DWin &win, const LaneMem &m, WinRec &wr, const u64 glb0, const u64 glb1, const bool skipSingle, u32 &nSkipped) {
    const u32 nA = win.nWA;
    PROF_T0();
    c.str = win.str;
    wr.nWinTr = 0; wr.top = 0; wr.overflow = false; wr.bestScore = 0;
    LDS SFrame *stack = m.stack; LDS staramd_exon *EX = m.EX, *LEAF = m.LEAF; LDS DWA *WA = m.WA;
    Hdr h; h.gStart = 0; h.tG2 = 0; h.nExons = 0; h.Score = 0; h.nMatch = h.nMM = h.nGap = h.lGap = h.nDel = h.lDel = h.nIns = h.lIns = 0;
    h.nUnique = h.nAnchor = 0; h.rStart = 0; h.tR2 = 0;
    u32 iA = 0; u32 sp = 0; u32 ex0R = 0; u64 ex0G = 0;
    // Seed B can never follow seed A of the same mate when it does not end behind A in the read and in the genome
    // (stitchAlignToTranscript.cpp:44-51 returns -1000001 / -1000002 before touching anything): ~3/4 of the reference's
    // stitch calls.  The pairs are known up front -- lane A builds the bit mask of the seeds that CAN follow A -- and the
    // walk steps over the others without visiting them (a failed include followed by the exclude changes nothing).
    if (lane < nA) {
        const DWA sa = ldsGet(&WA[lane]);
        const u32 rAe = (u32)sa.rStart + sa.L - 1; const u64 gAe = sa.gStart + sa.L - 1;
        u64 mk = 0;
        for (u32 b = lane + 1; b < nA; b++) {
            const u32 rBe = (u32)WA[b].rStart + WA[b].L - 1; const u64 gBe = WA[b].gStart + WA[b].L - 1;
            const bool fail = WA[b].iFrag == sa.iFrag && (rBe <= rAe || gBe <= gAe);
            if (!fail) mk |= 1ull << b;
        }
        m.compat[lane] = mk;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    PROF_ADD(c, 5);                                  // (window set-up: compat masks)
    u32 iLast = 0;                                   // last included seed of the working transcript
    u32 fragFirst = 0, fragLast = 0;                 // mates of its first and of its last seed = of its first and last exon (stitchAlignToTranscript.cpp:413-414)
    nSkipped = 0;
    // seeds of mate 2 (the list is sorted by read position: they follow the seeds of mate 1)
    const u64 mate2 = skipSingle ? __ballot(lane < nA && ldsGet(&WA[lane < nA ? lane : 0u]).iFrag != 0) : 0ull;
    const u64 allSeeds = nA >= 64u ? ~0ull : ((1ull << nA) - 1ull);
    u64 follow = ~0ull;                              // seeds that can follow the last included one (all, while the transcript is empty)
    DWA a = uni(ldsGet(&WA[0]));
    for (;;) {
        DIAG(c.nNodes++);
        // every leaf below this node is a single-mate transcript: the transcript holds one mate so far and no seed of the other mate is left
        bool onlySingle = false;
        if (skipSingle && h.nExons > 0 && fragFirst == fragLast && iA < nA) {
            const u64 other = (fragFirst ? (allSeeds & ~mate2) : mate2) >> iA;
            onlySingle = other == 0;
        }
        if (iA >= nA || onlySingle) {                // leaf (stitchWindowAligns.cpp:14-16: nothing to do when tR2==0)
            if (h.tR2 != 0 && skipSingle && fragFirst == fragLast) nSkipped++;
            else if (h.tR2 != 0) {
                DIAG(c.nLeaves++);
                if (!leafCanCount(c, *c.X, h, wr, fragFirst, fragLast)) { DIAG(c.nLeavesBound++); }
                else { PROF_T0(); finalizeTranscript(c, lane, h, EX, LEAF, win.chr, wr, glb0, glb1, m.rec, fragFirst, fragLast); PROF_ADD(c, 3); }
                if (wr.overflow) return false;
            }
            if (sp == 0) break;
            sp--;                                    // back to the frame that included a seed: now exclude it
            SFrame f = uni(ldsGet(&stack[sp]));
            h = f.h; iLast = f.pad & 255u; fragLast = f.pad >> 8; follow = h.nExons > 0 ? first64(m.compat[iLast]) : ~0ull;   // f.pad = last included seed of the restored transcript | its mate << 8
            iA = nextSeed(follow, f.iA, nA);
            if (h.nExons > 0 && lane == 0) ldsPut(&EX[h.nExons - 1], f.eA);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (iA < nA) a = uni(ldsGet(&WA[iA]));
            continue;
        }
        // ---- include branch (:311-345)
        Hdr hn = h; staramd_exon eA, eN; bool added = false; int dScore;
        if (h.nExons > 0) {
            eA = uni(ldsGet(&EX[h.nExons - 1]));
            staramd_exon eAold = eA;
#ifdef STARAMD_SHADOW
            Hdr hs = h; staramd_exon eAs = eA, eNs; bool addedS = false;
#endif
            { PROF_T0(); SPROF_KIND(4); dScore = coopStitch(c, lane, h.tR2, h.tG2, a.rStart, a.gStart, a.L, a.iFrag, a.sjA, hn, eA, eN, added, ex0R, ex0G); PROF_ADD(c, 1); SPROF_ADD(c); }
#ifdef STARAMD_SHADOW
            {   // every lane re-runs the call through the scalar restatement (same inputs): any disagreement is counted
                int dS = joinOnLane(c, h.tR2, h.tG2, a.rStart, a.gStart, a.L, a.iFrag, a.sjA, hs, eAs, eNs, addedS, ex0R, ex0G);
                bool bad = dS != dScore;
                if (!bad && dS > -1000000) {
                    bad = hs.nMatch != hn.nMatch || hs.nMM != hn.nMM || hs.nGap != hn.nGap || hs.lGap != hn.lGap || hs.nDel != hn.nDel || hs.lDel != hn.lDel || hs.nIns != hn.nIns || hs.lIns != hn.lIns
                          || addedS != added || eAs.L != eA.L || eAs.canonSJ != eA.canonSJ || eAs.sjAnnot != eA.sjAnnot || eAs.sjStr != eA.sjStr || eAs.iFrag != eA.iFrag || eAs.sjA != eA.sjA;
                    if (!bad && eA.canonSJ >= 0 && added) bad = eAs.shiftSJ[0] != eA.shiftSJ[0] || eAs.shiftSJ[1] != eA.shiftSJ[1];
                    if (!bad && added) bad = eNs.G != eN.G || eNs.R != eN.R || eNs.L != eN.L || eNs.iFrag != eN.iFrag || eNs.sjA != eN.sjA;
                }
                if (lane == 0) { if (bad) atomicAdd((unsigned long long *)&c.shadow[0], 1ull); atomicAdd((unsigned long long *)&c.shadow[1], 1ull); }
            }
#endif
            if (dScore > -1000000) {
                if (lane == 0) {
                    SFrame f; f.h = h; f.iA = iA; f.pad = iLast | (fragLast << 8); f.eA = eAold;
                    ldsPut(&stack[sp], f);
                    ldsPut(&EX[h.nExons - 1], eA);
                    if (added) ldsPut(&EX[h.nExons], eN);
                }
                if (added) hn.nExons = h.nExons + 1;
            }
        } else {                                     // first seed of the transcript (:318-334)
            eN.R = a.rStart; eN.G = a.gStart; eN.L = a.L; eN.iFrag = a.iFrag; eN.sjA = a.sjA;
            eN.canonSJ = 0; eN.sjAnnot = 0; eN.sjStr = 0; eN.shiftSJ[0] = eN.shiftSJ[1] = 0; eN.pad0 = 0; eN.pad1 = 0;
            if (lane == 0) { SFrame f; f.h = h; f.iA = iA; f.pad = 0; f.eA = eN; ldsPut(&stack[sp], f); ldsPut(&EX[0], eN); }     // empty transcript: pad unused
            hn.rStart = a.rStart; hn.gStart = a.gStart; hn.nExons = 1; hn.nMatch = a.L;
            dScore = a.L;
        }
        if (dScore > -1000000) {
            if (a.nrep == 1) hn.nUnique++;
            if (a.anchor > 0) hn.nAnchor++;
            hn.Score = h.Score + dScore; hn.tR2 = (u32)a.rStart + a.L - 1; hn.tG2 = a.gStart + a.L - 1;
            if (h.nExons == 0) { ex0R = a.rStart; ex0G = a.gStart; fragFirst = a.iFrag; }
            h = hn; sp++;
            iLast = iA; fragLast = a.iFrag; follow = first64(m.compat[iA]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        // include succeeded: continue below it; include failed: exclude branch (:348-351) = same transcript, next seed
        iA = nextSeed(follow, iA, nA);
        if (iA < nA) a = uni(ldsGet(&WA[iA]));
    }
    return true;
}

// copy the window's recorded transcripts (trAll[iW1][0..nWinTr-1]) into the result pools; false on pool overflow
// headChim: could the window's head be the main segment of a chimeric alignment -- the test chimericDetectionOld makes of trBest before it looks at any other window
// (ReadAlign_chimericDetectionOld.cpp:19-26): segment long enough, unmapped space of chimSegmentMin bases at one end of the read, no non-canonical junction, one strand
template <bool BIG> __device__ static bool flushWindowImpl(const DevBatch &B, u32 lane, const WinRec &wr, DWinOut &o, u32 chimSegmentMin, u32 Lread, bool &headChim) {
    u32 nTr = wr.nWinTr, nEx = 0;
    o.trOffset = 0; o.nTr = 0; o.exOffset = 0; o.nEx = 0; o.headScore = 0; o.headGlen = 0;
    headChim = false;
    if (nTr == 0) return true;
    for (u32 k = 0; k < nTr; k++) nEx += recT<BIG>(wr, k)->nExons;
    u32 to = 0, eo = 0;
    if (lane == 0) { to = atomicAdd(&B.cursors[CUR_TR], nTr); eo = atomicAdd(&B.cursors[CUR_EX], nEx); }
    to = first32(to); eo = first32(eo);
    if (to + nTr > B.trCap || eo + nEx > B.exCap) { if (lane == 0) atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_TRPOOL); return false; }
    typename AS<BIG>::trp hd = recT<BIG>(wr, 0);
    o.trOffset = to; o.nTr = nTr; o.exOffset = eo; o.nEx = nEx; o.headScore = hd->maxScore; o.headGlen = hd->gLength;
    if (chimSegmentMin) {
        typename AS<BIG>::exp hx = (typename AS<BIG>::exp)((typename AS<BIG>::u8p)hd + REC_HDR); const u32 hn = hd->nExons;
        headChim = hd->rLength >= chimSegmentMin && ((u32)hx[hn - 1].R + hx[hn - 1].L + chimSegmentMin <= Lread || hx[0].R >= chimSegmentMin)
                   && hd->intronMotifs[0] == 0 && (hd->intronMotifs[1] == 0 || hd->intronMotifs[2] == 0);
    }
    u32 eoff = 0;
    for (u32 k = 0; k < nTr; k++) {
        typename AS<BIG>::u64p s = (typename AS<BIG>::u64p)recT<BIG>(wr, k);
        u32 ne = recT<BIG>(wr, k)->nExons;
        u64 *dt = (u64 *)&B.trPool[to + k];
        if (lane < REC_HDR / 8) {
            u64 v = s[lane];
            if (lane == 0) v = (v & 0xFFFFFFFFull) | ((u64)eoff << 32);       // exonOffset relative to the window block; k_gather rebases it and sets iW
            dt[lane] = v;
        }
        u64 *de = (u64 *)&B.exPool[eo + eoff];
        for (u32 i = lane; i < ne * 4; i += NLANE) de[i] = s[REC_HDR / 8 + i];
        eoff += ne;
    }
    return true;
}

__device__ static bool flushWindow(const DevBatch &B, u32 lane, const WinRec &wr, DWinOut &o, u32 chimSegmentMin, u32 Lread, bool &headChim) {
    return wr.big ? flushWindowImpl<true>(B, lane, wr, o, chimSegmentMin, Lread, headChim) : flushWindowImpl<false>(B, lane, wr, o, chimSegmentMin, Lread, headChim);
}

__device__ __forceinline__ void laneSetup(LDS u8 *mine, u32 capDepth, u32 capRank, LaneMem &m) {
    m.stack = (LDS SFrame *)mine;
    m.EX = (LDS staramd_exon *)(mine + capDepth * (u32)sizeof(SFrame));
    m.LEAF = m.EX + STARAMD_MAX_N_EXONS;
    m.rank = (LDS u16 *)(m.LEAF + STARAMD_MAX_N_EXONS);
    m.WA = (LDS DWA *)((LDS u8 *)m.rank + ((capRank * 2u + 31u) & ~31u));
    m.compat = (LDS u64 *)((LDS u8 *)m.WA + waRows(capDepth) * 24u);
    m.rec = (LDS u64 *)((LDS u8 *)m.compat + waRows(capDepth) * 8u);      // staging slot for one output record header
    m.arena = (LDS u8 *)m.rec + REC_HDR_BYTES;
}

__device__ __forceinline__ void ctxLoadRead(StitchCtx &c, u32 lane, const DevBatch &B, const staramd_params &P, u32 ir) {
    c.Lread = first32((u32)(B.readOffset[ir + 1] - B.readOffset[ir]));
    c.readLength[0] = first32(B.mate1Length[ir]); c.readLength[1] = (P.readNmates == 2 && c.readLength[0] < c.Lread) ? c.Lread - c.readLength[0] - 1 : 0;      // (a read of merged mates in a paired-end run: one piece, no second mate)
    c.mmMaxTotal = first32(B.mmMaxTotal[ir]);
    const u32 *src = B.packed + (u64)ir * B.packWords;       // stage the 4-bit packed read in this wavefront's LDS slice
    u32 nw = (c.Lread + 7) / 8;
    LDS u32 *dst = (LDS u32 *)((LDS u8 *)ldsReads + c.ldsByte);
    for (u32 k = lane; k < nw; k += NLANE) dst[k] = src[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}

// ---- stitch, one wavefront per WINDOW -------------------------------------------------------------------------------
// Windows of a read are independent except for maxScoreMate[] (stitchWindowAligns.cpp:232-247), which a window only
// consults to decide whether a single-mate transcript that is NOT within range of the window's best is still recorded.
// Pass 0 stitches every window with an incoming maxScoreMate of 0 and remembers, per mate, the weakest leaf whose
// recording hung on that clause (DWinOut::sens).  k_stitch_verify then forms the true incoming values (a prefix maximum
// of DWinOut::mm over the windows of the read, which does not depend on what was recorded) and queues for pass 1 only
// the windows whose decisions could differ.  mode: 0 = pass 0, 1 = pass 1.
// A window whose recorded transcripts outgrow the LDS arena is walked again at once by the same wavefront with its
// arena in global memory (bigArena: one worst-case arena per wavefront).
#ifndef STITCH_WAVES
#define STITCH_WAVES 4      // minimum waves per SIMD the register allocation of the walk kernel is held to (Makefile: STITCH_WAVES)
#endif
extern "C" __global__ void __launch_bounds__(256, STITCH_WAVES) k_stitch_win(const DevIndex *__restrict__ Xp, DevBatch B, u8 *bigArena, u32 capDepth, u32 capRank, u32 arenaBytes,
                                                             u32 bigArenaBytes, u32 ldsWords, u32 mode, u32 pruneEnable) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch
    const DevIndex &X = *Xp;
    const staramd_params &P = X.P;
    const u32 lane = threadIdx.x & 63u;
    u32 waveInBlock = WAVE_INDEX(threadIdx.x >> 6), wavesPerBlock = blockDim.x >> 6;
    u32 stateBytes = stitchStateBytes(capDepth, capRank, arenaBytes);
    u32 readBytes = (ldsWords * 4u + 15u) & ~15u;
    LaneMem m;
    StitchCtx c; c.X = &X; c.nGstitch = 0; c.nStitchCalls = c.nExtendCalls = c.nNodes = c.nLeaves = c.nLeavesBound = c.nLeavesEarly = 0;
#ifdef STARAMD_SHADOW
    c.shadow = B.counters + DC_shadowBad;
#endif
    c.ldsByte = waveInBlock * (readBytes + stateBytes);
    laneSetup((LDS u8 *)ldsReads + c.ldsByte + readBytes, capDepth, capRank, m);
    const u32 waveId = blockIdx.x * wavesPerBlock + waveInBlock;
    WinRec wr; wr.rank = m.rank; wr.arenaL = m.arena; wr.arenaBytesL = arenaBytes; wr.arenaG = bigArena + (u64)waveId * bigArenaBytes; wr.arenaBytesG = bigArenaBytes;
    gcInit(c.ca); gcInit(c.cb);
    // mode 0: pass 0 over every item; mode 2: pass 0 over the items the lean-LDS launch (mode 0 with a small capDepth) handed on;
    // mode 1: pass 1, full re-walk of the windows on the redo list
    // mode 3: pass 0 with a lean LDS slice over the items the LANE kernel handed on (heavyList); what outgrows the slice goes to heavyList2, which
    // mode 4 = mode 2 over that list takes
    const bool pass0 = mode != 1;
    c.candBase = B.candPool + (u64)waveId * B.candWaveBytes; c.candTop = 0; c.candCap = (u32)B.candWaveBytes; c.nCand = 0; c.logOn = pass0; c.logOvf = false;
    if (mode == 2 || mode == 3 || mode == 4) {               // the log region of a wavefront is shared by the pass-0 launches: go on behind what the launch before wrote
        c.candTop = (u32)B.candTops[waveId];
    }
    // break points of the genomic-length score term: lane k keeps points k and k+64 in registers for the whole kernel
    const u64 glb0 = lane < X.nBreak ? X.glBreak[lane] : ~0ull, glb1 = lane + 64u < X.nBreak ? X.glBreak[lane + 64u] : ~0ull;
    const u32 *list; u32 nItems, ticketSlot;
    if (mode == 0) { list = B.order; nItems = ((B.cursors[CUR_ITEM] + 63u) / 64u) * 64u; ticketSlot = CUR_ST_TICKET0; }
    else if (mode == 2 || mode == 3) { list = B.heavyList; nItems = B.cursors[CUR_ST_HEAVY]; ticketSlot = CUR_ST_TICKETH; }
    else if (mode == 4) { list = B.heavyList2; nItems = B.cursors[CUR_ST_HEAVY2]; ticketSlot = CUR_ST_TICKETH2; }
    else { list = B.redoList; nItems = B.cursors[CUR_ST_REDO]; ticketSlot = CUR_ST_TICKET1; }
    u32 nOvf = 0, lastRead = 0xFFFFFFFFu, nPruned = 0, nRewalk = 0, nRewalkWin = 0, nSkippedLeaves = 0;
    const bool sweepEnable = (pruneEnable & 2u) != 0, skipEnable = (pruneEnable & 4u) != 0;
    // ---- window pruning (ours; exact for what is returned under resultSelect == 1): stitch_common.h pruneWindow / pruneSingleBar have the rule and the argument.
    // Once some window of the read has RECORDED a score `best`, a window whose score bound + range < best is not walked -- for a paired-end read provided that no
    // single-mate transcript of the read can be selected either (singleBar < best); windows of single-end reads do not depend on each other at all.  Off when every
    // transcript is wanted (resultSelect == 0: chimeric detection, merged mates), with a positive genomic-length term or positive indel scores, and for reads that
    // could reach alignTranscriptsPerReadNmax.
    const i32 perJ = max(0, P.sjdbScore) + max(0, max(max(P.scoreGap, P.scoreGapNoncan), max(P.scoreGapGCAG, P.scoreGapATAC)));
    // Chimeric detection with the partner chosen on the device (resultSelect 2) wants every window of a read -- IF the read's best alignment can be the main
    // segment of a chimera at all: chimericDetectionOld tests trBest alone before it looks at another window (flushWindowImpl headChim), and most reads fail that test (their
    // best alignment covers them end to end).  So such a run prunes like any other at first -- with every transcript recorded (stitchWindowAligns.cpp:247) the windows of a
    // read do not depend on each other at all, what is returned for multMapSelect is exact -- and a light read whose best head passes the test is walked again, every
    // window, nothing pruned (chimFull below).  Reads whose windows are separate work items are not pruned in such a run; single-mate leaves are never skipped in it (5.6).
    const bool chimMode = P.resultSelect == 2u && P.chimSegmentMinPositive && P.chimSegmentMin > 0;
    const bool pruneOn = pass0 && ((P.resultSelect == 1 && !P.chimSegmentMinPositive) || (P.resultSelect == 2u && (chimMode || !P.chimSegmentMinPositive) && (pruneEnable & 8u) != 0))
                         && X.glStep <= 0 && (pruneEnable & 3u) != 0
                         && P.scoreDelOpen <= 0 && P.scoreDelBase <= 0 && P.scoreInsOpen <= 0 && P.scoreInsBase <= 0;
#ifdef STARAMD_PROFILE
    for (int k = 0; k < 16; k++) c.prof[k] = 0;
    for (int k = 0; k < 12; k++) c.sprof[k] = 0;
    c.sprofKind = 0;
    const u64 profKernelStart = __builtin_readcyclecounter();
#endif
    for (;;) {
        u32 it = 0;
        if (lane == 0) it = atomicAdd(&B.cursors[ticketSlot], 1u);
        it = first32(it);
        if (it >= nItems) break;
        const u32 item = first32(list[it]);
        if (item == 0xFFFFFFFFu) continue;              // padding slot of the dealt order
        // a work item is a window, or (bit 31) a light read whose windows are walked in order with maxScoreMate carried
        const bool wholeRead = (item & 0x80000000u) != 0;
        u32 w0 = item, nWin = 1;
        i32 carry[2] = {0, 0};
        u32 maxSeeds; i32 bestSoFar = 0; u32 nWinRead = 0, maxSeedsRead = 0;
        if (wholeRead) { const DRead rd = uni(B.reads[item & 0x7FFFFFFFu]); w0 = rd.winOffset; nWin = rd.nWin; maxSeeds = rd.wtOffset; nWinRead = rd.nWin; maxSeedsRead = rd.wtOffset; }
        else maxSeeds = first32(B.winPool[item].nWA);
        if ((mode == 0 || mode == 3) && maxSeeds + 1u > capDepth && maxSeeds <= WA_MAX) {       // more seeds than this (lean) launch has LDS rows for: the full-size launch takes the item
            if (lane == 0) { u32 k = atomicAdd(&B.cursors[mode == 0 ? CUR_ST_HEAVY : CUR_ST_HEAVY2], 1u); (mode == 0 ? B.heavyList : B.heavyList2)[k] = item; }
            continue;
        }
        // A light read is walked in up to two sweeps.  Sweep 0 (only when some of its windows hold seeds of both mates and some do not): the
        // two-mate windows alone, in window order.  If the best score they record clears the pruning bar of EVERY other window of the read,
        // those are all skipped unwalked (sweep 1) -- the reasoning of the pruning note above holds for any order of the walked windows, because
        // all that order changes are maxScoreMate-dependent decisions about single-mate transcripts, none of which can be selected then.  If it
        // does not, nothing is kept: the read is walked again from scratch, every window in the reference's order (sweep 2).
        bool chimFull = false;                          // chimMode: the read is walked a second time, in full (its best head can be the main segment of a chimera)
        u32 sweep = 2;
        if (wholeRead && pruneOn && sweepEnable && nWin > 1 && (u64)(nWinRead + 1u) * P.alignTranscriptsPerWindowNmax < P.alignTranscriptsPerReadNmax) {
            bool anyPair = false, anySingle = false;
            for (u32 base = 0; base < nWin; base += NLANE) {
                const u32 k = base + lane; u32 mt = 0;
                if (k < nWin) mt = B.winPool[w0 + k].mates;
                anyPair |= __ballot(k < nWin && mt == 3u) != 0; anySingle |= __ballot(k < nWin && mt != 3u) != 0;
            }
            if (anyPair && anySingle) sweep = 0;
        }
        i32 runScore = 0; u64 runGlen = 0; bool runChim = false; u32 itemPruned = 0;      // chimMode: the read's best head so far by k_stitch_finish's rule, its test, windows not walked
        for (;;) {
        const bool pruneItem = pruneOn && !chimFull && (!chimMode || wholeRead);
        if (sweep == 2) { carry[0] = carry[1] = 0; bestSoFar = 0; runScore = 0; runGlen = 0; runChim = false; itemPruned = 0; }
        // The windows of the item 64 at a time, lane k = window chunk + k: one load brings the rows of all of them (a read has ~20 windows of which ~3 are
        // walked: going through them one dependent load at a time cost more round trips than the walks), the windows that are skipped unwalked get their
        // empty result from their own lane in one store, and only the windows a sweep has to look at are visited.
        for (u32 chunk = 0; chunk < nWin; chunk += NLANE) {
        DWin myWin; { u32 *z = (u32 *)&myWin; z[0] = z[1] = z[2] = z[3] = 0; }
        const bool haveW = chunk + lane < nWin;
        if (haveW) myWin = B.winPool[w0 + chunk + lane];
        const u64 validM = __ballot(haveW), pairM = __ballot(haveW && myWin.mates == 3u);
        u64 emptyM = sweep == 1 ? (validM & ~pairM) : 0ull;            // sweep 1: everything that was not walked in sweep 0
        for (u64 visit = sweep == 0 ? pairM : sweep == 1 ? 0ull : validM; visit; visit &= visit - 1) {
            const u32 il = firstLane(visit);
            const u32 w = w0 + chunk + il;
            DWin win; { u32 *d = (u32 *)&win; const u32 *sw = (const u32 *)&myWin; d[0] = laneGet32(sw[0], il); d[1] = laneGet32(sw[1], il); d[2] = laneGet32(sw[2], il); d[3] = laneGet32(sw[3], il); }
            if (win.read != lastRead) { ctxLoadRead(c, lane, B, P, win.read); lastRead = win.read; }
            if (win.nWA + 1u > capDepth || win.nWA > WA_MAX) { if (lane == 0) atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_HARD); continue; }
            if (!wholeRead && pruneItem) { nWinRead = first32(B.reads[win.read].nWin); maxSeedsRead = first32(B.reads[win.read].wtOffset); }
            const i32 singleBar = pruneSingleBar(P, perJ, c.readLength[0], c.readLength[1], maxSeedsRead);
            if (pruneItem && win.mates != 0 && sweep == 2) {
                if (!wholeRead) bestSoFar = firstI(__hip_atomic_load(&B.reads[win.read].pruneBest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if ((u64)(nWinRead + 1u) * P.alignTranscriptsPerWindowNmax < P.alignTranscriptsPerReadNmax
                    && pruneWindow(P, perJ, win.mates, win.nWA, c.readLength[0], c.readLength[1], singleBar, bestSoFar)) { emptyM |= 1ull << il; continue; }
            }
            {   // stage the window's seed list in LDS (6 dwords per row)
                const u32 *src = (const u32 *)(B.waPool + win.waOffset); LDS u32 *dst = (LDS u32 *)m.WA;
                for (u32 k = lane; k < win.nWA * 6u; k += NLANE) dst[k] = src[k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
            DWinOut o;
            if (wholeRead) { o.minIn[0] = carry[0]; o.minIn[1] = carry[1]; }
            else { o = uni(B.wout[w]); if (pass0) { o.minIn[0] = o.minIn[1] = 0; } }
            c.logOn = pass0 && !wholeRead;
            const u32 candStart = c.candTop;
            // Single-mate leaves of a two-mate window: first walked WITHOUT them.  Two-mate records do not depend on single-mate ones (a
            // single-mate transcript neither covers nor is covered-with-a-higher-score by... it cannot remove or block a two-mate one: stitchWindowAligns.cpp:267-285
            // act on containment), so the window's two-mate records and its best score H come out as in the full walk.  If every single-mate transcript
            // of the read is then below the selection bar (singleBar < H, stitch_common.h), none of them can be returned or change what
            // is, and the walk stands; else the window is walked again in full.  Same conditions as the window pruning (resultSelect == 1 ...).
            bool skipSingle = pruneItem && !chimMode && skipEnable && win.mates == 3u && (u64)(nWinRead + 1u) * P.alignTranscriptsPerWindowNmax < P.alignTranscriptsPerReadNmax;
            bool ok = false;
            for (;;) {
                u32 nSkipped = 0;
                ok = false;
                for (u32 attempt = 0; attempt < 2 && !ok; attempt++) {         // 2nd attempt: same walk, record arena in HBM
                    wr.big = attempt != 0; wr.arenaBytes = wr.big ? wr.arenaBytesG : wr.arenaBytesL;
                    c.maxScoreMate[0] = o.minIn[0]; c.maxScoreMate[1] = o.minIn[1];
                    c.sens[0] = c.sens[1] = 0x7FFFFFFF;
                    c.candTop = candStart; c.nCand = 0; c.logOvf = false;
                    { PROF_T0(); ok = stitchWindow(c, lane, win, m, wr, glb0, glb1, skipSingle, nSkipped); PROF_ADD(c, 0); }
                    if (!ok) nOvf++;
                }
                if (ok && skipSingle && nSkipped) {
                    const i32 bar = singleBar;
                    // (a full list may have pushed records out that the junk of the full walk would have pushed out differently: walked again as well)
                    // (the bar is measured against the best score RECORDED so far in the read -- this window's final head or an earlier window's: trBest can only be higher)
                    i32 readBest = wr.bestScore;
                    if (wholeRead) readBest = max(readBest, bestSoFar);
                    else readBest = max(readBest, firstI(__hip_atomic_load(&B.reads[win.read].pruneBest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
                    if (!(bar < readBest) || wr.nWinTr >= P.alignTranscriptsPerWindowNmax) { skipSingle = false; nRewalkWin++; continue; }
                    nSkippedLeaves += nSkipped;
                }
                break;
            }
            if (!ok) { if (lane == 0) atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_HARD); continue; }
            bool headChim = false;
            if (!flushWindow(B, lane, wr, o, chimMode ? P.chimSegmentMin : 0u, c.Lread, headChim)) continue;
            if (o.nTr && (o.headScore > runScore || (o.headScore == runScore && o.headGlen < runGlen))) { runScore = o.headScore; runGlen = o.headGlen; runChim = headChim; }
            o.mm[0] = c.maxScoreMate[0]; o.mm[1] = c.maxScoreMate[1];
            carry[0] = c.maxScoreMate[0]; carry[1] = c.maxScoreMate[1];
            if (pruneItem && o.headScore > 0) {              // the window recorded a transcript of this score: later windows are measured against it
                if (wholeRead) bestSoFar = max(bestSoFar, o.headScore);
                else if (lane == 0) atomicMax(&B.reads[win.read].pruneBest, o.headScore);
            }
            // the incoming maxScoreMate of a whole-read item is exact: its result is final
            o.sens[0] = wholeRead ? 0x7FFFFFFF : c.sens[0]; o.sens[1] = wholeRead ? 0x7FFFFFFF : c.sens[1];
            // keep the candidate log only if some decision of this window depends on the incoming maxScoreMate
            o.candOff32 = 0; o.nCand = 0; o.pad = 0;
            if (c.logOn) {
                bool sensitive = c.sens[0] != 0x7FFFFFFF || c.sens[1] != 0x7FFFFFFF;
                if (sensitive && !c.logOvf) { o.candOff32 = (u32)(((u64)waveId * B.candWaveBytes + candStart) / 32u); o.nCand = c.nCand; }
                else { c.candTop = candStart; if (sensitive) o.nCand = 0xFFFFFFFFu; }
            }
            if (lane == 0) B.wout[w] = o;
        }
        if (emptyM) {                                          // the windows of this chunk that are not walked: an empty result each
            if ((emptyM >> lane) & 1ull) {
                DWinOut z; z.trOffset = z.nTr = z.exOffset = z.nEx = 0; z.mm[0] = z.mm[1] = 0; z.sens[0] = z.sens[1] = 0x7FFFFFFF; z.minIn[0] = z.minIn[1] = 0;
                z.headScore = 0; z.candOff32 = 0; z.headGlen = 0; z.nCand = 0; z.pad = 0;
                B.wout[w0 + chunk + lane] = z;
            }
            nPruned += (u32)__popcll(emptyM); itemPruned += (u32)__popcll(emptyM);
        }
        }
        if (sweep == 0) {
            // the bar every single-mate transcript of the read must stay under (stitch_common.h pruneSingleBar; it bounds the single-mate windows as well)
            if (pruneSingleBar(P, perJ, c.readLength[0], c.readLength[1], maxSeedsRead) < bestSoFar) { sweep = 1; continue; }
            sweep = 2; nRewalk++;
            continue;
        }
        if (chimMode && wholeRead && !chimFull && itemPruned && runChim) { chimFull = true; sweep = 2; nRewalk++; continue; }      // (see pruneOn)
        break;
        }
    }
    if (lane == 0) {
        if (mode == 0 || mode == 3) B.candTops[waveId] = c.candTop;
        atomicAdd((unsigned long long *)&B.counters[DC_nGstitch], (unsigned long long)c.nGstitch);
        atomicAdd((unsigned long long *)&B.counters[DC_nStitchCalls], (unsigned long long)c.nStitchCalls);
        atomicAdd((unsigned long long *)&B.counters[DC_nExtendCalls], (unsigned long long)c.nExtendCalls);
        atomicAdd((unsigned long long *)&B.counters[DC_nNodes], (unsigned long long)c.nNodes);
        atomicAdd((unsigned long long *)&B.counters[DC_nLeaves], (unsigned long long)c.nLeaves);
        DIAG(atomicAdd((unsigned long long *)&B.counters[DC_nLeavesBound], (unsigned long long)c.nLeavesBound); atomicAdd((unsigned long long *)&B.counters[DC_nLeavesEarly], (unsigned long long)c.nLeavesEarly));
        if (nOvf) atomicAdd((unsigned long long *)&B.counters[DC_nOvfStitch], (unsigned long long)nOvf);
        if (nPruned) atomicAdd((unsigned long long *)&B.counters[DC_nPrunedWin], (unsigned long long)nPruned);
        if (nRewalk) atomicAdd((unsigned long long *)&B.counters[DC_nRewalkRead], (unsigned long long)nRewalk);
        if (nRewalkWin) atomicAdd((unsigned long long *)&B.counters[DC_nRewalkWin], (unsigned long long)nRewalkWin);
        if (nSkippedLeaves) atomicAdd((unsigned long long *)&B.counters[DC_nSkippedLeaves], (unsigned long long)nSkippedLeaves);
#ifdef STARAMD_PROFILE
        c.prof[7] = __builtin_readcyclecounter() - profKernelStart;      // whole wave life time
        for (int k = 0; k < 16; k++) atomicAdd((unsigned long long *)&B.counters[DC_prof0 + k], (unsigned long long)c.prof[k]);
        for (int k = 0; k < 12; k++) atomicAdd((unsigned long long *)&B.counters[DC_sprof0 + k], (unsigned long long)c.sprof[k]);
#endif
    }
}

// ---- pass 1 by replay: re-decide a window from its candidate log with the true incoming maxScoreMate ---------------
// The walk of a window (which leaves exist, their scores, the order) does not depend on maxScoreMate; only the record
// decision of each candidate does.  The log holds every leaf that passed (window-best clause OR mate clause with
// incoming 0), in walk order -- a superset, in order, of what any larger incoming value admits -- so replaying the
// decision + de-duplication + ranked insert over the log reproduces the sequential result exactly, with no genome or
// read access at all.
__device__ static bool replayWindow(const staramd_params &P, u32 lane, const DWinOut &o, const u8 *log, WinRec &wr) {
    wr.nWinTr = 0; wr.top = 0; wr.overflow = false; wr.bestScore = 0;
    i32 M[2] = {o.minIn[0], o.minIn[1]};
    const u8 *p = log;
    for (u32 ic = 0; ic < o.nCand; ic++) {
        staramd_transcript t;
        { const u64 *sw = (const u64 *)p; u64 *d = (u64 *)&t;
#pragma unroll
          for (u32 i = 0; i < REC_HDR / 8; i++) d[i] = sw[i]; }
        const u32 ne = t.nExons;
        const staramd_exon *ex = (const staramd_exon *)(p + REC_HDR);
        staramd_exon x;
        { u64 *z = (u64 *)&x; z[0] = z[1] = z[2] = z[3] = 0; }
        if (lane < ne) { const u64 *sw = (const u64 *)(ex + lane); u64 *d = (u64 *)&x; d[0] = sw[0]; d[1] = sw[1]; d[2] = sw[2]; d[3] = sw[3]; }
        p += REC_HDR + 32u * ne;
        const int Score = t.maxScore; const int f = t.iFrag;
        i32 Mf = 0;
        if (f == 0) { M[0] = max(M[0], Score); Mf = M[0]; } else if (f == 1) { M[1] = max(M[1], Score); Mf = M[1]; }
        i32 winBest = wr.bestScore;
        bool c1 = Score + P.outFilterMultimapScoreRange >= winBest || P.chimSegmentMinPositive;
        bool c2 = f >= 0 && Score + P.outFilterMultimapScoreRange >= Mf;
        if (!(c1 || c2)) continue;
        recordCandidate(P, lane, Score, (u64)t.gLength, (u32)t.mappedLength, ne, (const u64 *)&t, x, ex, wr);
        if (wr.overflow) return false;
    }
    return true;
}

extern "C" __global__ void __launch_bounds__(256) k_stitch_replay(const DevIndex *__restrict__ Xp, DevBatch B, u8 *bigArena, u32 capDepth, u32 capRank, u32 arenaBytes,
                                                                u32 bigArenaBytes, u32 ldsWords) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch
    const staramd_params &P = Xp->P;
    const u32 lane = threadIdx.x & 63u;
    u32 waveInBlock = WAVE_INDEX(threadIdx.x >> 6), wavesPerBlock = blockDim.x >> 6;
    u32 stateBytes = stitchStateBytes(capDepth, capRank, arenaBytes);
    u32 readBytes = (ldsWords * 4u + 15u) & ~15u;
    LaneMem m;
    laneSetup((LDS u8 *)ldsReads + waveInBlock * (readBytes + stateBytes) + readBytes, capDepth, capRank, m);
    WinRec wr; wr.rank = m.rank; wr.arenaL = m.arena; wr.arenaBytesL = arenaBytes; wr.arenaG = bigArena + (u64)(blockIdx.x * wavesPerBlock + waveInBlock) * bigArenaBytes; wr.arenaBytesG = bigArenaBytes;
    const u32 nItems = B.cursors[CUR_ST_REPLAY];
    for (;;) {
        u32 it = 0;
        if (lane == 0) it = atomicAdd(&B.cursors[CUR_ST_TICKETR], 1u);
        it = first32(it);
        if (it >= nItems) break;
        u32 w = B.replayList[it];
        DWinOut o = B.wout[w];
        const u8 *log = B.candPool + (u64)o.candOff32 * 32u;
        bool ok = false;
        for (u32 attempt = 0; attempt < 2 && !ok; attempt++) {
            wr.big = attempt != 0; wr.arenaBytes = wr.big ? wr.arenaBytesG : wr.arenaBytesL;
            ok = replayWindow(P, lane, o, log, wr);
        }
        if (!ok) { if (lane == 0) atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_HARD); continue; }
        bool headChimUnused = false;
        if (!flushWindow(B, lane, wr, o, 0u, 0u, headChimUnused)) continue;
        // mm (max over the leaves of this window) and the other fields keep their pass-0 values
        if (lane == 0) B.wout[w] = o;
    }
}

// ---- per read: true incoming maxScoreMate of every window; queue the windows whose pass-0 result may differ ----
extern "C" __global__ void __launch_bounds__(256) k_stitch_verify(const DevIndex *__restrict__ Xp, DevBatch B) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch
    u32 ir = blockIdx.x * blockDim.x + threadIdx.x;
    if (ir >= B.nReads) return;
    const DRead rd = B.reads[ir];
    if (rd.nWin == 0) return;
    i32 M0 = 0, M1 = 0; u32 nRedo = 0, nReplay = 0;
    for (u32 iw = 0; iw < rd.nWin; iw++) {
        u32 w = rd.winOffset + iw;
        DWinOut o = B.wout[w];
        if (o.sens[0] < M0 || o.sens[1] < M1) {
            o.minIn[0] = M0; o.minIn[1] = M1; B.wout[w] = o;
            if (o.nCand == 0xFFFFFFFFu) { u32 k = atomicAdd(&B.cursors[CUR_ST_REDO], 1u); B.redoList[k] = w; nRedo++; }
            else { u32 k = atomicAdd(&B.cursors[CUR_ST_REPLAY], 1u); B.replayList[k] = w; nReplay++; }
        }
        M0 = max(M0, o.mm[0]); M1 = max(M1, o.mm[1]);
    }
    if (nRedo) atomicAdd((unsigned long long *)&B.counters[DC_nRedoWin], (unsigned long long)nRedo);
    if (nReplay) atomicAdd((unsigned long long *)&B.counters[DC_nReplayWin], (unsigned long long)nReplay);
}

// ---- resultSelect 2: the partner loop of chimeric detection on the device (include/star_amd.h) --------------------------------------------------------------------------
// ReadAlign_chimericDetectionOld.cpp:50-108 over the records of the pools: trBest against the head of every other window and the other transcripts of its own.  One lane
// per read (k_stitch_finish); the arithmetic is the reference's (unsigned 64-bit read coordinates, the `roStart > readLength[0]` step over the mate spacer).
struct ChimSel { i32 scoreBest, scoreNext; u32 strBest; u32 win, rank; bool have; };       // win: index among the read's windows (winOffset + win), rank inside it
__device__ __forceinline__ u32 chimStrOf(const staramd_transcript &t) {                      // :40-46, :59-65
    if (t.intronMotifs[1] == 0 && t.intronMotifs[2] == 0) return 0u;
    return ((t.Str == 0) == (t.intronMotifs[1] > 0)) ? 1u : 2u;
}
// blocksOverlap.cpp:3-41 on two transcripts of the pools (ea / eb: their first exons)
__device__ static u64 chimBlocksOverlap(const staramd_transcript &a, const staramd_exon *ea, const staramd_transcript &b, const staramd_exon *eb) {
    u32 i1 = 0, i2 = 0; u64 n = 0;
    while (i1 < a.nExons && i2 < b.nExons) {
        const u64 rs1 = ea[i1].R, rs2 = eb[i2].R, re1 = rs1 + ea[i1].L, re2 = rs2 + eb[i2].L, gs1 = ea[i1].G, gs2 = eb[i2].G;
        if (rs1 >= re2) i2++;
        else if (rs2 >= re1) i1++;
        else if (gs1 - rs1 != gs2 - rs2) { if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
        else { n += min(re1, re2) - max(rs1, rs2); if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
    }
    return n;
}
__device__ static ChimSel chimSelectPartner(const staramd_params &P, const DevBatch &B, const DRead &rd, u32 ir, u32 bestWin) {
    ChimSel c; c.scoreBest = 0; c.scoreNext = 0; c.strBest = 0; c.win = 0; c.rank = 0; c.have = false;
    const u64 Lread = B.readOffset[ir + 1] - B.readOffset[ir], len0 = B.mate1Length[ir];
    const DWinOut ob = B.wout[rd.winOffset + bestWin];
    const staramd_transcript tb = B.trPool[ob.trOffset];
    const staramd_exon *xb = B.exPool + ob.exOffset + tb.exonOffset;
    const u32 nb = tb.nExons;
    u64 roStart1 = tb.Str == 0 ? (u64)xb[0].R : Lread - xb[nb - 1].R - xb[nb - 1].L;
    u64 roEnd1 = tb.Str == 0 ? (u64)xb[nb - 1].R + xb[nb - 1].L - 1 : Lread - xb[0].R - 1;
    if (roStart1 > len0) roStart1--;
    if (roEnd1 > len0) roEnd1--;
    const u32 chimStr = chimStrOf(tb);
    const u64 segMin = P.chimSegmentMin, gapMax = P.chimSegmentReadGapMax;
    for (u32 w = 0; w < rd.nWin; w++) {
        const DWinOut o = B.wout[rd.winOffset + w];
        const u32 nt = w == bestWin ? o.nTr : min(o.nTr, 1u);                       // other windows: their head only (:52)
        for (u32 k = (w == bestWin ? 1u : 0u); k < nt; k++) {                       // the best window: everything but trBest itself (:53)
            const staramd_transcript t = B.trPool[o.trOffset + k];
            if (t.intronMotifs[0] > 0) continue;
            const u32 chimStr1 = chimStrOf(t);
            if (chimStr != 0 && chimStr1 != 0 && chimStr != chimStr1) continue;
            const staramd_exon *x = B.exPool + o.exOffset + t.exonOffset; const u32 ne = t.nExons;
            u64 roStart2 = t.Str == 0 ? (u64)x[0].R : Lread - x[ne - 1].R - x[ne - 1].L;
            u64 roEnd2 = t.Str == 0 ? (u64)x[ne - 1].R + x[ne - 1].L - 1 : Lread - x[0].R - 1;
            if (roStart2 > len0) roStart2--;
            if (roEnd2 > len0) roEnd2--;
            const u64 chimOverlap = roStart2 > roStart1 ? (roStart2 > roEnd1 ? 0 : roEnd1 - roStart2 + 1) : (roEnd2 < roStart1 ? 0 : roEnd2 - roStart1 + 1);
            const bool diffMates = (roEnd1 < len0 && roStart2 >= len0) || (roEnd2 < len0 && roStart1 >= len0);
            if (!(roEnd1 > segMin + roStart1 + chimOverlap && roEnd2 > segMin + roStart2 + chimOverlap
                  && (diffMates || ((roEnd1 + gapMax + 1) >= roStart2 && (roEnd2 + gapMax + 1) >= roStart1)))) continue;
            const i32 chimScore = tb.maxScore + t.maxScore - (i32)chimOverlap;
            u64 overlap1 = 0;
            if (k > 0 && c.scoreBest > 0) {                                         // :84-88 (only ever non-zero for two transcripts of one window)
                const DWinOut op = B.wout[rd.winOffset + c.win];
                const staramd_transcript tp = B.trPool[op.trOffset + c.rank];
                overlap1 = chimBlocksOverlap(tp, B.exPool + op.exOffset + tp.exonOffset, t, x);
            }
            if (chimScore > c.scoreBest) {
                c.win = w; c.rank = k; c.have = true;
                if (overlap1 == 0) c.scoreNext = c.scoreBest;
                c.scoreBest = chimScore; c.strBest = chimStr1;
            } else if (chimScore > c.scoreNext && overlap1 == 0) c.scoreNext = chimScore;
        }
    }
    return c;
}

// ---- per read: totals, trBest, maxScoreMate (ReadAlign_stitchPieces.cpp:288-348) ----
// alignTranscriptsPerReadNmax (:290-294) stops the reference's walk before window k when the transcripts recorded so
// far reach the limit; windows < k do not depend on windows >= k, so the exact result is the prefix.
extern "C" __global__ void __launch_bounds__(256) k_stitch_finish(const DevIndex *__restrict__ Xp, DevBatch B) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch
    const staramd_params &P = Xp->P;
    u32 ir = blockIdx.x * blockDim.x + threadIdx.x;
    if (ir >= B.nReads) return;
    DRead rd = B.reads[ir];
    if (rd.nWin == 0) {
        if (rd.nSeeds > 0 && !(rd.status & (STARAMD_ST_SCRATCH_OVERFLOW | STARAMD_ST_NO_GOOD_WINDOW))) { rd.status |= STARAMD_ST_NO_GOOD_WINDOW; B.reads[ir] = rd; }
        return;
    }
    u32 nWt = 0, nTr = 0, nEx = 0; int bestScore = 0; u64 bestGlen = 0; i32 bestW = -1; i32 M0 = 0, M1 = 0;
    for (u32 iw = 0; iw < rd.nWin; iw++) {
        if (nTr + P.alignTranscriptsPerWindowNmax >= P.alignTranscriptsPerReadNmax) {
            rd.status |= STARAMD_ST_TR_PER_READ_LIMIT;
            for (; iw < rd.nWin; iw++) { B.wout[rd.winOffset + iw].nTr = 0; B.wout[rd.winOffset + iw].nEx = 0; }
            break;
        }
        const DWinOut o = B.wout[rd.winOffset + iw];
        M0 = max(M0, o.mm[0]); M1 = max(M1, o.mm[1]);
        if (o.nTr == 0) continue;
        if (o.headScore > bestScore || (o.headScore == bestScore && o.headGlen < bestGlen)) { bestW = (i32)nWt; bestScore = o.headScore; bestGlen = o.headGlen; }
        nWt++; nTr += o.nTr; nEx += o.nEx;
    }
    if (bestScore == 0) { rd.status |= STARAMD_ST_NO_GOOD_WINDOW; nWt = 0; nTr = 0; nEx = 0; bestW = -1; }   // :344-348
    else if (P.resultSelect) {
        // staramd_params::resultSelect: return only what multMapSelect can pick (ReadAlign_multMapSelect.cpp:26-44):
        // per window the best-first prefix with maxScore + outFilterMultimapScoreRange >= trBest->maxScore
        // (resultSelect 2: ... and the partner chimeric detection would choose -- the prefix of its window reaches up to it)
        ChimSel cs; cs.have = false; cs.scoreBest = cs.scoreNext = 0; cs.strBest = 0; cs.win = cs.rank = 0;
        if (P.resultSelect == 2u) {
            u32 ordB = 0, winB = 0;
            for (u32 iw = 0; iw < rd.nWin; iw++) { if (B.wout[rd.winOffset + iw].nTr == 0) continue; if ((i32)ordB == bestW) { winB = iw; break; } ordB++; }
            cs = chimSelectPartner(P, B, rd, ir, winB);
        }
        const int selMin = bestScore - P.outFilterMultimapScoreRange;
        i32 bestOrd = -1; u32 ord = 0, kept = 0; nTr = 0; nEx = 0; u32 partnerIdx = 0;
        for (u32 iw = 0; iw < rd.nWin; iw++) {
            DWinOut o = B.wout[rd.winOffset + iw];
            if (o.nTr == 0) continue;
            u32 sel = 0, selEx = 0;
            for (; sel < o.nTr; sel++) { const staramd_transcript &t = B.trPool[o.trOffset + sel]; if (t.maxScore < selMin && !(cs.have && cs.win == iw && sel <= cs.rank)) break; selEx += t.nExons; }
            if ((i32)ord == bestW) bestOrd = (i32)kept;
            ord++;
            if (cs.have && cs.win == iw) partnerIdx = nTr + cs.rank;
            if (sel != o.nTr) { B.wout[rd.winOffset + iw].nTr = sel; B.wout[rd.winOffset + iw].nEx = selEx; }
            if (sel) { kept++; nTr += sel; nEx += selEx; }
        }
        nWt = kept; bestW = bestOrd;
        if (cs.have) { rd.status |= STARAMD_ST_CHIM_PARTNER; M0 = cs.scoreBest; M1 = cs.scoreNext; rd.unmappedLength = partnerIdx | (cs.strBest << 30); }
    }
    rd.nWt = nWt; rd.nTr = nTr; rd.nEx = nEx; rd.bestW = bestW;
    // resultSelect 1: windows that cannot hold a selectable transcript are not walked, so the running maxima cover only some windows (and which ones depends on the order
    // wavefronts saw each other's bounds): the field is returned as 0, always (include/star_amd.h); resultSelect 0 returns ReadAlign::maxScoreMate[] exactly
    if (P.resultSelect && !(rd.status & STARAMD_ST_CHIM_PARTNER)) M0 = M1 = 0;
    rd.maxScoreMate[0] = M0; rd.maxScoreMate[1] = M1;
    B.reads[ir] = rd;
    if (nTr) atomicAdd((unsigned long long *)&B.counters[DC_nTrOut], (unsigned long long)nTr);
}

// 4-bit packing of the combined reads for LDS staging: one block per read
extern "C" __global__ void __launch_bounds__(64) k_pack_reads(DevBatch B, u32 *packed, u32 packWords) {
    u32 ir = blockIdx.x;
    const u8 *R = B.bases + B.readOffset[ir];
    u32 L = (u32)(B.readOffset[ir + 1] - B.readOffset[ir]);
    for (u32 w = threadIdx.x; w < packWords; w += blockDim.x) {
        u32 v = 0;
        for (u32 k = 0; k < 8; k++) { u32 j = w * 8 + k; u32 cde = j < L ? (R[j] & 15u) : 15u; v |= cde << (4 * k); }
        packed[(u64)ir * packWords + w] = v;
    }
}
