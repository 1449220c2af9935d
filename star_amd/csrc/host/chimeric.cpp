// chimeric.cpp -- chimeric alignment detection on the transcripts of all windows (--chimSegmentMin > 0, --chimMultimapNmax 0,
// --chimOutType Junctions): the best alignment plus the best-scoring alignment of another window that covers the rest of the read.
//   ReadAlign::chimericDetection            source/ReadAlign_chimericDetection.cpp:16-57
//   ReadAlign::chimericDetectionOld         source/ReadAlign_chimericDetectionOld.cpp:7-312
//   ReadAlign::chimericDetectionOldOutput   source/ReadAlign_chimericDetectionOldOutput.cpp:5-74  (Chimeric.out.junction line)
//   ReadAlign::outputTranscriptCIGARp       source/ReadAlign_outputTranscriptCIGARp.cpp:4-68
//   blocksOverlap                           source/blocksOverlap.cpp:3-41
// Host post-map code.  The device returns every recorded transcript of every window for it (resultSelect 0,
// chimSegmentMinPositive 1: stitchWindowAligns.cpp:247) -- or, for chimericDetectionOld on an engine that can (resultSelect 2), runs the partner loop itself and
// returns its outcome beside the transcripts multMapSelect can pick (k_stitch.hip chimSelectPartner).
#include "host.h"
#include <algorithm>
#include <cstring>
#include <cmath>

namespace staramd {

namespace {
void load(ChimTr &c, const staramd_transcript &t, const staramd_exon *ex) { c.t = t; memcpy(c.ex, ex, sizeof(staramd_exon) * t.nExons); }

uint64_t blocksOverlap(const ChimTr &a, const staramd_transcript &t2, const staramd_exon *e2) {
    uint64_t i1 = 0, i2 = 0, n = 0;
    while (i1 < a.t.nExons && i2 < t2.nExons) {
        uint64_t rs1 = a.ex[i1].R, rs2 = e2[i2].R, re1 = rs1 + a.ex[i1].L, re2 = rs2 + e2[i2].L, gs1 = a.ex[i1].G, gs2 = e2[i2].G;
        if (rs1 >= re2) i2++;
        else if (rs2 >= re1) i1++;
        else if (gs1 - rs1 != gs2 - rs2) { if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
        else { n += std::min(re1, re2) - std::max(rs1, rs2); if (re1 >= re2) i2++; if (re2 >= re1) i1++; }
    }
    return n;
}

inline void appendU(std::string &s, uint64_t v) { char b[24]; int n = 0; do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v); while (n) s.push_back(b[--n]); }

std::string cigarP(const ChimTr &c, const uint64_t readLength[2], uint64_t readLengthPair, int nMates) {
    std::string s;
    const uint64_t leftMate = nMates > 1 ? c.t.Str : 0;
    const uint32_t ne = c.t.nExons;
    uint64_t trimL = c.ex[0].R - (c.ex[0].R < readLength[leftMate] ? 0 : readLength[leftMate] + 1);
    if (trimL > 0) { appendU(s, trimL); s.push_back('S'); }
    for (uint32_t ii = 0; ii < ne; ii++) {
        if (ii > 0) {
            uint64_t prevEnd = c.ex[ii - 1].G + c.ex[ii - 1].L, gapG = c.ex[ii].G - prevEnd;
            if (c.ex[ii].G >= prevEnd) {
                if (c.ex[ii - 1].canonSJ == -3) {
                    uint64_t s1 = readLength[leftMate] - (c.ex[ii - 1].R + c.ex[ii - 1].L), s2 = c.ex[ii].R - (readLength[leftMate] + 1);
                    if (s1 > 0) { appendU(s, s1); s.push_back('S'); }
                    appendU(s, gapG); s.push_back('p');
                    if (s2 > 0) { appendU(s, s2); s.push_back('S'); }
                } else {
                    uint64_t gapR = (uint64_t)c.ex[ii].R - c.ex[ii - 1].R - c.ex[ii - 1].L;
                    if (gapR > 0) { appendU(s, gapR); s.push_back('I'); }
                    if (c.ex[ii - 1].canonSJ >= 0 || c.ex[ii - 1].sjAnnot == 1) { appendU(s, gapG); s.push_back('N'); }
                    else if (gapG > 0) { appendU(s, gapG); s.push_back('D'); }
                }
            } else { s.push_back('-'); appendU(s, prevEnd - c.ex[ii].G); s.push_back('p'); }
        }
        appendU(s, c.ex[ii].L); s.push_back('M');
    }
    trimL = (c.ex[ne - 1].R < readLength[leftMate] ? readLength[leftMate] : readLengthPair) - c.ex[ne - 1].R - c.ex[ne - 1].L;
    if (trimL > 0) { appendU(s, trimL); s.push_back('S'); }
    return s;
}

// the scan for the chimeric junction inside a mate (ReadAlign_chimericDetectionOld.cpp:143-229 = ChimericAlign_chimericStitching.cpp:40-122): every
// position between the start of segment 0's last block and the end of segment 1's first block is scored by which genome the read base agrees
// with; a GT/AG (CT/AC) pair at the position wins ties.  false = rejected (N in the read, or in the genome with banGenomicN)
struct JunctionScan { uint64_t roStart0, roStartB, jRbest; int motif; };
template <class GenomeAt>
bool scanJunction(const ChimParams &C, GenomeAt &G, const uint8_t *Read1, uint64_t Lread, uint64_t readLength0, const ChimTr &t0, const ChimTr &t1, uint32_t e0, uint32_t e1,
                  uint32_t chimStr, JunctionScan &js) {
    const staramd_exon &x0 = t0.ex[e0], &x1 = t1.ex[e1];
    const uint64_t roStart0 = t0.t.Str == 0 ? x0.R : Lread - x0.R - x0.L;
    const uint64_t roStartB = t1.t.Str == 0 ? x1.R : Lread - x1.R - x1.L;
    uint64_t jR, jRbest = 0; int jScore = 0, jMotif = 0, jScoreBest = -999999, jScoreJ = 0, chimMotif = 0;
    uint64_t jRmax = roStartB + x1.L;
    jRmax = jRmax > roStart0 ? jRmax - roStart0 - 1 : 0;
    for (jR = 0; jR < jRmax; jR++) {
        if (jR == readLength0) jR++;
        uint8_t bR = Read1[roStart0 + jR];
        uint8_t b0, b1;
        if (t0.t.Str == 0) b0 = G(x0.G + jR); else { b0 = G(x0.G + x0.L - 1 - jR); if (b0 < 4) b0 = 3 - b0; }
        if (t1.t.Str == 0) b1 = G(x1.G - roStartB + roStart0 + jR); else { b1 = G(x1.G + x1.L - 1 + roStartB - roStart0 - jR); if (b1 < 4) b1 = 3 - b1; }
        if ((C.filterGenomicN && (b0 > 3 || b1 > 3)) || bR > 3) return false;
        uint8_t b01, b02, b11, b12;
        if (t0.t.Str == 0) { b01 = G(x0.G + jR + 1); b02 = G(x0.G + jR + 2); }
        else { b01 = G(x0.G + x0.L - 1 - jR - 1); if (b01 < 4) b01 = 3 - b01; b02 = G(x0.G + x0.L - 1 - jR - 2); if (b02 < 4) b02 = 3 - b02; }
        if (t1.t.Str == 0) { b11 = G(x1.G - roStartB + roStart0 + jR - 1); b12 = G(x1.G - roStartB + roStart0 + jR); }
        else { b11 = G(x1.G + x1.L - 1 + roStartB - roStart0 - jR + 1); if (b11 < 4) b11 = 3 - b11; b12 = G(x1.G + x1.L - 1 + roStartB - roStart0 - jR); if (b12 < 4) b12 = 3 - b12; }
        jMotif = 0;
        if (b01 == 2 && b02 == 3 && b11 == 0 && b12 == 2) { if (chimStr != 2) jMotif = 1; }
        else if (b01 == 1 && b02 == 3 && b11 == 0 && b12 == 1) { if (chimStr != 1) jMotif = 2; }
        if (bR == b0 && bR != b1) jScore++; else if (bR != b0 && bR == b1) jScore--;
        jScoreJ = jMotif == 0 ? jScore + C.scoreJunctionNonGTAG : jScore;
        if (jScoreJ > jScoreBest || (jScoreJ == jScoreBest && jMotif > 0)) { chimMotif = jMotif; jRbest = jR; jScoreBest = jScoreJ; }
    }
    js.roStart0 = roStart0; js.roStartB = roStartB; js.jRbest = jRbest; js.motif = chimMotif;
    return true;
}

// the two blocks next to the junction are cut / extended to meet at it, then the repeat lengths around it (:231-283 = stitching :125-170)
template <class GenomeAt>
void shiftToJunction(GenomeAt &G, ChimTr &t0, ChimTr &t1, uint32_t e0, uint32_t e1, const JunctionScan &js, uint64_t &chimJ0, uint64_t &chimJ1, uint64_t &chimRepeat0, uint64_t &chimRepeat1) {
    staramd_exon &x0 = t0.ex[e0], &x1 = t1.ex[e1];
    const uint64_t roStart0 = js.roStart0, roStartB = js.roStartB, jRbest = js.jRbest;
    if (t0.t.Str == 1) { x0.R = (uint16_t)(x0.R + x0.L - jRbest - 1); x0.G += x0.L - jRbest - 1; x0.L = (uint16_t)(jRbest + 1); chimJ0 = x0.G - 1; }
    else { x0.L = (uint16_t)(jRbest + 1); chimJ0 = x0.G + x0.L; }
    if (t1.t.Str == 0) {
        x1.R = (uint16_t)(x1.R + roStart0 + jRbest + 1 - roStartB); x1.G += roStart0 + jRbest + 1 - roStartB;
        x1.L = (uint16_t)(roStartB + x1.L - roStart0 - jRbest - 1); chimJ1 = x1.G - 1;
    } else { x1.L = (uint16_t)(roStartB + x1.L - roStart0 - jRbest - 1); chimJ1 = x1.G + x1.L; }
    uint8_t b0, b1; uint64_t jR;
    for (jR = 0; jR < 100; jR++) {
        if (t0.t.Str == 0) b0 = G(chimJ0 + jR); else { b0 = G(chimJ0 - jR); if (b0 < 4) b0 = 3 - b0; }
        if (t1.t.Str == 0) b1 = G(chimJ1 + 1 + jR); else { b1 = G(chimJ1 - 1 - jR); if (b1 < 4) b1 = 3 - b1; }
        if (b0 != b1) break;
    }
    chimRepeat1 = jR;
    for (jR = 0; jR < 100; jR++) {
        if (t0.t.Str == 0) b0 = G(chimJ0 - 1 - jR); else { b0 = G(chimJ0 + 1 + jR); if (b0 < 4) b0 = 3 - b0; }
        if (t1.t.Str == 0) b1 = G(chimJ1 - jR); else { b1 = G(chimJ1 + jR); if (b1 < 4) b1 = 3 - b1; }
        if (b0 != b1) break;
    }
    chimRepeat0 = jR;
}
// score of an alignment recomputed from its blocks (after the junction shift)
} // namespace
int chimAlignScore(const staramd_params &D, const GenomeIndex &gi, const uint8_t *Read1, uint64_t Lread, ChimTr &c) {
    int maxScore = 0; uint32_t nMM = 0;
    c.t.maxScore = 0; c.t.nMM = 0;
    const uint32_t ne = c.t.nExons;
    if (ne == 0) return 0;
    for (uint32_t iex = 0; iex < ne; iex++)
        for (uint32_t ii = 0; ii < c.ex[iex].L; ii++) {
            uint64_t rp = (uint64_t)c.ex[iex].R + ii;
            uint8_t r1 = c.t.roStr == 0 ? Read1[rp] : Read1[Lread - 1 - rp];
            if (c.t.roStr != 0 && r1 < 4) r1 = 3 - r1;
            uint8_t g1 = gi.G[c.ex[iex].G + ii];
            if (r1 > 3 || g1 > 3) continue;
            if (r1 == g1) ++maxScore; else { --maxScore; ++nMM; }
        }
    for (uint32_t iex = 0; iex + 1 < ne; iex++) {
        if (c.ex[iex].sjAnnot == 1) { maxScore += D.sjdbScore; continue; }
        switch (c.ex[iex].canonSJ) {
            case -3: break;
            case -2: maxScore += (int)((int64_t)c.ex[iex + 1].R - c.ex[iex].R - c.ex[iex].L) * D.scoreInsBase + D.scoreInsOpen; break;
            case -1: maxScore += (int)((int64_t)(c.ex[iex + 1].G - c.ex[iex].G) - c.ex[iex].L) * D.scoreDelBase + D.scoreDelOpen; break;
            case 0: maxScore += D.scoreGapNoncan + D.scoreGap; break;
            case 1: case 2: maxScore += D.scoreGap; break;
            case 3: case 4: maxScore += D.scoreGapGCAG + D.scoreGap; break;
            case 5: case 6: maxScore += D.scoreGapATAC + D.scoreGap; break;
        }
    }
    if (D.scoreGenomicLengthLog2scale != 0) {
        unsigned long long gl = std::max(1ULL, (unsigned long long)(c.ex[ne - 1].G + c.ex[ne - 1].L - c.ex[0].G));
        maxScore += int(std::ceil(std::log2((double)gl) * D.scoreGenomicLengthLog2scale - 0.5));
    }
    c.t.maxScore = maxScore; c.t.nMM = nMM;
    return maxScore;
}
namespace {
// ReadAlign::peOverlapChimericSEtoPE (ReadAlign_peOverlapMergeMap.cpp:309-368): both segments of a chimera of merged mates cut back into the two mates; the
// shortest one-mate part of either segment (ties: the one whose junction lies deeper in its mate) is dropped, so that one mate is chimerically split, not both.
// nb / nir: the pair; Lread: length of the merged read
static void chimericMergedToPair(const ReadBatch &nb, uint32_t nir, const uint32_t *mateStart, uint64_t Lread, const ChimTr &se1, const ChimTr &se2, ChimTr tmp[2]) {
    const uint64_t LreadPE = nb.readOffset[nir + 1] - nb.readOffset[nir];
    const uint64_t readLengthPE[2] = {nb.mate1Length[nir], LreadPE - nb.mate1Length[nir] - 1};
    const uint64_t readLengthOriginalPE[2] = {nb.seqSpan[0][nir].len, nb.seqSpan[1][nir].len};
    mergedAlignToPair(tmp[0], mateStart, se1.t, se1.ex, Lread, readLengthPE, LreadPE);
    mergedAlignToPair(tmp[1], mateStart, se2.t, se2.ex, Lread, readLengthPE, LreadPE);
    uint64_t segLen[2][2] = {{0, 0}, {0, 0}}, segEx[2] = {0, 0}, i1 = 0, i2 = 0, posOfJunctionInRead = 0;
    for (uint64_t ii = 0; ii < 2; ii++) {
        for (uint32_t iex = 0; iex < tmp[ii].t.nExons; iex++) {
            if (tmp[ii].ex[iex].iFrag == tmp[ii].ex[0].iFrag) { segLen[ii][0] += tmp[ii].ex[iex].L; segEx[ii] = iex; } else segLen[ii][1] += tmp[ii].ex[iex].L;
        }
        const uint64_t readLen0 = readLengthOriginalPE[tmp[ii].ex[0].iFrag], readLen1 = readLengthOriginalPE[1 - tmp[ii].ex[0].iFrag];
        for (uint64_t jj = 0; jj < 2; jj++) {
            const uint64_t R = tmp[ii].ex[jj].R;
            const uint64_t cur = R > readLen0 ? readLen0 + readLen1 + 1 - R : R;
            if (segLen[ii][jj] < segLen[i1][i2] || (segLen[ii][jj] == segLen[i1][i2] && cur > posOfJunctionInRead)) { posOfJunctionInRead = cur; i1 = ii; i2 = jj; }
        }
    }
    ChimTr &c = tmp[i1];
    if (i2 == 1) c.t.nExons = (uint16_t)(segEx[i1] + 1);
    else {
        const uint32_t shift = (uint32_t)segEx[i1] + 1, nNew = c.t.nExons - shift;
        for (uint32_t iex = 0; iex < nNew; iex++) c.ex[iex] = c.ex[iex + shift];
        c.t.nExons = (uint16_t)nNew;
    }
}
} // namespace

// returns true when a chimeric alignment was recorded (Stats::chimericAll); the junction line is appended to `out`
bool chimericDetectionOld(const RunParams &P, const GenomeIndex &gi, const ReadBatch &b, uint32_t ir, const ReadAligns &ra,
                          const staramd_transcript *trBest, uint64_t nTr, const staramd_transcript *trMult0, const staramd_transcript *trMult1, std::string &out,
                          std::vector<ChimPair> *bamOut, const ReadBatch *nameBatch, uint32_t nameIr, const uint32_t *mateStart) {
    const ChimParams &C = P.chim;
    const bool merged = nameBatch != nullptr;                 // b holds merged mates (ReadAlign_chimericDetectionPEmerged.cpp:12-25): a single-end read for the detection
    const struct { uint32_t nTr; } rr = {ra.nTr};
    const staramd_transcript *T = ra.T;
    const struct { const staramd_exon *ex; } r = {ra.ex};
    const uint64_t Lread = b.readOffset[ir + 1] - b.readOffset[ir];
    const int nMates = merged ? 1 : (int)P.dev.readNmates;
    const uint64_t readLength[2] = {b.mate1Length[ir], nMates == 2 ? Lread - b.mate1Length[ir] - 1 : 0};
    const uint8_t *Read1 = b.bases.data() + b.readOffset[ir];
    const int64_t nG = (int64_t)gi.G.size();
    auto G = [&](uint64_t p) -> uint8_t { int64_t q = (int64_t)p; return q >= 0 && q < nG ? gi.G[(size_t)q] : 5; };

    if (nTr > C.mainSegmentMultNmax && nTr != 2) return false;
    const staramd_exon *exB = r.ex + trBest->exonOffset;
    const uint32_t neB = trBest->nExons;
    if (!(C.segmentMin > 0 && trBest->rLength >= C.segmentMin
          && ((uint64_t)exB[neB - 1].R + exB[neB - 1].L + C.segmentMin <= Lread || exB[0].R >= C.segmentMin)
          && trBest->intronMotifs[0] == 0 && (trBest->intronMotifs[1] == 0 || trBest->intronMotifs[2] == 0))) return false;
    int chimScoreBest = 0, chimScoreNext = 0;
    ChimTr trChim[2];
    load(trChim[0], *trBest, exB);
    const staramd_transcript *trChim1 = nullptr;
    uint64_t roStart1 = trBest->Str == 0 ? exB[0].R : Lread - exB[neB - 1].R - exB[neB - 1].L;
    uint64_t roEnd1 = trBest->Str == 0 ? (uint64_t)exB[neB - 1].R + exB[neB - 1].L - 1 : Lread - exB[0].R - 1;
    if (roStart1 > readLength[0]) roStart1--;
    if (roEnd1 > readLength[0]) roEnd1--;
    uint32_t chimStr, chimStrBest = 0;
    if (trBest->intronMotifs[1] == 0 && trBest->intronMotifs[2] == 0) chimStr = 0;
    else if ((trBest->Str == 0) == (trBest->intronMotifs[1] > 0)) chimStr = 1;
    else chimStr = 2;
    if (ra.pre) {             // the engine has run the loop below on all windows and returned its outcome with the partner (include/star_amd.h, resultSelect 2)
        chimScoreBest = ra.pre->scoreBest; chimScoreNext = ra.pre->scoreNext; chimStrBest = ra.pre->strBest;
        if (ra.pre->partner >= 0 && (uint32_t)ra.pre->partner < rr.nTr) { const staramd_transcript &t = T[ra.pre->partner]; load(trChim[1], t, r.ex + t.exonOffset); trChim1 = &t; }
    } else
    // windows: consecutive transcripts with the same iW, best first
    for (uint32_t k0 = 0; k0 < rr.nTr;) {
        uint32_t k1 = k0;
        while (k1 < rr.nTr && T[k1].iW == T[k0].iW) k1++;
        const bool bestWindow = trBest == T + k0;
        for (uint32_t k = k0; k < k1; k++) {
            const uint32_t iWt = k - k0;
            if (!bestWindow && iWt > 0) break;
            if (bestWindow && iWt == 0) continue;
            const staramd_transcript &t = T[k]; const staramd_exon *ex = r.ex + t.exonOffset; const uint32_t ne = t.nExons;
            if (t.intronMotifs[0] > 0) continue;
            uint32_t chimStr1;
            if (t.intronMotifs[1] == 0 && t.intronMotifs[2] == 0) chimStr1 = 0;
            else if ((t.Str == 0) == (t.intronMotifs[1] > 0)) chimStr1 = 1;
            else chimStr1 = 2;
            if (chimStr != 0 && chimStr1 != 0 && chimStr != chimStr1) continue;
            uint64_t roStart2 = t.Str == 0 ? ex[0].R : Lread - ex[ne - 1].R - ex[ne - 1].L;
            uint64_t roEnd2 = t.Str == 0 ? (uint64_t)ex[ne - 1].R + ex[ne - 1].L - 1 : Lread - ex[0].R - 1;
            if (roStart2 > readLength[0]) roStart2--;
            if (roEnd2 > readLength[0]) roEnd2--;
            uint64_t chimOverlap = roStart2 > roStart1 ? (roStart2 > roEnd1 ? 0 : roEnd1 - roStart2 + 1) : (roEnd2 < roStart1 ? 0 : roEnd2 - roStart1 + 1);
            bool diffMates = (roEnd1 < readLength[0] && roStart2 >= readLength[0]) || (roEnd2 < readLength[0] && roStart1 >= readLength[0]);
            if (roEnd1 > C.segmentMin + roStart1 + chimOverlap && roEnd2 > C.segmentMin + roStart2 + chimOverlap
                && (diffMates || ((roEnd1 + C.segmentReadGapMax + 1) >= roStart2 && (roEnd2 + C.segmentReadGapMax + 1) >= roStart1))) {
                int chimScore = trBest->maxScore + t.maxScore - (int)chimOverlap;
                uint64_t overlap1 = 0;
                if (iWt > 0 && chimScoreBest > 0) overlap1 = blocksOverlap(trChim[1], t, ex);
                if (chimScore > chimScoreBest) {
                    load(trChim[1], t, ex); trChim1 = &t;
                    if (overlap1 == 0) chimScoreNext = chimScoreBest;
                    chimScoreBest = chimScore;
                    chimStrBest = chimStr1;
                } else if (chimScore > chimScoreNext && overlap1 == 0) chimScoreNext = chimScore;
            }
        }
        k0 = k1;
    }
    const int readL = (int)(readLength[0] + readLength[1]);
    if (!(chimScoreBest >= C.scoreMin && chimScoreBest + C.scoreDropMax >= readL)) return false;
    if (nTr > C.mainSegmentMultNmax) { if (trChim1 != trMult0 && trChim1 != trMult1) return false; }
    if (chimStr == 0) chimStr = chimStrBest;
    if (chimScoreNext + C.scoreSeparation >= chimScoreBest) return false;
    auto roStartOf = [&](const ChimTr &c) { return c.t.roStr == 0 ? (uint64_t)c.t.rStart : Lread - c.t.rStart - c.t.rLength; };
    if (roStartOf(trChim[0]) > roStartOf(trChim[1])) std::swap(trChim[0], trChim[1]);
    VarOverlap varOrig[2];
    if (bamOut && P.var && !merged) for (int k = 0; k < 2; k++) P.var->overlap(trChim[k].t, trChim[k].ex, Read1, Lread, gi.chrStart[trChim[k].t.Chr], varOrig[k]);
    const uint32_t e0 = trChim[0].t.Str == 1 ? 0 : trChim[0].t.nExons - 1, e1 = trChim[1].t.Str == 0 ? 0 : trChim[1].t.nExons - 1;
    uint64_t chimRepeat0 = 0, chimRepeat1 = 0, chimJ0 = 0, chimJ1 = 0; int chimMotif = 0;
    staramd_exon &x0 = trChim[0].ex[e0], &x1 = trChim[1].ex[e1];
    if (x0.iFrag > x1.iFrag) return false;
    else if (x0.iFrag < x1.iFrag) {                                  // mates bracket the chimeric junction
        chimMotif = -1;
        chimJ0 = trChim[0].t.Str == 1 ? x0.G - 1 : x0.G + x0.L;
        chimJ1 = trChim[1].t.Str == 0 ? x1.G - 1 : x1.G + x1.L;
    } else {                                                         // junction inside one mate: find it, shift the segments (:143-283)
        if (!(x0.L >= C.junctionOverhangMin && x1.L >= C.junctionOverhangMin)) return false;
        JunctionScan js;
        if (!scanJunction(C, G, Read1, Lread, readLength[0], trChim[0], trChim[1], e0, e1, chimStr, js)) return false;
        chimMotif = js.motif;
        if (chimMotif == 0) {
            chimScoreBest += 1 + C.scoreJunctionNonGTAG;
            if (!(chimScoreBest >= C.scoreMin && chimScoreBest + C.scoreDropMax >= readL)) return false;
        }
        shiftToJunction(G, trChim[0], trChim[1], e0, e1, js, chimJ0, chimJ1, chimRepeat0, chimRepeat1);
    }
    // final check (:296-309): different chromosome / strand, or far apart
    if (trChim[0].t.Str != trChim[1].t.Str || trChim[0].t.Chr != trChim[1].t.Chr
        || (trChim[0].t.Str == 0 ? chimJ1 - chimJ0 + 1ull : chimJ0 - chimJ1 + 1ull) > (chimMotif >= 0 ? P.dev.alignIntronMax : P.dev.alignMatesGapMax)) {
        if (chimMotif >= 0 && (x0.L < C.junctionOverhangMin + chimRepeat0 || x1.L < C.junctionOverhangMin + chimRepeat1)) return false;
        if (bamOut && !merged) {                                     // chimericDetectionOldOutput :11-16: both segments re-scored, one chimera, the best by definition
            chimAlignScore(P.dev, gi, Read1, Lread, trChim[0]); chimAlignScore(P.dev, gi, Read1, Lread, trChim[1]);
            bamOut->push_back(ChimPair{trChim[0], trChim[1], true, varOrig[0], varOrig[1]});
        }
        if (bamOut && merged) {                                      // cut back into the pair first, then re-scored on the pair
            ChimTr tmp[2];
            chimericMergedToPair(*nameBatch, nameIr, mateStart, Lread, trChim[0], trChim[1], tmp);
            const uint64_t LreadPE = nameBatch->readOffset[nameIr + 1] - nameBatch->readOffset[nameIr];
            const uint8_t *Read1PE = nameBatch->bases.data() + nameBatch->readOffset[nameIr];
            chimAlignScore(P.dev, gi, Read1PE, LreadPE, tmp[0]); chimAlignScore(P.dev, gi, Read1PE, LreadPE, tmp[1]);
            bamOut->push_back(ChimPair{tmp[0], tmp[1], true, VarOverlap(), VarOverlap()});
        }
        if (!C.outJunctions) return true;
        // Chimeric.out.junction (chimericDetectionOldOutput :61-71)
        // the CIGARp is written against the lengths before clipping (ReadAlign_outputTranscriptCIGARp.cpp:13,25,55)
        const uint64_t readLengthOriginal[2] = {b.seqSpan[0][ir].len, nMates == 2 ? (uint64_t)b.seqSpan[1][ir].len : 0};
        const uint64_t readLengthPair = nMates == 2 ? readLengthOriginal[0] + readLengthOriginal[1] + 1 : readLengthOriginal[0];
        const uint64_t c0 = gi.chrStart[trChim[0].t.Chr], c1 = gi.chrStart[trChim[1].t.Chr];
        out += gi.chrName[trChim[0].t.Chr]; out.push_back('\t'); appendU(out, chimJ0 - c0 + 1); out.push_back('\t'); out.push_back(trChim[0].t.Str == 0 ? '+' : '-'); out.push_back('\t');
        out += gi.chrName[trChim[1].t.Chr]; out.push_back('\t'); appendU(out, chimJ1 - c1 + 1); out.push_back('\t'); out.push_back(trChim[1].t.Str == 0 ? '+' : '-'); out.push_back('\t');
        if (chimMotif < 0) { out.push_back('-'); appendU(out, (uint64_t)(-chimMotif)); } else appendU(out, (uint64_t)chimMotif);
        out.push_back('\t'); appendU(out, chimRepeat0); out.push_back('\t'); appendU(out, chimRepeat1); out.push_back('\t'); out += b.name(ir);
        out.push_back('\t'); appendU(out, trChim[0].ex[0].G - c0 + 1); out.push_back('\t'); out += cigarP(trChim[0], readLengthOriginal, readLengthPair, nMates);
        out.push_back('\t'); appendU(out, trChim[1].ex[0].G - c1 + 1); out.push_back('\t'); out += cigarP(trChim[1], readLengthOriginal, readLengthPair, nMates);
        if (std::find(P.outSAMattrOrder.begin(), P.outSAMattrOrder.end(), "RG") != P.outSAMattrOrder.end()) { out.push_back('\t'); out += P.outSAMattrRG.at(b.fileOf(ir)); }   // outSAMattrPresent.RG (:68)
        out.push_back('\n');
        return true;
    }
    return false;
}

// ---- the multimapping chimeric detection (--chimMultimapNmax > 0) ----
//   ChimericDetection::chimericDetectionMult   source/ChimericDetection_chimericDetectionMult.cpp:8-138
//   ChimericSegment                            source/ChimericSegment.cpp:3-31
//   ChimericAlign, chimericCheck               source/ChimericAlign.cpp:3-32
//   ChimericAlign::chimericStitching           source/ChimericAlign_chimericStitching.cpp:3-181
//   Transcript::alignScore                     source/Transcript_alignScore.cpp:4-58
//   ChimericAlign::chimericJunctionOutput      source/ChimericAlign_chimericJunctionOutput.cpp:4-23
// Every pair of recorded alignments of the read (all windows) is a candidate; pairs within --chimMultimapScoreRange of the best are all reported.
namespace {
struct Segment { const staramd_transcript *t; const staramd_exon *ex; uint64_t roS, roE; uint32_t str; bool good; };

struct ChimAlign { ChimTr a1, a2; uint64_t chimJ1, chimJ2, chimRepeat1, chimRepeat2; int chimMotif, chimScore; VarOverlap var1, var2; };

} // namespace

bool chimericDetectionMult(const RunParams &P, const GenomeIndex &gi, const ReadBatch &b, uint32_t ir, const ReadAligns &ra, const staramd_transcript *trBest, std::string &out,
                           std::vector<ChimPair> *bamOut, const ReadBatch *nameBatch, uint32_t nameIr, const uint32_t *mateStart) {
    const ChimParams &C = P.chim;
    const struct { uint32_t nTr; } rr = {ra.nTr};
    const staramd_transcript *T = ra.T;
    const struct { const staramd_exon *ex; } r = {ra.ex};
    const uint64_t Lread = b.readOffset[ir + 1] - b.readOffset[ir];
    const bool merged = nameBatch != nullptr;                 // b holds merged mates: a single-end read as far as the detection goes
    const int nMates = merged ? 1 : (int)P.dev.readNmates;
    const ReadBatch &nb = merged ? *nameBatch : b; const uint32_t nir = merged ? nameIr : ir;
    const uint64_t readLength[2] = {b.mate1Length[ir], nMates == 2 ? Lread - b.mate1Length[ir] - 1 : 0};
    const uint8_t *Read1 = b.bases.data() + b.readOffset[ir];
    const int64_t nG = (int64_t)gi.G.size();
    auto G = [&](uint64_t p) -> uint8_t { int64_t q = (int64_t)p; return q >= 0 && q < nG ? gi.G[(size_t)q] : 5; };
    const int maxNonChimAlignScore = trBest->maxScore;

    // ChimericSegment of every recorded alignment, in window order (trAll[iW][iA])
    std::vector<Segment> seg(rr.nTr);
    for (uint32_t k = 0; k < rr.nTr; k++) {
        const staramd_transcript &t = T[k]; const staramd_exon *ex = r.ex + t.exonOffset; const uint32_t ne = t.nExons;
        Segment &s = seg[k]; s.t = &t; s.ex = ex;
        if ((t.intronMotifs[1] == 0 && t.intronMotifs[2] == 0) || (t.intronMotifs[1] > 0 && t.intronMotifs[2] > 0)) s.str = 0;
        else if ((t.Str == 0) == (t.intronMotifs[1] > 0)) s.str = 1;
        else s.str = 2;
        s.roS = t.Str == 0 ? ex[0].R : Lread - ex[ne - 1].R - ex[ne - 1].L;
        s.roE = t.Str == 0 ? (uint64_t)ex[ne - 1].R + ex[ne - 1].L - 1 : Lread - ex[0].R - 1;
        if (s.roS > readLength[0]) s.roS--;
        if (s.roE > readLength[0]) s.roE--;
        s.good = t.rLength >= C.segmentMin && t.intronMotifs[0] == 0;
    }
    std::vector<ChimAlign> chimAligns;
    int chimScoreBest = 0; size_t bestChimAlign = 0;
    const int maxPossibleAlignScore = (int)(readLength[0] + readLength[1]);
    int minScoreToConsider = C.scoreMin;
    if (maxNonChimAlignScore >= minScoreToConsider) minScoreToConsider = maxNonChimAlignScore + 1;
    if (maxPossibleAlignScore - C.scoreDropMax > minScoreToConsider) minScoreToConsider = maxPossibleAlignScore - C.scoreDropMax;

    for (uint32_t k1 = 0; k1 < rr.nTr; k1++) {
        const Segment &s1 = seg[k1];
        if (!s1.good) continue;
        for (uint32_t k2 = k1 + 1; k2 < rr.nTr; k2++) {       // later alignments of the same window, then every alignment of the later windows
            const Segment &s2 = seg[k2];
            if (!s2.good) continue;
            if (s1.str != 0 && s2.str != 0 && s2.str != s1.str) continue;
            int chimScore = 0;
            {   // chimericAlignScore (:6-22)
                uint64_t chimOverlap = s2.roS > s1.roS ? (s2.roS > s1.roE ? 0 : s1.roE - s2.roS + 1) : (s2.roE < s1.roS ? 0 : s2.roE - s1.roS + 1);
                bool diffMates = (s1.roE < readLength[0] && s2.roS >= readLength[0]) || (s2.roE < readLength[0] && s1.roS >= readLength[0]);
                if (s1.roE > C.segmentMin + s1.roS + chimOverlap && s2.roE > C.segmentMin + s2.roS + chimOverlap
                    && (diffMates || ((s1.roE + C.segmentReadGapMax + 1) >= s2.roS && (s2.roE + C.segmentReadGapMax + 1) >= s1.roS)))
                    chimScore = s1.t->maxScore + s2.t->maxScore - (int)chimOverlap;
            }
            if (chimScore < minScoreToConsider) continue;
            const Segment *p1 = &s1, *p2 = &s2;
            if (p1->t->roStart > p2->t->roStart) std::swap(p1, p2);
            const uint32_t ex1 = p1->t->Str == 1 ? 0 : p1->t->nExons - 1, ex2 = p2->t->Str == 0 ? 0 : p2->t->nExons - 1;
            // chimericCheck
            if (!(p1->ex[ex1].iFrag <= p2->ex[ex2].iFrag)) continue;
            if (!(p1->ex[ex1].iFrag < p2->ex[ex2].iFrag || (p1->ex[ex1].L >= C.junctionOverhangMin && p2->ex[ex2].L >= C.junctionOverhangMin))) continue;
            // chimericStitching
            ChimAlign ca;
            load(ca.a1, *p1->t, p1->ex); load(ca.a2, *p2->t, p2->ex);
            if (bamOut && P.var && !merged) { P.var->overlap(ca.a1.t, ca.a1.ex, Read1, Lread, gi.chrStart[ca.a1.t.Chr], ca.var1); P.var->overlap(ca.a2.t, ca.a2.ex, Read1, Lread, gi.chrStart[ca.a2.t.Chr], ca.var2); }
            const uint32_t chimStr = std::max(s1.str, s2.str);
            ca.chimRepeat1 = ca.chimRepeat2 = ca.chimJ1 = ca.chimJ2 = 0; ca.chimMotif = 0; ca.chimScore = chimScore;
            staramd_exon &x1 = ca.a1.ex[ex1], &x2 = ca.a2.ex[ex2];
            bool alive = true;
            if (x1.iFrag < x2.iFrag) {
                ca.chimMotif = -1;
                ca.chimJ1 = ca.a1.t.Str == 1 ? x1.G - 1 : x1.G + x1.L;
                ca.chimJ2 = ca.a2.t.Str == 0 ? x2.G - 1 : x2.G + x2.L;
            } else {
                JunctionScan js;
                if (!scanJunction(C, G, Read1, Lread, readLength[0], ca.a1, ca.a2, ex1, ex2, chimStr, js)) { ca.chimScore = 0; alive = false; }
                else { ca.chimMotif = js.motif; shiftToJunction(G, ca.a1, ca.a2, ex1, ex2, js, ca.chimJ1, ca.chimJ2, ca.chimRepeat1, ca.chimRepeat2); }
            }
            if (alive) {
                if (ca.chimMotif >= 0 && (x1.L < C.junctionOverhangMin || x2.L < C.junctionOverhangMin)) ca.chimScore = 0;   // a linear junction too close to the chimeric one
                else ca.chimScore = chimAlignScore(P.dev, gi, Read1, Lread, ca.a1) + chimAlignScore(P.dev, gi, Read1, Lread, ca.a2) + (ca.chimMotif == 0 ? C.scoreJunctionNonGTAG : 0);
            }
            if (ca.chimScore >= minScoreToConsider) {
                chimAligns.push_back(ca);
                if (ca.chimScore > chimScoreBest) {
                    chimScoreBest = ca.chimScore; bestChimAlign = chimAligns.size() - 1;
                    if (chimScoreBest - (int)C.multimapScoreRange > minScoreToConsider) minScoreToConsider = chimScoreBest - (int)C.multimapScoreRange;
                }
            }
        }
    }
    if (chimScoreBest == 0) return false;
    uint64_t chimN = 0;
    for (const ChimAlign &ca : chimAligns) if (ca.chimScore >= minScoreToConsider) ++chimN;
    if (chimN > C.multimapNmax) return false;
    const uint64_t readLengthOriginal[2] = {merged ? Lread : (uint64_t)b.seqSpan[0][ir].len, nMates == 2 ? (uint64_t)b.seqSpan[1][ir].len : 0};
    const uint64_t readLengthPair = nMates == 2 ? readLengthOriginal[0] + readLengthOriginal[1] + 1 : readLengthOriginal[0];
    const bool rgColumn = std::find(P.outSAMattrOrder.begin(), P.outSAMattrOrder.end(), "RG") != P.outSAMattrOrder.end();
    auto appendI = [&](int v) { if (v < 0) { out.push_back('-'); appendU(out, (uint64_t)(-(int64_t)v)); } else appendU(out, (uint64_t)v); };
    for (size_t i = 0; i < chimAligns.size(); i++) {
        const ChimAlign &ca = chimAligns[i];
        if (ca.chimScore < minScoreToConsider) continue;
        if (bamOut && !merged) bamOut->push_back(ChimPair{ca.a1, ca.a2, i == bestChimAlign, ca.var1, ca.var2});
        if (bamOut && merged) {
            ChimTr tmp[2];
            chimericMergedToPair(nb, nir, mateStart, Lread, ca.a1, ca.a2, tmp);
            bamOut->push_back(ChimPair{tmp[0], tmp[1], i == bestChimAlign, VarOverlap(), VarOverlap()});
        }
        if (!C.outJunctions) continue;
        const uint64_t c1 = gi.chrStart[ca.a1.t.Chr], c2 = gi.chrStart[ca.a2.t.Chr];
        out += gi.chrName[ca.a1.t.Chr]; out.push_back('\t'); appendU(out, ca.chimJ1 - c1 + 1); out.push_back('\t'); out.push_back(ca.a1.t.Str == 0 ? '+' : '-'); out.push_back('\t');
        out += gi.chrName[ca.a2.t.Chr]; out.push_back('\t'); appendU(out, ca.chimJ2 - c2 + 1); out.push_back('\t'); out.push_back(ca.a2.t.Str == 0 ? '+' : '-'); out.push_back('\t');
        appendI(ca.chimMotif); out.push_back('\t'); appendU(out, ca.chimRepeat1); out.push_back('\t'); appendU(out, ca.chimRepeat2); out.push_back('\t'); out += nb.name(nir);
        out.push_back('\t'); appendU(out, ca.a1.ex[0].G - c1 + 1); out.push_back('\t'); out += cigarP(ca.a1, readLengthOriginal, readLengthPair, nMates);
        out.push_back('\t'); appendU(out, ca.a2.ex[0].G - c2 + 1); out.push_back('\t'); out += cigarP(ca.a2, readLengthOriginal, readLengthPair, nMates);
        out.push_back('\t'); appendU(out, chimN); out.push_back('\t'); appendI(maxPossibleAlignScore); out.push_back('\t'); appendI(maxNonChimAlignScore);
        out.push_back('\t'); appendI(ca.chimScore); out.push_back('\t'); appendI(chimScoreBest); out += merged ? "\t1" : "\t0";       // PEmerged_bool
        if (rgColumn) { out.push_back('\t'); out += P.outSAMattrRG.at(nb.fileOf(nir)); }
        out.push_back('\n');
    }
    return chimN > 0;
}

} // namespace staramd
