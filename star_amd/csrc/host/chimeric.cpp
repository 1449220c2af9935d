// chimeric.cpp -- chimeric alignment detection on the transcripts of all windows (--chimSegmentMin > 0, --chimMultimapNmax 0,
// --chimOutType Junctions): the best alignment plus the best-scoring alignment of another window that covers the rest of the read.
//   ReadAlign::chimericDetection            source/ReadAlign_chimericDetection.cpp:16-57
//   ReadAlign::chimericDetectionOld         source/ReadAlign_chimericDetectionOld.cpp:7-312
//   ReadAlign::chimericDetectionOldOutput   source/ReadAlign_chimericDetectionOldOutput.cpp:5-74  (Chimeric.out.junction line)
//   ReadAlign::outputTranscriptCIGARp       source/ReadAlign_outputTranscriptCIGARp.cpp:4-68
//   blocksOverlap                           source/blocksOverlap.cpp:3-41
// Host post-map code.  The device returns every recorded transcript of every window for it (resultSelect 0,
// chimSegmentMinPositive 1: stitchWindowAligns.cpp:247).
#include "host.h"
#include <algorithm>
#include <cstring>

namespace staramd {

namespace {
struct ChimTr { staramd_transcript t; staramd_exon ex[STARAMD_MAX_N_EXONS]; };

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
} // namespace

// returns true when a chimeric alignment was recorded (Stats::chimericAll); the junction line is appended to `out`
bool chimericDetectionOld(const RunParams &P, const GenomeIndex &gi, const ReadBatch &b, uint32_t ir, const staramd_results &r,
                          const staramd_transcript *trBest, uint64_t nTr, const staramd_transcript *trMult0, const staramd_transcript *trMult1, std::string &out) {
    const ChimParams &C = P.chim;
    const staramd_read_result &rr = r.reads[ir];
    const staramd_transcript *T = r.tr + rr.trOffset;
    const uint64_t Lread = b.readOffset[ir + 1] - b.readOffset[ir];
    const int nMates = (int)P.dev.readNmates;
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
        const uint64_t roStart0 = trChim[0].t.Str == 0 ? x0.R : Lread - x0.R - x0.L;
        const uint64_t roStartB = trChim[1].t.Str == 0 ? x1.R : Lread - x1.R - x1.L;
        uint64_t jR, jRbest = 0; int jScore = 0, jMotif = 0, jScoreBest = -999999, jScoreJ = 0;
        uint64_t jRmax = roStartB + x1.L;
        jRmax = jRmax > roStart0 ? jRmax - roStart0 - 1 : 0;
        bool rejected = false;
        for (jR = 0; jR < jRmax; jR++) {
            if (jR == readLength[0]) jR++;
            uint8_t bR = Read1[roStart0 + jR];
            uint8_t b0, b1;
            if (trChim[0].t.Str == 0) b0 = G(x0.G + jR); else { b0 = G(x0.G + x0.L - 1 - jR); if (b0 < 4) b0 = 3 - b0; }
            if (trChim[1].t.Str == 0) b1 = G(x1.G - roStartB + roStart0 + jR); else { b1 = G(x1.G + x1.L - 1 + roStartB - roStart0 - jR); if (b1 < 4) b1 = 3 - b1; }
            if ((C.filterGenomicN && (b0 > 3 || b1 > 3)) || bR > 3) { rejected = true; break; }
            uint8_t b01, b02, b11, b12;
            if (trChim[0].t.Str == 0) { b01 = G(x0.G + jR + 1); b02 = G(x0.G + jR + 2); }
            else { b01 = G(x0.G + x0.L - 1 - jR - 1); if (b01 < 4) b01 = 3 - b01; b02 = G(x0.G + x0.L - 1 - jR - 2); if (b02 < 4) b02 = 3 - b02; }
            if (trChim[1].t.Str == 0) { b11 = G(x1.G - roStartB + roStart0 + jR - 1); b12 = G(x1.G - roStartB + roStart0 + jR); }
            else { b11 = G(x1.G + x1.L - 1 + roStartB - roStart0 - jR + 1); if (b11 < 4) b11 = 3 - b11; b12 = G(x1.G + x1.L - 1 + roStartB - roStart0 - jR); if (b12 < 4) b12 = 3 - b12; }
            jMotif = 0;
            if (b01 == 2 && b02 == 3 && b11 == 0 && b12 == 2) { if (chimStr != 2) jMotif = 1; }
            else if (b01 == 1 && b02 == 3 && b11 == 0 && b12 == 1) { if (chimStr != 1) jMotif = 2; }
            if (bR == b0 && bR != b1) jScore++; else if (bR != b0 && bR == b1) jScore--;
            jScoreJ = jMotif == 0 ? jScore + C.scoreJunctionNonGTAG : jScore;
            if (jScoreJ > jScoreBest || (jScoreJ == jScoreBest && jMotif > 0)) { chimMotif = jMotif; jRbest = jR; jScoreBest = jScoreJ; }
        }
        if (rejected) return false;
        if (chimMotif == 0) {
            chimScoreBest += 1 + C.scoreJunctionNonGTAG;
            if (!(chimScoreBest >= C.scoreMin && chimScoreBest + C.scoreDropMax >= readL)) return false;
        }
        if (trChim[0].t.Str == 1) { x0.R = (uint16_t)(x0.R + x0.L - jRbest - 1); x0.G += x0.L - jRbest - 1; x0.L = (uint16_t)(jRbest + 1); chimJ0 = x0.G - 1; }
        else { x0.L = (uint16_t)(jRbest + 1); chimJ0 = x0.G + x0.L; }
        if (trChim[1].t.Str == 0) {
            x1.R = (uint16_t)(x1.R + roStart0 + jRbest + 1 - roStartB); x1.G += roStart0 + jRbest + 1 - roStartB;
            x1.L = (uint16_t)(roStartB + x1.L - roStart0 - jRbest - 1); chimJ1 = x1.G - 1;
        } else { x1.L = (uint16_t)(roStartB + x1.L - roStart0 - jRbest - 1); chimJ1 = x1.G + x1.L; }
        uint8_t b0, b1;
        for (jR = 0; jR < 100; jR++) {
            if (trChim[0].t.Str == 0) b0 = G(chimJ0 + jR); else { b0 = G(chimJ0 - jR); if (b0 < 4) b0 = 3 - b0; }
            if (trChim[1].t.Str == 0) b1 = G(chimJ1 + 1 + jR); else { b1 = G(chimJ1 - 1 - jR); if (b1 < 4) b1 = 3 - b1; }
            if (b0 != b1) break;
        }
        chimRepeat1 = jR;
        for (jR = 0; jR < 100; jR++) {
            if (trChim[0].t.Str == 0) b0 = G(chimJ0 - 1 - jR); else { b0 = G(chimJ0 + 1 + jR); if (b0 < 4) b0 = 3 - b0; }
            if (trChim[1].t.Str == 0) b1 = G(chimJ1 - jR); else { b1 = G(chimJ1 + jR); if (b1 < 4) b1 = 3 - b1; }
            if (b0 != b1) break;
        }
        chimRepeat0 = jR;
    }
    // final check (:296-309): different chromosome / strand, or far apart
    if (trChim[0].t.Str != trChim[1].t.Str || trChim[0].t.Chr != trChim[1].t.Chr
        || (trChim[0].t.Str == 0 ? chimJ1 - chimJ0 + 1ull : chimJ0 - chimJ1 + 1ull) > (chimMotif >= 0 ? P.dev.alignIntronMax : P.dev.alignMatesGapMax)) {
        if (chimMotif >= 0 && (x0.L < C.junctionOverhangMin + chimRepeat0 || x1.L < C.junctionOverhangMin + chimRepeat1)) return false;
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
        if (!P.outSAMattrRG.empty()) { out.push_back('\t'); out += P.outSAMattrRG.at(b.fileIndex); }
        out.push_back('\n');
        return true;
    }
    return false;
}

} // namespace staramd
