// quant.cpp -- --quantMode GeneCounts: reads per gene, three strandedness columns (ReadsPerGene.out.tab).
//   Transcriptome::Transcriptome (geneInfo.tab, exonGeTrInfo.tab)   source/Transcriptome.cpp:7-35,86-106
//   Transcriptome::geneCountsAddAlign                               source/Transcriptome_geneCountsAddAlign.cpp:4-63
//   Transcriptome::quantsOutput                                     source/Transcriptome.cpp:158-190
//   Quantifications                                                 source/Quantifications.cpp:3-36
// Host post-map code: it consumes the unique alignment multMapSelect picked, nothing of it runs on the device.
#include "host.h"
#include <fstream>
#include <sstream>
#include <cstring>
#include <algorithm>

namespace staramd {

std::string GeneAnnotation::load(const std::string &dir) {
    {
        std::ifstream ge(dir + "/geneInfo.tab");
        if (!ge.good()) return "EXITING because of fatal INPUT error: could not open input file " + dir + "/geneInfo.tab\nSOLUTION: utilize --sjdbGTFfile /path/to/annotations.gtf option at the genome generation step or mapping step\n";
        uint64_t n = 0; ge >> n;
        geID.resize(n);
        ge.ignore(999, '\n');
        for (uint64_t i = 0; i < n; i++) { std::string l; std::getline(ge, l); std::istringstream ls(l); ls >> geID[i]; }
    }
    std::ifstream ex(dir + "/exonGeTrInfo.tab");
    if (!ex.good()) return "EXITING because of fatal INPUT error: could not open input file " + dir + "/exonGeTrInfo.tab\nSOLUTION: utilize --sjdbGTFfile /path/to/annotantions.gtf option at the genome generation step or mapping step\n";
    uint64_t n = 0; ex >> n;
    s.resize(n); e.resize(n); eMax.resize(n); str.resize(n); g.resize(n);
    for (uint64_t i = 0; i < n; i++) { int st; uint64_t t; ex >> s[i] >> e[i] >> st >> g[i] >> t; str[i] = (uint8_t)st; }
    for (uint64_t i = 0; i < n; i++) eMax[i] = i == 0 ? e[0] : std::max(eMax[i - 1], e[i]);
    return "";
}

GeneCounts::GeneCounts(size_t nGe) { for (int t = 0; t < 3; t++) gCount[t].assign(nGe, 0); }

void GeneCounts::add(const GeneCounts &o) {
    cMulti += o.cMulti;
    for (int t = 0; t < 3; t++) { cAmbig[t] += o.cAmbig[t]; cNone[t] += o.cNone[t]; for (size_t i = 0; i < gCount[t].size(); i++) gCount[t][i] += o.gCount[t][i]; }
}

// binarySearch1a (serviceFuns.cpp:238-263): last element <= x, -1 if none
static int64_t lastNotAbove(uint64_t x, const std::vector<uint64_t> &X) {
    int64_t N = (int64_t)X.size();
    if (N == 0) return -1;
    if (x > X[N - 1]) return N - 1;
    if (x < X[0]) return -1;
    int64_t i1 = 0, i2 = N - 1;
    while (i2 > i1 + 1) { int64_t i3 = (i1 + i2) / 2; if (X[i3] > x) i2 = i3; else i1 = i3; }
    while (i1 < N - 1 && x == X[i1 + 1]) ++i1;
    return i1;
}

void GeneCounts::addAlign(const GeneAnnotation &A, uint64_t nA, const staramd_transcript &a, const staramd_exon *ex) {
    if (nA > 1) { cMulti++; return; }
    int32_t gene1[3] = {-1, -1, -1};
    for (int ib = (int)a.nExons - 1; ib >= 0; ib--) {
        uint64_t g1 = ex[ib].G + ex[ib].L - 1;
        int64_t e1 = lastNotAbove(g1, A.s);
        while (e1 >= 0 && A.eMax[e1] >= ex[ib].G) {
            if (A.e[e1] >= ex[ib].G) {
                uint32_t str1 = (uint32_t)A.str[e1] - 1;
                for (int itype = 0; itype < 3; itype++) {
                    if (itype == 1 && a.Str != str1 && str1 < 2) continue;
                    if (itype == 2 && a.Str == str1 && str1 < 2) continue;
                    if (gene1[itype] == -1) gene1[itype] = (int32_t)A.g[e1];
                    else if (gene1[itype] == -2) continue;
                    else if (gene1[itype] != (int32_t)A.g[e1]) gene1[itype] = -2;
                }
            }
            --e1;
        }
    }
    for (int itype = 0; itype < 3; itype++) {
        if (gene1[itype] == -1) cNone[itype]++;
        else if (gene1[itype] == -2) cAmbig[itype]++;
        else gCount[itype][gene1[itype]]++;
    }
}

std::string GeneCounts::write(const std::string &path, const GeneAnnotation &A, const Stats &st) const {
    std::ofstream q(path.c_str());
    if (!q.good()) return "EXITING because of fatal ERROR: could not create output file " + path;
    q << "N_unmapped";
    for (int t = 0; t < 3; t++) q << "\t" << st.unmappedMismatch + st.unmappedShort + st.unmappedOther + st.unmappedMulti;
    q << "\nN_multimapping";
    for (int t = 0; t < 3; t++) q << "\t" << cMulti;
    q << "\nN_noFeature";
    for (int t = 0; t < 3; t++) q << "\t" << cNone[t];
    q << "\nN_ambiguous";
    for (int t = 0; t < 3; t++) q << "\t" << cAmbig[t];
    q << "\n";
    for (size_t ig = 0; ig < A.geID.size(); ig++) { q << A.geID[ig]; for (int t = 0; t < 3; t++) q << "\t" << gCount[t][ig]; q << "\n"; }
    return "";
}


// ---------------------------------------------------------------------------------------------------------------------------
// --quantMode TranscriptomeSAM: alignments projected onto the annotated transcripts (Aligned.toTranscriptome.out.bam)
//   Transcriptome::Transcriptome (transcriptInfo.tab, exonInfo.tab)   source/Transcriptome.cpp:36-84
//   Transcriptome::quantAlign, alignToTranscript                      source/Transcriptome_quantAlign.cpp:5-114
std::string TranscriptAnnotation::load(const std::string &dir) {
    std::ifstream tr(dir + "/transcriptInfo.tab");
    if (!tr.good()) return "EXITING because of fatal INPUT error: could not open input file " + dir + "/transcriptInfo.tab\nSOLUTION: utilize --sjdbGTFfile /path/to/annotantions.gtf option at the genome generation step or mapping step\n";
    uint64_t n = 0; tr >> n;
    trID.resize(n); trS.resize(n); trE.resize(n); trEmax.resize(n); trStr.resize(n); trExN.resize(n); trExI.resize(n); trLen.resize(n);
    for (uint64_t i = 0; i < n; i++) {
        uint32_t str1; uint64_t gene;
        tr >> trID[i] >> trS[i] >> trE[i] >> trEmax[i] >> str1 >> trExN[i] >> trExI[i] >> gene;
        trStr[i] = (uint8_t)str1;
        if (!tr.good()) return "EXITING because of FATAL GENOME INDEX FILE error: transcriptInfo.tab is corrupt, or is incompatible with the current STAR version\nSOLUTION: re-generate genome index";
    }
    std::ifstream ex(dir + "/exonInfo.tab");
    if (!ex.good()) return "EXITING because of fatal INPUT error: could not open input file " + dir + "/exonInfo.tab";
    uint64_t ne = 0; ex >> ne;
    exSE.resize(2 * ne); exLenCum.resize(ne);
    for (uint64_t i = 0; i < ne; i++) ex >> exSE[2 * i] >> exSE[2 * i + 1] >> exLenCum[i];
    for (uint64_t i = 0; i < n; i++) { uint32_t l = trExI[i] + trExN[i] - 1; trLen[i] = exLenCum[l] + exSE[2 * l + 1] - exSE[2 * l] + 1; }
    return "";
}

namespace {
// binarySearch1 (serviceFuns.cpp:191-209): last element <= x, (uint32)-1 outside the range
uint32_t bs1(uint32_t x, const uint32_t *X, uint32_t N) {
    if (x > X[N - 1] || x < X[0]) return (uint32_t)-1;
    uint32_t i1 = 0, i2 = N - 1;
    while (i2 > i1 + 1) { uint32_t i3 = (i1 + i2) / 2; if (X[i3] > x) i2 = i3; else i1 = i3; }
    while (i1 < N - 1 && x == X[i1 + 1]) ++i1;
    return i1;
}
// alignToTranscript (Transcriptome_quantAlign.cpp:5-88).  canonSJ of the last genomic block is read as "last" instead of being overwritten.
int alignToTranscript(const GenomicAlign &aG, uint64_t trS1, uint8_t trStr1, const uint32_t *exSE1, const uint32_t *exLenCum1, uint16_t exN1, ProjectedAlign &aT) {
    uint32_t g1 = (uint32_t)(aG.ex[0].G - trS1);
    uint32_t ex1 = bs1(g1, exSE1, 2u * exN1);
    if (ex1 >= 2u * exN1) return 0;
    if (ex1 % 2 == 1) { if (exSE1[ex1] == g1) --ex1; else return 0; }
    ex1 = ex1 / 2;
    aT.nExons = 0;
    for (uint32_t iab = 0; iab < aG.nExons; iab++) {
        if (aG.ex[iab].G + aG.ex[iab].L > (uint64_t)exSE1[2 * ex1 + 1] + trS1 + 1) return 0;
        if (iab == 0 || aG.ex[iab - 1].canonSJ < 0) {
            staramd_exon &e = aT.ex[aT.nExons];
            memset(&e, 0, sizeof(e));
            e.R = aG.ex[iab].R; e.G = aG.ex[iab].G - trS1 - exSE1[2 * ex1] + exLenCum1[ex1]; e.L = aG.ex[iab].L; e.iFrag = aG.ex[iab].iFrag; e.sjA = -1;
            if (aT.nExons > 0) aT.ex[aT.nExons - 1].canonSJ = aG.ex[iab - 1].canonSJ;
            ++aT.nExons;
        } else aT.ex[aT.nExons - 1].L = (uint16_t)(aT.ex[aT.nExons - 1].L + aG.ex[iab].L);
        const int sj = iab + 1 == aG.nExons ? -999 : aG.ex[iab].canonSJ;
        if (sj == -999) {
            if (trStr1 == 2) {
                uint32_t trlength = exLenCum1[exN1 - 1] + exSE1[2 * exN1 - 1] - exSE1[2 * exN1 - 2] + 1;
                for (uint32_t i = 0; i < aT.nExons; i++) { aT.ex[i].R = (uint16_t)(aG.Lread - (aT.ex[i].R + aT.ex[i].L)); aT.ex[i].G = trlength - (aT.ex[i].G + aT.ex[i].L); }
                for (uint32_t i = 0; i < aT.nExons / 2; i++) {
                    staramd_exon &a = aT.ex[i], &b = aT.ex[aT.nExons - 1 - i];
                    std::swap(a.R, b.R); std::swap(a.G, b.G); std::swap(a.L, b.L); std::swap(a.iFrag, b.iFrag);
                }
                for (uint32_t i = 0; i + 1 < aT.nExons && i < (aT.nExons - 1) / 2; i++) std::swap(aT.ex[i].canonSJ, aT.ex[aT.nExons - 2 - i].canonSJ);
            }
            for (uint32_t i = 0; i < aT.nExons; i++) { aT.ex[i].sjAnnot = 0; aT.ex[i].shiftSJ[0] = aT.ex[i].shiftSJ[1] = 0; aT.ex[i].sjStr = 0; }
            return 1;
        } else if (sj == -3) {
            ex1 = bs1((uint32_t)(aG.ex[iab + 1].G - trS1), exSE1, 2u * exN1);
            if (ex1 % 2 == 1) return 0;
            ex1 = ex1 / 2;
        } else if (sj == -2 || sj == -1) {
        } else {
            if (aG.ex[iab].G + aG.ex[iab].L == (uint64_t)exSE1[2 * ex1 + 1] + trS1 + 1 && aG.ex[iab + 1].G == (uint64_t)exSE1[2 * (ex1 + 1)] + trS1) ++ex1;
            else return 0;
        }
    }
    return 0;
}
} // namespace

// Transcriptome::quantAlign (:90-114): projections of one genomic alignment, appended to `out`
uint32_t TranscriptAnnotation::quantAlign(const GenomicAlign &aG, std::vector<ProjectedAlign> &out) const {
    const int64_t N = (int64_t)trS.size();
    if (N == 0) return 0;
    const uint64_t x = aG.ex[0].G;
    int64_t tr1;                                            // binarySearch1a: last transcript start <= alignment start
    if (x > trS[N - 1]) tr1 = N - 1;
    else if (x < trS[0]) return 0;
    else {
        int64_t i1 = 0, i2 = N - 1;
        while (i2 > i1 + 1) { int64_t i3 = (i1 + i2) / 2; if (trS[i3] > x) i2 = i3; else i1 = i3; }
        while (i1 < N - 1 && x == trS[i1 + 1]) ++i1;
        tr1 = i1;
    }
    const uint64_t aGend = aG.ex[aG.nExons - 1].G;
    uint32_t n = 0;
    ++tr1;
    do {
        --tr1;
        if (aGend <= trE[tr1]) {
            ProjectedAlign p;
            if (alignToTranscript(aG, trS[tr1], trStr[tr1], exSE.data() + 2 * trExI[tr1], exLenCum.data() + trExI[tr1], trExN[tr1], p) == 1) {
                p.tr = (uint32_t)tr1; p.Str = trStr[tr1] == 1 ? aG.Str : 1 - aG.Str;
                out.push_back(p); ++n;
            }
        }
    } while (trEmax[tr1] >= aGend && tr1 > 0);
    return n;
}

} // namespace staramd
