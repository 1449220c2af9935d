// quant.cpp -- --quantMode GeneCounts: reads per gene, three strandedness columns (ReadsPerGene.out.tab).
//   Transcriptome::Transcriptome (geneInfo.tab, exonGeTrInfo.tab)   source/Transcriptome.cpp:7-35,86-106
//   Transcriptome::geneCountsAddAlign                               source/Transcriptome_geneCountsAddAlign.cpp:4-63
//   Transcriptome::quantsOutput                                     source/Transcriptome.cpp:158-190
//   Quantifications                                                 source/Quantifications.cpp:3-36
// Host post-map code: it consumes the unique alignment multMapSelect picked, nothing of it runs on the device.
#include "host.h"
#include <fstream>
#include <sstream>

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

} // namespace staramd
