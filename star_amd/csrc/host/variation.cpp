// variation.cpp -- --varVCFfile (SNVs of the sample: vA / vG attributes) and --waspOutputMode SAMtag (WASP re-mapping filter, vW attribute).
//   Variation::loadVCF, scanVCF          source/Variation.cpp:22-125
//   Transcript::variationAdjust          source/Transcript_variationAdjust.cpp:4-74   (the score is not changed: VAR_noScoreCorrection)
//   ReadAlign::waspMap                   source/ReadAlign_waspMap.cpp:3-112
// Nothing of this touches the hot path: the SNVs under an alignment are looked up after mapping, and the WASP filter maps, as ONE MORE BATCH through the same
// engine, every other combination of alleles of a uniquely mapped read and checks that each lands on the same blocks.
#include "host.h"
#include <fstream>
#include <sstream>
#include <algorithm>
#include <cstring>

namespace staramd {

static uint8_t nt01234(char c) { switch (c) { case 'a': case 'A': return 0; case 'c': case 'C': return 1; case 'g': case 'G': return 2; case 't': case 'T': return 3; default: return 4; } }

std::string Variation::load(const RunParams &P, const GenomeIndex &gi) {
    std::ifstream vcf(P.varVCFfile.c_str());
    if (!vcf.good()) return "EXITING because of fatal INPUT error: could not open input file " + P.varVCFfile + "\nSOLUTION: check the path and permissions of the VCF file: " + P.varVCFfile + "\n";
    std::vector<std::pair<uint64_t, std::array<uint8_t, 3> > > snps;
    std::string line;
    while (std::getline(vcf, line)) {
        std::istringstream ls(line);
        std::string chr, id, ref, alt, dummy, sample; uint64_t pos = 0;
        ls >> chr;
        if (chr.empty() || chr[0] == '#') continue;
        ls >> pos >> id >> ref >> alt >> dummy >> dummy >> dummy >> dummy >> sample;
        std::vector<std::string> altV; size_t maxL = 0;
        { std::stringstream ss(alt); std::string item; while (std::getline(ss, item, ',')) { maxL = std::max(maxL, item.size()); altV.push_back(item); } }
        if (!(ref.size() == 1 && maxL == 1)) continue;                     // SNVs only
        altV.insert(altV.begin(), ref);
        uint32_t ic = 0; for (; ic < gi.view.nChrReal; ic++) if (gi.chrName[ic] == chr) break;
        if (ic == gi.view.nChrReal) continue;                                // chromosome not in the genome (a WARNING in the reference's log)
        if (sample.size() < 3) continue;
        if (sample.size() > 3 && sample[3] != ':') continue;                 // more than 2 alleles
        if (sample[0] == '0' && sample[2] == '0') continue;
        const size_t a0 = (size_t)atoi(&sample[0]), a2 = (size_t)atoi(&sample[2]);
        if (a0 >= altV.size() || a2 >= altV.size()) return "EXITING because of fatal INPUT error: genotype refers to an allele that is not listed, VCF line: " + line;
        if (altV[a0][0] == ref[0] && altV[a2][0] == ref[0]) continue;
        if (P.varHeteroOnly && sample[0] == sample[2]) continue;             // homozygous: not used by WASP
        std::array<uint8_t, 3> nt1 = {nt01234(ref[0]), nt01234(altV[a0][0]), nt01234(altV[a2][0])};
        if (nt1[0] < 4 && nt1[1] < 4 && nt1[2] < 4) snps.emplace_back(pos - 1 + gi.chrStart[ic], nt1);
    }
    if (snps.empty()) return "EXITING because of FATAL INPUT FILE ERROR: could not find any SNPs in VCF file: " + P.varVCFfile + "\nSOLUTION: check formatting of the VCF file; unzip VCF file or use process substitution.\n";
    std::stable_sort(snps.begin(), snps.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    loci.resize(snps.size()); nt.resize(snps.size());
    for (size_t i = 0; i < snps.size(); i++) { loci[i] = snps[i].first; nt[i] = snps[i].second; }
    return "";
}

// SNVs under the blocks of an alignment; Read1 = the read as mapped (forward), turned into the alignment's orientation here
void Variation::overlap(const staramd_transcript &t, const staramd_exon *ex, const uint8_t *Read1, uint64_t Lread, uint64_t chrStart, VarOverlap &o) const {
    o.ind.clear(); o.genCoord.clear(); o.readCoord.clear(); o.allele.clear();
    const int64_t N = (int64_t)loci.size();
    if (N == 0) return;
    for (uint32_t ie = 0; ie < t.nExons; ie++) {
        const uint64_t x = ex[ie].G;
        int64_t isnp;                                                       // binarySearch1b: the first locus >= x, -1 if there is none
        if (x > loci[N - 1]) isnp = -1; else if (x <= loci[0]) isnp = 0;
        else { int64_t i1 = 0, i2 = N - 1; while (i2 > i1 + 1) { int64_t i3 = (i1 + i2) / 2; if (loci[i3] >= x) i2 = i3; else i1 = i3; } isnp = i2; }
        if (isnp < 0) continue;
        while (isnp < N && ex[ie].G + ex[ie].L > loci[isnp]) {
            o.ind.push_back((uint32_t)isnp);
            o.genCoord.push_back((int32_t)(loci[isnp] - chrStart));
            const uint64_t rp = ex[ie].R + loci[isnp] - ex[ie].G;
            o.readCoord.push_back((uint32_t)rp);
            uint8_t ntR = t.roStr == 0 ? Read1[rp] : Read1[Lread - 1 - rp];
            if (t.roStr != 0 && ntR < 4) ntR = 3 - ntR;
            uint8_t igt;
            if (ntR > 3) igt = 4; else { for (igt = 1; igt < 3; igt++) if (nt[isnp][igt] == ntR) break; }
            o.allele.push_back((char)igt);
            ++isnp;
        }
    }
}

// ---- WASP ----
// the re-mapping batch of one batch of reads: for every read whose best alignment is unique and covers 1..10 SNVs with known alleles, the read with every OTHER
// combination of the two alleles at those SNVs.  type[] gets the verdicts that need no mapping (-1 no variants, 2 multimapper, 7 too many variants, 3 N at a variant)
void WaspBatch::build(const RunParams &P, const GenomeIndex &gi, const Variation &var, const ReadBatch &b, const staramd_results &r) {
    const uint32_t n = b.n;
    reads.clear(); first.assign(n, 0); count.assign(n, 0); type.assign(n, -1);
    VarOverlap vo; std::vector<uint8_t> rd;
    for (uint32_t ir = 0; ir < n; ir++) {
        const staramd_read_result &rr = r.reads[ir];
        if (rr.nW == 0 || rr.trBest < 0) continue;
        const staramd_transcript *T = r.tr + rr.trOffset, &tb = T[rr.trBest];
        const uint8_t *Read1 = b.bases.data() + b.readOffset[ir];
        const uint64_t Lread = b.readOffset[ir + 1] - b.readOffset[ir];
        var.overlap(tb, r.ex + tb.exonOffset, Read1, Lread, gi.chrStart[tb.Chr], vo);
        if (vo.allele.empty()) continue;
        uint64_t nTr = 0;
        for (uint32_t k = 0; k < rr.nTr; k++) if (T[k].maxScore + P.dev.outFilterMultimapScoreRange >= tb.maxScore) nTr++;
        if (nTr > P.outFilterMultimapNmax) nTr = 0;                         // multMapSelect leaves nTr above the limit; it is > 1 either way
        if (nTr != 1) { type[ir] = 2; if (nTr == 0) type[ir] = 2; continue; }
        if (vo.allele.size() > 10) { type[ir] = 7; continue; }
        bool hasN = false; for (char a : vo.allele) if (a > 3) hasN = true;
        if (hasN) { type[ir] = 3; continue; }
        type[ir] = 0;                                                       // decided after the re-mapping
        first[ir] = reads.n;
        const size_t nv = vo.allele.size();
        rd.assign(Read1, Read1 + Lread);
        for (uint32_t combo = 0; combo < (1u << nv); combo++) {             // order of the reference's nested loops: the first variant is the most significant bit
            bool same = true;
            for (size_t iv = 0; iv < nv; iv++) { const char a = (char)(1 + ((combo >> (nv - 1 - iv)) & 1)); if (a != vo.allele[iv]) same = false; }
            if (same) continue;
            for (size_t iv = 0; iv < nv; iv++) {
                const uint32_t a = 1 + ((combo >> (nv - 1 - iv)) & 1);
                uint8_t nt2 = var.nt[vo.ind[iv]][a]; uint64_t vr = vo.readCoord[iv];
                if (tb.Str == 1) { nt2 = 3 - nt2; vr = Lread - 1 - vr; }
                rd[vr] = nt2;
            }
            reads.bases.insert(reads.bases.end(), rd.begin(), rd.end());
            reads.readOffset.push_back(reads.readOffset.back() + Lread);
            reads.mate1Length.push_back(b.mate1Length[ir]); reads.mmMaxTotal.push_back(b.mmMaxTotal[ir]);
            reads.n++; count[ir]++;
        }
    }
}

// verdicts after the re-mapping: 4 a combination does not map, 5 maps to several places, 6 maps differently, 1 all combinations map like the read itself
void WaspBatch::finish(const RunParams &P, const ReadBatch &b, const staramd_results &r, const staramd_results &rw) {
    for (uint32_t ir = 0; ir < b.n; ir++) {
        if (type[ir] != 0) continue;
        const staramd_read_result &rr = r.reads[ir];
        const staramd_transcript &t1 = (r.tr + rr.trOffset)[rr.trBest]; const staramd_exon *e1 = r.ex + t1.exonOffset;
        const uint64_t Lread = b.readOffset[ir + 1] - b.readOffset[ir];
        int8_t verdict = 1;
        for (uint32_t k = first[ir]; k < first[ir] + count[ir] && verdict == 1; k++) {
            const staramd_read_result &w = rw.reads[k];
            const staramd_transcript *T = rw.tr + w.trOffset;
            if (w.nW == 0 || w.trBest < 0) { verdict = 4; break; }
            const staramd_transcript &t2 = T[w.trBest];
            uint64_t nTr2 = 0;
            for (uint32_t q = 0; q < w.nTr; q++) if (T[q].maxScore + P.dev.outFilterMultimapScoreRange >= t2.maxScore) nTr2++;
            // mappedFilter (ReadAlign_mappedFilter.cpp:4-22) on the re-mapped read
            bool unmapped = (t2.maxScore < P.outFilterScoreMin) || (t2.maxScore < (int)(P.outFilterScoreMinOverLread * (double)(Lread - 1)))
                            || (t2.nMatch < P.outFilterMatchNmin) || (t2.nMatch < (uint64_t)(P.outFilterMatchNminOverLread * (double)(Lread - 1)))
                            || (t2.nMM > b.mmMaxTotal[ir]) || (double(t2.nMM) / double(t2.rLength) > P.dev.outFilterMismatchNoverLmax) || nTr2 > P.outFilterMultimapNmax;
            if (unmapped) { verdict = 4; break; }
            if (nTr2 > 1) { verdict = 5; break; }
            if (t2.nExons != t1.nExons) { verdict = 6; break; }
            const staramd_exon *e2 = rw.ex + t2.exonOffset;
            for (uint32_t ii = 0; ii < t1.nExons; ii++) if (e1[ii].R != e2[ii].R || e1[ii].G != e2[ii].G || e1[ii].L != e2[ii].L) { verdict = 6; break; }
        }
        type[ir] = verdict;
    }
}

} // namespace staramd
