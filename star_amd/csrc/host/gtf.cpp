// gtf.cpp -- --sjdbGTFfile at the mapping stage: exon lines of a GTF -> annotated junctions (+ the transcript / exon / gene tables
// the reference writes next to them).  Restates
//   GTF::GTF                   source/GTF.cpp:7-171                 (exon lines, transcript / gene numbering)
//   GTF::transcriptGeneSJ      source/GTF_transcriptGeneSJ.cpp:23-183 (sorting, tables, junctions between consecutive exons)
// The junctions enter sjdbInsertJunctions with priority 20 (sjdb_insert.cpp).
#include "host.h"
#include <algorithm>
#include <array>
#include <fstream>
#include <map>
#include <set>
#include <sstream>

namespace staramd {

std::string loadGTFjunctions(const RunParams &P, const GenomeIndex &gi, SjdbLoci &loci, const std::string &dirOut, std::string &log) {
    if (P.sjdbGTFfile.empty()) return "";
    std::ifstream in(P.sjdbGTFfile.c_str());
    if (in.fail()) return "FATAL error, could not open file pGe.sjdbGTFfile=" + P.sjdbGTFfile + "\n";
    std::map<std::string, uint64_t> chrIndex, transcriptIDnumber, geneIDnumber;
    for (uint32_t i = 0; i < gi.view.nChrReal; i++) chrIndex[gi.chrName[i]] = i;
    enum { exT, exS, exE, exG, exL };
    std::vector<std::array<uint64_t, exL> > exonLoci;
    std::vector<uint32_t> transcriptStrand;
    std::vector<std::string> transcriptID, geneID;
    std::vector<std::array<std::string, 2> > geneAttr;
    uint64_t exonLines = 0;
    std::string oneLine;
    while (in.good()) {
        std::getline(in, oneLine);
        std::istringstream ls(oneLine);
        std::string chr1, ddd2, featureType;
        ls >> chr1 >> ddd2 >> featureType;
        if (chr1.substr(0, 1) == "#" || featureType != P.sjdbGTFfeatureExon) continue;
        exonLines++;
        if (P.sjdbGTFchrPrefix != "-") chr1 = P.sjdbGTFchrPrefix + chr1;
        if (chrIndex.count(chr1) == 0) { log += "WARNING: while processing sjdbGTFfile=" + P.sjdbGTFfile + ": chromosome '" + chr1 + "' not found in Genome fasta files for line:\n" + oneLine + "\n"; continue; }
        uint64_t ex1 = 0, ex2 = 0; char str1 = '.';
        ls >> ex1 >> ex2 >> ddd2 >> str1 >> ddd2;
        if (ex2 > gi.chrLength[chrIndex[chr1]]) { log += "WARNING: exon end is larger than the chromosome length, will skip this exon: " + oneLine + "\n"; continue; }
        std::string rest;
        std::getline(ls, rest);
        for (char &c : rest) if (c == ';' || c == '=' || c == '\t' || c == '"') c = ' ';
        const std::vector<std::vector<std::string> > names = {{P.sjdbGTFtagExonParentTranscript}, {P.sjdbGTFtagExonParentGene}, P.sjdbGTFtagExonParentGeneName, P.sjdbGTFtagExonParentGeneType};
        std::string exAttr[4];
        for (size_t ii = 0; ii < names.size(); ii++)
            for (const std::string &a : names[ii]) {
                size_t pos1 = rest.find(" " + a + " ");
                if (pos1 != std::string::npos) pos1 = rest.find_first_not_of(" ", pos1 + a.size() + 1);
                if (pos1 != std::string::npos) exAttr[ii] = rest.substr(pos1, rest.find_first_of(" ", pos1) - pos1);
            }
        if (exAttr[0].empty()) exAttr[0] = "tr_" + chr1 + "_" + std::to_string(ex1) + "_" + std::to_string(ex2) + "_" + std::to_string(exonLoci.size());
        if (exAttr[1].empty()) exAttr[1] = "MissingGeneID";
        if (exAttr[2].empty()) exAttr[2] = exAttr[1];
        if (exAttr[3].empty()) exAttr[3] = "MissingGeneType";
        transcriptIDnumber.insert(std::make_pair(exAttr[0], (uint64_t)transcriptIDnumber.size()));
        if (transcriptID.size() < transcriptIDnumber.size()) { transcriptID.push_back(exAttr[0]); transcriptStrand.push_back(str1 == '+' ? 1 : str1 == '-' ? 2 : 0); }
        geneIDnumber.insert(std::make_pair(exAttr[1], (uint64_t)geneIDnumber.size()));
        if (geneID.size() < geneIDnumber.size()) { geneID.push_back(exAttr[1]); geneAttr.push_back({exAttr[2], exAttr[3]}); }
        uint64_t cs = gi.chrStart[chrIndex[chr1]];
        exonLoci.push_back({transcriptIDnumber[exAttr[0]], ex1 + cs - 1, ex2 + cs - 1, geneIDnumber[exAttr[1]]});
    }
    if (exonLines == 0)
        return "Fatal INPUT FILE error, no exon lines in the GTF file: " + P.sjdbGTFfile + "\nSolution: check the formatting of the GTF file, it must contain some lines with exon in the 3rd column.\n          Make sure the GTF file is unzipped.\n          If exons are marked with a different word, use --sjdbGTFfeatureExon .\n";
    if (exonLoci.empty())
        return "Fatal INPUT FILE error, no valid exon lines in the GTF file: " + P.sjdbGTFfile + "\nSolution: check the formatting of the GTF file. One likely cause is the difference in chromosome naming between GTF and FASTA file.\n";
    const uint64_t exonN = exonLoci.size();
    // transcriptGeneSJ: by (transcript, exon start); glibc qsort = merge sort, i.e. stable
    std::stable_sort(exonLoci.begin(), exonLoci.end(), [](const std::array<uint64_t, exL> &a, const std::array<uint64_t, exL> &b) { return a[0] != b[0] ? a[0] < b[0] : a[1] < b[1]; });
    {
        std::vector<std::array<uint64_t, 5> > exge(exonN);
        for (uint64_t i = 0; i < exonN; i++) exge[i] = {exonLoci[i][exS], exonLoci[i][exE], transcriptStrand[exonLoci[i][exT]], exonLoci[i][exG], exonLoci[i][exT]};
        std::stable_sort(exge.begin(), exge.end());
        std::ofstream o(dirOut + "/exonGeTrInfo.tab");
        o << exonN << "\n";
        for (auto &e : exge) o << e[0] << "\t" << e[1] << "\t" << e[2] << "\t" << e[3] << "\t" << e[4] << "\n";
        std::ofstream g(dirOut + "/geneInfo.tab");
        g << geneID.size() << "\n";
        for (size_t i = 0; i < geneID.size(); i++) g << geneID[i] << "\t" << geneAttr[i][0] << "\t" << geneAttr[i][1] << "\n";
    }
    {
        // (trStart, trEnd, trID, exStart, exEnd, geID); sorted on the first five
        std::vector<std::array<uint64_t, 6> > extr(exonN);
        uint64_t trex1 = 0;
        for (uint64_t iex = 0; iex <= exonN; iex++) {
            if (iex == exonN || exonLoci[iex][exT] != exonLoci[trex1][exT]) {
                for (uint64_t k = trex1; k < iex; k++) extr[k][1] = exonLoci[iex - 1][exE];
                if (iex == exonN) break;
                trex1 = iex;
            }
            extr[iex][0] = exonLoci[trex1][exS]; extr[iex][2] = exonLoci[iex][exT]; extr[iex][3] = exonLoci[iex][exS]; extr[iex][4] = exonLoci[iex][exE]; extr[iex][5] = exonLoci[iex][exG];
        }
        std::stable_sort(extr.begin(), extr.end(), [](const std::array<uint64_t, 6> &a, const std::array<uint64_t, 6> &b) { for (int k = 0; k < 5; k++) if (a[k] != b[k]) return a[k] < b[k]; return false; });
        std::ofstream trOut(dirOut + "/transcriptInfo.tab"), exOut(dirOut + "/exonInfo.tab");
        trOut << transcriptID.size() << "\n"; exOut << exonN << "\n";
        uint64_t trid = extr[0][2], trex = 0, trstart = extr[0][0], trend = extr[0][1], exlen = 0;
        for (uint64_t iex = 0; iex <= exonN; iex++) {
            if (iex == exonN || extr[iex][2] != trid) {
                trOut << transcriptID.at(trid) << "\t" << extr[iex - 1][0] << "\t" << extr[iex - 1][1] << "\t" << trend << "\t" << (uint64_t)transcriptStrand[trid] << "\t" << iex - trex << "\t" << trex << "\t" << extr[iex - 1][5] << "\n";
                if (iex == exonN) break;
                trid = extr[iex][2]; trstart = extr[iex][0]; trex = iex; trend = std::max(trend, extr[iex - 1][1]); exlen = 0;
            }
            exOut << extr[iex][3] - trstart << "\t" << extr[iex][4] - trstart << "\t" << exlen << "\n";
            exlen += extr[iex][4] - extr[iex][3] + 1;
        }
    }
    // junctions between consecutive exons of a transcript
    std::vector<std::array<uint64_t, 4> > sjLoci;
    uint64_t trIDn = exonLoci[0][exT];
    for (uint64_t iex = 1; iex < exonN; iex++) {
        if (trIDn == exonLoci[iex][exT]) {
            if (exonLoci[iex][exS] <= exonLoci[iex - 1][exE] + 1) {}                    // touching or overlapping: nothing to add
            else sjLoci.push_back({exonLoci[iex - 1][exE] + 1, exonLoci[iex][exS] - 1, (uint64_t)transcriptStrand[trIDn], exonLoci[iex][exG] + 1});
        } else trIDn = exonLoci[iex][exT];
    }
    std::stable_sort(sjLoci.begin(), sjLoci.end(), [](const std::array<uint64_t, 4> &a, const std::array<uint64_t, 4> &b) { return a[0] != b[0] ? a[0] < b[0] : a[1] < b[1]; });
    const char strandChar[3] = {'.', '+', '-'};
    const size_t n0 = loci.chr.size();
    std::vector<std::set<uint64_t> > genes;
    for (size_t ii = 0; ii < sjLoci.size(); ii++) {
        if (ii == 0 || sjLoci[ii][0] != sjLoci[ii - 1][0] || sjLoci[ii][1] != sjLoci[ii - 1][1] || sjLoci[ii][2] != sjLoci[ii - 1][2]) {
            uint32_t chr1 = gi.chrBin[sjLoci[ii][0] >> gi.view.gChrBinNbits];
            loci.chr.push_back(gi.chrName[chr1]);
            loci.start.push_back(sjLoci[ii][0] + 1 - gi.chrStart[chr1]); loci.end.push_back(sjLoci[ii][1] + 1 - gi.chrStart[chr1]);
            loci.str.push_back(strandChar[sjLoci[ii][2]]);
            genes.push_back({sjLoci[ii][3]});
        } else genes.back().insert(sjLoci[ii][3]);
    }
    {
        std::ofstream o(dirOut + "/sjdbList.fromGTF.out.tab");
        for (size_t ii = n0; ii < loci.chr.size(); ii++) {
            o << loci.chr[ii] << "\t" << loci.start[ii] << "\t" << loci.end[ii] << "\t" << loci.str[ii];
            auto gg = genes[ii - n0].cbegin();
            o << "\t" << *gg;
            for (++gg; gg != genes[ii - n0].cend(); ++gg) o << "," << *gg;
            o << "\n";
        }
    }
    loci.priority.resize(loci.chr.size(), 20);
    log += "Processing pGe.sjdbGTFfile=" + P.sjdbGTFfile + ", found:\n\t\t" + std::to_string(transcriptID.size()) + " transcripts\n\t\t" + std::to_string(exonN) +
           " exons (non-collapsed)\n\t\t" + std::to_string(loci.chr.size() - n0) + " collapsed junctions\n";
    return "";
}

} // namespace staramd
