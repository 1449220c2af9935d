// genome_generate.cpp -- host side of `--runMode genomeGenerate`: everything of Genome::genomeGenerate
// (source/Genome_genomeGenerate.cpp:96-416) except the two stages that are arrays-in, arrays-out and run on the device
// (suffix array sort + SAindex: include/star_amd_index.h).  FASTA scan, chromosome tables, index geometry, junction
// insertion through the same code the mapping stage uses (sjdb_insert.cpp / gtf.cpp), and the files of a genomeDir.
#include "host.h"
#include <fstream>
#include <sstream>
#include <cmath>
#include <cstring>
#include <sys/stat.h>

namespace staramd {

static inline uint8_t ntCode(unsigned char c) {          // convertNucleotidesToNumbersRemoveControls (SequenceFuns.cpp:170-192)
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

std::string genomeGenerateScan(RunParams &P, GenomeIndex &gi, GenerateJob &job) {
    gi.dir = P.genomeDir;
    memset(&gi.view, 0, sizeof(gi.view));
    if (P.genomeSAsparseD != 1) return "EXITING because of fatal PARAMETERS error: --genomeSAsparseD " + std::to_string(P.genomeSAsparseD) + ": only 1 is built on the MI355X (use the reference for sparse suffix arrays)";
    // createDirectory(pGe.gDir, ...) (:101)
    if (mkdir(P.genomeDir.c_str(), S_IRWXU | S_IRGRP | S_IXGRP | S_IROTH | S_IXOTH) != 0 && errno != EEXIST) {
        std::string perr = P.genomeDir;
        for (size_t i = perr.find('/', 1); i != std::string::npos; i = perr.find('/', i + 1)) mkdir(perr.substr(0, i).c_str(), S_IRWXU | S_IRGRP | S_IXGRP | S_IROTH | S_IXOTH);
        if (mkdir(P.genomeDir.c_str(), S_IRWXU | S_IRGRP | S_IXGRP | S_IROTH | S_IXOTH) != 0 && errno != EEXIST)
            return "EXITING because of fatal ERROR: could not create output directory: " + P.genomeDir + " for --genomeDir";
    }
    const uint64_t binBases = 1ull << P.genomeChrBinNbits;
    // genomeScanFastaFiles: chromosomes padded to bin boundaries with at least one spacer (:47-49, :74)
    std::vector<uint8_t> &G = gi.G;
    G.clear();
    gi.chrName.clear(); gi.chrStart.clear(); gi.chrLength.clear();
    uint64_t N = 0;
    for (const std::string &fa : P.genomeFastaFiles) {
        FILE *f = fopen(fa.c_str(), "rb");
        if (!f) return "EXITING because of INPUT ERROR: could not open genomeFastaFile: " + fa + "\n";
        std::vector<char> buf(1 << 24);
        bool lineStart = true, inHeader = false, first = true; std::string header;
        size_t got;
        while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) {
            if (first) {
                if (buf[0] != '>') { fclose(f); return "EXITING because of INPUT ERROR: the file format of the genomeFastaFile: " + fa + " is not fasta: the first character is '" + std::string(1, buf[0]) + "' (" + std::to_string((int)buf[0]) + "), not '>'.\n Solution: check formatting of the fasta file. Make sure the file is uncompressed (unzipped).\n"; }
                first = false;
            }
            for (size_t i = 0; i < got; i++) {
                unsigned char c = (unsigned char)buf[i];
                if (inHeader) {
                    if (c == '\n') {
                        inHeader = false; lineStart = true;
                        std::istringstream ls(header); ls.ignore(1, ' ');
                        std::string name; ls >> name;
                        gi.chrName.push_back(name);
                        if (!gi.chrStart.empty()) gi.chrLength.push_back(N - gi.chrStart.back());
                        if (N > 0) N = ((N + 1) / binBases + 1) * binBases;
                        gi.chrStart.push_back(N);
                        if (G.size() < N) G.resize(N, 5);
                    } else header.push_back((char)c);
                    continue;
                }
                if (c == '\n') { lineStart = true; continue; }
                if (lineStart && c == '>') { inHeader = true; header.assign(1, '>'); lineStart = false; continue; }
                lineStart = false;
                if ((signed char)c < 32) continue;                     // control characters -- and, as in the reference, which tests a signed char (genomeScanFastaFiles.cpp), bytes >= 0x80 -- are skipped
                G.push_back(ntCode(c)); N++;
            }
        }
        fclose(f);
        if (first) return "EXITING because of INPUT ERROR: could not read from genomeFastaFile: " + fa + "\n";
        if (inHeader) return "EXITING because of INPUT ERROR: genomeFastaFile ends inside a header line: " + fa + "\n";
    }
    if (gi.chrStart.empty()) return "EXITING because of INPUT ERROR: no sequences in --genomeFastaFiles";
    gi.chrLength.push_back(N - gi.chrStart.back());
    N = ((N + 1) / binBases + 1) * binBases;
    G.resize(N, 5);
    const uint32_t nChr = (uint32_t)gi.chrName.size();
    gi.chrStart.push_back(N);
    // writeChrInfo (:418-432)
    {
        std::ofstream chrN(P.genomeDir + "/chrName.txt"), chrS(P.genomeDir + "/chrStart.txt"), chrL(P.genomeDir + "/chrLength.txt"), chrNL(P.genomeDir + "/chrNameLength.txt");
        if (!chrN.good() || !chrS.good() || !chrL.good() || !chrNL.good()) return "EXITING because of fatal ERROR: could not write into --genomeDir " + P.genomeDir;
        for (uint32_t i = 0; i < nChr; i++) { chrN << gi.chrName[i] << "\n"; chrS << gi.chrStart[i] << "\n"; chrL << gi.chrLength[i] << "\n"; chrNL << gi.chrName[i] << "\t" << gi.chrLength[i] << "\n"; }
        chrS << gi.chrStart[nChr] << "\n";
    }
    // geometry (:160-181): sjdbLength comes from the parameter even when no annotation is given (Genome.cpp:23-24)
    const bool annot = P.sjdbInsertPass1();
    const uint64_t sjdbLengthPar = P.sjdbOverhang == 0 ? 0 : 2ull * P.sjdbOverhang + 1;
    uint32_t GstrandBit = (uint32_t)(uint64_t)std::floor(std::log((double)(N + P.limitSjdbInsertNsj * sjdbLengthPar)) / std::log(2.0)) + 1;
    if (GstrandBit < 32) GstrandBit = 32;
    job.GstrandBit = GstrandBit;
    uint64_t nACGT = 0;
    for (uint64_t i = 0; i < N; i++) nACGT += G[i] < 4;
    job.nSA = 2 * nACGT;
    const uint32_t L = P.genomeSAindexNbases;
    job.saiStart[0] = 0;
    for (uint32_t i = 1; i <= L; i++) job.saiStart[i] = job.saiStart[i - 1] + (1ull << (2 * i));
    auto packedBytes = [](uint64_t n, uint32_t bits) { return n == 0 ? (uint64_t)8 : (n - 1) * bits / 8 + 8; };
    job.saBytes = packedBytes(job.nSA, GstrandBit + 1); job.saiBytes = packedBytes(job.saiStart[L], GstrandBit + 3);
    gi.SA.assign(job.saBytes + 8, 0); gi.SAi.assign(job.saiBytes + 8, 0);
    // chrBinFill (Genome.cpp:209-215)
    uint64_t chrBinN = gi.chrStart[nChr] / binBases + 1;
    gi.chrBin.assign(chrBinN, 0);
    for (uint64_t ii = 0, ichr = 1; ii < chrBinN; ++ii) { if (ii * binBases >= gi.chrStart[ichr]) ichr++; gi.chrBin[ii] = (uint32_t)(ichr - 1); }
    staramd_genome &V = gi.view;
    V.nGenome = N; V.nSA = job.nSA; V.nSAbyte = job.saBytes; V.nSAi = job.saiStart[L]; V.nSAibyte = job.saiBytes;
    V.GstrandBit = GstrandBit; V.gSAindexNbases = L; V.gSAsparseD = 1; V.gChrBinNbits = P.genomeChrBinNbits;
    for (uint32_t i = 0; i <= L; i++) V.genomeSAindexStart[i] = job.saiStart[i];
    V.nChrReal = nChr; V.chrBinN = chrBinN;
    V.sjdbOverhang = annot ? P.sjdbOverhang : 0; V.sjdbLength = annot ? 2 * P.sjdbOverhang + 1 : 0;
    V.sjdbN = 0; V.sjGstart = gi.chrStart[nChr] + 1;
    gi.refreshView();
    return "";
}

// after the device build filled gi.SA / gi.SAi
std::string genomeGenerateFinish(RunParams &P, GenomeIndex &gi, GenerateJob &job, SjdbLoci &loci, std::string &log) {
    staramd_genome &V = gi.view;
    if (P.sjdbInsertPass1()) {                                               // :326-333
        P.sjdbInsertOutDir = P.genomeDir + "/";
        V.sjGstart = gi.chrStart[V.nChrReal];
        std::string e = sjdbInsertJunctions(P, gi, loci, false, "", log);
        if (!e.empty()) return e;
    }
    // genomeParametersWrite (genomeParametersWrite.cpp:4-46)
    {
        std::ofstream gp(P.genomeDir + "/genomeParameters.txt");
        if (!gp.good()) return "EXITING because of fatal ERROR: could not write " + P.genomeDir + "/genomeParameters.txt";
        gp << "### " << P.commandLine << "\n";
        gp << "### GstrandBit " << V.GstrandBit << "\n";
        gp << "versionGenome\t2.7.4a\n" << "genomeType\tFull\n" << "genomeFastaFiles\t";
        for (auto &f : P.genomeFastaFiles) gp << f << " ";
        gp << "\n" << "genomeSAindexNbases\t" << V.gSAindexNbases << "\n" << "genomeChrBinNbits\t" << V.gChrBinNbits << "\n" << "genomeSAsparseD\t" << V.gSAsparseD << "\n";
        gp << "genomeTransformType\tNone\n" << "genomeTransformVCF\t-\n";
        gp << "sjdbOverhang\t" << V.sjdbOverhang << "\n" << "sjdbFileChrStartEnd\t";
        if (P.sjdbFileChrStartEnd.empty()) gp << "- "; else for (auto &f : P.sjdbFileChrStartEnd) gp << f << " ";
        gp << "\n" << "sjdbGTFfile\t" << (P.sjdbGTFfile.empty() ? "-" : P.sjdbGTFfile) << "\n" << "sjdbGTFchrPrefix\t" << P.sjdbGTFchrPrefix << "\n"
           << "sjdbGTFfeatureExon\t" << P.sjdbGTFfeatureExon << "\n" << "sjdbGTFtagExonParentTranscript\t" << P.sjdbGTFtagExonParentTranscript << "\n"
           << "sjdbGTFtagExonParentGene\t" << P.sjdbGTFtagExonParentGene << "\n" << "sjdbInsertSave\t" << (P.sjdbInsertSaveAll ? "All" : "Basic") << "\n";
        gp << "genomeFileSizes\t" << V.nGenome << " " << V.nSAbyte << "\n";
    }
    auto dump = [&](const std::string &name, const void *p, uint64_t n, const void *hdr = nullptr, uint64_t nh = 0) {
        FILE *f = fopen((P.genomeDir + "/" + name).c_str(), "wb");
        if (!f) return false;
        bool ok = true;
        if (nh) ok = fwrite(hdr, 1, nh, f) == nh;
        const uint8_t *q = (const uint8_t *)p;
        for (uint64_t off = 0; ok && off < n; ) { uint64_t k = std::min<uint64_t>(n - off, 1ull << 30); ok = fwrite(q + off, 1, k, f) == k; off += k; }
        return fclose(f) == 0 && ok;
    };
    std::vector<uint64_t> hdr(1, V.gSAindexNbases);
    for (uint32_t i = 0; i <= V.gSAindexNbases; i++) hdr.push_back(V.genomeSAindexStart[i]);
    if (!dump("Genome", gi.G.data(), V.nGenome) || !dump("SA", gi.SA.data(), V.nSAbyte) || !dump("SAindex", gi.SAi.data(), V.nSAibyte, hdr.data(), hdr.size() * 8))
        return "EXITING because of fatal ERROR: could not write the genome files into " + P.genomeDir;
    (void)job;
    return "";
}

} // namespace staramd
