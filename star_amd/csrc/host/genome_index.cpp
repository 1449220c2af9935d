// genome_index.cpp -- parse an unmodified STAR genomeDir.
// Follows Genome::genomeLoad (source/Genome_genomeLoad.cpp:18-467), Genome::chrInfoLoad / chrBinFill
// (source/Genome.cpp:139-215) and Genome::loadSJDB (source/Genome_genomeLoad.cpp:471-521).
#include "host.h"
#include <fstream>
#include <sstream>
#include <cmath>
#include <cstring>
#include <chrono>
#include <sys/stat.h>

namespace staramd {

// headroom: capacity beyond the file (address space only, untouched) -- the genome text grows by the inserted junction sequences in a 2-pass run, and a
// vector that has to move for that copies 3 GB on one thread
static bool readFile(const std::string &path, std::vector<uint8_t> &out, size_t extra = 0, size_t headroom = 0) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (headroom) out.reserve((size_t)n + extra + headroom);
    out.assign((size_t)n + extra, 0);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    out.resize((size_t)n + extra);
    return got == (size_t)n;
}

std::string GenomeIndex::load(const std::string &genomeDir) {
    auto t0 = std::chrono::steady_clock::now();
    dir = genomeDir;
    memset(&view, 0, sizeof(view));
    // --- genomeParameters.txt (:33-62): "### GstrandBit N" comment + key/value lines
    uint32_t GstrandBit = 0, gSAindexNbases = 14, gChrBinNbits = 18, gSAsparseD = 1, sjdbOverhang = 0;
    std::string versionGenome;
    {
        std::ifstream pf(dir + "/genomeParameters.txt");
        if (!pf.good()) return "EXITING because of FATAL ERROR: could not open genome file " + dir + "/genomeParameters.txt";
        std::string line;
        while (std::getline(pf, line)) {
            std::istringstream ls(line);
            std::string k;
            ls >> k;
            if (k == "###") { ls >> k; if (k == "GstrandBit") ls >> GstrandBit; continue; }
            if (k == "versionGenome") ls >> versionGenome;
            else if (k == "genomeSAindexNbases") ls >> gSAindexNbases;
            else if (k == "genomeChrBinNbits") ls >> gChrBinNbits;
            else if (k == "genomeSAsparseD") ls >> gSAsparseD;
            else if (k == "sjdbOverhang") ls >> sjdbOverhang;
            else if (k == "genomeType") { std::string v; ls >> v; if (v != "Full") return "EXITING because of FATAL ERROR: the index in " + dir + " was generated with --genomeType " + v + "; only Full genomes are supported by the MI355X engine"; }
            else if (k == "genomeTransformType") { std::string v; ls >> v; if (v != "None") return "EXITING because of FATAL ERROR: the index in " + dir + " is a transformed genome (--genomeTransformType " + v + "); genome transformation is not supported by the MI355X engine"; }
        }
    }
    if (versionGenome != "2.7.4a")       // parametersDefault:2 ; Genome_genomeLoad.cpp:71-83
        return "EXITING because of FATAL ERROR: Genome version: " + versionGenome + " is INCOMPATIBLE with running STAR version: 2.7.11b";
    // --- chrName/Length/Start (Genome.cpp:139-192)
    {
        std::ifstream in(dir + "/chrName.txt");
        if (!in.good()) return "EXITING because of FATAL error, could not open file " + dir + "/chrName.txt";
        std::string l;
        while (std::getline(in, l)) { if (l.empty()) break; chrName.push_back(l); }
    }
    uint32_t nChrReal = (uint32_t)chrName.size();
    chrStart.assign(nChrReal + 1, 0); chrLength.assign(nChrReal, 0);
    {
        std::ifstream in(dir + "/chrLength.txt");
        if (!in.good()) return "EXITING because of FATAL error, could not open file " + dir + "/chrLength.txt";
        for (uint32_t i = 0; i < nChrReal; i++) in >> chrLength[i];
    }
    {
        std::ifstream in(dir + "/chrStart.txt");
        if (!in.good()) return "EXITING because of FATAL error, could not open file " + dir + "/chrStart.txt";
        for (uint32_t i = 0; i <= nChrReal; i++) in >> chrStart[i];
    }
    // --- Genome / SA / SAindex (:139-169, 315-336)
    if (!readFile(dir + "/Genome", G, 0, (size_t)320 << 20)) return "EXITING because of FATAL ERROR: could not open genome file " + dir + "/Genome";
    if (!readFile(dir + "/SA", SA, 8)) return "EXITING because of FATAL ERROR: could not open genome file " + dir + "/SA";
    uint64_t nSAbyte = SA.size() - 8;
    std::vector<uint8_t> sai;
    if (!readFile(dir + "/SAindex", sai, 8)) return "EXITING because of FATAL ERROR: could not open genome file " + dir + "/SAindex";
    uint64_t nb;
    memcpy(&nb, sai.data(), 8);
    if (nb > 16) return "SAindex: unsupported genomeSAindexNbases";
    gSAindexNbases = (uint32_t)nb;
    memcpy(view.genomeSAindexStart, sai.data() + 8, 8 * (nb + 1));
    SAi.assign(sai.begin() + 8 * (nb + 2), sai.end());
    uint64_t nGenome = G.size();
    if (GstrandBit == 0) {       // :149-153
        GstrandBit = (uint32_t)std::floor(std::log((double)nGenome) / std::log(2.0)) + 1;
        if (GstrandBit < 32) GstrandBit = 32;
    }
    view.G = G.data(); view.nGenome = nGenome;
    view.SA = SA.data(); view.nSAbyte = nSAbyte; view.nSA = (nSAbyte * 8) / (GstrandBit + 1);
    view.SAi = SAi.data(); view.nSAibyte = SAi.size() - 8; view.nSAi = view.genomeSAindexStart[nb];
    view.GstrandBit = GstrandBit; view.gSAindexNbases = gSAindexNbases; view.gSAsparseD = gSAsparseD;
    view.gChrBinNbits = gChrBinNbits;
    view.chrStart = chrStart.data(); view.chrLength = chrLength.data(); view.nChrReal = nChrReal;
    // --- chrBinFill (Genome.cpp:209-215)
    uint64_t binBases = 1ull << gChrBinNbits;
    uint64_t chrBinN = chrStart[nChrReal] / binBases + 1;
    chrBin.assign(chrBinN, 0);
    for (uint64_t ii = 0, ichr = 1; ii < chrBinN; ++ii) {
        if (ii * binBases >= chrStart[ichr]) ichr++;
        chrBin[ii] = (uint32_t)(ichr - 1);
    }
    view.chrBin = chrBin.data(); view.chrBinN = chrBinN;
    { struct stat st1; sjdbInfoExists = stat((dir + "/sjdbInfo.txt").c_str(), &st1) == 0; }
    // --- loadSJDB (:471-521)
    view.sjdbOverhang = sjdbOverhang;
    view.sjdbLength = sjdbOverhang == 0 ? 0 : sjdbOverhang * 2 + 1;
    if (nGenome == chrStart[nChrReal]) {
        view.sjdbN = 0; view.sjGstart = chrStart[nChrReal] + 1;
    } else {
        std::ifstream in(dir + "/sjdbInfo.txt");
        if (!in.good()) return "EXITING because of FATAL error, could not open file " + dir + "/sjdbInfo.txt";
        uint64_t n, ov;
        in >> n >> ov;
        view.sjdbN = (uint32_t)n; view.sjdbOverhang = (uint32_t)ov; view.sjdbLength = (uint32_t)(ov * 2 + 1);
        view.sjGstart = chrStart[nChrReal];
        sjDstart.resize(n); sjAstart.resize(n); sjdbStart.resize(n); sjdbEnd.resize(n);
        sjdbMotif.resize(n); sjdbShiftLeft.resize(n); sjdbShiftRight.resize(n); sjdbStrand.resize(n);
        for (uint64_t ii = 0; ii < n; ii++) {
            uint32_t d1, d2, d3, d4;
            in >> sjdbStart[ii] >> sjdbEnd[ii] >> d1 >> d2 >> d3 >> d4;
            sjdbMotif[ii] = (uint8_t)d1; sjdbShiftLeft[ii] = (uint8_t)d2; sjdbShiftRight[ii] = (uint8_t)d3; sjdbStrand[ii] = (uint8_t)d4;
            sjDstart[ii] = sjdbStart[ii] - ov; sjAstart[ii] = sjdbEnd[ii] + 1;
            if (sjdbMotif[ii] == 0) { sjDstart[ii] += sjdbShiftLeft[ii]; sjAstart[ii] += sjdbShiftLeft[ii]; }
        }
        view.sjDstart = sjDstart.data(); view.sjAstart = sjAstart.data(); view.sjdbStart = sjdbStart.data(); view.sjdbEnd = sjdbEnd.data();
        view.sjdbMotif = sjdbMotif.data(); view.sjdbShiftLeft = sjdbShiftLeft.data(); view.sjdbShiftRight = sjdbShiftRight.data(); view.sjdbStrand = sjdbStrand.data();
    }
    loadSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return "";
}

} // namespace staramd
