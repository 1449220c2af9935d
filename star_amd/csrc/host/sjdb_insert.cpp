// sjdb_insert.cpp -- insertion of splice junctions into a loaded index.
// Used between the two passes of --twopassMode Basic and for --sjdbFileChrStartEnd at the mapping stage; afterwards
// the caller re-uploads the index with staramd_update_index().  Restates, on the host:
//   sjdbLoadFromStream                      source/sjdbLoadFromStream.cpp:2-28
//   sjdbPrepare                             source/sjdbPrepare.cpp:5-225
//   sjdbBuildIndex                          source/sjdbBuildIndex.cpp:16-333
//   suffixArraySearch1, compareSeqToGenome1, compareRefEnds, funCalcSAi   source/SuffixArrayFuns.cpp:211-351,397-410
//   funCompareUintAndSuffixes               source/funCompareUintAndSuffixes.cpp:6-40
//   sjdbInsertJunctions                     source/sjdbInsertJunctions.cpp:11-102
// The suffix searches are independent and run on --runThreadN host threads (the reference uses OpenMP there).
#include "host.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <thread>
#include <chrono>
#include <array>
#include <sys/stat.h>

namespace staramd {

namespace {

const uint8_t SPACER = 5;                    // GENOME_spacingChar, IncludeDefine.h:62
const uint64_t GP = 512;                     // padding of spacers either side of the host genome copy (Genome_genomeLoad.cpp:27,320-323)

struct Packed {                              // PackedArray (PackedArray.h:24-32, PackedArray.cpp:17-25) over a byte vector with 8 spare bytes
    uint8_t *a; uint32_t w; uint64_t mask;
    Packed(uint8_t *p, uint32_t bits) : a(p), w(bits), mask(bits >= 64 ? ~0ull : ((1ull << bits) - 1)) {}
    uint64_t get(uint64_t i) const { uint64_t b = i * w, v; memcpy(&v, a + b / 8, 8); return (v >> (b % 8)) & mask; }
    void put(uint64_t i, uint64_t x) { uint64_t b = i * w, v; memcpy(&v, a + b / 8, 8); uint32_t s = (uint32_t)(b % 8); v = (v & ~(mask << s)) | (x << s); memcpy(a + b / 8, &v, 8); }
    static uint64_t lengthByte(uint64_t n, uint32_t bits) { return (n - 1) * bits / 8 + 8; }
};

struct SearchCtx {                           // the OLD index, as the searches see it
    const uint8_t *G;                        // padded: G[-GP .. nGenome+GP)
    uint64_t nGenome, nSA; uint32_t GstrandBit; uint64_t GstrandMask;
    Packed SA;
};

// compareRefEnds with gInsert = -1 (SuffixArrayFuns.cpp:211-220): a new suffix that ties up to its spacer goes after every
// + strand suffix of the old index and before every - strand one
inline int compareRefEnds(bool strG, bool /*strR*/) { return strG ? 1 : -1; }

// compareSeqToGenome1 (SuffixArrayFuns.cpp:221-294): s = junction text, sc = its complement; spacers are allowed in s
uint64_t compareSeqToGenome1(const SearchCtx &X, const uint8_t *s0, const uint8_t *s1, uint64_t N, uint64_t L, uint64_t iSA, bool dirR, int &compRes) {
    uint64_t SAstr = X.SA.get(iSA);
    bool dirG = (SAstr >> X.GstrandBit) == 0;
    SAstr &= X.GstrandMask;
    if (dirG) {
        const uint8_t *s = s0 + L, *g = X.G + SAstr + L;
        for (uint64_t ii = 0; ii < N - L; ii++) {
            if (s[ii] != g[ii]) { compRes = s[ii] > g[ii] ? 1 : -1; return ii + L; }
            if (s[ii] == SPACER) { compRes = compareRefEnds(dirG, dirR); return ii + L; }
        }
        return N;
    }
    const uint8_t *s = s1 + L, *g = X.G + (X.nGenome - 1 - SAstr - L);
    for (uint64_t ii = 0; ii < N - L; ii++) {
        uint8_t sv = s[ii], gv = *(g - (int64_t)ii);
        if (sv != gv) {
            if (sv < 4) sv = 3 - sv;
            if (gv < 4) gv = 3 - gv;
            compRes = sv > gv ? 1 : -1;
            return ii + L;
        }
        if (sv == SPACER) { compRes = compareRefEnds(dirG, dirR); return ii + L; }
    }
    return N;
}

inline uint64_t medianUint2(uint64_t a, uint64_t b) { return a / 2 + b / 2 + (a % 2 + b % 2) / 2; }

// suffixArraySearch1 (SuffixArrayFuns.cpp:297-351) with S folded into the pointers, N = 10000, gInsert = -1, strR = true
uint64_t suffixArraySearch1(const SearchCtx &X, const uint8_t *s0, const uint8_t *s1) {
    const uint64_t N = 10000;
    uint64_t i1 = 0, i2 = X.nSA - 1, L = 0;
    int compRes = 0;
    uint64_t L1 = compareSeqToGenome1(X, s0, s1, N, L, i1, true, compRes);
    if (compRes < 0) return 0;
    uint64_t L2 = compareSeqToGenome1(X, s0, s1, N, L, i2, true, compRes);
    if (compRes > 0) return (uint64_t)-2ll;
    L = std::min(L1, L2);
    while (i1 + 1 < i2) {
        uint64_t i3 = medianUint2(i1, i2);
        uint64_t L3 = compareSeqToGenome1(X, s0, s1, N, L, i3, true, compRes);
        if (L3 == N) return i3;
        if (compRes > 0) { i1 = i3; L1 = L3; } else if (compRes < 0) { i2 = i3; L2 = L3; }
        L = std::min(L1, L2);
    }
    return i2;
}

// funCalcSAi (SuffixArrayFuns.cpp:397-410)
inline int64_t funCalcSAi(const uint8_t *g, uint32_t iL) {
    int64_t ind1 = 0;
    for (uint32_t k = 0; k <= iL; k++) { uint32_t c = g[k]; if (c > 3) return -ind1; ind1 = (ind1 << 2) + c; }
    return ind1;
}

// binarySearch2 (source/binarySearch2.cpp:3-43) over the OLD junction table
int64_t binarySearch2(uint64_t x, uint64_t y, const std::vector<uint64_t> &X, const std::vector<uint64_t> &Y) {
    int64_t N = (int64_t)X.size();
    if (N == 0 || x < X[0] || x > X[N - 1]) return -1;
    int64_t i1 = 0, i2 = N - 1, i3;
    while (i2 > i1 + 1) { i3 = i1 + (i2 - i1) / 2; if (X[i3] > x) i2 = i3; else i1 = i3; }
    if (x == X[i2]) i1 = i2; else if (x != X[i1]) return -1;
    for (int64_t jj = i1; jj >= 0; jj--) { if (x != X[jj]) break; if (y == Y[jj]) return jj; }
    return -2;
}

void removeDirRecursive(const std::string &d) {          // sysRemoveDir: the run-time directories hold flat files only
    std::string cmd = "rm -rf '" + d + "'";
    int rc = system(cmd.c_str()); (void)rc;
}

} // namespace

void sjdbLoadFromStream(std::istream &in, SjdbLoci &loci) {
    std::string line;
    while (in.good()) {
        std::getline(in, line);
        // `ls >> chr1 >> u1 >> u2 >> str1` of the reference (sjdbLoadFromStream.cpp:9-12), by hand: a stream object per line was 0.7 s for the two tables of a 2-pass run
        std::string chr1; uint64_t u1 = 0, u2 = 0; char str1 = '.';
        {
            const char *p = line.c_str();
            auto ws = [&] { while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\v' || *p == '\f') p++; };
            ws(); const char *b = p; while (*p && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\v' || *p == '\f')) p++;
            chr1.assign(b, p);
            bool ok = !chr1.empty();
            auto num = [&](uint64_t &v) { ws(); if (!ok || *p < '0' || *p > '9') { ok = false; return; } char *e; v = strtoull(p, &e, 10); p = e; };     // (a failed extraction leaves the rest at its default, as the stream does)
            num(u1); num(u2);
            if (ok) { ws(); if (*p) str1 = *p; }
        }
        if (chr1.empty()) continue;
        loci.chr.push_back(chr1); loci.start.push_back(u1); loci.end.push_back(u2);
        loci.str.push_back((str1 == '1' || str1 == '+') ? '+' : (str1 == '2' || str1 == '-') ? '-' : '.');
    }
}

// device junction insertion (include/star_amd_index.h staramd_sjdb_insert), handed in by the front end that links the engine: the host
// library itself links no GPU code.  Null: the threaded host restatement below does the work.
static SjdbDeviceFn g_sjdbDeviceFn = nullptr; static int g_sjdbDevice = 0;
void setSjdbDeviceFn(SjdbDeviceFn fn, int device) { g_sjdbDeviceFn = fn; g_sjdbDevice = device; }
// the same on the index RESIDENT in the engine contexts (staramd_insert_junctions): SA / SAindex stay in HBM, the host copies are dropped
static SjdbResidentFn g_sjdbResidentFn = nullptr; static void *g_sjdbResidentUser = nullptr;
void setSjdbResidentFn(SjdbResidentFn fn, void *user) { g_sjdbResidentFn = fn; g_sjdbResidentUser = user; }

void GenomeIndex::refreshView() {
    view.G = G.data(); view.nGenome = G.size();
    view.SA = SA.data(); view.SAi = SAi.data();
    view.chrStart = chrStart.data(); view.chrLength = chrLength.data(); view.chrBin = chrBin.data();
    view.sjDstart = sjDstart.data(); view.sjAstart = sjAstart.data(); view.sjdbStart = sjdbStart.data(); view.sjdbEnd = sjdbEnd.data();
    view.sjdbMotif = sjdbMotif.data(); view.sjdbShiftLeft = sjdbShiftLeft.data(); view.sjdbShiftRight = sjdbShiftRight.data(); view.sjdbStrand = sjdbStrand.data();
}

// sjdbPrepare + sjdbBuildIndex; `gi` is rewritten in place
static std::string sjdbPrepareAndBuild(const RunParams &P, GenomeIndex &gi, const SjdbLoci &loci, const std::string &outDir, std::string &log) {
    staramd_genome &V = gi.view;
    const uint64_t nLoci = loci.chr.size();
    const uint32_t nChr = V.nChrReal;
    const uint64_t nGenomeReal = gi.chrStart[nChr];
    const uint64_t ov = V.sjdbOverhang, sjdbLength = V.sjdbLength;
    // the old index: genome text with spacer padding on both sides, junction table
    const bool timing = getenv("STARAMD_HOST_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (timing) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  sjdb insert %-12s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count()); T0 = t; } };
    // (3 GB for a human genome: not zero-filled first, copied on threads -- the two single-threaded passes over it were 1.5 s of every insertion)
    std::vector<uint8_t, NoInitAlloc<uint8_t>> Gp(GP + V.nGenome + GP);
    memset(Gp.data(), SPACER, GP); memset(Gp.data() + GP + V.nGenome, SPACER, GP);
    {
        const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(P.runThreadN, 1), V.nGenome >> 24));
        const uint64_t per = (V.nGenome + T - 1) / T;
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t] { const uint64_t lo = std::min<uint64_t>(V.nGenome, (uint64_t)t * per), hi = std::min<uint64_t>(V.nGenome, lo + per); if (hi > lo) memcpy(Gp.data() + GP + lo, gi.G.data() + lo, hi - lo); });
        for (auto &x : th) x.join();
    }
    const uint8_t *G = Gp.data() + GP;
    const std::vector<uint64_t> oldStart = gi.sjdbStart, oldEnd = gi.sjdbEnd;
    const uint64_t oldSjdbN = V.sjdbN, oldNSA = V.nSA, oldNGenome = V.nGenome;

    lap("genome copy");
    // ---------------- sjdbPrepare (sjdbPrepare.cpp:5-225)
    std::vector<uint64_t> sjdbS(nLoci), sjdbE(nLoci);
    std::vector<uint8_t> motif(nLoci), shL(nLoci), shR(nLoci);
    {
        std::string chrOld; uint32_t iChr = 0;
        for (uint64_t ii = 0; ii < nLoci; ii++) {
            if (chrOld != loci.chr[ii]) {
                for (iChr = 0; iChr < nChr; iChr++) if (loci.chr[ii] == gi.chrName[iChr]) break;
                if (iChr >= nChr) {
                    std::ostringstream e;
                    e << "EXITING because of FATAL error, the sjdb chromosome " << loci.chr[ii] << " is not found among the genomic chromosomes\n"
                      << "SOLUTION: fix your file(s) --sjdbFileChrStartEnd or --sjdbGTFfile, offending junction:" << loci.chr[ii] << "\t" << loci.start[ii] << "\t" << loci.end[ii] << "\n";
                    return e.str();
                }
                chrOld = loci.chr[ii];
            }
            uint64_t s = loci.start[ii] + gi.chrStart[iChr] - 1, e = loci.end[ii] + gi.chrStart[iChr] - 1;
            if (s == 0 || e + 1 >= nGenomeReal || s > e) return "EXITING because of FATAL error: sjdb junction outside of the chromosomes: " + loci.chr[ii];
            auto is = [&](uint8_t a, uint8_t b, uint8_t c, uint8_t d) { return G[s] == a && G[s + 1] == b && G[e - 1] == c && G[e] == d; };
            motif[ii] = is(2, 3, 0, 2) ? 1 : is(1, 3, 0, 1) ? 2 : is(2, 1, 0, 2) ? 3 : is(1, 3, 2, 1) ? 4 : is(0, 3, 0, 1) ? 5 : is(2, 3, 0, 3) ? 6 : 0;
            uint64_t jjL = 0, jjR = 0;
            while (jjL <= s - 1 && G[s - 1 - jjL] == G[e - jjL] && G[s - 1 - jjL] < 4 && jjL < 255) jjL++;
            while (s + jjR < nGenomeReal && G[s + jjR] == G[e + 1 + jjR] && G[s + jjR] < 4 && jjR < 255) jjR++;
            shL[ii] = (uint8_t)jjL; shR[ii] = (uint8_t)jjR;
            if (jjR == 255 || jjL == 255) log += "WARNING: long repeat for junction # " + std::to_string(ii + 1) + "\n";
            sjdbS[ii] = s - jjL; sjdbE[ii] = e - jjL;
        }
    }
    typedef std::array<uint64_t, 3> T3;
    auto cmp2 = [](const T3 &a, const T3 &b) { return a[0] != b[0] ? a[0] < b[0] : a[1] < b[1]; };   // funCompareUint2; glibc qsort = merge sort -> stable
    std::vector<T3> srt(nLoci);
    for (uint64_t ii = 0; ii < nLoci; ii++) {
        uint64_t shift1 = loci.str[ii] == '+' ? 0 : loci.str[ii] == '-' ? nGenomeReal : 2 * nGenomeReal;
        srt[ii] = {sjdbS[ii] + shift1, sjdbE[ii] + shift1, ii};
    }
    std::stable_sort(srt.begin(), srt.end(), cmp2);
    std::vector<uint64_t> I; I.reserve(nLoci);
    for (uint64_t ii = 0; ii < nLoci; ii++) {
        uint64_t isj = srt[ii][2];
        if (I.empty()) { I.push_back(isj); continue; }
        uint64_t isj0 = I.back();
        if (sjdbS[isj] != sjdbS[isj0] || sjdbE[isj] != sjdbE[isj0]) I.push_back(isj);
        else if (loci.priority[isj] < loci.priority[isj0]) {}
        else if (loci.priority[isj] > loci.priority[isj0]) I.back() = isj;
        else if ((motif[isj] > 0 && motif[isj0] == 0) || (((motif[isj] > 0) == (motif[isj0] > 0)) && shL[isj] < shL[isj0])) I.back() = isj;
    }
    const uint64_t nsj = I.size();
    srt.resize(nsj);
    for (uint64_t ii = 0; ii < nsj; ii++) {
        uint64_t k = I[ii], back = motif[k] == 0 ? 0 : shL[k];
        srt[ii] = {sjdbS[k] + back, sjdbE[k] + back, k};
    }
    std::stable_sort(srt.begin(), srt.end(), cmp2);
    std::vector<uint64_t> nStart(nsj), nEnd(nsj);
    std::vector<uint8_t> nMotif(nsj), nShL(nsj), nShR(nsj), nStrand(nsj);
    uint64_t nsj1 = 0;
    for (uint64_t ii = 0; ii < nsj; ii++) {
        uint64_t isj = srt[ii][2];
        if (nsj1 > 0 && nStart[nsj1 - 1] == srt[ii][0] && nEnd[nsj1 - 1] == srt[ii][1]) {       // same loci on opposite strands
            uint64_t isj0 = srt[ii - 1][2];
            if (loci.priority[isj] < loci.priority[isj0]) continue;
            else if (loci.priority[isj] > loci.priority[isj0]) nsj1--;
            else if (nStrand[nsj1 - 1] > 0 && loci.str[isj] == '.') continue;
            else if (nStrand[nsj1 - 1] == 0 && loci.str[isj] != '.') nsj1--;
            else if (nMotif[nsj1 - 1] == 0 && motif[isj] == 0) { nStrand[nsj1 - 1] = 0; continue; }
            else if ((nMotif[nsj1 - 1] > 0 && motif[isj] == 0) || (nMotif[nsj1 - 1] % 2 == (2 - nStrand[nsj1 - 1]))) continue;
            else nsj1--;
        }
        nStart[nsj1] = srt[ii][0]; nEnd[nsj1] = srt[ii][1];
        nMotif[nsj1] = motif[isj]; nShL[nsj1] = shL[isj]; nShR[nsj1] = shR[isj];
        if (loci.str[isj] == '+') nStrand[nsj1] = 1;
        else if (loci.str[isj] == '-') nStrand[nsj1] = 2;
        else nStrand[nsj1] = nMotif[nsj1] == 0 ? 0 : 2 - nMotif[nsj1] % 2;
        nsj1++;
    }
    const uint64_t sjdbN = nsj1;
    nStart.resize(sjdbN); nEnd.resize(sjdbN); nMotif.resize(sjdbN); nShL.resize(sjdbN); nShR.resize(sjdbN); nStrand.resize(sjdbN);
    if (sjdbN > P.limitSjdbInsertNsj) {
        std::ostringstream e;
        e << "Fatal LIMIT error: the number of junctions to be inserted on the fly =" << sjdbN << " is larger than the limitSjdbInsertNsj=" << P.limitSjdbInsertNsj << "\n"
          << "SOLUTION: re-run with at least --limitSjdbInsertNsj " << sjdbN << "\n";
        return e.str();
    }
    std::vector<uint64_t> nD(sjdbN), nA(sjdbN);
    const uint64_t nGsj = sjdbLength * sjdbN;
    std::vector<uint8_t> Gsj(2 * nGsj + 1 + 64, SPACER);
    {
        std::ofstream info(outDir + "/sjdbInfo.txt"), list(outDir + "/sjdbList.out.tab");
        if (!info.good() || !list.good()) return "EXITING because of fatal ERROR: could not write into " + outDir;
        const char strandChar[3] = {'.', '+', '-'};
        info << sjdbN << "\t" << ov << "\n";
        uint64_t pos = 0;
        for (uint64_t ii = 0; ii < sjdbN; ii++) {
            nD[ii] = nStart[ii] - ov; nA[ii] = nEnd[ii] + 1;
            if (nMotif[ii] == 0) { nD[ii] += nShL[ii]; nA[ii] += nShL[ii]; }
            memcpy(Gsj.data() + pos, G + (int64_t)nD[ii], ov);          // may reach into the left padding for a junction at the very start
            memcpy(Gsj.data() + pos + ov, G + nA[ii], ov);
            pos += sjdbLength;
            Gsj[pos - 1] = SPACER;
            info << nStart[ii] << "\t" << nEnd[ii] << "\t" << (int)nMotif[ii] << "\t" << (int)nShL[ii] << "\t" << (int)nShR[ii] << "\t" << (int)nStrand[ii] << "\n";
            uint32_t chr1 = gi.chrBin[nStart[ii] >> V.gChrBinNbits];
            uint64_t back = nMotif[ii] > 0 ? 0 : nShL[ii];
            list << gi.chrName[chr1] << "\t" << nStart[ii] - gi.chrStart[chr1] + 1 + back << "\t" << nEnd[ii] - gi.chrStart[chr1] + 1 + back << "\t" << strandChar[nStrand[ii]] << "\n";
        }
    }
    lap("prepare");
    if (sjdbN == 0) {                                                    // sjdbBuildIndex.cpp:20-23: nothing to insert, the index stays as it is
        // ... also in the engine contexts: they need the new tables / parameters only, not a re-upload (the host copy of the suffix array is gone by then)
        if (g_sjdbResidentFn != nullptr && gi.engineHoldsIndex) gi.indexInEngine = true;
        return "";
    }

    // ---------------- sjdbBuildIndex (sjdbBuildIndex.cpp:16-333)
    Gsj[nGsj * 2] = SPACER;
    for (uint64_t ii = 0; ii < nGsj; ii++) Gsj[nGsj * 2 - 1 - ii] = Gsj[ii] < 4 ? 3 - Gsj[ii] : Gsj[ii];
    std::vector<uint8_t> G1c(Gsj.size());
    for (size_t ii = 0; ii < Gsj.size(); ii++) G1c[ii] = Gsj[ii] < 4 ? 3 - Gsj[ii] : Gsj[ii];
    SearchCtx X{G, oldNGenome, oldNSA, V.GstrandBit, (1ull << V.GstrandBit) - 1, Packed(gi.SA.data(), V.GstrandBit + 1)};
    std::vector<uint32_t> oldSJind(oldSjdbN, 0);
    std::vector<int64_t> sjdbInd(2 * sjdbN);
    uint64_t sjNew2 = 0;
    for (uint64_t isj = 0; isj < 2 * sjdbN; isj++) {
        uint64_t isj1 = isj < sjdbN ? isj : 2 * sjdbN - 1 - isj;
        sjdbInd[isj] = oldSjdbN == 0 ? -1 : binarySearch2(nStart[isj1], nEnd[isj1], oldStart, oldEnd);
        if (sjdbInd[isj] < 0) sjNew2++; else oldSJind[sjdbInd[isj]] = (uint32_t)isj1;
    }
    const uint64_t sjNew = sjNew2 / 2;
    const uint64_t nGenomeNew = nGenomeReal + nGsj;
    {
        uint32_t bit1 = (uint32_t)std::floor(std::log((double)nGenomeNew) / std::log(2.0)) + 1;
        if (bit1 < 32) bit1 = 32;
        if (bit1 > V.GstrandBit)
            return "EXITING because of FATAL ERROR: cannot insert junctions on the fly because of strand GstrandBit problem\nSOLUTION: please contact STAR author at https://groups.google.com/forum/#!forum/rna-star\n";
    }
    const uint32_t wSA = V.GstrandBit + 1;
    std::vector<uint8_t> SA2v; uint64_t nSAnew = 0;
    const bool resident = g_sjdbResidentFn != nullptr && gi.engineHoldsIndex;
    const bool hostArrays = !resident || P.sjdbInsertSaveAll || P.runModeGenerate;      // does the host need the new SA / SAindex at all?
    if (g_sjdbDeviceFn || resident) {
        // ---- on the device: search, sort, merge and SAindex (star_amd/csrc/index/sjdb_core.h)
        std::vector<uint8_t> isOld(sjdbN);
        for (uint64_t isj = 0; isj < sjdbN; isj++) isOld[isj] = sjdbInd[isj] >= 0;
        uint64_t nInd = 0;
        for (uint64_t isj = 0; isj < 2 * sjdbN; isj++) {
            if (sjdbInd[isj] >= 0) continue;
            const uint8_t *q = Gsj.data() + isj * sjdbLength;
            for (uint64_t k = 0; k < sjdbLength; k++) nInd += q[k] < 4;
        }
        nSAnew = oldNSA + nInd;
        if (hostArrays) SA2v.assign(Packed::lengthByte(nSAnew + 1, wSA) + 16, 0);
        std::vector<uint8_t> SAiNew(hostArrays ? V.nSAibyte + 24 : 0, 0);
        staramd_sjdb_args a; memset(&a, 0, sizeof(a));
        a.G = gi.G.data(); a.nGenomeOld = oldNGenome; a.nGenomeReal = nGenomeReal;
        a.SA = gi.SA.data(); a.nSAold = oldNSA; a.nSAbyteOld = V.nSAbyte; a.GstrandBit = V.GstrandBit; a.gSAindexNbases = V.gSAindexNbases;
        a.Gsj = Gsj.data(); a.sjdbN = (uint32_t)sjdbN; a.sjdbLength = (uint32_t)sjdbLength;
        a.isOld = isOld.data(); a.oldSJind = oldSJind.data(); a.oldSjdbN = (uint32_t)oldSjdbN; a.sjNew = sjNew;
        a.SAout = hostArrays ? SA2v.data() : nullptr; a.saOutCapacity = SA2v.size(); a.SAiOut = hostArrays ? SAiNew.data() : nullptr; a.saiOutCapacity = SAiNew.size();
        staramd_sjdb_result r;
        int rc = resident ? g_sjdbResidentFn(g_sjdbResidentUser, &a, &r) : g_sjdbDeviceFn(g_sjdbDevice, &a, &r);
        if (rc) return "EXITING because of FATAL ERROR: junction insertion on the MI355X failed (code " + std::to_string(rc) + ")";
        if (r.nInd != nInd || r.nSAibyte != V.nSAibyte) return "EXITING because of FATAL ERROR: junction insertion on the MI355X returned an index of unexpected size";
        if (hostArrays) { SAiNew.resize(V.nSAibyte + 8); gi.SAi.swap(SAiNew); }
        gi.indexInEngine = resident;
        lap("device insert");
        log += "   Finished SA search: number of new junctions=" + std::to_string(sjNew) + ", old junctions=" + std::to_string(sjdbN - sjNew) + "\n";
    } else {
    typedef std::array<uint64_t, 2> T2;                                   // (insertion point in the old SA, offset in Gsj)
    std::vector<T2> ind;
    {
        int T = std::max(1, std::min(P.runThreadN, 256));
        std::vector<std::vector<T2> > part(T);
        std::atomic<uint64_t> next(0);
        auto work = [&](int t) {
            for (;;) {
                uint64_t lo = next.fetch_add(64), hi = std::min<uint64_t>(lo + 64, 2 * sjdbN);
                if (lo >= 2 * sjdbN) break;
                for (uint64_t isj = lo; isj < hi; isj++) {
                    if (sjdbInd[isj] >= 0) continue;                      // no new suffixes for junctions that are already in the index
                    for (uint64_t istart = 0; istart < sjdbLength; istart++) {
                        uint64_t off = isj * sjdbLength + istart;
                        if (Gsj[off] > 3) continue;                       // nor for suffixes that start with N / spacer
                        part[t].push_back({suffixArraySearch1(X, Gsj.data() + off, G1c.data() + off), off});
                    }
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        size_t tot = 0; for (auto &p : part) tot += p.size();
        ind.reserve(tot + 1);
        for (auto &p : part) ind.insert(ind.end(), p.begin(), p.end());
    }
    lap("search");
    log += "   Finished SA search: number of new junctions=" + std::to_string(sjNew) + ", old junctions=" + std::to_string(sjdbN - sjNew) + "\n";
    const uint8_t *gs = Gsj.data();
    std::sort(ind.begin(), ind.end(), [gs](const T2 &a, const T2 &b) {      // funCompareUintAndSuffixes: a total order
        if (a[0] != b[0]) return a[0] < b[0];
        const uint8_t *ga = gs + a[1], *gb = gs + b[1];
        for (uint64_t ig = 0;; ig++) {
            if (ga[ig] != gb[ig]) return ga[ig] < gb[ig];
            if (ga[ig] == SPACER) return a[1] < b[1];
        }
    });
    lap("sort");
    const uint64_t nInd = ind.size();
    ind.push_back({(uint64_t)-999ll, (uint64_t)-999ll});                  // sentinel (:103-104)
    nSAnew = oldNSA + nInd;
    SA2v.assign(Packed::lengthByte(nSAnew + 1, wSA) + 8, 0);
    Packed SA2(SA2v.data(), wSA);
    const uint64_t nGsjNew = sjNew * sjdbLength, N2bit = 1ull << V.GstrandBit, strandMask = ~N2bit;
    {
        auto sjEntry = [&](uint64_t off) { return off < nGsj ? off + nGenomeReal : ((off - nGsj) | N2bit); };
        auto oldEntry = [&](uint64_t isa) {
            uint64_t ind1 = X.SA.get(isa);
            if (ind1 & N2bit) {
                uint64_t ind1s = oldNGenome - (ind1 & strandMask);
                if (ind1s >= nGenomeReal) {                               // an old junction: its index may have moved
                    uint64_t sj1 = (ind1s - nGenomeReal) / sjdbLength;
                    ind1s += ((uint64_t)oldSJind[sj1] - sj1) * sjdbLength;
                    ind1 = (nGenomeNew - ind1s) | N2bit;
                } else ind1 += nGsjNew;
            } else if (ind1 >= nGenomeReal) {
                uint64_t sj1 = (ind1 - nGenomeReal) / sjdbLength;
                ind1 += ((uint64_t)oldSJind[sj1] - sj1) * sjdbLength;
            }
            return ind1;
        };
        // The merge (:141-196) is a single pass in the reference.  Here the old SA is cut into slices that are merged by threads:
        // the inserts in front of old entry `isa` are those with insertion point == isa, so a slice that starts at old index a
        // starts at output position a + (number of inserts with insertion point < a).  Packed entries of neighbouring slices
        // share 64-bit words, so the few entries next to a slice boundary are written after the threads have joined.
        const uint64_t sliceMin = getenv("STARAMD_SJDB_SLICE") ? std::max<uint64_t>(16, strtoull(getenv("STARAMD_SJDB_SLICE"), nullptr, 10)) : 1000000;
        const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(1, std::min(P.runThreadN, 64)), oldNSA / sliceMin + 1));
        std::vector<uint64_t> cutA(T + 1), cutJ(T + 1);
        for (int t = 0; t <= T; t++) {
            cutA[t] = t == T ? oldNSA : oldNSA / T * t;
            cutJ[t] = (uint64_t)(std::lower_bound(ind.begin(), ind.begin() + nInd, cutA[t], [](const T2 &x, uint64_t v) { return x[0] < v; }) - ind.begin());
        }
        std::vector<std::vector<T2> > deferred(T);
        auto slice = [&](int t) {
            uint64_t isj = cutJ[t], isa2 = cutA[t] + cutJ[t];
            const uint64_t outLo = isa2, outHi = cutA[t + 1] + cutJ[t + 1];
            auto emit = [&](uint64_t v) { if (isa2 < outLo + 3 || isa2 + 3 >= outHi) deferred[t].push_back({isa2, v}); else SA2.put(isa2, v); isa2++; };
            for (uint64_t isa = cutA[t]; isa < cutA[t + 1]; isa++) {
                while (isa == ind[isj][0]) { emit(sjEntry(ind[isj][1])); ++isj; }
                emit(oldEntry(isa));
            }
        };
        {
            std::vector<std::thread> th;
            for (int t = 1; t < T; t++) th.emplace_back(slice, t);
            slice(0);
            for (auto &x : th) x.join();
        }
        for (auto &dv : deferred) for (const T2 &e : dv) SA2.put(e[0], e[1]);
        uint64_t isa2 = oldNSA + cutJ[T];
        for (uint64_t isj = cutJ[T]; isj < nInd; isj++) SA2.put(isa2++, sjEntry(ind[isj][1]));     // suffixes larger than every old one
    }
    lap("SA merge");
    // SAi (:209-284)
    const uint32_t nb = V.gSAindexNbases, wSAi = V.GstrandBit + 3;
    const uint64_t absentC = 1ull << (V.GstrandBit + 2), NmaskC = 1ull << (V.GstrandBit + 1);        // Genome_genomeLoad.cpp:157-169
    const uint64_t absentMask = ~absentC, Nmask = ~NmaskC;
    Packed SAi(gi.SAi.data(), wSAi);
    const uint64_t *st = V.genomeSAindexStart;
    auto calc = [&](uint64_t iSJ, uint32_t iL) -> int64_t {
        if (iSJ >= nInd) return INT64_MIN;       // the reference reads past its arrays here; the value it gets is compared, never kept
        return funCalcSAi(gs + ind[iSJ][1], iL);
    };
    for (uint32_t iL = 0; iL < nb; iL++) {
        uint64_t iSJ = 0, ind0 = st[iL] - 1;
        for (uint64_t ii = st[iL]; ii < st[iL + 1]; ii++) {
            uint64_t iSA1 = SAi.get(ii), iSA2 = iSA1 & Nmask & absentMask;
            if (iSJ < nInd && (iSA1 & absentC) > 0) {                     // prefix absent from the old genome
                uint64_t iSJ1 = iSJ;
                int64_t ind1 = calc(iSJ, iL);
                while (ind1 < (int64_t)(ii - st[iL]) && ind[iSJ][0] - 1 < iSA2) { ++iSJ; ind1 = calc(iSJ, iL); }
                if (ind1 == (int64_t)(ii - st[iL])) {
                    uint64_t v = ind[iSJ][0] - 1 + iSJ + 1;
                    SAi.put(ii, v);
                    for (uint64_t ii0 = ind0 + 1; ii0 < ii; ii0++) SAi.put(ii0, v | absentC);
                    ++iSJ; ind0 = ii;
                } else iSJ = iSJ1;
            } else {
                while (iSJ < nInd && ind[iSJ][0] - 1 + 1 < iSA2) ++iSJ;
                while (iSJ < nInd && ind[iSJ][0] - 1 + 1 == iSA2) {
                    if (funCalcSAi(gs + ind[iSJ][1], iL) >= (int64_t)(ii - st[iL])) break;
                    ++iSJ;
                }
                SAi.put(ii, iSA1 + iSJ);
                for (uint64_t ii0 = ind0 + 1; ii0 < ii; ii0++) SAi.put(ii0, (iSA2 + iSJ) | absentC);
                ind0 = ii;
            }
        }
    }
    for (uint64_t isj = 0; isj < nInd; isj++) {
        int64_t ind1 = 0;
        for (uint32_t iL = 0; iL < nb; iL++) {
            uint32_t g = gs[ind[isj][1] + iL];
            ind1 <<= 2;
            if (g > 3) {
                for (uint32_t iL1 = iL; iL1 < nb; iL1++) {
                    ind1 += 3;
                    int64_t ind2 = (int64_t)st[iL1] + ind1;
                    for (; ind2 >= 0; ind2--) if ((SAi.get((uint64_t)ind2) & absentC) == 0) break;
                    if (ind2 >= 0) SAi.put((uint64_t)ind2, SAi.get((uint64_t)ind2) | NmaskC);
                    ind1 <<= 2;
                }
                break;
            } else ind1 += g;
        }
    }
    lap("SAi");
    }
    // ---------------- the index is now the new one
    // the chromosomes stay where they are; the inserted sequences behind them are replaced (capacity for later insertions is taken once)
    if (gi.G.capacity() < nGenomeNew) gi.G.reserve(nGenomeNew + (uint64_t)P.limitSjdbInsertNsj * sjdbLength / 4);
    gi.G.resize(nGenomeNew);
    memcpy(gi.G.data() + nGenomeReal, Gsj.data(), nGsj);
    if (!SA2v.empty()) {
        { Packed SA2f(SA2v.data(), wSA); SA2f.put(nSAnew, 0); }            // sjdbInsertJunctions.cpp:66-68
        SA2v.resize(Packed::lengthByte(nSAnew, wSA) + 8);
        gi.SA.swap(SA2v);
    } else { std::vector<uint8_t>().swap(gi.SA); std::vector<uint8_t>().swap(gi.SAi); }      // the arrays live in HBM only (staramd_insert_junctions)
    gi.sjdbStart.swap(nStart); gi.sjdbEnd.swap(nEnd); gi.sjdbMotif.swap(nMotif); gi.sjdbShiftLeft.swap(nShL); gi.sjdbShiftRight.swap(nShR); gi.sjdbStrand.swap(nStrand);
    gi.sjDstart.swap(nD); gi.sjAstart.swap(nA);
    V.sjdbN = (uint32_t)sjdbN; V.sjGstart = nGenomeReal;
    V.nSA = nSAnew; V.nSAbyte = Packed::lengthByte(nSAnew, wSA);
    gi.refreshView();
    lap("new index");
    log += "Genome size with junctions=" + std::to_string(nGenomeNew) + "  " + std::to_string(nGenomeReal) + "   " + std::to_string(nGsj) + "\n";
    return "";
}

static bool copyFile(const std::string &a, const std::string &b) {
    std::ifstream in(a, std::ios::binary); std::ofstream out(b, std::ios::binary);
    if (!in.good() || !out.good()) return false;
    out << in.rdbuf();
    return true;
}

std::string sjdbInsertJunctions(RunParams &P, GenomeIndex &gi, SjdbLoci &loci, bool pass2, const std::string &pass1sjFile, std::string &log) {
    const std::string &outDir = P.sjdbInsertOutDir;
    const bool timing = getenv("STARAMD_HOST_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (timing) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  sjdbInsertJunctions: %-18s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count()); T0 = t; } };
    if (gi.view.sjdbN > 0 && loci.chr.empty()) {                           // junctions of the generated genome, once
        std::ifstream in(gi.dir + "/sjdbList.out.tab");
        if (!in.good()) return "EXITING because of fatal INPUT error: could not open " + gi.dir + "/sjdbList.out.tab\nSOLUTION: re-generate the genome in " + gi.dir;
        sjdbLoadFromStream(in, loci);
        loci.priority.resize(loci.chr.size(), 30);
    }
    if (pass2) {
        std::ifstream in(pass1sjFile);
        if (in.fail()) return "FATAL INPUT error, could not open input file with junctions from the 1st pass=" + pass1sjFile + "\n";
        sjdbLoadFromStream(in, loci);
        loci.priority.resize(loci.chr.size(), 0);
    } else {
        for (const std::string &f : P.sjdbFileChrStartEnd) {               // sjdbLoadFromFiles.cpp:6-26
            std::ifstream in(f);
            if (in.fail()) return "FATAL INPUT error, could not open input file pGe.sjdbFileChrStartEnd=" + f + "\n";
            sjdbLoadFromStream(in, loci);
            loci.priority.resize(loci.chr.size(), 10);
        }
        std::string e = loadGTFjunctions(P, gi, loci, outDir, log);        // GTF gtf(...); gtf.transcriptGeneSJ(...) (:45-46)
        if (!e.empty()) return e;
    }
    lap("junction tables read");
    std::string err = sjdbPrepareAndBuild(P, gi, loci, outDir, log);
    if (!err.empty()) return err;
    lap("prepare + build");
    if (P.sjdbInsertSaveAll) {                                             // sjdbInsertJunctions.cpp:70-98
        if (gi.dir != outDir)
            for (const char *f : {"chrName.txt", "chrStart.txt", "chrNameLength.txt", "chrLength.txt"}) copyFile(gi.dir + "/" + f, outDir + "/" + f);
        {
            std::ifstream in(gi.dir + "/genomeParameters.txt"); std::ofstream out(outDir + "/genomeParameters.txt");
            out << "### " << P.commandLine << "\n";
            std::string l;
            while (std::getline(in, l)) {
                if (l.compare(0, 4, "### ") == 0 && l.find("GstrandBit") == std::string::npos) continue;
                std::string k = l.substr(0, l.find('\t'));
                if (k == "sjdbOverhang") out << "sjdbOverhang\t" << gi.view.sjdbOverhang << "\n";
                else if (k == "sjdbInsertSave") out << "sjdbInsertSave\tAll\n";
                else if (k == "genomeFileSizes") out << "genomeFileSizes\t" << gi.view.nGenome << " " << gi.view.nSAbyte << "\n";
                else out << l << "\n";
            }
        }
        auto dump = [&](const std::string &name, const void *p, uint64_t n, const void *hdr = nullptr, uint64_t nh = 0) {
            FILE *f = fopen((outDir + "/" + name).c_str(), "wb");
            if (!f) return false;
            if (nh) fwrite(hdr, 1, nh, f);
            fwrite(p, 1, n, f);
            fclose(f);
            return true;
        };
        std::vector<uint64_t> hdr(1, gi.view.gSAindexNbases);
        for (uint32_t i = 0; i <= gi.view.gSAindexNbases; i++) hdr.push_back(gi.view.genomeSAindexStart[i]);
        if (gi.SA.size() < gi.view.nSAbyte) return "EXITING because of fatal ERROR: --sjdbInsertSave All needs the suffix array on the host, which was released after the upload";
        if (!dump("Genome", gi.G.data(), gi.view.nGenome) || !dump("SA", gi.SA.data(), gi.view.nSAbyte) ||
            !dump("SAindex", gi.SAi.data(), gi.view.nSAibyte, hdr.data(), hdr.size() * 8))
            return "EXITING because of fatal ERROR: could not write the genome files into " + outDir;
    }
    P.dev.winBinN = gi.view.nGenome / (1ull << P.dev.winBinNbits) + 1;    // sjdbInsertJunctions.cpp:101
    return "";
}

std::string makeRunDir(const std::string &d, bool allRWX) {               // Parameters.cpp:817-824, 1027-1034; --runDirPerm User_RWX | All_RWX (:471-481)
    removeDirRecursive(d);
    if (mkdir(d.c_str(), allRWX ? (S_IRWXU | S_IRWXG | S_IRWXO) : S_IRWXU) != 0) return "EXITING because of fatal ERROR: could not make run-time directory: " + d + "\nSOLUTION: please check the path and writing permissions \n";
    return "";
}

} // namespace staramd
