// sjout_stats.cpp -- junction collapse / filter / SJ.out.tab and Log.final.out.
//   OutSJ::collapseSJ, Junction::collapseOneSJ, Junction::outputStream   source/OutSJ.cpp:42-123
//   outputSJ (filters, two-sided distance filter, file)                  source/outputSJ.cpp:20-138
//   Stats::reportFinal                                                   source/Stats.cpp:99-145
#include "host.h"
#include <algorithm>
#include <fstream>
#include <iomanip>
#include <ctime>
#include <cstdio>
#include <functional>
#include <thread>

namespace staramd {

static inline bool sjLess(const Junction &a, const Junction &b) { return a.start != b.start ? a.start < b.start : a.gap < b.gap; }

// records of one (start, gap) folded into the first: counts add up, overhangs take the maximum (Junction::collapseOneSJ, OutSJ.cpp:74-98);
// strand / motif / annot are functions of the locus, so the order among equal keys does not matter
static size_t foldSorted(Junction *d, size_t n) {
    if (n == 0) return 0;
    size_t k = 0;
    for (size_t i = 1; i < n; i++) {
        if (d[i].start == d[k].start && d[i].gap == d[k].gap) {
            d[k].countUnique += d[i].countUnique; d[k].countMultiple += d[i].countMultiple;
            if (d[k].overhangLeft < d[i].overhangLeft) d[k].overhangLeft = d[i].overhangLeft;
            if (d[k].overhangRight < d[i].overhangRight) d[k].overhangRight = d[i].overhangRight;
        } else { ++k; if (k != i) d[k] = d[i]; }
    }
    return k + 1;
}

// OutSJ::collapseSJ (OutSJ.cpp:42-72).  A run keeps millions of records between two collapses (one per junction per read); they are
// split into key ranges by sampled splitters, every range is sorted and folded by its own thread, the folded ranges are concatenated.
void OutSJ::collapse() {
    const size_t n = data.size();
    if (n == 0) return;
    unsigned T = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n < 200000 || T < 2) { std::sort(data.begin(), data.end(), sjLess); data.resize(foldSorted(data.data(), n)); return; }
    std::vector<Junction> sample;
    const size_t nSample = 64 * T, step = n / nSample;
    for (size_t i = 0; i < nSample; i++) sample.push_back(data[i * step]);
    std::sort(sample.begin(), sample.end(), sjLess);
    std::vector<Junction> split;                                  // T-1 splitters; equal keys never straddle two ranges
    for (unsigned t = 1; t < T; t++) split.push_back(sample[t * 64]);
    auto rangeOf = [&](const Junction &j) { return (unsigned)(std::upper_bound(split.begin(), split.end(), j, sjLess) - split.begin()); };
    // counting scatter, the input cut into T slices
    std::vector<std::vector<size_t>> cnt(T, std::vector<size_t>(T + 1, 0));
    std::vector<uint8_t> rng(n);
    auto each = [&](const std::function<void(unsigned)> &f) { std::vector<std::thread> th; for (unsigned t = 1; t < T; t++) th.emplace_back(f, t); f(0); for (auto &x : th) x.join(); };
    each([&](unsigned t) { std::vector<size_t> c(T + 1, 0); for (size_t i = n * t / T; i < n * (t + 1) / T; i++) { unsigned r = rangeOf(data[i]); rng[i] = (uint8_t)r; c[r]++; } cnt[t] = c; });
    std::vector<size_t> rangeStart(T + 1, 0);
    for (unsigned r = 0; r < T; r++) { size_t tot = 0; for (unsigned t = 0; t < T; t++) { size_t c = cnt[t][r]; cnt[t][r] = rangeStart[r] + tot; tot += c; } rangeStart[r + 1] = rangeStart[r] + tot; }
    std::vector<Junction> tmp(n);
    each([&](unsigned t) { std::vector<size_t> pos = cnt[t]; for (size_t i = n * t / T; i < n * (t + 1) / T; i++) tmp[pos[rng[i]]++] = data[i]; });
    std::vector<size_t> folded(T, 0);
    each([&](unsigned r) { Junction *d = tmp.data() + rangeStart[r]; size_t m = rangeStart[r + 1] - rangeStart[r]; std::sort(d, d + m, sjLess); folded[r] = foldSorted(d, m); });
    size_t k = 0;
    for (unsigned r = 0; r < T; r++) { std::copy(tmp.begin() + rangeStart[r], tmp.begin() + rangeStart[r] + folded[r], data.begin() + k); k += folded[r]; }
    data.resize(k);
}

// outputSJ.cpp:56-120: collapse, per-junction filter, then (unless `skipDistanceFilter`, 2nd stage of BySJout :84) the distance
// to other junctions' donors / acceptors
std::vector<Junction> OutSJ::filtered(const RunParams &P, bool skipDistanceFilter) {
    collapse();
    // per-junction filter (outputSJ.cpp:59-65)
    std::vector<Junction> all;
    for (const Junction &j : data) {
        int m = (j.motif + 1) / 2;
        uint32_t tot = j.countMultiple + j.countUnique;
        bool ok = j.annot > 0 ||
                  ((j.countUnique >= (uint32_t)P.outSJfilterCountUniqueMin[m] || tot >= (uint32_t)P.outSJfilterCountTotalMin[m])
                   && j.overhangLeft >= (uint32_t)P.outSJfilterOverhangMin[m] && j.overhangRight >= (uint32_t)P.outSJfilterOverhangMin[m]
                   && (tot > P.outSJfilterIntronMaxVsReadN.size() || j.gap <= (uint32_t)P.outSJfilterIntronMaxVsReadN[tot - 1]));
        if (ok) all.push_back(j);
    }
    if (skipDistanceFilter) return all;
    // distance to other junctions' donors / acceptors (:85-120)
    size_t N = all.size();
    std::vector<char> keep(N, 0);
    struct Acc { uint64_t a; uint64_t idx; uint64_t motif; };
    std::vector<Acc> sjA(N);
    for (size_t ii = 0; ii < N; ii++) {
        uint64_t x1 = 0, x2 = (uint64_t)-1;
        if (ii > 0) x1 = all[ii - 1].start;
        if (ii + 1 < N) x2 = all[ii + 1].start;
        uint64_t minDist = std::min(all[ii].start - x1, x2 - all[ii].start);
        keep[ii] = minDist >= (uint64_t)P.outSJfilterDistToOtherSJmin[(all[ii].motif + 1) / 2];
        sjA[ii].a = all[ii].start + (uint64_t)all[ii].gap; sjA[ii].idx = ii;
        sjA[ii].motif = all[ii].annot == 0 ? (uint64_t)all[ii].motif : 8;   // SJ_MOTIF_SIZE+1
    }
    // the reference qsorts triples by the acceptor only (compareUint): ties keep an unspecified order, which cannot
    // change the outcome because equal acceptors give minDist 0 for both members of the tie
    std::sort(sjA.begin(), sjA.end(), [](const Acc &x, const Acc &y) { return x.a != y.a ? x.a < y.a : x.idx < y.idx; });
    for (size_t ii = 0; ii < N; ii++) {
        if (sjA[ii].motif == 8) keep[sjA[ii].idx] = 1;
        else {
            uint64_t x1 = 0, x2 = (uint64_t)-1;
            if (ii > 0) x1 = sjA[ii - 1].a;
            if (ii + 1 < N) x2 = sjA[ii + 1].a;
            uint64_t minDist = std::min(sjA[ii].a - x1, x2 - sjA[ii].a);
            keep[sjA[ii].idx] = keep[sjA[ii].idx] && (minDist >= (uint64_t)P.outSJfilterDistToOtherSJmin[(sjA[ii].motif + 1) / 2]);
        }
    }
    std::vector<Junction> out;
    for (size_t ii = 0; ii < N; ii++) if (keep[ii]) out.push_back(all[ii]);
    return out;
}

// P.sjNovelStart/End after the 1st stage of --outFilterType BySJout (outputSJ.cpp:139-161): unannotated junctions that pass
void OutSJ::novelWhitelist(const RunParams &P, std::vector<uint64_t> &start, std::vector<uint64_t> &end) {
    start.clear(); end.clear();
    for (const Junction &j : filtered(P, false)) if (j.annot == 0) { start.push_back(j.start); end.push_back(j.start + (uint64_t)j.gap - 1); }
}

std::string OutSJ::filterAndWrite(const RunParams &P, const GenomeIndex &gi, const std::string &path, bool skipDistanceFilter) {
    std::vector<Junction> all = filtered(P, skipDistanceFilter);
    size_t N = all.size();
    std::vector<char> keep(N, 1);
    FILE *out = fopen(path.c_str(), "wb");
    if (!out) return "EXITING because of fatal ERROR: could not create output file " + path;
    // Junction::outputStream (OutSJ.cpp:100-123): decimal fields separated by tabs; formatted by hand into one buffer (half a million lines)
    std::string buf; buf.reserve(N * 48 + 64);
    char num[24];
    auto dec = [&](uint64_t v) { int k = 24; do { num[--k] = (char)('0' + v % 10); v /= 10; } while (v); buf.append(num + k, 24 - k); };
    auto sdec = [&](int v) { if (v < 0) { buf.push_back('-'); dec((uint64_t)(-(int64_t)v)); } else dec((uint64_t)v); };
    for (size_t ii = 0; ii < N; ii++) {
        if (!keep[ii]) continue;
        const Junction &j = all[ii];
        uint32_t c = gi.chrBin[j.start >> gi.view.gChrBinNbits];
        buf += gi.chrName.at(c); buf.push_back('\t'); dec(j.start + 1 - gi.chrStart[c]); buf.push_back('\t'); dec(j.start + j.gap - gi.chrStart[c]);
        buf.push_back('\t'); sdec(j.strand); buf.push_back('\t'); sdec(j.motif); buf.push_back('\t'); sdec(j.annot); buf.push_back('\t'); dec(j.countUnique);
        buf.push_back('\t'); dec(j.countMultiple); buf.push_back('\t'); dec(j.overhangLeft); buf.push_back('\n');
    }
    const bool ok = fwrite(buf.data(), 1, buf.size(), out) == buf.size();
    if (fclose(out) != 0 || !ok) return "EXITING because of fatal ERROR: could not write output file " + path;
    return "";
}

void Stats::add(const Stats &s) {
    readN += s.readN; readBases += s.readBases; mappedMismatchesN += s.mappedMismatchesN; mappedInsN += s.mappedInsN; mappedDelN += s.mappedDelN;
    mappedInsL += s.mappedInsL; mappedDelL += s.mappedDelL; mappedBases += s.mappedBases; mappedPortion += s.mappedPortion;
    mappedReadsU += s.mappedReadsU; mappedReadsM += s.mappedReadsM; unmappedOther += s.unmappedOther; unmappedShort += s.unmappedShort;
    unmappedMismatch += s.unmappedMismatch; unmappedMulti += s.unmappedMulti; unmappedAll += s.unmappedAll; chimericAll += s.chimericAll;
    splicesNsjdb += s.splicesNsjdb;
    for (int i = 0; i < 7; i++) splicesN[i] += s.splicesN[i];
}

static std::string timeMonthDayTime(time_t t) {      // TimeFunctions.cpp
    char buf[64]; strftime(buf, sizeof(buf), "%b %d %H:%M:%S", localtime(&t)); return buf;
}

void Stats::reportFinal(const std::string &path) {
    std::ofstream so(path.c_str());
    int w1 = 50;
    time(&timeFinish);
    so << std::setiosflags(std::ios::fixed) << std::setprecision(2)
       << std::setw(w1) << "Started job on |\t" << timeMonthDayTime(timeStart) << "\n"
       << std::setw(w1) << "Started mapping on |\t" << timeMonthDayTime(timeStartMap) << "\n"
       << std::setw(w1) << "Finished on |\t" << timeMonthDayTime(timeFinish) << "\n"
       << std::setw(w1) << "Mapping speed, Million of reads per hour |\t" << double(readN) / 1e6 / difftime(timeFinish, timeStartMap) * 3600 << "\n"
       << "\n"
       << std::setw(w1) << "Number of input reads |\t" << readN << "\n"
       << std::setw(w1) << "Average input read length |\t" << (readN > 0 ? readBases / readN : 0) << "\n"
       << std::setw(w1) << "UNIQUE READS:\n"
       << std::setw(w1) << "Uniquely mapped reads number |\t" << mappedReadsU << "\n"
       << std::setw(w1) << "Uniquely mapped reads % |\t" << (readN > 0 ? double(mappedReadsU) / double(readN) * 100 : 0) << '%' << "\n"
       << std::setw(w1) << "Average mapped length |\t" << (mappedReadsU > 0 ? double(mappedBases) / double(mappedReadsU) : 0) << "\n";
    so << std::setw(w1) << "Number of splices: Total |\t" << splicesN[0] + splicesN[1] + splicesN[2] + splicesN[3] + splicesN[4] + splicesN[5] + splicesN[6] << "\n"
       << std::setw(w1) << "Number of splices: Annotated (sjdb) |\t" << splicesNsjdb << "\n"
       << std::setw(w1) << "Number of splices: GT/AG |\t" << splicesN[1] + splicesN[2] << "\n"
       << std::setw(w1) << "Number of splices: GC/AG |\t" << splicesN[3] + splicesN[4] << "\n"
       << std::setw(w1) << "Number of splices: AT/AC |\t" << splicesN[5] + splicesN[6] << "\n"
       << std::setw(w1) << "Number of splices: Non-canonical |\t" << splicesN[0] << "\n";
    so << std::setw(w1) << "Mismatch rate per base, % |\t" << double(mappedMismatchesN) / double(mappedBases) * 100 << '%' << "\n"
       << std::setw(w1) << "Deletion rate per base |\t" << (mappedBases > 0 ? double(mappedDelL) / double(mappedBases) * 100 : 0) << '%' << "\n"
       << std::setw(w1) << "Deletion average length |\t" << (mappedDelN > 0 ? double(mappedDelL) / double(mappedDelN) : 0) << "\n"
       << std::setw(w1) << "Insertion rate per base |\t" << (mappedBases > 0 ? double(mappedInsL) / double(mappedBases) * 100 : 0) << '%' << "\n"
       << std::setw(w1) << "Insertion average length |\t" << (mappedInsN > 0 ? double(mappedInsL) / double(mappedInsN) : 0) << "\n"
       << std::setw(w1) << "MULTI-MAPPING READS:\n"
       << std::setw(w1) << "Number of reads mapped to multiple loci |\t" << mappedReadsM << "\n"
       << std::setw(w1) << "% of reads mapped to multiple loci |\t" << (readN > 0 ? double(mappedReadsM) / double(readN) * 100 : 0) << '%' << "\n"
       << std::setw(w1) << "Number of reads mapped to too many loci |\t" << unmappedMulti << "\n"
       << std::setw(w1) << "% of reads mapped to too many loci |\t" << (readN > 0 ? double(unmappedMulti) / double(readN) * 100 : 0) << '%' << "\n"
       << std::setw(w1) << "UNMAPPED READS:\n"
       << std::setw(w1) << "Number of reads unmapped: too many mismatches |\t" << unmappedMismatch << "\n"
       << std::setw(w1) << "% of reads unmapped: too many mismatches |\t" << (readN > 0 ? double(unmappedMismatch) / double(readN) * 100 : 0) << '%' << "\n"
       << std::setw(w1) << "Number of reads unmapped: too short |\t" << unmappedShort << "\n"
       << std::setw(w1) << "% of reads unmapped: too short |\t" << (readN > 0 ? double(unmappedShort) / double(readN) * 100 : 0) << '%' << "\n"
       << std::setw(w1) << "Number of reads unmapped: other |\t" << unmappedOther << "\n"
       << std::setw(w1) << "% of reads unmapped: other |\t" << (readN > 0 ? double(unmappedOther) / double(readN) * 100 : 0) << '%' << "\n"
       << std::setw(w1) << "CHIMERIC READS:\n"
       << std::setw(w1) << "Number of chimeric reads |\t" << chimericAll << "\n"
       << std::setw(w1) << "% of chimeric reads |\t" << (readN > 0 ? double(chimericAll) / double(readN) * 100 : 0) << '%' << "\n" << std::flush;
}

} // namespace staramd
