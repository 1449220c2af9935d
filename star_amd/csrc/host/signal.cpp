// signal.cpp -- --outWigType bedGraph | wiggle [read1_5p | read2], --outWigStrand, --outWigNorm, --outWigReferencesPrefix: read coverage ("signal")
// from the coordinate-sorted alignments, written at the end of an alignReads run as Signal.{Unique,UniqueMultiple}.str{1,2}.out.{bg,wig}.
//   signalFromBAM     source/signalFromBAM.cpp:5-209   (the reference re-reads its sorted BAM; here the sorted records are still in memory)
#include "host.h"
#include <cstring>
#include <cstdio>
#include <zlib.h>

namespace staramd {

namespace {
inline uint32_t rd32(const char *p) { uint32_t v; memcpy(&v, p, 4); return v; }

// integer value of an attribute of a BAM record (bam_aux_get + bam_aux2i); false when the tag is absent
bool auxInt(const char *rec, const char tag[2], int64_t &val) {
    const uint32_t recSize = rd32(rec), bmn = rd32(rec + 12), fnc = rd32(rec + 16), lseq = rd32(rec + 20);
    const char *a = rec + 36 + (bmn & 0xff) + 4 * (size_t)(fnc & 0xffff) + (lseq + 1) / 2 + lseq, *end = rec + 4 + recSize;
    while (a + 3 <= end) {
        const char t = a[2]; const bool hit = a[0] == tag[0] && a[1] == tag[1];
        a += 3;
        size_t w;
        switch (t) {
            case 'A': case 'c': case 'C': w = 1; break;
            case 's': case 'S': w = 2; break;
            case 'i': case 'I': case 'f': w = 4; break;
            case 'Z': case 'H': w = strlen(a) + 1; break;
            case 'B': { const char bt = a[0]; const uint32_t n = rd32(a + 1); w = 5 + (size_t)n * (bt == 'c' || bt == 'C' ? 1 : bt == 's' || bt == 'S' ? 2 : 4); break; }
            default: return false;
        }
        if (hit) {
            if (t == 'c') val = (int8_t)a[0]; else if (t == 'C') val = (uint8_t)a[0];
            else if (t == 's') { int16_t v; memcpy(&v, a, 2); val = v; } else if (t == 'S') { uint16_t v; memcpy(&v, a, 2); val = v; }
            else if (t == 'i') { int32_t v; memcpy(&v, a, 4); val = v; } else if (t == 'I') { uint32_t v; memcpy(&v, a, 4); val = v; }
            else val = 0;
            return true;
        }
        a += w;
    }
    return false;
}
}

std::string writeSignal(const RunParams &P, const std::vector<std::string> &chrName, const std::vector<uint64_t> &chrLength, const std::string &sigFileName, const std::vector<const char *> &recs) {
    const WigParams &W = P.wig;
    auto wanted = [&](int32_t tid) { return W.referencesPrefix.empty() || chrName[tid].compare(0, W.referencesPrefix.size(), W.referencesPrefix) == 0; };
    double nMult = 0, nUniq = 0;
    if (W.norm == 1) {
        for (const char *rec : recs) {
            const int32_t tid = (int32_t)rd32(rec + 4);
            if (tid < 0 || !wanted(tid)) continue;
            int64_t nh;
            if (auxInt(rec, "NH", nh)) { if ((uint32_t)nh == 1) ++nUniq; else if ((uint32_t)nh > 1) nMult += 1.0 / (uint32_t)nh; }
        }
    }
    const int sigN = W.strand ? 4 : 2;
    double normFactor[4] = {1, 1, 1, 1};
    std::string names[4] = {sigFileName + ".Unique.str1.out", sigFileName + ".UniqueMultiple.str1.out", sigFileName + ".Unique.str2.out", sigFileName + ".UniqueMultiple.str2.out"};
    FILE *out[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int ii = 0; ii < sigN; ii++) {
        names[ii] += W.format == 0 ? ".bg" : ".wig";
        out[ii] = fopen(names[ii].c_str(), "wb");
        if (!out[ii]) { for (int k = 0; k < ii; k++) fclose(out[k]); return "EXITING because of fatal ERROR: could not create output file " + names[ii]; }
    }
    if (W.norm == 1) { normFactor[0] = 1.0e6 / nUniq; normFactor[1] = 1.0e6 / (nUniq + nMult); }
    if (W.strand) { normFactor[2] = normFactor[0]; normFactor[3] = normFactor[1]; }
    const char *numFormat = W.norm == 1 ? "%.5f" : "%g";      // ostream default vs fixed << setprecision(5) (:64-66)
    auto num = [&](FILE *f, double v) { fprintf(f, numFormat, v); };

    int32_t iChr = -999; std::vector<double> sigAll; uint32_t chrLen = 0;
    auto flushChr = [&]() {
        if (iChr == -999) return;
        for (int is = 0; is < sigN; is++) {
            FILE *f = out[is];
            if (W.format == 1) fprintf(f, "variableStep chrom=%s\n", chrName[iChr].c_str());
            double prevSig = 0;
            for (uint32_t ig = 0; ig < chrLen; ig++) {
                const double newSig = sigAll[(size_t)sigN * ig + is];
                if (W.format == 0) {
                    if (newSig != prevSig) {
                        if (prevSig != 0) { fprintf(f, "%u\t", ig); num(f, prevSig * normFactor[is]); fputc('\n', f); }
                        if (newSig != 0) fprintf(f, "%s\t%u\t", chrName[iChr].c_str(), ig);
                        prevSig = newSig;
                    }
                } else if (newSig != 0) { fprintf(f, "%u\t", ig + 1); num(f, newSig * normFactor[is]); fputc('\n', f); }
            }
        }
    };
    for (size_t ir = 0; ir <= recs.size(); ir++) {
        const char *rec = ir < recs.size() ? recs[ir] : nullptr;
        const int32_t tid = rec ? (int32_t)rd32(rec + 4) : -1;
        if (!rec || tid != iChr) {
            flushChr();
            if (!rec) break;
            iChr = tid;
            if (iChr == -1 || !wanted(iChr)) { iChr = -999; continue; }
            chrLen = (uint32_t)chrLength[iChr] + 1;                 // one extra base at the end, always 0
            sigAll.assign((size_t)sigN * chrLen, 0.0);
        }
        if (iChr == -999) continue;
        const uint32_t bmn = rd32(rec + 12), fnc = rd32(rec + 16), flag = fnc >> 16, nCigar = fnc & 0xffff;
        if (flag & 0x400) continue;
        int64_t nh; uint32_t aNH = 1;
        if (auxInt(rec, "NH", nh)) aNH = (uint32_t)nh;
        if (aNH == 0) continue;
        uint32_t aG = rd32(rec + 8), iStrand = 0;
        if (W.strand) iStrand = ((flag & 0x10) > 0) == ((flag & 0x80) == 0);
        if (W.type == 1) {
            if (flag & 0x80) continue;
            if (iStrand == 0) { if (aNH == 1) sigAll[(size_t)aG * sigN + 0 + 2 * iStrand]++; sigAll[(size_t)aG * sigN + 1 + 2 * iStrand] += 1.0 / aNH; continue; }
        }
        const char *cigar = rec + 36 + (bmn & 0xff);
        for (uint32_t ic = 0; ic < nCigar; ic++) {
            const uint32_t c = rd32(cigar + 4 * ic), op = c & 0xf, len = c >> 4;
            if (op == 2 || op == 3) aG += len;
            else if (op == 0) {
                if (W.type == 0 || (W.type == 2 && (flag & 0x80))) {
                    for (uint32_t ig = 0; ig < len; ig++) {
                        if (aG >= chrLen) { for (int k = 0; k < sigN; k++) fclose(out[k]); return "BUG: alignment extends past chromosome in the signal output"; }
                        if (aNH == 1) sigAll[(size_t)aG * sigN + 0 + 2 * iStrand]++;
                        sigAll[(size_t)aG * sigN + 1 + 2 * iStrand] += 1.0 / aNH;
                        aG++;
                    }
                } else aG += len;
            }
        }
        if (W.type == 1) { --aG; if (aNH == 1) sigAll[(size_t)aG * sigN + 0 + 2 * iStrand]++; sigAll[(size_t)aG * sigN + 1 + 2 * iStrand] += 1.0 / aNH; }
    }
    for (int is = 0; is < sigN; is++) fclose(out[is]);
    return "";
}

// --runMode inputAlignmentsFromBAM --inputBAMfile x.bam --outWigType ... (Parameters.cpp:585-592): the same tracks from a coordinate-sorted BAM file.
// The file is inflated block by block (BGZF = concatenated gzip members) and held in memory; chromosome names and lengths come from its header.
std::string signalFromBamFile(const RunParams &P, const std::string &bamPath, const std::string &sigFileName) {
    gzFile in = gzopen(bamPath.c_str(), "rb");
    if (!in) return "EXITING because of fatal INPUT error: could not open --inputBAMfile " + bamPath;
    std::vector<char> d;
    for (;;) {
        const size_t old = d.size(), block = 1u << 24;
        d.resize(old + block);
        int got = gzread(in, d.data() + old, (unsigned)block);
        if (got < 0) { gzclose(in); return "EXITING because of fatal INPUT error: could not decompress --inputBAMfile " + bamPath; }
        d.resize(old + (size_t)got);
        if ((size_t)got < block) break;
    }
    gzclose(in);
    if (d.size() < 12 || memcmp(d.data(), "BAM\1", 4) != 0) return "EXITING because of fatal INPUT error: --inputBAMfile " + bamPath + " is not a BAM file";
    size_t p = 8 + rd32(d.data() + 4);
    if (p + 4 > d.size()) return "EXITING because of fatal INPUT error: truncated BAM header in " + bamPath;
    const uint32_t nRef = rd32(d.data() + p); p += 4;
    std::vector<std::string> chrName; std::vector<uint64_t> chrLength;
    for (uint32_t i = 0; i < nRef; i++) {
        if (p + 4 > d.size()) return "EXITING because of fatal INPUT error: truncated BAM header in " + bamPath;
        const uint32_t ln = rd32(d.data() + p); p += 4;
        if (p + ln + 4 > d.size()) return "EXITING because of fatal INPUT error: truncated BAM header in " + bamPath;
        chrName.emplace_back(d.data() + p, ln ? ln - 1 : 0); p += ln;
        chrLength.push_back(rd32(d.data() + p)); p += 4;
    }
    std::vector<const char *> recs;
    while (p + 4 <= d.size()) {
        const uint32_t bs = rd32(d.data() + p);
        if (p + 4 + bs > d.size()) break;
        recs.push_back(d.data() + p);
        p += 4 + bs;
    }
    return writeSignal(P, chrName, chrLength, sigFileName, recs);
}

} // namespace staramd
