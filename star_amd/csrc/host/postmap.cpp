// postmap.cpp -- everything the reference does with the result of mapOneRead:
//   multMapSelect          source/ReadAlign_multMapSelect.cpp:8-95
//   mappedFilter           source/ReadAlign_mappedFilter.cpp:3-21
//   outputAlignments       source/ReadAlign_outputAlignments.cpp:5-72  (recordSJ :76-87, writeSAM :132-256)
//   outputTranscriptSAM    source/ReadAlign_outputTranscriptSAM.cpp:5-359
//   outputTranscriptSJ     source/ReadAlign_outputTranscriptSJ.cpp:4-56
//   Stats::transcriptStats source/Stats.cpp:35-56
// Host-side integer code; defines the parity surface (Aligned.out.sam, SJ.out.tab, Log.final.out).
#include "host.h"
#include <fstream>
#include <cstring>
#include <immintrin.h>
#include <algorithm>

namespace staramd {

namespace {
inline void appendUint(std::string &s, uint64_t v) {
    char buf[24]; int n = 0;
    do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) s.push_back(buf[--n]);
}
inline void appendInt(std::string &s, int64_t v) { if (v < 0) { s.push_back('-'); appendUint(s, (uint64_t)(-v)); } else appendUint(s, (uint64_t)v); }


struct TrView {                 // one candidate alignment = header + exon slice of the result arrays
    const staramd_transcript *t; const staramd_exon *ex;
    bool primary;
};

struct ReadCtx {
    const ReadBatch *b; uint32_t i;
    uint64_t Lread, readLength[2];
    uint64_t readLengthOriginal[2], clip[2][2];   // lengths before clipping; clip[mate][0 = 5', 1 = 3']
    int waspType = -1;                            // vW (--waspOutputMode SAMtag), -1: none
    int nMates;
    // bases soft-clipped on the left of the mate's alignment because of --clip* (ReadAlign_calcCIGAR.cpp:14-23)
    uint64_t trimL(uint32_t Str, uint32_t Mate) const { return clip[Mate][Str == Mate ? 0 : 1]; }
};
} // namespace

// a segment of a chimeric alignment (--chimOutType WithinBAM / SeparateSAMold): alignType -10 = the representative one, -11 / -12 = supplementary with the hard
// clip on the left / right, -13 = supplementary with soft clips; mateChr (> nChrReal: none), mateStart (0-based in the chromosome), mateStrand: the other segment
struct ChimBam { int alignType; uint32_t mateChr, mateStart; uint8_t mateStrand; const VarOverlap *var = nullptr; };

std::string PostMap::samHeader() const {                 // samHeaders.cpp:27-98
    std::string h;
    if (P.outSAMheaderHD.empty()) h = "@HD\tVN:1.4";
    else for (size_t i = 0; i < P.outSAMheaderHD.size(); i++) { if (i) h.push_back('\t'); h += P.outSAMheaderHD[i]; }
    h.push_back('\n');
    for (uint32_t i = 0; i < gi.view.nChrReal; i++) { h += "@SQ\tSN:" + gi.chrName[i] + "\tLN:"; appendUint(h, gi.chrLength[i]); h += "\n"; }
    if (!P.outSAMheaderPG.empty()) { for (size_t i = 0; i < P.outSAMheaderPG.size(); i++) { if (i) h.push_back('\t'); h += P.outSAMheaderPG[i]; } h.push_back('\n'); }
    h += "@PG\tID:STAR\tPN:STAR\tVN:2.7.11b\tCL:" + P.commandLine + "\n";
    if (!P.outSAMheaderCommentFile.empty()) {            // non-blank lines of the file, as they are
        std::ifstream com(P.outSAMheaderCommentFile.c_str());
        std::string line;
        while (std::getline(com, line)) if (line.find_first_not_of(" \t\n\v\f\r") != std::string::npos) h += line + "\n";
    }
    for (const std::string &rg : P.outSAMattrRGlineSplit) h += "@RG\t" + rg + "\n";                  // samHeaders.cpp:78-80
    h += "@CO\tuser command line: " + P.commandLine + "\n";
    return h;
}

// ---- ReadAlign::outputTranscriptSAM, mapped branch (:57-356) ----
// The text of a record is written through a raw pointer into space the output string is grown by once per alignment (an upper bound of both mates'
// lines), numbers two digits at a time, the reverse complement through a byte table: with std::string appends character by character this function
// was 97 % of the post-map stage (1.9 us per pair and thread; tools/host_bench.py).
namespace {
struct Digits2 { char d[200]; Digits2() { for (int i = 0; i < 100; i++) { d[2 * i] = (char)('0' + i / 10); d[2 * i + 1] = (char)('0' + i % 10); } } };
const Digits2 DIG2;
struct RcTable { char t[256]; RcTable() { for (int c = 0; c < 256; c++) t[c] = rcNt((char)c); } };
const RcTable RCT;
inline char *putUint(char *p, uint64_t v) {
    char buf[24]; int n = 24;
    while (v >= 100) { const uint64_t q = v / 100; const unsigned r = (unsigned)(v - q * 100); v = q; n -= 2; buf[n] = DIG2.d[2 * r]; buf[n + 1] = DIG2.d[2 * r + 1]; }
    if (v >= 10) { n -= 2; buf[n] = DIG2.d[2 * v]; buf[n + 1] = DIG2.d[2 * v + 1]; } else buf[--n] = (char)('0' + v);
    memcpy(p, buf + n, (size_t)(24 - n));
    return p + (24 - n);
}
inline char *putInt(char *p, int64_t v) { if (v < 0) { *p++ = '-'; return putUint(p, (uint64_t)(-v)); } return putUint(p, (uint64_t)v); }
inline char *putStr(char *p, const char *s, size_t n) { memcpy(p, s, n); return p + n; }
inline char *putSv(char *p, std::string_view s) { memcpy(p, s.data(), s.size()); return p + s.size(); }
template <size_t N> inline char *putLit(char *p, const char (&s)[N]) { memcpy(p, s, N - 1); return p + (N - 1); }
// dst[k] = src[n - 1 - k] (qualities of a reverse-strand mate) and dst[k] = complement of src[n - 1 - k] (its bases), 32 characters per step when the CPU has AVX2.
// The complement of a chunk that holds nothing but A C G T N (upper case: what a sequencer writes) is one table look-up on the low four bits of the character
// code (0x41 0x43 0x47 0x54 0x4E: 1 3 7 4 E); any other character in the chunk sends that chunk through the byte table (IUPAC codes, lower case)
__attribute__((target("avx2"))) static inline __m256i rev32(__m256i c) {
    const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const __m256i x = _mm256_shuffle_epi8(c, rev);
    return _mm256_permute2x128_si256(x, x, 1);
}
__attribute__((target("avx2"))) static void revCopyAvx2(char *dst, const char *src, size_t n) {
    size_t k = 0;
    for (; k + 32 <= n; k += 32) _mm256_storeu_si256((__m256i *)(dst + k), rev32(_mm256_loadu_si256((const __m256i *)(src + n - 32 - k))));
    for (; k < n; k++) dst[k] = src[n - 1 - k];
}
__attribute__((target("avx2"))) static void revCompCopyAvx2(char *dst, const char *src, size_t n) {
    const __m256i orig = _mm256_setr_epi8(0, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 'N', 0, 0, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 'N', 0);
    const __m256i comp = _mm256_setr_epi8(0, 'T', 0, 'G', 'A', 0, 0, 'C', 0, 0, 0, 0, 0, 0, 'N', 0, 0, 'T', 0, 'G', 'A', 0, 0, 'C', 0, 0, 0, 0, 0, 0, 'N', 0);
    size_t k = 0;
    for (; k + 32 <= n; k += 32) {
        const __m256i c = rev32(_mm256_loadu_si256((const __m256i *)(src + n - 32 - k)));
        const __m256i nib = _mm256_and_si256(c, _mm256_set1_epi8(0x0F));
        const __m256i ok = _mm256_cmpeq_epi8(_mm256_shuffle_epi8(orig, nib), c);          // (a character code with bit 7 set never equals a table entry)
        if (_mm256_movemask_epi8(ok) == -1) _mm256_storeu_si256((__m256i *)(dst + k), _mm256_shuffle_epi8(comp, nib));
        else for (size_t j = 0; j < 32; j++) dst[k + j] = RCT.t[(uint8_t)src[n - 1 - k - j]];
    }
    for (; k < n; k++) dst[k] = RCT.t[(uint8_t)src[n - 1 - k]];
}
static const bool HAVE_AVX2 = __builtin_cpu_supports("avx2") && !getenv("STARAMD_NO_AVX2");
inline void revCopy(char *dst, const char *src, size_t n) { if (HAVE_AVX2) revCopyAvx2(dst, src, n); else for (size_t k = 0; k < n; k++) dst[k] = src[n - 1 - k]; }
inline void revCompCopy(char *dst, const char *src, size_t n) { if (HAVE_AVX2) revCompCopyAvx2(dst, src, n); else for (size_t k = 0; k < n; k++) dst[k] = RCT.t[(uint8_t)src[n - 1 - k]]; }
enum SamAttr : uint8_t { A_NH, A_HI, A_AS, A_nM, A_jM, A_jI, A_XS, A_NM, A_MD, A_MC, A_RG, A_OTHER };
inline SamAttr samAttrCode(const std::string &a) {
    static const char *names[] = {"NH", "HI", "AS", "nM", "jM", "jI", "XS", "NM", "MD", "MC", "RG"};
    for (int k = 0; k < 11; k++) if (a == names[k]) return (SamAttr)k;
    return A_OTHER;
}
} // namespace

static void samMapped(std::string &out, const RunParams &P, const GenomeIndex &gi, const ReadCtx &rc, const TrView &tv, uint64_t nTrOut, uint64_t iTrOut, const ChimBam *chim = nullptr) {
    const staramd_transcript &t = *tv.t; const staramd_exon *ex = tv.ex;
    const ReadBatch &b = *rc.b; uint32_t ir = rc.i;
    bool flagPaired = rc.nMates == 2;
    uint32_t nEx = t.nExons;
    uint32_t iExMate, nMates = 1;
    for (iExMate = 0; iExMate + 1 < nEx; iExMate++) if (ex[iExMate].canonSJ == -3) { nMates = 2; break; }
    uint32_t samFlagCommon = 0;
    uint64_t Lread = rc.Lread;
    if (flagPaired) {
        samFlagCommon = 0x0001;
        if (iExMate == nEx - 1) { if (!chim || chim->mateChr > gi.view.nChrReal) samFlagCommon += 0x0008; }      // no mate given: (uint)-1 > nChrReal (:75)
        else if (P.dev.alignEndsProtrudeConcordantPair ||
                 ((ex[0].G <= ex[iExMate + 1].G + ex[0].R) && (ex[iExMate].G + ex[iExMate].L <= ex[nEx - 1].G + Lread - ex[nEx - 1].R)))
            samFlagCommon += 0x0002;
    }
    if (b.filter[ir] == 'Y') samFlagCommon += 0x200;
    uint32_t Str = t.Str;
    uint32_t leftMate = flagPaired ? Str : 0;
    uint64_t chrS = gi.chrStart[t.Chr];
    // the attributes of the run as codes (the list is a parameter: the same for every record)
    // (decoded once per thread and run: eleven string compares per attribute and record were a tenth of the formatter's time)
    static thread_local const RunParams *attrOf = nullptr; static thread_local SamAttr attr[32]; static thread_local uint32_t nAttr = 0; static thread_local bool wantJ = false;
    static thread_local uint64_t attrKey = 0;
    uint64_t key = P.outSAMattrOrder.size();                                  // the tags are two characters: a hash of them, not the address of P alone, says whether the list is the one decoded
    for (const std::string &a : P.outSAMattrOrder) key = key * 1000003ull + (a.size() >= 2 ? (uint64_t)(uint8_t)a[0] * 257u + (uint8_t)a[1] : 0xFFFFu) + a.size();
    if (attrOf != &P || attrKey != key) {
        nAttr = 0; wantJ = false;
        for (const std::string &a : P.outSAMattrOrder) { if (nAttr < 32) { attr[nAttr] = samAttrCode(a); wantJ = wantJ || attr[nAttr] == A_jM || attr[nAttr] == A_jI; nAttr++; } }
        attrOf = &P; attrKey = key;
    }
    // ReadAlign::calcCIGAR (ReadAlign_calcCIGAR.cpp:3-58): the CIGAR of both mates first, the MC tag needs the other mate's.  At most 3 operations per exon + 2 clips,
    // 21 characters each at the very most
    char cig[2][(3 * STARAMD_MAX_N_EXONS + 2) * 21]; size_t cigLen[2] = {0, 0};
    for (uint32_t imate = 0; imate < nMates; imate++) {
        uint32_t iEx1 = imate == 0 ? 0 : iExMate + 1, iEx2 = imate == 0 ? iExMate : nEx - 1;
        uint32_t Mate = ex[iEx1].iFrag;
        char *c = cig[imate];
        const uint64_t trimL = rc.trimL(Str, Mate);
        uint64_t trimL1 = trimL + ex[iEx1].R - (ex[iEx1].R < rc.readLength[leftMate] ? 0 : rc.readLength[leftMate] + 1);
        if (trimL1 > 0) { c = putUint(c, trimL1); *c++ = 'S'; }
        for (uint32_t ii = iEx1; ii <= iEx2; ii++) {
            if (ii > iEx1) {
                uint64_t gapG = ex[ii].G - (ex[ii - 1].G + ex[ii - 1].L);
                uint64_t gapR = (uint64_t)ex[ii].R - ex[ii - 1].R - ex[ii - 1].L;
                if (gapR > 0) { c = putUint(c, gapR); *c++ = 'I'; }
                if (ex[ii - 1].canonSJ >= 0 || ex[ii - 1].sjAnnot == 1) { c = putUint(c, gapG); *c++ = 'N'; }
                else if (gapG > 0) { c = putUint(c, gapG); *c++ = 'D'; }
            }
            c = putUint(c, ex[ii].L); *c++ = 'M';
        }
        uint64_t trimR1 = (ex[iEx1].R < rc.readLength[leftMate] ? rc.readLengthOriginal[leftMate] : rc.readLength[leftMate] + 1 + rc.readLengthOriginal[Mate])
                          - ex[iEx2].R - ex[iEx2].L - trimL;
        if (trimR1 > 0) { c = putUint(c, trimR1); *c++ = 'S'; }
        cigLen[imate] = (size_t)(c - cig[imate]);
    }
    for (uint32_t imate = 0; imate < nMates; imate++) {
        uint32_t samFLAG = samFlagCommon;
        uint32_t iEx1 = imate == 0 ? 0 : iExMate + 1, iEx2 = imate == 0 ? iExMate : nEx - 1;
        uint32_t Mate = ex[iEx1].iFrag;
        if (Mate == 0) { samFLAG |= Str * 0x10; if (nMates == 2) samFLAG |= (1 - Str) * 0x20; }
        else { samFLAG |= (1 - Str) * 0x10; if (nMates == 2) samFLAG |= Str * 0x20; }
        if (flagPaired) { samFLAG |= (Mate == 0 ? 0x0040 : 0x0080); if (nMates == 1 && chim && chim->mateStrand == 1) samFLAG |= 0x20; }     // :120-123
        if (!tv.primary) samFLAG |= 0x100;
        // jM / jI (only when asked for): at most one junction per exon, 4 + 2 x 21 characters each
        char jm[STARAMD_MAX_N_EXONS * 5 + 8], ji[STARAMD_MAX_N_EXONS * 44 + 8]; size_t jmLen = 0, jiLen = 0;
        if (wantJ) {
            char *m = jm, *q = ji;
            for (uint32_t ii = iEx1 + 1; ii <= iEx2; ii++) {
                if (ex[ii - 1].canonSJ >= 0 || ex[ii - 1].sjAnnot == 1) {
                    *m++ = ','; m = putInt(m, ex[ii - 1].canonSJ + (ex[ii - 1].sjAnnot == 0 ? 0 : 20));   // SJ_SAM_AnnotatedMotifShift
                    *q++ = ','; q = putUint(q, ex[ii - 1].G + ex[ii - 1].L + 1 - chrS);
                    *q++ = ','; q = putUint(q, ex[ii].G - chrS);
                }
            }
            if (m == jm) { m = putLit(m, ",-1"); q = putLit(q, ",-1"); }
            jmLen = (size_t)(m - jm); jiLen = (size_t)(q - ji);
        }
        // NM / MD (ReadAlign_outputTranscriptSAM.cpp:242-276): read in the orientation of the alignment against the genome text
        uint64_t tagNM = 0; std::string tagMD;
        if (P.attrNMorMD) {
            const uint8_t *rd = b.bases.data() + b.readOffset[ir];
            uint64_t matchN = 0;
            for (uint32_t iex = iEx1; iex <= iEx2; iex++) {
                for (uint32_t ii = 0; ii < ex[iex].L; ii++) {
                    uint64_t rp = (uint64_t)ex[iex].R + ii;
                    uint8_t r1 = t.roStr == 0 ? rd[rp] : rd[Lread - 1 - rp];
                    if (t.roStr != 0 && r1 < 4) r1 = 3 - r1;
                    uint8_t g1 = gi.G[ex[iex].G + ii];
                    if (r1 != g1 || r1 == 4 || g1 == 4) { ++tagNM; appendUint(tagMD, matchN); tagMD.push_back("ACGTN"[g1 < 5 ? g1 : 4]); matchN = 0; }
                    else matchN++;
                }
                if (iex < iEx2) {
                    if (ex[iex].canonSJ == -1) {
                        tagNM += ex[iex + 1].G - (ex[iex].G + ex[iex].L);
                        appendUint(tagMD, matchN); tagMD.push_back('^');
                        for (uint64_t ii = ex[iex].G + ex[iex].L; ii < ex[iex + 1].G; ii++) { uint8_t g1 = gi.G[ii]; tagMD.push_back("ACGTN"[g1 < 5 ? g1 : 4]); }
                        matchN = 0;
                    } else if (ex[iex].canonSJ == -2) tagNM += (uint64_t)ex[iex + 1].R - ex[iex].R - ex[iex].L;
                }
            }
            appendUint(tagMD, matchN);
        }
        int MAPQ = P.outSAMmapqUnique;
        if (nTrOut >= 5) MAPQ = 0; else if (nTrOut >= 3) MAPQ = 1; else if (nTrOut == 2) MAPQ = 3;
        const std::string_view nm = b.name(ir), sq = b.seq((int)Mate, ir), ql = b.qual((int)Mate, ir), xt = b.extra((int)imate, ir);
        const std::string &chrN = gi.chrName[t.Chr];
        const std::string *mateChrN = (nMates == 1 && chim && chim->mateChr < gi.view.nChrReal) ? &gi.chrName[chim->mateChr] : nullptr;
        const std::string *rg = nullptr;
        for (uint32_t k = 0; k < nAttr; k++) if (attr[k] == A_RG) rg = &P.outSAMattrRG.at(b.fileOf(ir));
        // upper bound of the line: the variable-length columns + 10 numbers of at most 21 characters + per attribute (a list may repeat an attribute) its tag,
        // a number and the longest variable-length payload an attribute can carry
        const size_t attrVar = std::max(std::max((size_t)jmLen, (size_t)jiLen), std::max(std::max((size_t)cigLen[0], (size_t)cigLen[1]), std::max(tagMD.size(), rg ? rg->size() : (size_t)0)));
        const size_t bound = nm.size() + chrN.size() + (mateChrN ? mateChrN->size() : 0) + cigLen[0] + cigLen[1] + 2 * sq.size() + xt.size()
                             + 10 * 21 + (size_t)nAttr * (8 + 21 + attrVar) + 64;
        // the line is built in a buffer on the stack and appended once: growing `out` by the BOUND first (std::string::resize) zero-fills ~3x the bytes the line ends up with
        char lineBuf[4096];
        const bool onStack = bound <= sizeof(lineBuf);
        const size_t base = out.size();
        if (!onStack) out.resize(base + bound);
        char *const p0 = onStack ? lineBuf : &out[base];
        char *p = p0;
        p = putSv(p, nm); *p++ = '\t';
        p = putUint(p, (samFLAG & P.outSAMflagAND) | P.outSAMflagOR); *p++ = '\t';
        p = putStr(p, chrN.data(), chrN.size()); *p++ = '\t';
        p = putUint(p, ex[iEx1].G + 1 - chrS); *p++ = '\t';
        p = putInt(p, MAPQ); *p++ = '\t';
        p = putStr(p, cig[imate], cigLen[imate]);
        if (nMates > 1) {
            p = putLit(p, "\t=\t"); p = putUint(p, ex[imate == 0 ? iExMate + 1 : 0].G + 1 - chrS); *p++ = '\t';
            if (imate != 0) *p++ = '-';
            p = putUint(p, ex[nEx - 1].G + ex[nEx - 1].L - ex[0].G);        // (--outSAMtlen 2 changes BAM output only, as in the reference)
        } else if (mateChrN) { *p++ = '\t'; p = putStr(p, mateChrN->data(), mateChrN->size()); *p++ = '\t'; p = putUint(p, (uint64_t)chim->mateStart + 1); p = putLit(p, "\t0"); }
        else p = putLit(p, "\t*\t0\t0");
        *p++ = '\t';
        const bool noQS = P.outSAMmodeNoQS || b.fasta;           // readFileType==2 ? Qual : "*" (ReadAlign_outputTranscriptSAM.cpp:215)
        if (Mate == Str) { p = putSv(p, sq); *p++ = '\t'; if (!noQS) p = putSv(p, ql); else *p++ = '*'; }
        else {
            const size_t n = sq.size();
            revCompCopy(p, sq.data(), n);
            p += n; *p++ = '\t';
            if (!noQS) { revCopy(p, ql.data(), n); p += n; } else *p++ = '*';
        }
        for (uint32_t k = 0; k < nAttr; k++) {
            switch (attr[k]) {
                case A_NH: p = putLit(p, "\tNH:i:"); p = putUint(p, nTrOut); break;
                case A_HI: p = putLit(p, "\tHI:i:"); p = putInt(p, (int64_t)iTrOut + P.outSAMattrIHstart); break;
                case A_AS: p = putLit(p, "\tAS:i:"); p = putInt(p, t.maxScore); break;
                case A_nM: p = putLit(p, "\tnM:i:"); p = putUint(p, t.nMM); break;
                case A_jM: p = putLit(p, "\tjM:B:c"); p = putStr(p, jm, jmLen); break;
                case A_jI: p = putLit(p, "\tjI:B:i"); p = putStr(p, ji, jiLen); break;
                case A_XS: if (t.sjMotifStrand == 1) p = putLit(p, "\tXS:A:+"); else if (t.sjMotifStrand == 2) p = putLit(p, "\tXS:A:-"); break;
                case A_NM: p = putLit(p, "\tNM:i:"); p = putUint(p, tagNM); break;
                case A_MD: p = putLit(p, "\tMD:Z:"); p = putStr(p, tagMD.data(), tagMD.size()); break;
                case A_MC: if (nMates > 1) { p = putLit(p, "\tMC:Z:"); p = putStr(p, cig[1 - imate], cigLen[1 - imate]); } break;
                case A_RG: p = putLit(p, "\tRG:Z:"); p = putStr(p, rg->data(), rg->size()); break;
                default: break;
            }
        }
        // SAM input: its attributes go out again; indexed by the position of the mate in the alignment, not by the mate, as the reference does (:351-353)
        if (!xt.empty()) { *p++ = '\t'; p = putSv(p, xt); }
        *p++ = '\n';
        if (onStack) out.append(lineBuf, (size_t)(p - p0)); else out.resize(base + (size_t)(p - p0));
    }
}

// ================= BAM records (ReadAlign::alignBAM, source/ReadAlign_alignBAM.cpp:49-614; BAMfunctions.cpp) =================
namespace {
inline void put32(std::string &o, uint32_t v) { o.append((const char *)&v, 4); }
inline int reg2bin(int beg, int end) {                   // BAMfunctions.cpp:101-110
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (beg >> 26);
    return 0;
}
inline void attrInt(std::string &a, const char *tag, int64_t x) {        // bamAttrArrayWriteInt (BAMfunctions.h:45-81): smallest type that holds x
    a.push_back(tag[0]); a.push_back(tag[1]);
    if (x < 0) {
        if (x >= -127) { a.push_back('c'); int8_t v = (int8_t)x; a.append((const char *)&v, 1); }
        else if (x >= -32767) { a.push_back('s'); int16_t v = (int16_t)x; a.append((const char *)&v, 2); }
        else { a.push_back('i'); int32_t v = (int32_t)x; a.append((const char *)&v, 4); }
    } else {
        if (x <= 255) { a.push_back('C'); uint8_t v = (uint8_t)x; a.append((const char *)&v, 1); }
        else if (x <= 65535) { a.push_back('S'); uint16_t v = (uint16_t)x; a.append((const char *)&v, 2); }
        else { a.push_back('I'); uint32_t v = (uint32_t)x; a.append((const char *)&v, 4); }
    }
}
inline void attrChar(std::string &a, const char *tag, char c) { a.push_back(tag[0]); a.push_back(tag[1]); a.push_back('A'); a.push_back(c); }
inline void attrStr(std::string &a, const char *tag, const std::string &s) { a.push_back(tag[0]); a.push_back(tag[1]); a.push_back('Z'); a += s; a.push_back(0); }
inline uint8_t nuclToNumBAM(char c) {                    // SequenceFuns.cpp:99-120  =ACMGRSVTWYHKDBN
    switch (c) {
        case '=': return 0; case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'M': case 'm': return 3; case 'G': case 'g': return 4;
        case 'R': case 'r': return 5; case 'S': case 's': return 6; case 'V': case 'v': return 7; case 'T': case 't': return 8; case 'W': case 'w': return 9;
        case 'Y': case 'y': return 10; case 'H': case 'h': return 11; case 'K': case 'k': return 12; case 'D': case 'd': return 13; case 'B': case 'b': return 14;
        default: return 15;
    }
}
// bamAttrArrayWriteSAMtags (BAMfunctions.cpp:147-192): attributes of the input SAM record as BAM attributes; i -> int32, A, Z, f; others dropped
void attrFromSAMtags(std::string &a, std::string_view tags, const RunParams &P) {
    if (tags.empty() || P.samAttrKeepNone) return;
    size_t pos1 = 0, pos2;
    do {
        pos2 = tags.find('\t', pos1);
        std::string_view t = tags.substr(pos1, pos2 == std::string_view::npos ? std::string_view::npos : pos2 - pos1);
        pos1 = pos2 + 1;
        if (t.size() < 5) continue;
        if (!P.samAttrKeepAll && std::find(P.samAttrKeep.begin(), P.samAttrKeep.end(), std::string(t.substr(0, 2))) == P.samAttrKeep.end()) continue;
        const std::string val(t.substr(5));
        switch (t[3]) {
            case 'i': { int32_t v = (int32_t)strtol(val.c_str(), nullptr, 10); a.push_back(t[0]); a.push_back(t[1]); a.push_back('i'); a.append((const char *)&v, 4); break; }
            case 'A': { a.push_back(t[0]); a.push_back(t[1]); a.push_back('A'); a.push_back(val.empty() ? 0 : val[0]); break; }
            case 'Z': { a.push_back(t[0]); a.push_back(t[1]); a.push_back('Z'); a += val; a.push_back(0); break; }
            case 'f': { float v = strtof(val.c_str(), nullptr); a.push_back(t[0]); a.push_back(t[1]); a.push_back('f'); a.append((const char *)&v, 4); break; }
            default: break;
        }
    } while (pos2 != std::string_view::npos);
}
// tail of a record: name, CIGAR, packed sequence, qualities, attributes (:547-590); the 9 core words come first
void bamFinish(std::string &out, const uint32_t core[8], std::string_view name, const std::vector<uint32_t> &cigar, std::string_view seq, std::string_view qual, bool rev,
               bool noQS, const std::string &attr, size_t hardL = 0, size_t hardR = 0) {
    if (rev) std::swap(hardL, hardR);                    // hard clips are counted on the sequence as it is written (ReadAlign_alignBAM.cpp:503-510)
    seq = seq.substr(hardL, seq.size() - hardL - hardR); qual = qual.substr(hardL, qual.size() - hardL - hardR);
    size_t L = seq.size();
    uint32_t recSize = 8 * 4 + (uint32_t)name.size() + 1 + (uint32_t)cigar.size() * 4 + (uint32_t)(L + 1) / 2 + (uint32_t)L + (uint32_t)attr.size();
    put32(out, recSize);
    out.append((const char *)core, 32);
    out += name; out.push_back(0);
    if (!cigar.empty()) out.append((const char *)cigar.data(), cigar.size() * 4);
    auto base = [&](size_t k) { return rev ? rcNt(seq[L - 1 - k]) : seq[k]; };
    for (size_t j = 0; j < L / 2; j++) out.push_back((char)(nuclToNumBAM(base(2 * j)) << 4 | nuclToNumBAM(base(2 * j + 1))));
    if (L % 2 == 1) out.push_back((char)(nuclToNumBAM(base(L - 1)) << 4));
    if (!noQS) { for (size_t k = 0; k < L; k++) out.push_back((char)((rev ? qual[L - 1 - k] : qual[k]) - 33)); }
    else out.append(L, (char)0xFF);
    out += attr;
}
} // namespace

// mapped mates of one alignment (alignType -1)
// quant = a projection onto a transcript (ReadAlign_quantTranscriptome.cpp:71-76): coordinates start at 0, attributes NH HI (+ RG, MC)
static void bamMapped(std::string &out, const RunParams &P, const GenomeIndex &gi, const ReadCtx &rc, const TrView &tv, uint64_t nTrOut, uint64_t iTrOut, std::vector<BamKey> *keys,
                      bool quant = false, std::vector<uint64_t> *recOffsets = nullptr, const ChimBam *chim = nullptr) {
    const int alignType = chim ? chim->alignType : -1;
    const uint32_t mateChr = chim ? chim->mateChr : (uint32_t)-1;
    const staramd_transcript &t = *tv.t; const staramd_exon *ex = tv.ex;
    const ReadBatch &b = *rc.b; uint32_t ir = rc.i;
    const bool flagPaired = rc.nMates == 2;
    const uint32_t nEx = t.nExons;
    uint32_t iExMate, nMates = 1;
    for (iExMate = 0; iExMate + 1 < nEx; iExMate++) if (ex[iExMate].canonSJ == -3) { nMates = 2; break; }
    const uint32_t Str = t.Str, leftMate = flagPaired ? Str : 0;
    const uint64_t chrS = quant ? 0 : gi.chrStart[t.Chr], Lread = rc.Lread;
    // CIGAR strings of both mates for MC (calcCIGAR)
    std::string matesCIGAR[2];
    std::vector<uint32_t> packed[2]; std::vector<int32_t> SJintron[2]; std::vector<char> SJmotif[2];
    VarOverlap vo; bool varDone = false;
    uint64_t hardClip[2][2];
    for (uint32_t imate = 0; imate < nMates; imate++) {
        uint32_t iEx1 = imate == 0 ? 0 : iExMate + 1, iEx2 = imate == 0 ? iExMate : nEx - 1;
        uint32_t Mate = ex[iEx1].iFrag;
        std::string &cg = matesCIGAR[imate]; std::vector<uint32_t> &pc = packed[imate];
        auto op = [&](uint64_t len, char c, uint32_t code, bool inString) { pc.push_back((uint32_t)len << 4 | code); if (inString) { appendUint(cg, len); cg.push_back(c); } };
        const uint64_t trimL = rc.trimL(Str, Mate);
        uint64_t trimL1 = trimL + ex[iEx1].R - (ex[iEx1].R < rc.readLength[leftMate] ? 0 : rc.readLength[leftMate] + 1);
        hardClip[imate][0] = hardClip[imate][1] = 0;
        if (trimL1 > 0) { op(trimL1, 'S', 4, true); if (alignType == -11) { pc.back() = (uint32_t)trimL1 << 4 | 5; hardClip[imate][0] = trimL1; } }     // H in the record, S in the MC string of calcCIGAR
        for (uint32_t ii = iEx1; ii <= iEx2; ii++) {
            if (ii > iEx1) {
                uint64_t gapG = ex[ii].G - (ex[ii - 1].G + ex[ii - 1].L);
                uint64_t gapR = (uint64_t)ex[ii].R - ex[ii - 1].R - ex[ii - 1].L;
                if (gapR > 0) op(gapR, 'I', 1, true);
                if (ex[ii - 1].canonSJ >= 0 || ex[ii - 1].sjAnnot == 1) {
                    op(gapG, 'N', 3, true);
                    SJmotif[imate].push_back((char)(ex[ii - 1].canonSJ + (ex[ii - 1].sjAnnot == 0 ? 0 : 20)));
                    SJintron[imate].push_back((int32_t)(ex[ii - 1].G + ex[ii - 1].L + 1 - chrS)); SJintron[imate].push_back((int32_t)(ex[ii].G - chrS));
                } else if (gapG > 0) op(gapG, 'D', 2, true);
            }
            // the CIGAR string of calcCIGAR always has the M; the packed one skips 0-length blocks (:223-224)
            appendUint(cg, ex[ii].L); cg.push_back('M');
            if (ex[ii].L > 0) pc.push_back((uint32_t)ex[ii].L << 4 | 0);
        }
        if (SJmotif[imate].empty()) { SJmotif[imate].push_back(-1); SJintron[imate].push_back(-1); }
        uint64_t trimR1 = (ex[iEx1].R < rc.readLength[leftMate] ? rc.readLengthOriginal[leftMate] : rc.readLength[leftMate] + 1 + rc.readLengthOriginal[Mate]) - ex[iEx2].R - ex[iEx2].L - trimL;
        if (trimR1 > 0) { op(trimR1, 'S', 4, true); if (alignType == -12) { pc.back() = (uint32_t)trimR1 << 4 | 5; hardClip[imate][1] = trimR1; } }
    }
    for (uint32_t imate = 0; imate < nMates; imate++) {
        uint32_t iEx1 = imate == 0 ? 0 : iExMate + 1, iEx2 = imate == 0 ? iExMate : nEx - 1;
        uint32_t Mate = ex[iEx1].iFrag;
        uint32_t samFLAG = 0;
        if (flagPaired) { samFLAG = 0x0001; if (iExMate == nEx - 1) { if (mateChr > gi.view.nChrReal) samFLAG |= 0x0008; } else samFLAG |= 0x0002; }     // without a chimeric mate: (uint)-1 > nChrReal
        if (b.filter[ir] == 'Y') samFLAG |= 0x200;
        if (alignType == -11 || alignType == -12 || alignType == -13) samFLAG |= 0x800;
        if (!tv.primary) samFLAG |= 0x100;
        if (Mate == 0) { samFLAG |= Str * 0x10; if (nMates == 2) samFLAG |= (1 - Str) * 0x20; }
        else { samFLAG |= (1 - Str) * 0x10; if (nMates == 2) samFLAG |= Str * 0x20; }
        if (flagPaired) { samFLAG |= (Mate == 0 ? 0x0040 : 0x0080); if (nMates == 1 && chim && chim->mateStrand == 1) samFLAG |= 0x20; }
        int MAPQ = P.outSAMmapqUnique;
        if (nTrOut >= 5) MAPQ = 0; else if (nTrOut >= 3) MAPQ = 1; else if (nTrOut == 2) MAPQ = 3;
        // NM / MD of the BAM path (samAttrNM_MD :8-47): insertions of every gap and deletions of every non-junction gap count
        uint64_t tagNM = 0; std::string tagMD;
        if (P.attrNMorMD && !quant) {
            const uint8_t *rd = b.bases.data() + b.readOffset[ir];
            uint64_t matchN = 0, nMM = 0, nI = 0, nD = 0;
            for (uint32_t iex = iEx1; iex <= iEx2; iex++) {
                for (uint32_t ii = 0; ii < ex[iex].L; ii++) {
                    uint64_t rp = (uint64_t)ex[iex].R + ii;
                    uint8_t r1 = t.roStr == 0 ? rd[rp] : rd[Lread - 1 - rp];
                    if (t.roStr != 0 && r1 < 4) r1 = 3 - r1;
                    uint8_t g1 = gi.G[ex[iex].G + ii];
                    if (r1 != g1 || r1 == 4 || g1 == 4) { ++nMM; appendUint(tagMD, matchN); tagMD.push_back("ACGTN"[g1 < 5 ? g1 : 4]); matchN = 0; }
                    else matchN++;
                }
                if (iex < iEx2) {
                    if (ex[iex].canonSJ < 0) nD += ex[iex + 1].G - (ex[iex].G + ex[iex].L);
                    nI += (uint64_t)ex[iex + 1].R - ex[iex].R - ex[iex].L;
                    if (ex[iex].canonSJ == -1) {
                        appendUint(tagMD, matchN); tagMD.push_back('^');
                        for (uint64_t ii = ex[iex].G + ex[iex].L; ii < ex[iex + 1].G; ii++) { uint8_t g1 = gi.G[ii]; tagMD.push_back("ACGTN"[g1 < 5 ? g1 : 4]); }
                        matchN = 0;
                    }
                }
            }
            appendUint(tagMD, matchN);
            tagNM = nMM + nI + nD;
        }
        std::string attr;
        for (const std::string &a : (quant ? P.outSAMattrOrderQuant : P.outSAMattrOrder)) {
            if (a == "NH") attrInt(attr, "NH", (int64_t)nTrOut);
            else if (a == "HI") attrInt(attr, "HI", (int64_t)iTrOut + P.outSAMattrIHstart);
            else if (a == "AS") attrInt(attr, "AS", t.maxScore);
            else if (a == "nM") attrInt(attr, "nM", t.nMM);
            else if (a == "jM") { attr += "jMBc"; uint32_t n = (uint32_t)SJmotif[imate].size(); attr.append((const char *)&n, 4); attr.append(SJmotif[imate].data(), n); }
            else if (a == "jI") { attr += "jIBi"; uint32_t n = (uint32_t)SJintron[imate].size(); attr.append((const char *)&n, 4); attr.append((const char *)SJintron[imate].data(), 4 * (size_t)n); }
            else if (a == "XS") { if (t.sjMotifStrand == 1) attrChar(attr, "XS", '+'); else if (t.sjMotifStrand == 2) attrChar(attr, "XS", '-'); }
            else if (a == "NM") attrInt(attr, "NM", (int64_t)tagNM);
            else if (a == "MD") attrStr(attr, "MD", tagMD);
            else if (a == "MC") { if (nMates > 1) attrStr(attr, "MC", matesCIGAR[1 - imate]); }
            else if (a == "RG") attrStr(attr, "RG", P.outSAMattrRG.at(b.fileOf(ir)));
            else if (a == "ch") { if (alignType <= -10) attrChar(attr, "ch", '1'); }
            else if ((a == "vA" || a == "vG") && P.var) {          // ReadAlign_alignBAM.cpp:347-360: the SNVs under the whole alignment, on every record of it
                if (!varDone && chim && chim->var) { vo = *chim->var; varDone = true; }
                if (!varDone) { P.var->overlap(t, ex, b.bases.data() + b.readOffset[ir], Lread, quant ? 0 : gi.chrStart[t.Chr], vo); varDone = true; }
                if (!vo.allele.empty()) {
                    uint32_t nv = (uint32_t)vo.allele.size();
                    if (a == "vA") { attr += "vABc"; attr.append((const char *)&nv, 4); attr.append(vo.allele.data(), nv); }
                    else { attr += "vGBi"; attr.append((const char *)&nv, 4); attr.append((const char *)vo.genCoord.data(), 4 * (size_t)nv); }
                }
            }
            else if (a == "rB") {                                  // :335-346 read and genome coordinates of every block of this mate
                std::vector<int32_t> rb;
                for (uint32_t ii = iEx1; ii <= iEx2; ii++) { rb.push_back((int32_t)ex[ii].R + 1); rb.push_back((int32_t)ex[ii].R + ex[ii].L); rb.push_back((int32_t)(ex[ii].G - chrS + 1)); rb.push_back((int32_t)(ex[ii].G - chrS + ex[ii].L)); }
                attr += "rBBi"; uint32_t n = (uint32_t)rb.size(); attr.append((const char *)&n, 4); attr.append((const char *)rb.data(), 4 * (size_t)n);
            }
            else if (a == "cN") {                                  // :318-322 clipped bases at the 5' and 3' end; indexed by the position of the mate in the alignment, as in the reference
                int32_t v1[2] = {(int32_t)rc.clip[imate][0], (int32_t)rc.clip[imate][1]};
                attr += "cNBi"; uint32_t n = 2; attr.append((const char *)&n, 4); attr.append((const char *)v1, 8);
            }
            else if (a == "vW") { if (rc.waspType != -1) { int32_t w = rc.waspType; attr += "vWi"; attr.append((const char *)&w, 4); } }
        }
        attrFromSAMtags(attr, b.extra((int)Mate, ir), P);
        uint32_t core[8];
        core[0] = t.Chr;
        core[1] = (uint32_t)(ex[iEx1].G - chrS);
        core[2] = ((uint32_t)reg2bin((int)(ex[iEx1].G - chrS), (int)(ex[iEx2].G + ex[iEx2].L - chrS)) << 16) | ((uint32_t)MAPQ << 8) | (uint32_t)(b.name(ir).size() + 1);
        core[3] = (((samFLAG & P.outSAMflagAND) | P.outSAMflagOR) << 16) | (uint32_t)packed[imate].size();
        core[4] = (uint32_t)(b.seq((int)Mate, ir).size() - hardClip[imate][0] - hardClip[imate][1]);
        if (nMates > 1) {
            core[5] = t.Chr; core[6] = (uint32_t)(ex[imate == 0 ? iExMate + 1 : 0].G - chrS);
            int32_t tlen = (int32_t)(ex[nEx - 1].G + ex[nEx - 1].L - ex[0].G);                 // outSAMtlen 1
            if (P.outSAMtlen == 2) {                                                             // :78-82: leftmost base of any mate to rightmost base of any mate
                tlen = (int32_t)(std::max(ex[nEx - 1].G + ex[nEx - 1].L, ex[iExMate].G + ex[iExMate].L) - std::min(ex[0].G, ex[iExMate + 1].G));
                core[7] = (uint32_t)(imate == (ex[0].G <= ex[iExMate + 1].G ? 0u : 1u) ? tlen : -tlen);
            } else core[7] = (uint32_t)(imate == 0 ? tlen : -tlen);
        } else if (mateChr < gi.view.nChrReal) { core[5] = mateChr; core[6] = chim->mateStart; core[7] = 0; }
        else { core[5] = (uint32_t)-1; core[6] = (uint32_t)-1; core[7] = 0; }
        const size_t off0 = out.size();
        if (recOffsets) recOffsets->push_back(off0);
        bamFinish(out, core, b.name(ir), packed[imate], b.seq((int)Mate, ir), b.qual((int)Mate, ir), Mate != Str, P.outSAMmodeNoQS || b.fasta, attr, hardClip[imate][0], hardClip[imate][1]);
        // BAMoutput::coordOneAlign key (ReadAlign_outputAlignments.cpp:196-199): iReadAll << 32 | iTr << 8 | mate of the first exon; chimeric segments: iReadAll << 32
        if (keys && !chim) keys->push_back(BamKey{((uint64_t)core[0] << 32) | core[1], (b.readIndex(ir) << 32) | (iTrOut << 8) | ex[0].iFrag, off0, (uint32_t)(out.size() - off0), 0});
    }
}

// ChimericAlign::chimericBAMoutput (ChimericAlign_chimericBAMoutput.cpp:7-105): the two segments of a chimeric alignment as BAM records.  The segment
// that holds both mates (or, for single-end reads, the better one) is the representative alignment, the other one is supplementary (0x800, hard- or
// soft-clipped); two one-mate segments are written as an ordinary, if distant, pair.  The chimerically split mate and the supplementary record point at
// each other through SA tags.
static uint32_t rd32(const char *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static std::string saTagOf(const GenomeIndex &gi, const char *rec) {
    const uint32_t tid = rd32(rec + 4), pos = rd32(rec + 8), bmn = rd32(rec + 12), fnc = rd32(rec + 16), lseq = rd32(rec + 20), recSize = rd32(rec);
    const uint32_t lname = bmn & 0xff, mapq = (bmn >> 8) & 0xff, ncig = fnc & 0xffff, flag = fnc >> 16;
    const char *cig = rec + 36 + lname;
    std::string s = "SAZ" + gi.chrName[tid] + ","; appendUint(s, (uint64_t)pos + 1); s.push_back(','); s.push_back((flag & 0x10) == 0 ? '+' : '-'); s.push_back(',');
    for (uint32_t k = 0; k < ncig; k++) { uint32_t c = rd32(cig + 4 * k); appendUint(s, c >> 4); s.push_back("MIDNSHP=X"[c & 0xf]); }
    s.push_back(','); appendUint(s, mapq); s.push_back(',');
    // NM of that record (bam_aux_get + bam_aux2i)
    const char *a = cig + 4 * ncig + (lseq + 1) / 2 + lseq, *end = rec + 4 + recSize;
    int64_t nm = 0;
    while (a + 3 <= end) {
        const char t = a[2]; const bool isNM = a[0] == 'N' && a[1] == 'M';
        a += 3;
        size_t w = 0;
        switch (t) {
            case 'A': case 'c': case 'C': w = 1; break;
            case 's': case 'S': w = 2; break;
            case 'i': case 'I': case 'f': w = 4; break;
            case 'Z': case 'H': w = strlen(a) + 1; break;
            case 'B': { const char bt = a[0]; const uint32_t n = rd32(a + 1); w = 5 + (size_t)n * (bt == 'c' || bt == 'C' ? 1 : bt == 's' || bt == 'S' ? 2 : 4); break; }
            default: w = (size_t)(end - a); break;
        }
        if (isNM) {
            if (t == 'c') nm = (int8_t)a[0]; else if (t == 'C') nm = (uint8_t)a[0];
            else if (t == 's') { int16_t v; memcpy(&v, a, 2); nm = v; } else if (t == 'S') { uint16_t v; memcpy(&v, a, 2); nm = v; }
            else if (t == 'i') { int32_t v; memcpy(&v, a, 4); nm = v; } else if (t == 'I') { uint32_t v; memcpy(&v, a, 4); nm = v; }
            break;
        }
        a += w;
    }
    appendUint(s, (uint64_t)(uint32_t)nm); s.push_back(';');
    return s;
}
static void chimBamOutput(std::string &out, const RunParams &P, const GenomeIndex &gi, const ReadCtx &rc, const ChimPair &cp, uint64_t iTr, uint64_t chimN, std::vector<BamKey> *keys) {
    const ChimTr *trChim[2] = {&cp.a1, &cp.a2};
    auto frag0 = [&](int i) { return trChim[i]->ex[0].iFrag; };
    auto fragN = [&](int i) { return trChim[i]->ex[trChim[i]->t.nExons - 1].iFrag; };
    int chimRepresent, chimType;
    if (frag0(0) != fragN(0)) { chimRepresent = 0; chimType = 1; }
    else if (frag0(1) != fragN(1)) { chimRepresent = 1; chimType = 1; }
    else if (frag0(0) != frag0(1)) { chimRepresent = -1; chimType = 2; }
    else { chimRepresent = trChim[0]->t.maxScore > trChim[1]->t.maxScore ? 0 : 1; chimType = 3; }
    std::vector<std::string> recs;                       // one BAM record each, in the order of production
    int bamIsuppl = -1, bamIrepr = -1;
    for (int itr = 0; itr < 2; itr++) {
        ChimBam cb; uint64_t mateStartG = 0;
        if (chimType == 2) {
            cb.mateChr = trChim[1 - itr]->t.Chr; mateStartG = trChim[1 - itr]->ex[0].G; cb.mateStrand = (uint8_t)(trChim[1 - itr]->t.Str != trChim[1 - itr]->ex[0].iFrag); cb.alignType = -10;
        } else {
            cb.mateChr = (uint32_t)-1; cb.mateStrand = 0;
            if (chimRepresent == itr) {
                cb.alignType = -10;
                bamIrepr = (int)recs.size();
                if (frag0(itr) != frag0(1 - itr)) ++bamIrepr;           // the next mate is the chimerically split one
            } else {
                cb.alignType = P.chim.bamHardClip ? ((uint32_t)(itr % 2) == trChim[itr]->t.Str ? -12 : -11) : -13;
                bamIsuppl = (int)recs.size();
                if (chimType == 1) {                                 // the supplementary record of a paired read points at the other mate in the representative alignment
                    const ChimTr &rp = *trChim[chimRepresent];
                    uint32_t iex = 0;
                    for (; iex + 1 < rp.t.nExons; iex++) if (rp.ex[iex].iFrag != frag0(itr)) break;
                    cb.mateChr = rp.t.Chr; mateStartG = rp.ex[iex].G; cb.mateStrand = (uint8_t)(rp.t.Str != rp.ex[iex].iFrag);
                }
            }
        }
        cb.mateStart = (uint32_t)(mateStartG - gi.chrStart[cb.mateChr < gi.view.nChrReal ? cb.mateChr : 0]);
        if (!(cb.mateChr < gi.view.nChrReal)) cb.mateStart = (uint32_t)((uint64_t)-1 - gi.chrStart[0]);
        TrView v; v.t = &trChim[itr]->t; v.ex = trChim[itr]->ex; v.primary = cp.best;
        cb.var = itr == 0 ? &cp.var1 : &cp.var2;
        std::string raw; std::vector<uint64_t> offs;
        bamMapped(raw, P, gi, rc, v, chimN, iTr, nullptr, false, &offs, &cb);
        for (size_t k = 0; k < offs.size(); k++) recs.push_back(raw.substr(offs[k], (k + 1 < offs.size() ? offs[k + 1] : raw.size()) - offs[k]));
    }
    const std::vector<std::string> plain = recs;
    for (int ii = 0; ii < (int)recs.size(); ii++) {
        int tagI = -1;
        if (ii == bamIrepr) tagI = bamIsuppl; else if (ii == bamIsuppl) tagI = bamIrepr;
        if (tagI >= 0) {
            std::string sa = saTagOf(gi, plain[tagI].data());
            recs[ii] += sa; recs[ii].push_back(0);
            uint32_t sz = (uint32_t)recs[ii].size() - 4; memcpy(&recs[ii][0], &sz, 4);
        }
        const size_t off0 = out.size();
        out += recs[ii];
        if (keys) keys->push_back(BamKey{((uint64_t)rd32(recs[ii].data() + 4) << 32) | rd32(recs[ii].data() + 8), rc.b->readIndex(rc.i) << 32, off0, (uint32_t)recs[ii].size(), 0});
    }
}

// --chimOutType SeparateSAMold (ReadAlign_chimericDetectionOldOutput.cpp:18-59): the two segments as SAM records of Chimeric.out.sam, each pointing at the other
static void chimSamOldOutput(std::string &out, const RunParams &P, const GenomeIndex &gi, const ReadCtx &rc, const ChimPair &cp) {
    const ChimTr *trChim[2] = {&cp.a1, &cp.a2};
    auto frag0 = [&](int i) { return trChim[i]->ex[0].iFrag; };
    auto fragN = [&](int i) { return trChim[i]->ex[trChim[i]->t.nExons - 1].iFrag; };
    bool primary[2];
    if (frag0(0) != fragN(0)) { primary[0] = true; primary[1] = false; }
    else if (frag0(1) != fragN(1)) { primary[1] = true; primary[0] = false; }
    else if (frag0(0) != frag0(1)) { primary[0] = primary[1] = true; }
    else { int r = trChim[0]->t.maxScore > trChim[1]->t.maxScore ? 0 : 1; primary[r] = true; primary[1 - r] = false; }
    for (int iTr = 0; iTr < 2; iTr++) {
        TrView v; v.t = &trChim[iTr]->t; v.ex = trChim[iTr]->ex; v.primary = primary[iTr];
        if (rc.nMates == 2) {
            const ChimTr &o = *trChim[1 - iTr];
            uint32_t iex = 0;
            if (frag0(1 - iTr) != fragN(1 - iTr)) for (; iex < o.t.nExons; iex++) if (o.ex[iex].iFrag != frag0(iTr)) break;
            ChimBam cb; cb.alignType = -1; cb.mateChr = o.t.Chr; cb.mateStart = (uint32_t)(o.ex[iex].G - gi.chrStart[o.t.Chr]); cb.mateStrand = (uint8_t)(o.t.Str != o.ex[iex].iFrag);
            samMapped(out, P, gi, rc, v, 2, (uint64_t)iTr, &cb);
        } else samMapped(out, P, gi, rc, v, 2, (uint64_t)iTr);
    }
}

// unmapped mates (alignType >= 0): both mates of an unmapped read, or the missing mate of a single-end alignment (:120-188)
static void bamUnmapped(std::string &out, const RunParams &P, const GenomeIndex &gi, const ReadCtx &rc, const staramd_transcript *trBest, const staramd_exon *exBest,
                        int unmapType, const bool mateMap[2], std::vector<BamKey> *keys, bool mappedMateSecondary = false) {
    const ReadBatch &b = *rc.b; uint32_t ir = rc.i;
    for (int imate = 0; imate < rc.nMates; imate++) {
        if (mateMap[imate]) continue;
        uint32_t samFLAG = 0x4; uint32_t mateChr = (uint32_t)-1, mateStart = (uint32_t)-1;
        if (rc.nMates == 2) {
            samFLAG |= 0x1 + (imate == 0 ? 0x40 : 0x80);
            if (mateMap[1 - imate]) {
                if (trBest->Str != (uint32_t)(1 - imate)) samFLAG |= 0x20;
                mateChr = trBest->Chr; mateStart = (uint32_t)(exBest[0].G - gi.chrStart[mateChr]);
                if (mappedMateSecondary) samFLAG |= 0x100;       // KeepPairs: the unmapped mate of a secondary alignment is secondary too (:136-139)
            } else samFLAG |= 0x8;
        }
        if (b.filter[ir] == 'Y') samFLAG |= 0x200;
        std::string attr;
        attrInt(attr, "NH", 0); attrInt(attr, "HI", 0); attrInt(attr, "AS", trBest ? trBest->maxScore : 0); attrInt(attr, "nM", trBest ? trBest->nMM : 0);
        attrChar(attr, "uT", (char)('0' + unmapType));
        if (!P.outSAMattrRG.empty()) attrStr(attr, "RG", P.outSAMattrRG.at(b.fileOf(ir)));
        if (std::find(P.outSAMattrOrder.begin(), P.outSAMattrOrder.end(), "cN") != P.outSAMattrOrder.end()) { int32_t v1[2] = {(int32_t)rc.clip[imate][0], (int32_t)rc.clip[imate][1]}; attr += "cNBi"; uint32_t n = 2; attr.append((const char *)&n, 4); attr.append((const char *)v1, 8); }   // :181-184
        attrFromSAMtags(attr, b.extra(imate, ir), P);
        uint32_t core[8];
        core[0] = (uint32_t)-1; core[1] = (uint32_t)-1;
        core[2] = ((uint32_t)reg2bin(-1, 0) << 16) | (uint32_t)(b.name(ir).size() + 1);
        core[3] = (((samFLAG & P.outSAMflagAND) | P.outSAMflagOR) << 16);
        core[4] = (uint32_t)b.seq(imate, ir).size();
        if (mateChr < gi.view.nChrReal) { core[5] = mateChr; core[6] = mateStart; } else { core[5] = (uint32_t)-1; core[6] = (uint32_t)-1; }
        core[7] = 0;
        const size_t off0 = out.size();
        bamFinish(out, core, b.name(ir), std::vector<uint32_t>(), b.seq(imate, ir), b.qual(imate, ir), false, P.outSAMmodeNoQS || b.fasta, attr);
        if (keys) keys->push_back(BamKey{~0ull, b.readIndex(ir) << 32, off0, (uint32_t)(out.size() - off0), 0});      // unmapped: last, in read order
    }
}

std::string PostMap::quantBamHeader() const {
    std::string samh, h = "BAM\1";
    for (size_t i = 0; i < transcripts->trID.size(); i++) { samh += "@SQ\tSN:" + transcripts->trID[i] + "\tLN:"; appendUint(samh, transcripts->trLen[i]); samh += "\n"; }
    for (const std::string &rg : P.outSAMattrRGlineSplit) samh += "@RG\t" + rg + "\n";
    put32(h, (uint32_t)samh.size()); h += samh;
    put32(h, (uint32_t)transcripts->trID.size());
    for (size_t i = 0; i < transcripts->trID.size(); i++) { put32(h, (uint32_t)transcripts->trID[i].size() + 1); h += transcripts->trID[i]; h.push_back(0); put32(h, transcripts->trLen[i]); }
    return h;
}

std::string PostMap::bamHeader(bool sortedByCoordinate) const {
    std::string samh = samHeader(), h = "BAM\1";
    if (sortedByCoordinate) samh.insert(samh.find('\n'), "\tSO:coordinate");     // samHeaders.cpp:99
    put32(h, (uint32_t)samh.size()); h += samh;
    put32(h, gi.view.nChrReal);
    for (uint32_t i = 0; i < gi.view.nChrReal; i++) { put32(h, (uint32_t)gi.chrName[i].size() + 1); h += gi.chrName[i]; h.push_back(0); put32(h, (uint32_t)gi.chrLength[i]); }
    return h;
}

// ---- ReadAlign::outputTranscriptSAM, unmapped branch (:11-54) ----
static void samUnmapped(std::string &out, const RunParams &P, const GenomeIndex &gi, const ReadCtx &rc, const staramd_transcript *trBest, const staramd_exon *exBest,
                        int unmapType, const bool mateMap[2], bool mappedMateSecondary = false) {
    const ReadBatch &b = *rc.b; uint32_t ir = rc.i;
    for (int imate = 0; imate < rc.nMates; imate++) {
        if (mateMap[imate]) continue;
        uint32_t samFLAG = 0x4;
        if (rc.nMates == 2) {
            samFLAG |= 0x1 + (imate == 0 ? 0x40 : 0x80);
            if (mateMap[1 - imate]) { if (trBest->Str != (uint32_t)(1 - imate)) samFLAG |= 0x20; }
            else samFLAG |= 0x8;
        }
        if (b.filter[ir] == 'Y') samFLAG |= 0x200;
        if (rc.nMates == 2 && mateMap[1 - imate] && mappedMateSecondary) samFLAG |= 0x100;      // :31-33 (KeepPairs)
        out += b.name(ir); out.push_back('\t'); appendUint(out, samFLAG); out += "\t*\t0\t0\t*";
        if (rc.nMates == 2 && mateMap[1 - imate]) { out.push_back('\t'); out += gi.chrName[trBest->Chr]; out.push_back('\t'); appendUint(out, exBest[0].G + 1 - gi.chrStart[trBest->Chr]); }
        else out += "\t*\t0";
        out += "\t0\t"; out += b.seq(imate, ir); out.push_back('\t'); if (b.fasta) out.push_back('*'); else out += b.qual(imate, ir);      // :44
        out += "\tNH:i:0\tHI:i:0\tAS:i:"; appendInt(out, trBest ? trBest->maxScore : 0);
        out += "\tnM:i:"; appendUint(out, trBest ? trBest->nMM : 0); out += "\tuT:A:"; appendInt(out, unmapType);
        if (!P.outSAMattrRG.empty()) { out += "\tRG:Z:"; out += P.outSAMattrRG.at(b.fileOf(ir)); }
        if (!b.extra(imate, ir).empty()) { out.push_back('\t'); out += b.extra(imate, ir); }               // :47-49
        out.push_back('\n');
    }
}

// ---- merged mates back to a pair (--peOverlapNbasesMin) ----
//   Transcript::peOverlapSEtoPE    source/ReadAlign_peOverlapMergeMap.cpp:136-265
//   ReadAlign::peOverlapSEtoPE     source/ReadAlign_peOverlapMergeMap.cpp:267-307
// An alignment of the merged read is cut into the blocks of mate 1 and of mate 2 (the overlap appears in both); scores are recomputed on the pair.
// false = more than MAX_N_EXONS blocks, the alignment is dropped
bool mergedAlignToPair(ChimTr &o, const uint32_t mateStart[2], const staramd_transcript &t, const staramd_exon *tex, uint64_t tLread, const uint64_t readLength[2], uint64_t Lread) {
    uint64_t mLen[2] = {readLength[t.Str], readLength[1 - t.Str]};
    uint64_t mSta2[2] = {0, mLen[0] + 1};
    uint64_t mSta[2] = {mateStart[0], mateStart[1]};
    if (t.Str == 1) { for (int ii = 0; ii < 2; ii++) mSta[ii] = tLread - readLength[ii] - mSta[ii]; std::swap(mSta[0], mSta[1]); }
    uint64_t mEnd[2] = {mSta[0] + mLen[0], mSta[1] + mLen[1]};
    memset(&o, 0, sizeof(o));
    uint32_t nExons = 0;
    for (int imate = 0; imate < 2; imate++) {
        for (uint32_t iex = 0; iex < t.nExons; iex++) {
            const staramd_exon &e = tex[iex];
            if (e.R >= mEnd[imate] || (uint64_t)e.R + e.L <= mSta[imate]) continue;
            if (nExons >= STARAMD_MAX_N_EXONS) return false;
            staramd_exon &x = o.ex[nExons];
            x.iFrag = (uint8_t)(imate == 0 ? t.Str : 1 - t.Str);
            x.sjA = e.sjA;
            if (iex + 1 < t.nExons) { x.canonSJ = e.canonSJ; x.sjAnnot = e.sjAnnot; x.sjStr = e.sjStr; x.shiftSJ[0] = e.shiftSJ[0]; x.shiftSJ[1] = e.shiftSJ[1]; }
            if (e.R >= mSta[imate]) { x.G = e.G; x.L = e.L; x.R = (uint16_t)(e.R - mSta[imate] + mSta2[imate]); }
            else { x.R = (uint16_t)mSta2[imate]; uint64_t delta = mSta[imate] - e.R; x.L = (uint16_t)(e.L - delta); x.G = e.G + delta; }
            if ((uint64_t)e.R + e.L > mEnd[imate]) x.L = (uint16_t)(x.L - ((uint64_t)e.R + e.L - mEnd[imate]));
            ++nExons;
        }
        if (nExons > 0) { staramd_exon &x = o.ex[nExons - 1]; x.canonSJ = -3; x.sjAnnot = 0; x.sjStr = 0; x.shiftSJ[0] = x.shiftSJ[1] = 0; }
    }
    if (nExons == 0) return false;
    staramd_transcript &a = o.t;
    a.nExons = (uint16_t)nExons;
    for (int ii = 0; ii < 3; ii++) a.intronMotifs[ii] = t.intronMotifs[ii];
    a.sjMotifStrand = t.sjMotifStrand; a.Chr = t.Chr; a.Str = t.Str; a.roStr = t.roStr; a.gStart = t.gStart; a.gLength = t.gLength; a.iFrag = -1;
    uint32_t rLength = 0;
    for (uint32_t iex = 0; iex < nExons; iex++) rLength += o.ex[iex].L;
    a.rLength = (uint16_t)rLength; a.mappedLength = rLength; a.rStart = o.ex[0].R;
    a.roStart = (uint16_t)(a.roStr == 0 ? a.rStart : Lread - a.rStart - rLength);
    a.nGap = t.nGap; a.lGap = t.lGap; a.nDel = t.nDel; a.nIns = t.nIns; a.lDel = t.nDel; a.lIns = t.lIns;      // lDel = nDel, as in the reference (:257)
    a.nUnique = t.nUnique; a.nAnchor = t.nAnchor;
    return true;
}

// all alignments of the merged read, window by window; the best of a window moves to its front.  Returns the index of the best one, -1 if nothing converted
static int mergedReadToPair(const RunParams &P, const GenomeIndex &gi, const uint8_t *Read1, uint64_t Lread, const uint64_t readLength[2], const uint32_t mateStart[2],
                            const ReadAligns &se, uint64_t seLread, std::vector<staramd_transcript> &T, std::vector<staramd_exon> &E, uint32_t &nW) {
    T.clear(); E.clear(); nW = 0;
    int best = -1; int64_t bestScore = -10 * (int64_t)Lread;
    std::vector<ChimTr> win;
    for (uint32_t k0 = 0; k0 < se.nTr;) {
        uint32_t k1 = k0;
        while (k1 < se.nTr && se.T[k1].iW == se.T[k0].iW) k1++;
        win.clear();
        for (uint32_t k = k0; k < k1; k++) {
            ChimTr c;
            if (!mergedAlignToPair(c, mateStart, se.T[k], se.ex + se.T[k].exonOffset, seLread, readLength, Lread)) continue;
            chimAlignScore(P.dev, gi, Read1, Lread, c);
            {   // nMatch of Transcript::alignScore (mappedFilter needs it)
                uint32_t nMatch = 0;
                for (uint32_t iex = 0; iex < c.t.nExons; iex++) for (uint32_t ii = 0; ii < c.ex[iex].L; ii++) {
                    uint64_t rp = (uint64_t)c.ex[iex].R + ii;
                    uint8_t r1 = c.t.roStr == 0 ? Read1[rp] : Read1[Lread - 1 - rp];
                    if (c.t.roStr != 0 && r1 < 4) r1 = 3 - r1;
                    if (r1 < 4 && r1 == gi.G[c.ex[iex].G + ii]) ++nMatch;
                }
                c.t.nMatch = nMatch;
            }
            win.push_back(c);
            if (win.back().t.maxScore > win[0].t.maxScore) std::swap(win.back(), win[0]);
        }
        if (!win.empty()) {
            for (ChimTr &c : win) {
                c.t.iW = nW; c.t.exonOffset = (uint32_t)E.size();
                E.insert(E.end(), c.ex, c.ex + c.t.nExons);
                T.push_back(c.t);
            }
            const int head = (int)(T.size() - win.size());
            if (T[head].maxScore > bestScore) { best = head; bestScore = T[head].maxScore; }
            ++nW;
        }
        k0 = k1;
    }
    return best;
}

void PostMap::multCounts(const ReadBatch &b, const staramd_results &r, uint32_t ir, const MergedBatch *merged, const staramd_results *mergedRes, uint64_t &nTr, uint64_t &nbest) const {
    nTr = nbest = 0;
    const staramd_read_result &rr = r.reads[ir];
    const staramd_transcript *T = r.tr + rr.trOffset; uint32_t nTrAll = rr.nTr; int best = (rr.nW > 0 && rr.trBest >= 0) ? rr.trBest : -1;
    std::vector<staramd_transcript> pairT; std::vector<staramd_exon> pairE;
    if (merged && merged->index[ir] >= 0 && mergedRes->reads[merged->index[ir]].nW > 0) {
        const uint32_t mi = (uint32_t)merged->index[ir];
        const staramd_read_result &mr = mergedRes->reads[mi];
        const ReadAligns se{mergedRes->tr + mr.trOffset, mr.nTr, mergedRes->ex};
        const uint64_t Lread = b.readOffset[ir + 1] - b.readOffset[ir];
        const uint64_t readLength[2] = {b.mate1Length[ir], Lread - b.mate1Length[ir] - 1};
        uint32_t nW = 0;
        best = mergedReadToPair(P, gi, b.bases.data() + b.readOffset[ir], Lread, readLength, merged->mateStart[ir].data(), se, merged->reads.readOffset[mi + 1] - merged->reads.readOffset[mi], pairT, pairE, nW);
        T = pairT.data(); nTrAll = (uint32_t)pairT.size();
    }
    if (best < 0) return;
    const int maxScore = T[best].maxScore;
    for (uint32_t k = 0; k < nTrAll; k++) if (T[k].maxScore + P.dev.outFilterMultimapScoreRange >= maxScore) { nTr++; if (T[k].maxScore == maxScore) nbest++; }
}

std::string PostMap::process(const ReadBatch &b, const staramd_results &r, std::string &sam, OutSJ &sj, Stats &st) {
    RangeOut o; o.sam = &sam; o.sj = &sj; o.st = &st;
    return processRange(b, r, 0, b.n, o);
}

// reads [lo, hi) of the batch: the reference's per-thread ReadAlign loop body (ReadAlign_oneRead.cpp:87-111); ranges of one
// batch are independent (per-thread SAM buffer, junction table and Stats, merged by the caller in read order)
// ReadAlign::quantTranscriptome (ReadAlign_quantTranscriptome.cpp:7-91): every alignment of the read projected onto the transcripts that
// contain it; BAM records with the primary flag left off (it is chosen afterwards, see QuantPatch)
static void quantTranscriptome(const RunParams &P, const GenomeIndex &gi, const ReadCtx &rc, const TranscriptAnnotation &A, const std::vector<TrView> &trMult, uint64_t nTr,
                               uint64_t mmMaxTotal, std::string &out, std::vector<QuantPatch> &patches) {
    const ReadBatch &b = *rc.b; const uint32_t ir = rc.i;
    const uint64_t Lread = rc.Lread;
    std::vector<ProjectedAlign> alignT;
    std::vector<const staramd_transcript *> src;
    for (uint64_t iag = 0; iag < nTr; iag++) {
        const staramd_transcript &t = *trMult[iag].t; const staramd_exon *ex = trMult[iag].ex;
        if (!P.quantTrIndel && (t.nDel > 0 || t.nIns > 0)) continue;
        if (!P.quantTrSingleEnd && rc.nMates == 2 && ex[0].iFrag == ex[t.nExons - 1].iFrag) continue;
        GenomicAlign g; g.nExons = t.nExons; g.Str = t.Str; g.Lread = Lread;
        memcpy(g.ex, ex, sizeof(staramd_exon) * t.nExons);
        if (!P.quantTrSoftClip) {                                          // soft clips are extended instead (:23-61)
            uint64_t nMM1 = 0;
            const uint8_t *rd = b.bases.data() + b.readOffset[ir];
            auto R = [&](uint64_t p) -> uint8_t { uint8_t c = t.roStr == 0 ? rd[p] : rd[Lread - 1 - p]; return (t.roStr != 0 && c < 4) ? (uint8_t)(3 - c) : c; };
            // the reference's G sits inside G1 with spacer padding either side (Genome_genomeLoad.cpp:27,320-323): a soft clip that hangs over
            // an end of the genome array compares against the padding code there
            const uint64_t nG = gi.view.nGenome;
            auto GG = [&](uint64_t p) -> uint8_t { return p < nG ? gi.G[p] : (uint8_t)5; };      // p - k below base 0 wraps to a huge value: padding
            for (uint32_t iab = 0; iab < g.nExons; iab++) {
                uint64_t left1 = 0, right1 = 0;
                if (iab == 0) left1 = g.ex[iab].R;
                else if (g.ex[iab - 1].canonSJ == -3) left1 = g.ex[iab].R - rc.readLength[g.ex[iab - 1].iFrag] - 1;
                if (iab == g.nExons - 1) right1 = Lread - g.ex[iab].R - g.ex[iab].L;
                else if (g.ex[iab].canonSJ == -3) right1 = rc.readLength[g.ex[iab].iFrag] - g.ex[iab].R - g.ex[iab].L;
                for (uint64_t k = 1; k <= left1; k++) { uint8_t r1 = R(g.ex[iab].R - k), g1 = GG(g.ex[iab].G - k); if (r1 != g1 && r1 < 4 && g1 < 4) ++nMM1; }
                for (uint64_t k = 0; k < right1; k++) { uint8_t r1 = R((uint64_t)g.ex[iab].R + g.ex[iab].L + k), g1 = GG(g.ex[iab].G + g.ex[iab].L + k); if (r1 != g1 && r1 < 4 && g1 < 4) ++nMM1; }
                g.ex[iab].R = (uint16_t)(g.ex[iab].R - left1); g.ex[iab].G -= left1; g.ex[iab].L = (uint16_t)(g.ex[iab].L + left1 + right1);
            }
            if (t.nMM + nMM1 > std::min<uint64_t>(mmMaxTotal, (uint64_t)(P.dev.outFilterMismatchNoverLmax * (double)(Lread - 1)))) continue;
        }
        size_t n0 = alignT.size();
        A.quantAlign(g, alignT);
        for (size_t k = n0; k < alignT.size(); k++) src.push_back(&t);
    }
    QuantPatch qp; qp.ir = ir; qp.nAlignT = (uint32_t)alignT.size();
    for (size_t iatr = 0; iatr < alignT.size(); iatr++) {
        staramd_transcript tq = *src[iatr];
        tq.Chr = alignT[iatr].tr; tq.Str = (uint8_t)alignT[iatr].Str; tq.nExons = (uint16_t)alignT[iatr].nExons;
        TrView v; v.t = &tq; v.ex = alignT[iatr].ex; v.primary = false;
        size_t nBefore = qp.recOffset.size();
        bamMapped(out, P, gi, rc, v, alignT.size(), iatr, nullptr, true, &qp.recOffset);
        for (size_t k = nBefore; k < qp.recOffset.size(); k++) qp.recAlign.push_back((uint32_t)iatr);
    }
    patches.push_back(std::move(qp));
}

// recordSJ (ReadAlign_outputAlignments.cpp:76-87) -> outputTranscriptSJ (ReadAlign_outputTranscriptSJ.cpp:14-55)
static void recordSJ(const RunParams &P, const std::vector<TrView> &trMult, uint64_t nTr, OutSJ &sj) {
    if (P.outSJfilterReadsUnique && nTr != 1) return;
    size_t sjReadStartN = sj.data.size();
    for (uint64_t it = 0; it < nTr; it++) {
        const staramd_transcript &t = *trMult[it].t; const staramd_exon *ex = trMult[it].ex;
        for (uint32_t iex = 0; iex + 1 < t.nExons; iex++) {
            if (ex[iex].canonSJ < 0) continue;
            Junction j; j.start = ex[iex].G + ex[iex].L; j.gap = (uint32_t)(ex[iex + 1].G - j.start);
            j.overhangLeft = (uint16_t)std::min<uint32_t>(ex[iex].L, ex[iex + 1].L); j.overhangRight = j.overhangLeft;
            bool dup = false;
            for (size_t ii = sjReadStartN; ii < sj.data.size(); ii++) {
                if (sj.data[ii].start == j.start && sj.data[ii].gap == j.gap) {
                    dup = true;
                    if (sj.data[ii].overhangLeft < j.overhangLeft) { sj.data[ii].overhangLeft = j.overhangLeft; sj.data[ii].overhangRight = j.overhangLeft; }
                    break;
                }
            }
            if (dup) continue;
            j.motif = ex[iex].canonSJ; j.strand = (int8_t)(ex[iex].canonSJ == 0 ? 0 : (ex[iex].canonSJ + 1) % 2 + 1); j.annot = (int8_t)ex[iex].sjAnnot;
            if (nTr == 1) { j.countUnique = 1; j.countMultiple = 0; } else { j.countMultiple = 1; j.countUnique = 0; }
            sj.data.push_back(j);
        }
    }
}

std::string PostMap::processRange(const ReadBatch &b, const staramd_results &r, uint32_t lo, uint32_t hi, const RangeOut &out) const { return processRange(b, r, lo, hi, out, RangeIn()); }

std::string PostMap::processRange(const ReadBatch &b, const staramd_results &r, uint32_t lo, uint32_t hi, const RangeOut &out, const RangeIn &in) const {
    std::string &sam = *out.sam; OutSJ &sj = *out.sj; Stats &st = *out.st;
    OutSJ *const sj1 = out.sj1; std::vector<uint32_t> *const held = out.held; GeneCounts *const gc = out.gc; std::vector<BamKey> *const bamKeys = out.bamKeys;
    std::string *const unmappedFastx = out.unmappedFastx, *const chimJunction = out.chimJunction, *const chimSam = out.chimSam, *const quantBam = out.quantBam;
    std::vector<QuantPatch> *const quantPatches = out.quantPatches;
    const MultOrder *const order = in.order; const bool dry = in.dry; const MergedBatch *const merged = in.merged; const staramd_results *const mergedRes = in.mergedRes;
    const std::vector<int8_t> *const waspType = in.waspType;
    const bool bam = P.outBAMunsorted || P.outBAMcoord;
    std::vector<staramd_transcript> pairT; std::vector<staramd_exon> pairE;
    const bool samOff = this->samOff || dry;
    std::vector<TrView> trMult;
    int waspPrev = -1;
    if (waspType && !in.probeChimBam) {
        // The BAM records of a chimeric read (--chimOutType WithinBAM) show the verdict of the read BEFORE it (ReadAlign_oneRead.cpp:99-103:
        // waspMap does not run for such a read) -- i.e. of the nearest earlier read that was not such a read itself.  Ranges of a batch run
        // on threads: the reads in front of this range are probed (chimeric detection only, no output) until one of the ordinary kind is found.
        int64_t j = (int64_t)lo - 1;
        if (P.chim.outBam && chimJunction)
            for (; j >= 0; j--) {
                bool chimBam = false;
                std::string sam0, cj0; OutSJ sj0; Stats st0;
                RangeOut po; po.sam = &sam0; po.sj = &sj0; po.st = &st0; po.chimJunction = &cj0;
                RangeIn pi = in; pi.dry = true; pi.probeChimBam = &chimBam;
                processRange(b, r, (uint32_t)j, (uint32_t)j + 1, po, pi);
                if (!chimBam) break;
            }
        waspPrev = j >= 0 ? (*waspType)[j] : waspCarry;
    }
    for (uint32_t ir = lo; ir < hi; ir++) {
        const staramd_read_result &rr = r.reads[ir];
        if (rr.status & STARAMD_ST_FATAL_SEEDS_PER_READ)
            return "EXITING because of FATAL error: too many pieces pere read\nSOLUTION: increase input parameter --seedPerReadNmax";
        if (rr.status & STARAMD_ST_SCRATCH_OVERFLOW)
            return "EXITING because of FATAL error: a device work-space cap was exceeded for read " + std::string(b.name(ir));
        ReadCtx rc; rc.b = &b; rc.i = ir; rc.nMates = (int)P.dev.readNmates;
        rc.Lread = b.readOffset[ir + 1] - b.readOffset[ir];
        rc.readLength[0] = b.mate1Length[ir]; rc.readLength[1] = rc.nMates == 2 ? rc.Lread - rc.readLength[0] - 1 : 0;
        for (int m = 0; m < 2; m++) {
            rc.readLengthOriginal[m] = m < rc.nMates ? b.seqSpan[m][ir].len : 0;
            for (int q = 0; q < 2; q++) rc.clip[m][q] = m < rc.nMates ? b.clipped(m, q, ir) : 0;
        }
        rc.waspType = waspType ? (*waspType)[ir] : -1;
        const int waspOfRead = rc.waspType;
        st.readN++; st.readBases += rc.readLength[0] + rc.readLength[1];
        const staramd_transcript *T = r.tr + rr.trOffset;
        const staramd_exon *EX = r.ex;
        uint32_t nTrAll = rr.nTr;
        uint64_t nW = rr.nW;
        const staramd_transcript *trBest = (nW > 0 && rr.trBest >= 0) ? T + rr.trBest : nullptr;
        // ---- peOverlapMergeMap (ReadAlign_peOverlapMergeMap.cpp:4-75): where the merged mates mapped, their alignments, cut back into the two mates, replace the pair's
        bool peOvYes = false, chimRecord = false;
        std::vector<ChimPair> chimPairs, *cpp = (chimJunction && (P.chim.outBam || P.chim.outSamOld)) ? &chimPairs : nullptr;
        if (merged && merged->index[ir] >= 0 && mergedRes->reads[merged->index[ir]].nW > 0) {
            const uint32_t mi = (uint32_t)merged->index[ir];
            const staramd_read_result &mr = mergedRes->reads[mi];
            const ReadAligns se{mergedRes->tr + mr.trOffset, mr.nTr, mergedRes->ex};
            const uint64_t seLread = merged->reads.readOffset[mi + 1] - merged->reads.readOffset[mi];
            const int peScore = trBest ? trBest->maxScore : 0;
            uint32_t nWpair = 0;
            const int best = mergedReadToPair(P, gi, b.bases.data() + b.readOffset[ir], rc.Lread, rc.readLength, merged->mateStart[ir].data(), se, seLread, pairT, pairE, nWpair);
            T = pairT.data(); EX = pairE.data(); nTrAll = (uint32_t)pairT.size(); nW = nWpair;
            trBest = best >= 0 ? T + best : nullptr;
            if (chimJunction) {                                  // chimericDetectionPEmerged (ReadAlign_chimericDetectionPEmerged.cpp:5-39): on the merged read's own alignments
                const staramd_transcript *seBest = mr.trBest >= 0 ? se.T + mr.trBest : nullptr;
                if (P.chim.multimapNmax == 0 && seBest) {
                    // the merged read's multMapSelect: how many alignments are within the score range of its best, and the first two of them
                    uint64_t nTrSE = 0; const staramd_transcript *m0 = nullptr, *m1 = nullptr;
                    for (uint32_t k = 0; k < se.nTr; k++) if (se.T[k].maxScore + P.dev.outFilterMultimapScoreRange >= seBest->maxScore) { if (nTrSE == 0) m0 = se.T + k; else if (nTrSE == 1) m1 = se.T + k; nTrSE++; }
                    chimRecord = chimericDetectionOld(P, gi, merged->reads, mi, se, seBest, nTrSE, m0, m1, *chimJunction, cpp, &b, ir, merged->mateStart[ir].data());
                } else if (P.chim.multimapNmax > 0 && seBest && (trBest ? trBest->maxScore : 0) <= (int)(rc.readLength[0] + rc.readLength[1]) - (int)P.chim.nonchimScoreDropMin)
                    chimRecord = chimericDetectionMult(P, gi, merged->reads, mi, se, seBest, *chimJunction, cpp, &b, ir, merged->mateStart[ir].data());
                if (chimRecord) st.chimericAll++;
            }
            if (peScore <= (trBest ? trBest->maxScore : 0) || chimRecord) peOvYes = true;
        }
        // ---- multMapSelect
        trMult.clear();
        uint64_t nTr = 0;
        if (nW > 0) {
            int maxScore = trBest->maxScore;     // == max over windows' heads (asserted in the reference :20-24)
            for (uint32_t k = 0; k < nTrAll; k++)
                if (T[k].maxScore + P.dev.outFilterMultimapScoreRange >= maxScore) { TrView v; v.t = T + k; v.ex = EX + T[k].exonOffset; v.primary = false; trMult.push_back(v); }
            nTr = trMult.size();
            if (!(nTr > P.outFilterMultimapNmax || nTr == 0)) {
                if (nTr == 1) trMult[0].primary = true;
                else {
                    if (P.outSAMmultNmax >= 0 || P.outMultimapperRandom) {      // :61-68 the best alignments move to the top of the list (they are the ones that get written)
                        uint64_t nbest = 0;
                        for (uint64_t it = 0; it < nTr; it++) if (trMult[it].t->maxScore == maxScore) { std::swap(trMult[it], trMult[nbest]); ++nbest; }
                        if (P.outMultimapperRandom && order) {                  // :71-80 the best ones and the rest are shuffled separately
                            const uint32_t *partner = order->partner.data() + order->offset[ir];
                            for (int itr = (int)nbest - 1; itr >= 1; itr--) std::swap(trMult[itr], trMult[*partner++]);
                            for (int itr = (int)(nTr - nbest) - 1; itr >= 1; itr--) std::swap(trMult[nbest + itr], trMult[nbest + *partner++]);
                        }
                    }
                    if (P.outSAMprimaryAllBest) { for (auto &v : trMult) if (v.t->maxScore == maxScore) v.primary = true; }
                    else if (P.outSAMmultNmax >= 0 || P.outMultimapperRandom) trMult[0].primary = true;
                    else { for (auto &v : trMult) if (v.t == trBest) v.primary = true; }
                }
            }
        }
        // ---- mappedFilter
        int unmapType = -1;
        if (nW == 0) { st.unmappedOther++; unmapType = 0; }
        else if ((trBest->maxScore < P.outFilterScoreMin) || (trBest->maxScore < (int)(P.outFilterScoreMinOverLread * (double)(rc.Lread - 1)))
                 || (trBest->nMatch < P.outFilterMatchNmin) || (trBest->nMatch < (uint64_t)(P.outFilterMatchNminOverLread * (double)(rc.Lread - 1)))) { st.unmappedShort++; unmapType = 1; }
        else if ((trBest->nMM > b.mmMaxTotal[ir]) || (double(trBest->nMM) / double(trBest->rLength) > P.dev.outFilterMismatchNoverLmax)) { st.unmappedMismatch++; unmapType = 2; }
        else if (nTr > P.outFilterMultimapNmax) { st.unmappedMulti++; unmapType = 3; }
        // ---- chimericDetection (ReadAlign_oneRead.cpp:95-97; not in the 2nd stage of BySJout, ReadAlign_chimericDetection.cpp:23)
        if (chimJunction && nW > 0 && P.dev.outFilterBySJoutStage <= 1 && !peOvYes) {       // ReadAlign_oneRead.cpp:95-97
            ReadAligns ra{T, nTrAll, EX};
            ChimPre pre{-1, 0, 0, 0};
            if (P.dev.resultSelect == 2) {              // the engine chose the partner (include/star_amd.h)
                if (rr.status & STARAMD_ST_CHIM_PARTNER) { pre.partner = (int32_t)(rr.unmappedLength & 0x3FFFFFFFu); pre.strBest = rr.unmappedLength >> 30; pre.scoreBest = rr.maxScoreMate[0]; pre.scoreNext = rr.maxScoreMate[1]; }
                ra.pre = &pre;
            }
            chimRecord = false;
            if (P.chim.multimapNmax == 0) chimRecord = chimericDetectionOld(P, gi, b, ir, ra, trBest, nTr, nTr > 0 ? trMult[0].t : nullptr, nTr > 1 ? trMult[1].t : nullptr, *chimJunction, cpp);
            else if (trBest->maxScore <= (int)(rc.readLength[0] + rc.readLength[1]) - (int)P.chim.nonchimScoreDropMin)       // ReadAlign_chimericDetection.cpp:48
                chimRecord = chimericDetectionMult(P, gi, b, ir, ra, trBest, *chimJunction, cpp);
            if (chimRecord) st.chimericAll++;
        }
        if (in.probeChimBam) { *in.probeChimBam = chimRecord && P.chim.outBam; return ""; }
        if (chimRecord && P.chim.outSamOld && chimSam) for (const ChimPair &cp : chimPairs) chimSamOldOutput(*chimSam, P, gi, rc, cp);
        if (chimRecord && P.chim.outBam) rc.waspType = waspPrev;     // waspMap does not run for such a read: its records show the verdict of the read before it (ReadAlign_oneRead.cpp:99-103)
        else waspPrev = waspOfRead;
        if (chimRecord && P.chim.outBam) {          // the chimera stands for the read in the BAM: nothing else is output or counted for it (ReadAlign_oneRead.cpp:99-101)
            if (!samOff) for (size_t k = 0; k < chimPairs.size(); k++) chimBamOutput(sam, P, gi, rc, chimPairs[k], k, chimPairs.size(), bamKeys);
            continue;
        }
        // ---- outFilterBySJout, 1st stage (ReadAlign_outputAlignments.cpp:90-124)
        if (sj1 && unmapType <= 0) {
            bool pass = true;
            for (uint64_t it = 0; it < nTr && pass; it++)
                for (uint32_t k = 0; k + 1 < trMult[it].t->nExons; k++) if (trMult[it].ex[k].canonSJ >= 0 && trMult[it].ex[k].sjAnnot == 0) { pass = false; break; }
            recordSJ(P, trMult, nTr, *sj1);                 // junctions of every read, held or not
            if (!pass) { st.readN--; st.readBases -= rc.readLength[0] + rc.readLength[1]; held->push_back(ir); continue; }
        }
        // ---- outputAlignments
        bool mateMapped[2] = {false, false};
        if (unmapType < 0) {
            if (nTr > 1) { st.mappedReadsM++; unmapType = -2; }
            else if (nTr == 1) {
                st.mappedReadsU++;
                const staramd_transcript &t = *trMult[0].t; const staramd_exon *ex = trMult[0].ex;
                st.mappedMismatchesN += t.nMM; st.mappedInsN += t.nIns; st.mappedDelN += t.nDel; st.mappedInsL += t.lIns; st.mappedDelL += t.lDel;
                uint64_t mappedL = 0;
                for (uint32_t k = 0; k < t.nExons; k++) mappedL += ex[k].L;
                for (uint32_t k = 0; k + 1 < t.nExons; k++) { if (ex[k].canonSJ >= 0) st.splicesN[ex[k].canonSJ]++; if (ex[k].sjAnnot == 1) st.splicesNsjdb++; }
                st.mappedBases += mappedL; st.mappedPortion += double(mappedL) / double(rc.Lread);
            }
            recordSJ(P, trMult, nTr, sj);
            if (gc && nTr > 0) gc->addAlign(*genes, nTr, *trMult[0].t, trMult[0].ex);        // alignedAnnotation (ReadAlign_outputAlignments.cpp:298-308)
            if (quantBam) quantTranscriptome(P, gi, rc, *transcripts, trMult, nTr, b.mmMaxTotal[ir], *quantBam, *quantPatches);
            // writeSAM (:132-256), default outSAMmultNmax=-1: all nTr
            // writeSAM (:132-256): at most --outSAMmultNmax alignments are written (NH keeps the full count)
            const uint64_t nTrWrite = P.outSAMmultNmax < 0 ? nTr : std::min<uint64_t>(nTr, (uint64_t)P.outSAMmultNmax);
            const bool keepPairs = P.outSAMunmappedKeepPairs;
            if (!samOff) for (uint64_t it = 0; it < nTrWrite; it++) {
                if (bam) bamMapped(sam, P, gi, rc, trMult[it], nTr, it, bamKeys); else samMapped(sam, P, gi, rc, trMult[it], nTr, it);
                if (keepPairs && (!P.outBAMcoord || P.outBAMunsorted)) {   // ReadAlign_outputAlignments.cpp:178-195: the unmapped mate right after every one-mate alignment (no sort key: not in the sorted BAM)
                    const staramd_transcript &t = *trMult[it].t;
                    bool mateMapped1[2] = {false, false};
                    mateMapped1[trMult[it].ex[0].iFrag] = true; mateMapped1[trMult[it].ex[t.nExons - 1].iFrag] = true;
                    if (!(mateMapped1[0] && mateMapped1[1])) {
                        if (bam) bamUnmapped(sam, P, gi, rc, &t, trMult[it].ex, 4, mateMapped1, nullptr, !trMult[it].primary); else samUnmapped(sam, P, gi, rc, &t, trMult[it].ex, 4, mateMapped1, !trMult[it].primary);
                    }
                }
            }
            const staramd_exon *exB = EX + trBest->exonOffset;
            mateMapped[exB[0].iFrag] = true; mateMapped[exB[trBest->nExons - 1].iFrag] = true;
            if (rc.nMates > 1 && !(mateMapped[0] && mateMapped[1])) unmapType = 4;
            if (unmapType == 4 && P.outSAMunmappedWithin && !samOff && (!keepPairs || P.outBAMcoord)) {     // :216-233
                bool trBestSecondary = false;
                if (keepPairs) { trBestSecondary = true; for (uint64_t it = 0; it < nTr; it++) if (trMult[it].t == trBest && trMult[it].primary) trBestSecondary = false; }
                if (bam) {
                    const size_t k0 = bamKeys ? bamKeys->size() : 0;
                    bamUnmapped(sam, P, gi, rc, trBest, exB, unmapType, mateMapped, bamKeys, trBestSecondary);
                    if (keepPairs && P.outBAMunsorted && bamKeys) for (size_t k = k0; k < bamKeys->size(); k++) (*bamKeys)[k].len |= 0x80000000u;   // for the sorted BAM only: cut out of the unsorted stream
                } else samUnmapped(sam, P, gi, rc, trBest, exB, unmapType, mateMapped);
            }
        } else if (P.outSAMunmappedWithin && quantBam && samOff) {        // no alignment output, but the transcriptome BAM still gets the unmapped reads (:237-245)
            staramd_transcript t0; memset(&t0, 0, sizeof(t0));
            bamUnmapped(*quantBam, P, gi, rc, trBest ? trBest : &t0, trBest ? EX + trBest->exonOffset : nullptr, unmapType, mateMapped, nullptr);
        } else if (P.outSAMunmappedWithin && !samOff) {
            staramd_transcript t0; memset(&t0, 0, sizeof(t0));
            if (quantBam) bamUnmapped(*quantBam, P, gi, rc, trBest ? trBest : &t0, trBest ? EX + trBest->exonOffset : nullptr, unmapType, mateMapped, nullptr);   // :243-245
            if (bam) bamUnmapped(sam, P, gi, rc, trBest ? trBest : &t0, trBest ? EX + trBest->exonOffset : nullptr, unmapType, mateMapped, bamKeys);
            else samUnmapped(sam, P, gi, rc, trBest ? trBest : &t0, trBest ? EX + trBest->exonOffset : nullptr, unmapType, mateMapped);
        }
        if (unmapType >= 0) {
            st.unmappedAll++;
            if (unmappedFastx) {                 // outReadsUnmapped (ReadAlign_outputAlignments.cpp:259-275): both mates, also of one-mate alignments
                for (int im = 0; im < rc.nMates; im++) {
                    std::string &u = unmappedFastx[im];
                    u.push_back(b.fasta ? '>' : '@'); u += b.name(ir); u.push_back(' '); u.push_back((char)('0' + im)); u.push_back(':'); u.push_back(b.filter[ir]); u += ": "; u += b.extra(im, ir);
                    if (rc.nMates > 1) { u.push_back(' '); u.push_back(mateMapped[0] ? '1' : '0'); u.push_back(mateMapped[1] ? '1' : '0'); }
                    u.push_back('\n'); u += b.seq(im, ir); u.push_back('\n');
                    if (!b.fasta) { u += "+\n"; u += b.qual(im, ir); u.push_back('\n'); }
                }
            }
        }
    }
    if (out.waspEnd) *out.waspEnd = waspPrev;
    return "";
}

} // namespace staramd
