// reads.cpp -- FASTQ text -> numeric read batches.
// Reproduces, for FASTQ input, what reaches ReadAlign::mapOneRead in the reference:
//   ReadAlignChunk::processChunks FASTQ branch   source/ReadAlignChunk_processChunks.cpp:111-157
//   readLoad                                      source/readLoad.cpp:4-100
//   convertNucleotidesToNumbers                   source/SequenceFuns.cpp:131-146
//   PE concatenation in ReadAlign::oneRead        source/ReadAlign_oneRead.cpp:35-78
#include "host.h"
#include <cstring>
#include <algorithm>

namespace staramd {

staramd_batch ReadBatch::view() const {
    staramd_batch b;
    b.nReads = n; b.bases = bases.data(); b.readOffset = readOffset.data();
    b.mate1Length = mate1Length.data(); b.mmMaxTotal = mmMaxTotal.data();
    return b;
}
void ReadBatch::clear() {
    n = 0; bases.clear(); readOffset.assign(1, 0); mate1Length.clear(); mmMaxTotal.clear();
    name.clear(); filter.clear();
    for (int i = 0; i < 2; i++) { seq[i].clear(); qual[i].clear(); }
}

FastqReader::~FastqReader() { for (int i = 0; i < 2; i++) if (f[i]) fclose(f[i]); }

std::string FastqReader::open(const std::vector<std::string> &paths) {
    nMates = (int)paths.size();
    for (int i = 0; i < nMates; i++) {
        f[i] = fopen(paths[i].c_str(), "rb");
        if (!f[i]) return "EXITING because of fatal input ERROR: could not open readFilesIn=" + paths[i];
        setvbuf(f[i], nullptr, _IOFBF, 1 << 22);
    }
    lineBuf.resize(1 << 16);
    return "";
}

std::string FastqReader::reopen() {
    for (int i = 0; i < nMates; i++) if (fseek(f[i], 0, SEEK_SET) != 0) return "EXITING because of fatal input ERROR: could not rewind the read file";
    readsSoFar = 0;
    return "";
}

bool FastqReader::getLine(int im, std::string &out) {
    out.clear();
    for (;;) {
        if (!fgets(lineBuf.data(), (int)lineBuf.size(), f[im])) return !out.empty();
        size_t l = strlen(lineBuf.data());
        bool eol = l > 0 && lineBuf[l - 1] == '\n';
        out.append(lineBuf.data(), eol ? l - 1 : l);
        if (eol) break;
    }
    // fastqReadOneLine strips a trailing control character (\r)
    if (!out.empty() && (unsigned char)out.back() < 33) out.pop_back();
    return true;
}

static inline uint8_t nt2num(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

bool FastqReader::nextBatch(ReadBatch &b, const RunParams &P, uint64_t maxReads, std::string &err) {
    b.clear();
    b.firstReadIndex = readsSoFar;
    std::string l1[2], s[2], plus, q[2];
    while (b.n < maxReads) {
        if (P.readMapNumber >= 0 && (int64_t)readsSoFar >= P.readMapNumber) break;
        if (!getLine(0, l1[0]) || l1[0].empty()) break;
        if (l1[0][0] != '@') { err = "EXITING because of FATAL ERROR in input reads: wrong read ID line format: the read ID lines should start with @ (FASTA/SAM input: out of scope)"; return false; }
        if (nMates == 2 && !getLine(1, l1[1])) { err = "EXITING because of FATAL ERROR: read files are not consistent, reached the end of the one before the other one"; return false; }
        for (int im = 0; im < nMates; im++) {
            if (!getLine(im, s[im]) || !getLine(im, plus) || !getLine(im, q[im])) { err = "EXITING because of FATAL ERROR in reads input: truncated FASTQ record"; return false; }
            if (s[im].size() < 1) { err = "EXITING because of FATAL ERROR in reads input: short read sequence line: 0"; return false; }
            if (s[im].size() > STARAMD_READ_LEN_MAX) { err = "EXITING because of FATAL ERROR in reads input: Lread>DEF_readSeqLengthMax"; return false; }
            if (q[im].size() != s[im].size()) { err = "EXITING because of FATAL ERROR in reads input: quality string length is not equal to sequence length"; return false; }
        }
        // read ID: first white-space token of mate 1's line, then trimmed at readNameSeparator chars
        size_t e = l1[0].find_first_of(" \t");
        std::string id = l1[0].substr(1, e == std::string::npos ? std::string::npos : e - 1);
        char pf = 'N';
        if (e != std::string::npos) {
            size_t f2 = l1[0].find_first_not_of(" \t", e);
            if (f2 != std::string::npos) {
                std::string field2 = l1[0].substr(f2, l1[0].find_first_of(" \t", f2) - f2);
                if (field2.length() >= 3 && field2[1] == ':' && field2[2] == 'Y' && field2[3] == ':') pf = 'Y';
            }
        }
        for (char c : P.readNameSeparator) { size_t p = id.find(c); if (p != std::string::npos) id.resize(p); }
        uint64_t len0 = s[0].size(), len1 = nMates == 2 ? s[1].size() : 0;
        uint64_t Lread = nMates == 2 ? len0 + len1 + 1 : len0;
        if (Lread > STARAMD_READ_LEN_MAX) { err = "EXITING because of FATAL ERROR in reads input: Lread of the pair > DEF_readSeqLengthMax"; return false; }
        size_t off = b.bases.size();
        b.bases.resize(off + Lread);
        uint8_t *r = b.bases.data() + off;
        for (uint64_t i = 0; i < len0; i++) r[i] = nt2num(s[0][i]);
        if (nMates == 2) {
            r[len0] = STARAMD_SPACER_BASE;
            for (uint64_t i = 0; i < len1; i++) { uint8_t c = nt2num(s[1][len1 - 1 - i]); r[len0 + 1 + i] = c < 4 ? 3 - c : c; }
        }
        b.readOffset.push_back(off + Lread);
        b.mate1Length.push_back((uint16_t)len0);
        // ReadAlign_oneRead.cpp:78
        b.mmMaxTotal.push_back((uint16_t)std::min<uint64_t>(P.outFilterMismatchNmax, (uint64_t)(P.outFilterMismatchNoverReadLmax * (double)(len0 + len1))));
        b.name.push_back(id); b.filter.push_back(pf);
        for (int im = 0; im < nMates; im++) { b.seq[im].push_back(s[im]); b.qual[im].push_back(q[im]); }
        b.n++; readsSoFar++;
    }
    return b.n > 0;
}

} // namespace staramd
