// reads.cpp -- FASTQ text -> numeric read batches.
// Reproduces, for FASTQ input, what reaches ReadAlign::mapOneRead in the reference:
//   ReadAlignChunk::processChunks FASTQ branch   source/ReadAlignChunk_processChunks.cpp:111-157
//   readLoad                                      source/readLoad.cpp:4-100
//   convertNucleotidesToNumbers                   source/SequenceFuns.cpp:131-146
//   PE concatenation in ReadAlign::oneRead        source/ReadAlign_oneRead.cpp:35-78
// The reference reads line by line under a mutex; here the text comes in blocks, line boundaries are
// found with memchr, and the records of a batch are converted on --runThreadN threads.  The batch keeps the text itself;
// names, sequences and qualities are spans into it (nothing is copied per read until the SAM line is formatted).
#include "host.h"
#include <sys/stat.h>
#include <sys/mman.h>
#include <unistd.h>
#include <cstring>
#include <algorithm>
#include <thread>
#include <atomic>
#include <mutex>
#include <functional>
#include <immintrin.h>
#include <chrono>
#include <unistd.h>

namespace staramd {
static std::atomic<uint64_t> g_cpuNs[CPU_NSTAGE];
void cpuAdd(int stage, uint64_t ns) { if (stage >= 0 && stage < CPU_NSTAGE) g_cpuNs[stage] += ns; }
uint64_t cpuTake(int stage, bool reset) { if (stage < 0 || stage >= CPU_NSTAGE) return 0; return reset ? g_cpuNs[stage].exchange(0) : g_cpuNs[stage].load(); }


void *(*g_batchAllocFn)(uint64_t bytes) = nullptr;
void (*g_batchFreeFn)(void *p) = nullptr;
// blocks that came from malloc although a hook is set (the hook returned nothing: page-locked memory can run out; or they were allocated before the hook was set)
static std::mutex g_plainM; static std::vector<void *> g_plain; static std::atomic<size_t> g_plainN{0};
void *batchAllocate(size_t bytes) {
    if (g_batchAllocFn) { if (void *p = g_batchAllocFn((uint64_t)bytes)) return p; }
    void *p = malloc(bytes);
    if (p) { std::lock_guard<std::mutex> l(g_plainM); g_plain.push_back(p); g_plainN++; }
    return p;
}
void batchRelease(void *p) {
    if (!p) return;
    if (g_plainN.load() > 0) {
        std::lock_guard<std::mutex> l(g_plainM);
        for (size_t i = 0; i < g_plain.size(); i++) if (g_plain[i] == p) { g_plain[i] = g_plain.back(); g_plain.pop_back(); g_plainN--; free(p); return; }
    }
    if (g_batchFreeFn) g_batchFreeFn(p); else free(p);
}

staramd_batch ReadBatch::view() const {
    staramd_batch b;
    b.nReads = n; b.bases = bases.data(); b.readOffset = readOffset.data();
    b.mate1Length = mate1Length.data(); b.mmMaxTotal = mmMaxTotal.data();
    return b;
}
void ReadBatch::clear() {
    n = 0; bases.clear(); readOffset.assign(1, 0); mate1Length.clear(); mmMaxTotal.clear();
    nameSpan.clear(); filter.clear(); origIndex.clear(); heldFile.clear();
    for (int m = 0; m < 2; m++) for (int q = 0; q < 2; q++) clipN[m][q].clear();
    for (int i = 0; i < 2; i++) { text[i].clear(); mapped[i] = nullptr; seqSpan[i].clear(); qualSpan[i].clear(); extraSpan[i].clear(); lineStart[i].clear(); lineEnd[i].clear(); }
}

void FastqReader::closeFiles() {
    for (int i = 0; i < 2; i++) if (f[i]) { if (command_.empty()) fclose(f[i]); else pclose(f[i]); f[i] = nullptr; }
}
void FastqReader::dropMaps() { for (Mapped &m : allMaps) if (m.p) munmap((void *)m.p, m.n); allMaps.clear(); curMap[0] = curMap[1] = Mapped(); }
FastqReader::~FastqReader() { closeFiles(); dropMaps(); }

std::string FastqReader::openCurrent() {
    for (int i = 0; i < (samMates_ > 0 ? 1 : nMates); i++) {
        const std::string &path = files_[i][curFile];
        if (command_.empty()) f[i] = fopen(path.c_str(), "rb");
        else {
            if (access(path.c_str(), R_OK) != 0) return "EXITING because of fatal input ERROR: could not open readFilesIn=" + path;
            std::string esc;
            for (char c : path) { if (c == '\'') esc += "'\\''"; else esc.push_back(c); }
            f[i] = popen((command_ + " '" + esc + "'").c_str(), "r");
        }
        if (!f[i]) return "EXITING because of fatal input ERROR: could not open readFilesIn=" + path;
        setvbuf(f[i], nullptr, _IONBF, 0);          // blocks are read straight into the batch text
        carry[i].clear(); eof[i] = false;
        curMap[i] = Mapped(); mapPos[i] = 0; useMap = -1;
        if (command_.empty() && samMates_ == 0 && !getenv("STARAMD_NO_INPUT_MMAP")) {
            struct stat st;
            if (fstat(fileno(f[i]), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
                void *p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f[i]), 0);
                if (p != MAP_FAILED) { (void)madvise(p, (size_t)st.st_size, MADV_SEQUENTIAL); curMap[i].p = (const char *)p; curMap[i].n = (size_t)st.st_size; allMaps.push_back(curMap[i]); }
            }
        }
    }
    {   // the first character of the first mate's file decides the format (ReadAlignChunk_processChunks.cpp:111,158)
        char c = 0;
        size_t got = fread(&c, 1, 1, f[0]);
        if (got == 1) carry[0].push_back(c);
        fasta = got == 1 && c == '>';
    }
    return "";
}

std::string FastqReader::open(const std::vector<std::string> &paths, const std::string &readCommand, int samMates) {
    closeFiles(); dropMaps(); useMap = -1;          // (every batch of the pass before has been written by now)
    ioError = 0;                                    // (a read error of an earlier pass / file was reported then: not sticky)
    nMates = (int)paths.size(); paths_ = paths; command_ = readCommand; fromMemory = false; samMates_ = samMates; extras = samMates > 0;
    for (int i = 0; i < nMates; i++) {              // --readFilesIn a1,a2,... b1,b2,...: the files of a mate are read one after the other
        files_[i].clear();
        size_t p0 = 0;
        for (;;) { size_t c = paths[i].find(',', p0); files_[i].push_back(paths[i].substr(p0, c == std::string::npos ? std::string::npos : c - p0)); if (c == std::string::npos) break; p0 = c + 1; }
    }
    if (nMates == 2 && files_[0].size() != files_[1].size()) return "EXITING: because of fatal INPUT ERROR: number of input files for mate 1 is not equal to that for mate 2";
    curFile = 0; readsSoFar = 0;
    std::string e = openCurrent();
    if (samMates_ > 0) { nMates = samMates_; fasta = false; }      // one stream, both mates
    return e;
}

void FastqReader::openMemory(std::string mate1, std::string mate2, int nMatesIn) {
    nMates = nMatesIn; fromMemory = true; useMap = 0;      // (the mappings of the files stay: batches of the stage before may still be in flight)
    noQualities = noQualities || fasta; fasta = false; samMates_ = 0;     // held reads are kept as four-line records whatever the input format was
    mem[0].swap(mate1); mem[1].swap(mate2);
    for (int i = 0; i < 2; i++) { memPos[i] = 0; carry[i].clear(); eof[i] = false; }
    readsSoFar = 0;
}

std::string FastqReader::reopen() {
    if (fromMemory) { for (int i = 0; i < 2; i++) { memPos[i] = 0; carry[i].clear(); eof[i] = false; } readsSoFar = 0; return ""; }
    lastExtra[0].clear(); lastExtra[1].clear();    // the next pass runs on fresh ReadAlign objects
    return open(paths_, command_, samMates_);      // first file again (a pipe cannot be rewound: the command is run again)
}

static bool isRegularFile(FILE *f) { struct stat st; return f && fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode); }

// offsets of the '\n' bytes in [p+from, p+to), appended to `out` until `out` holds `maxOut` entries; returns the offset where
// the scan stopped.  32 bytes per step when the CPU has AVX2 (lines are ~100 bytes: one memchr call per line is mostly call overhead).
__attribute__((target("avx2"))) static uint64_t scanNewlinesAvx2(const char *p, uint64_t from, uint64_t to, std::vector<uint64_t> &out, uint64_t maxOut) {
    const __m256i nl = _mm256_set1_epi8('\n');
    uint64_t i = from;
    for (; i + 32 <= to && out.size() < maxOut; i += 32) {
        uint32_t mask = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i)), nl));
        while (mask) {
            out.push_back(i + (uint64_t)__builtin_ctz(mask));
            mask &= mask - 1;
            if (out.size() >= maxOut) return out.back() + 1;
        }
    }
    for (; i < to && out.size() < maxOut; i++) if (p[i] == '\n') out.push_back(i);
    return out.size() >= maxOut ? out.back() + 1 : i;
}
static uint64_t scanNewlines(const char *p, uint64_t from, uint64_t to, std::vector<uint64_t> &out, uint64_t maxOut) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return scanNewlinesAvx2(p, from, to, out, maxOut);
    uint64_t i = from;
    while (i < to && out.size() < maxOut) {
        const char *q = (const char *)memchr(p + i, '\n', to - i);
        if (!q) return to;
        out.push_back((uint64_t)(q - p)); i = (uint64_t)(q - p) + 1;
    }
    return i;
}

// FASTA reads: every record is rewritten as four lines (ID, the sequence on one line, +, a quality line of 'A's as readLoad.cpp:84-88 assigns) so that the
// rest of the batcher is the same for both formats; ReadBatch::fasta tells the writers that there are no real qualities
uint64_t FastqReader::fillFasta(int m, uint64_t want, ReadBatch &b) {
    TextBuf &text = b.text[m];
    std::vector<uint64_t> &ls = b.lineStart[m], &le = b.lineEnd[m];
    ls.clear(); le.clear(); text.clear();
    std::vector<char> &raw = carry[m];              // unparsed input text
    size_t p = 0; uint64_t nRec = 0;
    auto more = [&]() {                              // append a block; false at the end of the input
        if (eof[m]) return false;
        const size_t block = 1u << 22, old = raw.size();
        raw.resize(old + block);
        size_t got;
        if (fromMemory) { got = std::min<size_t>(block, mem[m].size() - memPos[m]); memcpy(raw.data() + old, mem[m].data() + memPos[m], got); memPos[m] += got; }
        else got = fread(raw.data() + old, 1, block, f[m]);
        raw.resize(old + got);
        if (got < block) eof[m] = true;
        return got > 0;
    };
    auto line = [&](size_t from, size_t &end) {      // [from, end) up to the newline; false when the line is not complete yet
        for (;;) {
            const char *nl = from < raw.size() ? (const char *)memchr(raw.data() + from, '\n', raw.size() - from) : nullptr;
            if (nl) { end = (size_t)(nl - raw.data()); return true; }
            if (!more()) { end = raw.size(); return from < raw.size(); }
        }
    };
    auto emit = [&](const char *a, size_t n) { ls.push_back(text.size()); text.insert(text.end(), a, a + n); le.push_back(text.size()); text.push_back('\n'); };
    while (nRec < want) {
        size_t e;
        if (!line(p, e)) break;
        if (e == p) { ls.push_back(text.size()); le.push_back(text.size()); text.push_back('\n'); p = raw.size(); break; }   // an empty line ends the input
        size_t he = e; if (he > p && (unsigned char)raw[he - 1] < 33) he--;
        ls.push_back(text.size()); text.push_back('@'); text.insert(text.end(), raw.begin() + p + 1, raw.begin() + he); le.push_back(text.size()); text.push_back('\n');
        p = std::min(e + 1, raw.size());
        ls.push_back(text.size());
        for (;;) {                                   // sequence lines until the next '>' or the end
            if (p >= raw.size() && !more()) break;
            if (p < raw.size() && raw[p] == '>') break;
            size_t se;
            if (!line(p, se)) break;
            size_t s1 = se; if (s1 > p && (unsigned char)raw[s1 - 1] < 33) s1--;
            text.insert(text.end(), raw.begin() + p, raw.begin() + s1);
            p = std::min(se + 1, raw.size());
        }
        const size_t L = text.size() - ls.back();
        le.push_back(text.size()); text.push_back('\n');
        emit("+", 1);
        ls.push_back(text.size()); text.insert(text.end(), L, 'A'); le.push_back(text.size()); text.push_back('\n');
        nRec++;
    }
    raw.erase(raw.begin(), raw.begin() + std::min(p, raw.size()));
    return ls.size();
}

// SAM text input (--readFilesType SAM SE|PE, ReadAlignChunk_processChunks.cpp:28-107): header lines skipped; one record per mate, the two of a pair on
// consecutive lines; sequences of reverse-strand records are turned back; the attributes go onto the ID line (after a \x01 here) and come out again with
// every alignment of the read.  Rewritten as four-line records for both mates at once; fill(1) hands out mate 2.
uint64_t FastqReader::fillSam(uint64_t want, ReadBatch &b) {
    TextBuf &text = b.text[0];
    std::vector<uint64_t> *LS[2] = {&b.lineStart[0], &samLs2}, *LE[2] = {&b.lineEnd[0], &samLe2};
    TextBuf *TX[2] = {&text, &samText2};
    for (int m = 0; m < 2; m++) { LS[m]->clear(); LE[m]->clear(); TX[m]->clear(); }
    std::vector<char> &raw = carry[0];
    size_t p = 0; uint64_t nRec = 0;
    auto more = [&]() {
        if (eof[0]) return false;
        const size_t block = 1u << 22, old = raw.size();
        raw.resize(old + block);
        size_t got = fread(raw.data() + old, 1, block, f[0]);
        raw.resize(old + got);
        if (got < block) eof[0] = true;
        return got > 0;
    };
    auto line = [&](size_t from, size_t &end) {
        for (;;) {
            const char *nl = from < raw.size() ? (const char *)memchr(raw.data() + from, '\n', raw.size() - from) : nullptr;
            if (nl) { end = (size_t)(nl - raw.data()); return true; }
            if (!more()) { end = raw.size(); return from < raw.size(); }
        }
    };
    auto put = [&](int m, const char *a, size_t n) { LS[m]->push_back(TX[m]->size()); TX[m]->insert(TX[m]->end(), a, a + n); LE[m]->push_back(TX[m]->size()); TX[m]->push_back('\n'); };
    std::string name1, idLine, seq, qual; int imate1 = 0;
    while (nRec < want) {
        // the samMates_ records of one read
        bool got = true; size_t pRec = p;
        for (int imate = 0; imate < samMates_; imate++) {
            size_t e;
            for (;;) {                                   // skip header lines
                if (!line(p, e)) { got = false; break; }
                if (e > p && raw[p] == '@') { p = e + 1; if (imate == 0) pRec = p; continue; }
                break;
            }
            if (!got || e == p) { got = false; break; }
            size_t le = e; if (le > p && (unsigned char)raw[le - 1] < 33) le--;
            // fields: QNAME FLAG (7 skipped) SEQ QUAL attributes
            size_t fs[12], fe[12]; int nf = 0; size_t q = p;
            while (nf < 11 && q <= le) { fs[nf] = q; while (q < le && raw[q] != '\t' && raw[q] != ' ') q++; fe[nf] = q; nf++; if (q >= le) break; q++; }
            if (nf < 11) { p = e + 1; got = false; samError = "EXITING because of FATAL ERROR in input SAM file: a record has fewer than 11 fields"; break; }
            std::string nm(raw.data() + fs[0], fe[0] - fs[0]);
            const uint64_t flag = strtoull(std::string(raw.data() + fs[1], fe[1] - fs[1]).c_str(), nullptr, 10);
            if (imate == 0) { name1 = nm; imate1 = (samMates_ == 2 && (flag & 0x80)) ? 1 : 0; firstFlag = flag; }
            else {
                if (nm != name1) { samError = "EXITING because of FATAL ERROR in input BAM file: the consecutive lines in paired-end BAM have different read IDs:\n" + name1 + "   vs   " + nm + "\n\n SOLUTION: fix BAM file formatting. Paired-end reads should be always consecutive lines, with exactly 2 lines per paired-end read"; got = false; break; }
                if (!(((firstFlag & 0x40) && (flag & 0x80)) || ((flag & 0x40) && (firstFlag & 0x80)))) { samError = "EXITING because of FATAL ERROR in input BAM file: the consecutive lines in paired-end BAM have wrong mate FLAG bits"; got = false; break; }
                imate1 = 1 - imate1;
            }
            seq.assign(raw.data() + fs[9], fe[9] - fs[9]); qual.assign(raw.data() + fs[10], fe[10] - fs[10]);
            if (flag & 0x10) {
                std::reverse(seq.begin(), seq.end());
                for (char &c : seq) c = rcNt(c);
                std::reverse(qual.begin(), qual.end());
            }
            idLine = "@" + nm + " 0:" + ((flag & 0x800) ? "Y" : "N") + ":0";
            size_t as = fe[10]; while (as < le && (raw[as] == '\t' || raw[as] == ' ')) as++;
            if (as < le) { idLine.push_back('\x01'); idLine.append(raw.data() + as, le - as); }
            put(imate1, idLine.data(), idLine.size()); put(imate1, seq.data(), seq.size()); put(imate1, "+", 1); put(imate1, qual.data(), qual.size());
            p = std::min(e + 1, raw.size());
        }
        if (!got) {
            if (samError.empty() && LS[0]->size() != LS[1]->size() && samMates_ == 2) samError = "EXITING because of FATAL ERROR in input SAM file: the last paired-end read has one record only";
            (void)pRec; p = raw.size() > p && samError.empty() && !eof[0] ? p : p;
            break;
        }
        nRec++;
    }
    raw.erase(raw.begin(), raw.begin() + std::min(p, raw.size()));
    return LS[0]->size();
}

// The batch as a range of the file's mapping: line ends are looked for in place, in slices on threads, from where the batch before ended; nothing is copied.
uint64_t FastqReader::fillMapped(int m, uint64_t want, ReadBatch &b) {
    std::vector<uint64_t> &ls = b.lineStart[m], &le = b.lineEnd[m], &nlp = lineRaw[m];
    ls.clear(); le.clear(); nlp.clear(); b.text[m].clear();
    // a file that shrank under its mapping (truncated or rewritten while the run reads it) would end the process with SIGBUS inside the scan: looked at before every
    // batch and reported like any other read error (it cannot close the window between this look and the scan; it does turn the common accident into a message)
    if (curMap[m].p && f[m]) { struct stat st; if (fstat(fileno(f[m]), &st) != 0 || (uint64_t)st.st_size < (uint64_t)curMap[m].n) { ioError = EIO; eof[m] = true; return 0; } }
    const char *tx = curMap[m].p ? curMap[m].p + mapPos[m] : nullptr;
    const uint64_t avail = curMap[m].p ? curMap[m].n - mapPos[m] : 0;
    b.mapped[m] = tx;
    const uint64_t wantLines = want * 4;
    static const uint64_t sliceMin = getenv("STARAMD_READ_SLICE_MIN") ? strtoull(getenv("STARAMD_READ_SLICE_MIN"), nullptr, 10) : (8u << 20);
    auto onThreads = [&](unsigned K, const std::function<void(unsigned)> &fn) {
        std::vector<std::thread> th;
        for (unsigned k = 1; k < K; k++) th.emplace_back([&fn, k] { CpuScope cs(CPU_FILL); fn(k); });
        fn(0);
        for (auto &x : th) x.join();
    };
    uint64_t scanned = 0;
    while (nlp.size() < wantLines && scanned < avail) {
        const uint64_t missing = (wantLines - nlp.size() + 3) / 4;
        const uint64_t block = std::min<uint64_t>(avail - scanned, std::max<uint64_t>(1u << 16, std::min<uint64_t>(256u << 20, (uint64_t)((double)missing * bytesPerRecord[m] * 1.01) + 4096)));
        if (block >= sliceMin) {
            const unsigned K = std::max(1u, readSlices);
            std::vector<std::vector<uint64_t>> nl(K);
            const uint64_t per = (block + K - 1) / K;
            onThreads(K, [&](unsigned k) {
                const uint64_t lo = std::min(block, k * per), hi = std::min(block, lo + per);
                nl[k].reserve((size_t)((double)(hi - lo) / bytesPerRecord[m] * 4.2) + 16);
                scanNewlines(tx, scanned + lo, scanned + hi, nl[k], UINT64_MAX);
            });
            std::vector<size_t> off(K + 1, nlp.size());
            for (unsigned k = 0; k < K; k++) off[k + 1] = off[k] + nl[k].size();
            const size_t total = std::min<size_t>(off[K], wantLines);
            nlp.resize(total);
            onThreads(K, [&](unsigned k) { if (off[k] < total) memcpy(nlp.data() + off[k], nl[k].data(), (std::min(off[k + 1], total) - off[k]) * sizeof(uint64_t)); });
            slicedBlocks++;
        } else scanNewlines(tx, scanned, scanned + block, nlp, wantLines);
        scanned += block;
    }
    if (nlp.size() >= 4) bytesPerRecord[m] = (double)(nlp.back() + 1) / (double)(nlp.size() / 4);
    const size_t nNl = nlp.size();
    const uint64_t lineBeg = nNl == 0 ? 0 : nlp.back() + 1;
    const bool tailLine = nNl < wantLines && lineBeg < avail;           // end of file without a final newline: the tail is a line
    ls.resize(nNl + (tailLine ? 1 : 0)); le.resize(ls.size());
    {
        const unsigned K = nNl >= (1u << 16) ? std::max(1u, readSlices) : 1u;
        const size_t per = (nNl + K - 1) / K;
        onThreads(K, [&](unsigned k) {
            const size_t lo = std::min(nNl, k * per), hi = std::min(nNl, lo + per);
            for (size_t i = lo; i < hi; i++) {
                const uint64_t s0 = i ? nlp[i - 1] + 1 : 0; uint64_t e0 = nlp[i];
                if (e0 > s0 && (unsigned char)tx[e0 - 1] < 33) e0--;
                ls[i] = s0; le[i] = e0;
            }
        });
    }
    if (tailLine) {
        uint64_t e0 = avail;
        if (e0 > lineBeg && (unsigned char)tx[e0 - 1] < 33) e0--;
        ls[nNl] = lineBeg; le[nNl] = e0;
    }
    mapPos[m] += nNl >= wantLines ? lineBeg : avail;                     // what follows belongs to the next batches
    eof[m] = mapPos[m] >= curMap[m].n;
    return ls.size();
}

uint64_t FastqReader::fill(int m, uint64_t want, ReadBatch &b) {
    TextBuf &text = b.text[m];
    b.mapped[m] = nullptr;          // (only fillMapped points the batch at a file mapping: a file of the list that cannot be mapped -- a FIFO -- after one that was must not inherit the pointer)
    if (samMates_ > 0) {
        if (m == 0) return fillSam(want, b);
        text.swap(samText2); b.lineStart[1].swap(samLs2); b.lineEnd[1].swap(samLe2);
        return b.lineStart[1].size();
    }
    if (fasta) return fillFasta(m, want, b);
    if (useMap == 1) return fillMapped(m, want, b);
    std::vector<uint64_t> &ls = b.lineStart[m], &le = b.lineEnd[m], &nlp = lineRaw[m];
    ls.clear(); le.clear(); nlp.clear();
    text.assign(carry[m].begin(), carry[m].end());      // (not swap: every buffer keeps its capacity, so no fresh pages per batch)
    carry[m].clear();
    uint64_t scanned = 0;
    const uint64_t wantLines = want * 4;
    static const bool timing = getenv("STARAMD_HOST_TIMING") != nullptr;
    auto Tf = std::chrono::steady_clock::now(); double msRead = 0, msMerge = 0;
    auto lapf = [&](double &acc) { if (timing) { auto t = std::chrono::steady_clock::now(); acc += std::chrono::duration<double, std::milli>(t - Tf).count(); Tf = t; } };
    static const uint64_t sliceMin = getenv("STARAMD_READ_SLICE_MIN") ? strtoull(getenv("STARAMD_READ_SLICE_MIN"), nullptr, 10) : (8u << 20);   // (tests lower it)
    auto onThreads = [&](unsigned K, const std::function<void(unsigned)> &fn) {
        std::vector<std::thread> th;
        for (unsigned k = 1; k < K; k++) th.emplace_back([&fn, k] { CpuScope cs(CPU_FILL); fn(k); });
        fn(0);
        for (auto &x : th) x.join();
    };
    for (;;) {
        scanned = scanNewlines(text.data(), scanned, text.size(), nlp, wantLines);
        if (nlp.size() >= wantLines || eof[m]) break;
        // read about as much as the missing records need (little is left over to carry into the next batch)
        uint64_t missing = (wantLines - nlp.size() + 3) / 4;
        uint64_t block = std::max<uint64_t>(1u << 16, std::min<uint64_t>(64u << 20, (uint64_t)((double)missing * bytesPerRecord[m] * 1.01) + 4096));
        size_t old = text.size();
        text.resize(old + block);
        size_t got;
        if (fromMemory) { got = std::min<size_t>(block, mem[m].size() - memPos[m]); memcpy(text.data() + old, mem[m].data() + memPos[m], got); memPos[m] += got; }
        else if (block >= sliceMin && isRegularFile(f[m])) {
            // a regular file (not a pipe of --readFilesCommand): the block is cut into slices that threads read at their own positions
            // and scan for line ends; one thread copies ~100 MB per batch and mate at memcpy speed, which is most of the parse time
            const unsigned K = readSlices;
            const off_t pos0 = ftello(f[m]);
            const int fd = fileno(f[m]);
            std::vector<std::vector<uint64_t>> nl(K);
            std::vector<size_t> gotK(K, 0);
            const size_t per = (block + K - 1) / K;
            lapf(msMerge);
            onThreads(K, [&](unsigned k) {
                const size_t lo = std::min<size_t>(block, k * per), hi = std::min<size_t>(block, lo + per);
                size_t done = 0;
                while (lo + done < hi) {
                    ssize_t r = pread(fd, text.data() + old + lo + done, hi - lo - done, pos0 + (off_t)(lo + done));
                    if (r < 0 && errno == EINTR) continue;
                    if (r < 0) { ioError = errno; break; }                 // a read error is not the end of the input
                    if (r == 0) break;
                    done += (size_t)r;
                }
                gotK[k] = done;
                nl[k].reserve((size_t)((double)done / bytesPerRecord[m] * 4.2) + 16);
                scanNewlines(text.data(), old + lo, old + lo + done, nl[k], UINT64_MAX);
            });
            lapf(msRead);
            got = 0; unsigned Kgood = 0;
            for (unsigned k = 0; k < K; k++) { got += gotK[k]; Kgood = k + 1; if (gotK[k] < std::min<size_t>(block, (k + 1) * per) - std::min<size_t>(block, k * per)) break; }   // a short slice ends the input
            fseeko(f[m], pos0 + (off_t)got, SEEK_SET);
            // everything before `old` has been scanned; the line ends of the slices follow in order: each slice is copied to its place on its thread
            std::vector<size_t> off(Kgood + 1, nlp.size());
            for (unsigned k = 0; k < Kgood; k++) off[k + 1] = off[k] + nl[k].size();
            const size_t total = std::min<size_t>(off[Kgood], wantLines);
            nlp.resize(total);
            onThreads(Kgood, [&](unsigned k) { if (off[k] < total) memcpy(nlp.data() + off[k], nl[k].data(), (std::min(off[k + 1], total) - off[k]) * sizeof(uint64_t)); });
            scanned = nlp.size() >= wantLines ? nlp.back() + 1 : old + got;
            slicedBlocks++;
        }
        else got = fread(text.data() + old, 1, block, f[m]);      // (fread itself loops over short pipe reads until EOF)
        text.resize(old + got);
        if (got < block) eof[m] = true;
    }
    lapf(msMerge);
    if (nlp.size() >= 4) bytesPerRecord[m] = (double)(nlp.back() + 1) / (double)(nlp.size() / 4);
    // the line table: starts (behind the previous newline) and ends; fastqReadOneLine strips one trailing control character (\r) from every line
    const size_t nNl = nlp.size();
    uint64_t lineBeg = nNl == 0 ? 0 : nlp.back() + 1;
    const bool tailLine = nNl < wantLines && lineBeg < text.size();      // end of file without a final newline: the tail is a line
    ls.resize(nNl + (tailLine ? 1 : 0)); le.resize(ls.size());
    {
        const char *tx = text.data();
        const unsigned K = nNl >= (1u << 16) ? std::max(1u, readSlices) : 1u;
        const size_t per = (nNl + K - 1) / K;
        onThreads(K, [&](unsigned k) {
            const size_t lo = std::min(nNl, k * per), hi = std::min(nNl, lo + per);
            for (size_t i = lo; i < hi; i++) {
                const uint64_t s0 = i ? nlp[i - 1] + 1 : 0; uint64_t e0 = nlp[i];
                if (e0 > s0 && (unsigned char)tx[e0 - 1] < 33) e0--;
                ls[i] = s0; le[i] = e0;
            }
        });
    }
    if (nNl >= wantLines) {                             // the rest belongs to the next batches
        carry[m].assign(text.begin() + lineBeg, text.end());
        text.resize(lineBeg);
    } else if (tailLine) {
        uint64_t e0 = text.size();
        if (e0 > lineBeg && (unsigned char)text[e0 - 1] < 33) e0--;
        ls[nNl] = lineBeg; le[nNl] = e0;
    }
    lapf(msMerge);
    if (timing) fprintf(stderr, "  fill mate %d: read + scan on %u threads %.2f ms, serial (line table, carry) %.2f ms\n", m, readSlices, msRead, msMerge);
    return ls.size();
}

struct NtTable {               // convertNucleotidesToNumbers (SequenceFuns.cpp:131-146) as a table; [1] = code of the complement
    uint8_t fwd[256], rc[256];
    NtTable() {
        memset(fwd, 4, 256); memset(rc, 4, 256);
        const char *nt = "ACGTacgt";
        for (int k = 0; k < 8; k++) { fwd[(uint8_t)nt[k]] = (uint8_t)(k & 3); rc[(uint8_t)nt[k]] = (uint8_t)(3 - (k & 3)); }
    }
};
static const NtTable NT;

// the same two conversions 32 bases per step: a base is A/C/G/T in either case or it is code 4; (c | 0x20) folds the cases, four compares pick the code
// (exactly one can match): 4 ^ (eqA & 4 | eqC & 5 | eqG & 6 | eqT & 7) = 0 / 1 / 2 / 3, else 4; the complement has the constants 7 / 6 / 5 / 4
__attribute__((target("avx2"))) static inline __m256i ntCodes32(__m256i c, bool comp) {
    const __m256i lower = _mm256_or_si256(c, _mm256_set1_epi8(0x20));
    const __m256i eA = _mm256_cmpeq_epi8(lower, _mm256_set1_epi8('a')), eC = _mm256_cmpeq_epi8(lower, _mm256_set1_epi8('c'));
    const __m256i eG = _mm256_cmpeq_epi8(lower, _mm256_set1_epi8('g')), eT = _mm256_cmpeq_epi8(lower, _mm256_set1_epi8('t'));
    const __m256i m = _mm256_or_si256(_mm256_or_si256(_mm256_and_si256(eA, _mm256_set1_epi8(comp ? 7 : 4)), _mm256_and_si256(eC, _mm256_set1_epi8(comp ? 6 : 5))),
                                      _mm256_or_si256(_mm256_and_si256(eG, _mm256_set1_epi8(comp ? 5 : 6)), _mm256_and_si256(eT, _mm256_set1_epi8(comp ? 4 : 7))));
    return _mm256_xor_si256(m, _mm256_set1_epi8(4));
}
__attribute__((target("avx2"))) static void ntForwardAvx2(const char *s, uint8_t *r, uint64_t n) {
    uint64_t k = 0;
    for (; k + 32 <= n; k += 32) _mm256_storeu_si256((__m256i *)(r + k), ntCodes32(_mm256_loadu_si256((const __m256i *)(s + k)), false));
    for (; k < n; k++) r[k] = NT.fwd[(uint8_t)s[k]];
}
// r[k] = complement code of s[n - 1 - k]
__attribute__((target("avx2"))) static void ntRevCompAvx2(const char *s, uint8_t *r, uint64_t n) {
    const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    uint64_t k = 0;
    for (; k + 32 <= n; k += 32) {
        __m256i c = _mm256_loadu_si256((const __m256i *)(s + n - 32 - k));
        c = _mm256_permute2x128_si256(_mm256_shuffle_epi8(c, rev), _mm256_shuffle_epi8(c, rev), 1);      // bytes reversed within the halves, halves swapped
        _mm256_storeu_si256((__m256i *)(r + k), ntCodes32(c, true));
    }
    for (; k < n; k++) r[k] = NT.rc[(uint8_t)s[n - 1 - k]];
}

bool FastqReader::fillBatch(ReadBatch &b, const RunParams &P, uint64_t maxReads, std::string &err) {
    b.clear();
    b.firstReadIndex = readsSoFar;
    uint64_t want = maxReads;
    if (P.readMapNumber >= 0) {
        if ((int64_t)readsSoFar >= P.readMapNumber) return false;
        want = std::min<uint64_t>(want, (uint64_t)P.readMapNumber - readsSoFar);
    }
    if (want == 0) return false;
    static const bool timing = getenv("STARAMD_HOST_TIMING") != nullptr;
    {   // a block of ~100 MB per mate and batch is copied out of the page cache and scanned for line ends by this many threads per mate
        static const unsigned fixed = getenv("STARAMD_READ_SLICES") ? (unsigned)atoi(getenv("STARAMD_READ_SLICES")) : 0u;
        // (measured on a GPU box with 16 CPUs, profiles/r04_host_stages_on_the_gpu_box.txt: read + scan of one mate's block 20.6 ms with 4 slices, 7.3 ms with 8 -- the
        // fill stage is the longest host stage of a batch and runs once per batch on its own thread, so its slices are what the other stages leave idle)
        // (all helper thread counts follow --runThreadN: a rank of an 8-GPU node that is given 2 host threads must not start 8 slice readers beside them)
        readSlices = fixed ? std::max(1u, std::min(fixed, 64u)) : (unsigned)std::max(1, std::min(16, P.runThreadN / 2));
    }
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (timing) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  parse %-8s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count()); T0 = t; } };
    uint64_t nLines[2] = {0, 0};
    for (;;) {
        if (useMap < 0) useMap = (!fromMemory && samMates_ == 0 && !fasta && command_.empty() && P.outQSconversionAdd == 0 && !P.outSAMreadIDnumber && curMap[0].p && (nMates < 2 || curMap[1].p)) ? 1 : 0;
        if (nMates == 2 && samMates_ == 0 && !fasta) {       // the two mate files are read and scanned for line ends side by side
            std::thread second([&] { CpuScope cs(CPU_FILL); nLines[1] = fill(1, want, b); });
            nLines[0] = fill(0, want, b);
            second.join();
        } else
            for (int m = 0; m < nMates; m++) nLines[m] = fill(m, want, b);
        b.fileIndex = (uint32_t)curFile; b.fasta = fasta || noQualities;
        // a batch never spans two input files: when this one is exhausted the next batch starts with the next file
        if (!fromMemory && nLines[0] == 0 && curFile + 1 < files_[0].size()) { closeFiles(); curFile++; std::string e = openCurrent(); if (!e.empty()) { err = e; return false; } continue; }
        break;
    }
    if (ioError.load()) { err = std::string("EXITING because of INPUT ERROR: read error in --readFilesIn: ") + strerror(ioError.load()) + "\n"; return false; }
    lap("fill");
    if (timing) fprintf(stderr, "  parse blocks read in slices so far: %llu\n", (unsigned long long)slicedBlocks.load());
    // records of mate 1 decide the batch; an empty ID line ends the input
    uint64_t n = nLines[0] / 4;
    bool partial = nLines[0] % 4 != 0;
    for (uint64_t i = 0; i < n; i++) if (b.lineEnd[0][4 * i] == b.lineStart[0][4 * i]) { n = i; partial = false; eof[0] = true; carry[0].clear(); mapPos[0] = curMap[0].n; break; }     // (mapped input: nothing behind the empty line is read either)
    if (partial && b.lineEnd[0][4 * n] == b.lineStart[0][4 * n]) partial = false;
    if (partial) { err = "EXITING because of FATAL ERROR in reads input: truncated FASTQ record"; return false; }
    if (n == 0) return false;
    if (nMates == 2 && nLines[1] < 4 * n) {
        err = nLines[1] / 4 < n && nLines[1] % 4 == 0 ? "EXITING because of FATAL ERROR: read files are not consistent, reached the end of the one before the other one"
                                                       : "EXITING because of FATAL ERROR in reads input: truncated FASTQ record";
        return false;
    }
    if (!samError.empty()) { err = samError; return false; }
    b.n = (uint32_t)n;
    readsSoFar += n;
    if (b.mapped[0]) mappedBatches++;
    return true;
}

bool FastqReader::convertBatch(ReadBatch &b, const RunParams &P, std::string &err) {
    static const bool timing = getenv("STARAMD_HOST_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (timing) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  parse %-8s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count()); T0 = t; } };
    const uint64_t n = b.n;
    const std::vector<uint64_t> *lineStart = b.lineStart, *lineEnd = b.lineEnd;
    b.readOffset.assign(n + 1, 0); b.mate1Length.assign(n, 0); b.mmMaxTotal.assign(n, 0);
    b.nameSpan.assign(n, TextSpan{0, 0}); b.filter.assign(n, 'N');
    if (fromMemory) { b.origIndex.assign(n, 0); b.heldFile.assign(n, 0); }
    if (P.clipYes) for (int m = 0; m < nMates; m++) for (int q = 0; q < 2; q++) b.clipN[m][q].assign(n, 0);
    for (int m = 0; m < nMates; m++) { b.seqSpan[m].assign(n, TextSpan{0, 0}); b.qualSpan[m].assign(n, TextSpan{0, 0}); if (extras) b.extraSpan[m].assign(n, TextSpan{0, 0}); }
    lap("alloc");
    const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::min(std::max(P.runThreadN, 1), 32), n / 2048));
    std::vector<uint32_t> Lread(n);
    std::atomic<uint64_t> firstBad(UINT64_MAX);
    std::vector<std::string> errs(T);
    auto inRanges = [&](const std::function<void(uint64_t, uint64_t, int)> &fn) {
        if (T == 1) { fn(0, n, 0); return; }
        std::vector<std::thread> th;
        uint64_t per = (n + T - 1) / T;
        for (int t = 0; t < T; t++) th.emplace_back([&, t] { CpuScope cs(CPU_CONVERT); uint64_t lo = std::min<uint64_t>(n, t * per), hi = std::min<uint64_t>(n, lo + per); fn(lo, hi, t); });
        for (auto &x : th) x.join();
    };
    // pass 1: spans, checks, lengths
    inRanges([&](uint64_t lo, uint64_t hi, int t) {
        auto bad = [&](uint64_t i, const char *msg) { uint64_t cur = firstBad.load(); while (i < cur && !firstBad.compare_exchange_weak(cur, i)) {} if (errs[t].empty()) errs[t] = msg; };
        for (uint64_t i = lo; i < hi; i++) {
            const char *t0 = b.txt(0);
            uint64_t s0 = lineStart[0][4 * i], e0 = lineEnd[0][4 * i];
            if (t0[s0] != '@') { bad(i, "EXITING because of FATAL ERROR in input reads: wrong read ID line format: the read ID lines should start with @ (FASTA/SAM input: out of scope)"); return; }
            uint64_t len[2] = {0, 0};
            for (int m = 0; m < nMates; m++) {
                uint64_t ss = lineStart[m][4 * i + 1], se = lineEnd[m][4 * i + 1], qs = lineStart[m][4 * i + 3], qe = lineEnd[m][4 * i + 3];
                len[m] = se - ss;
                if (len[m] < 1) { bad(i, "EXITING because of FATAL ERROR in reads input: short read sequence line: 0"); return; }
                if (len[m] > STARAMD_READ_LEN_MAX) { bad(i, "EXITING because of FATAL ERROR in reads input: Lread>DEF_readSeqLengthMax"); return; }
                if (qe - qs != len[m]) { bad(i, "EXITING because of FATAL ERROR in reads input: quality string length is not equal to sequence length"); return; }
                b.seqSpan[m][i] = TextSpan{ss, (uint32_t)len[m]}; b.qualSpan[m][i] = TextSpan{qs, (uint32_t)len[m]};
                if (extras) {                           // attributes of the input SAM record, kept on the ID line after a \x01
                    const char *tm = b.txt(m); const uint64_t is = lineStart[m][4 * i], ie = lineEnd[m][4 * i];
                    const char *x = (const char *)memchr(tm + is, '\x01', ie - is);
                    b.extraSpan[m][i] = x ? TextSpan{(uint64_t)(x + 1 - tm), (uint32_t)(tm + ie - x - 1)} : TextSpan{0, 0};
                }
                if (P.outQSconversionAdd != 0) {        // readLoad.cpp:71-82, in place: every output of the qualities sees the converted ones
                    char *q = &b.text[m][qs];
                    for (uint64_t k = 0; k < len[m]; k++) { int v = int(q[k]) + P.outQSconversionAdd; q[k] = (char)(v < 33 ? 33 : v > 126 ? 126 : v); }
                }
                if (P.clipYes) {                        // ClipMate::clip (ClipMate_clip.cpp:5-78), 5' then 3' (readLoad.cpp:57-58); len[] becomes the clipped length
                    const char *sq = b.txt(m) + ss;
                    uint64_t L = len[m], cN[2] = {0, 0};
                    for (int ip = 0; ip < 2; ip++) {
                        const RunParams::ClipEnd &c = P.clip[m][ip];
                        if (!c.active) continue;
                        const uint64_t Lold = L;
                        if (c.N > 0) { if (L > c.N) { L -= c.N; cN[ip] += c.N; } else { L = 0; cN[ip] = Lold; } }
                        if (!c.adSeq.empty()) {          // 3' only. localSearch (SequenceFuns.cpp:293-315): best ungapped placement of the adapter, Ns of the read skipped
                            uint64_t nMatchBest = 0, nMMbest = 0, ixBest = L;
                            for (uint64_t ix = 0; ix < L; ix++) {
                                uint64_t nMatch = 0, nMM = 0;
                                for (uint64_t iy = 0; iy < std::min<uint64_t>(c.adSeq.size(), L - ix); iy++) {
                                    uint8_t x = NT.fwd[(uint8_t)sq[cN[0] + ix + iy]];
                                    if (x > 3) continue;
                                    if (x == NT.fwd[(uint8_t)c.adSeq[iy]]) nMatch++; else nMM++;
                                }
                                if ((nMatch > nMatchBest || (nMatch == nMatchBest && nMM < nMMbest)) && double(nMM) / double(nMatch) <= c.adMMp) { ixBest = ix; nMatchBest = nMatch; nMMbest = nMM; }
                            }
                            cN[ip] += L - ixBest; L = ixBest;
                        }
                        if (c.NafterAd > 0) { if (L > c.NafterAd) { L -= c.NafterAd; cN[ip] += c.NafterAd; } else { L = 0; cN[ip] = Lold; } }
                    }
                    const uint64_t c5 = cN[0], c3 = cN[1];
                    b.clipN[m][0][i] = (uint16_t)c5; b.clipN[m][1][i] = (uint16_t)c3;
                    len[m] = L;
                }
            }
            // read ID: first white-space token of mate 1's line, then trimmed at readNameSeparator chars
            uint64_t p = s0 + 1, e = p;
            while (e < e0 && t0[e] != ' ' && t0[e] != '\t' && t0[e] != '\x01') e++;
            if (e < e0 && t0[e] != '\x01') {
                uint64_t f2 = e;
                while (f2 < e0 && (t0[f2] == ' ' || t0[f2] == '\t')) f2++;
                uint64_t f3 = f2;
                while (f3 < e0 && t0[f3] != ' ' && t0[f3] != '\t') f3++;
                if (f3 - f2 >= 3 && t0[f2 + 1] == ':' && t0[f2 + 2] == 'Y' && (f2 + 3 < f3 ? t0[f2 + 3] == ':' : false)) b.filter[i] = 'Y';
                if (fromMemory) {                       // held reads carry their index in the original input as a third field
                    uint64_t f4 = f3; while (f4 < e0 && (t0[f4] == ' ' || t0[f4] == '\t')) f4++;
                    uint64_t v = 0; while (f4 < e0 && t0[f4] >= '0' && t0[f4] <= '9') { v = v * 10 + (uint64_t)(t0[f4] - '0'); f4++; }
                    b.origIndex[i] = v;
                    while (f4 < e0 && (t0[f4] == ' ' || t0[f4] == '\t')) f4++;       // 4th field: the input file of the read (ReadAlign_outputAlignments.cpp:113)
                    uint64_t fi = 0; while (f4 < e0 && t0[f4] >= '0' && t0[f4] <= '9') { fi = fi * 10 + (uint64_t)(t0[f4] - '0'); f4++; }
                    b.heldFile[i] = (uint32_t)fi;
                }
            }
            uint64_t ne = e;
            for (char c : P.readNameSeparator) { const void *q = memchr(t0 + p, c, ne - p); if (q) ne = (uint64_t)((const char *)q - t0); }
            b.nameSpan[i] = TextSpan{p, (uint32_t)(ne - p)};
            uint64_t L = nMates == 2 ? len[0] + len[1] + 1 : len[0];
            if (L > STARAMD_READ_LEN_MAX) { bad(i, "EXITING because of FATAL ERROR in reads input: Lread of the pair > DEF_readSeqLengthMax"); return; }
            Lread[i] = (uint32_t)L;
            b.mate1Length[i] = (uint16_t)len[0];
            // ReadAlign_oneRead.cpp:78
            b.mmMaxTotal[i] = (uint16_t)std::min<uint64_t>(P.outFilterMismatchNmax, (uint64_t)(P.outFilterMismatchNoverReadLmax * (double)(len[0] + len[1])));
        }
    });
    if (firstBad.load() != UINT64_MAX) {
        // report the error of the earliest offending read (each range stops at its first one)
        uint64_t per = (n + T - 1) / T;
        err = errs[T == 1 ? 0 : (int)(firstBad.load() / per)];
        return false;
    }
    lap("pass1");
    if (P.outSAMreadIDnumber && !fromMemory)           // --outSAMreadID Number: the read's 1-based index in the input (ReadAlignChunk_processChunks.cpp:117-119)
        for (uint64_t i = 0; i < n; i++) {
            std::string nm = std::to_string(b.firstReadIndex + i + 1);
            for (char c : P.readNameSeparator) { size_t q = nm.find(c); if (q != std::string::npos) nm.resize(q); }   // readLoad trims the number like any other name (readLoad.cpp:95-98)
            b.nameSpan[i] = TextSpan{(uint64_t)b.text[0].size(), (uint32_t)nm.size()};
            b.text[0].insert(b.text[0].end(), nm.begin(), nm.end());
        }
    for (uint64_t i = 0; i < n; i++) b.readOffset[i + 1] = b.readOffset[i] + Lread[i];
    b.bases.resize(b.readOffset[n]);
    lap("prefix");
    // pass 2: numeric combined reads
    static const bool avx2 = __builtin_cpu_supports("avx2") && !getenv("STARAMD_NO_AVX2");
    inRanges([&](uint64_t lo, uint64_t hi, int) {
        for (uint64_t i = lo; i < hi; i++) {
            uint8_t *r = b.bases.data() + b.readOffset[i];
            const char *s0 = b.txt(0) + b.seqSpan[0][i].off + b.clipped(0, 0, (uint32_t)i);
            uint64_t len0 = b.mate1Length[i];
            if (avx2) ntForwardAvx2(s0, r, len0); else for (uint64_t k = 0; k < len0; k++) r[k] = NT.fwd[(uint8_t)s0[k]];
            if (nMates == 2) {
                const char *s1 = b.txt(1) + b.seqSpan[1][i].off + b.clipped(1, 0, (uint32_t)i);
                uint64_t len1 = Lread[i] - len0 - 1;
                r[len0] = STARAMD_SPACER_BASE;
                if (avx2) ntRevCompAvx2(s1, r + len0 + 1, len1); else for (uint64_t k = 0; k < len1; k++) r[len0 + 1 + k] = NT.rc[(uint8_t)s1[len1 - 1 - k]];
            }
        }
    });
    lap("pass2");
    if (extras) {
        // a record without attributes keeps those of the read before it: readLoad's getline on the exhausted ID line leaves readNameExtra as it was
        // (readLoad.cpp:28-29); reproduced for the one-thread order of the reference
        for (uint64_t i = 0; i < n; i++) for (int m = 0; m < nMates; m++) {
            TextSpan &x = b.extraSpan[m][i];
            if (x.len > 0) lastExtra[m].assign(b.txt(m) + x.off, x.len);
            else if (!lastExtra[m].empty()) { x = TextSpan{(uint64_t)b.text[m].size(), (uint32_t)lastExtra[m].size()}; b.text[m].insert(b.text[m].end(), lastExtra[m].begin(), lastExtra[m].end()); }
        }
    }
    return true;
}

} // namespace staramd

namespace staramd {
namespace {
// localSearchNisMM (SequenceFuns.cpp:317-339): best ungapped placement of y on the tail of x; N anywhere counts as a mismatch
uint64_t localSearchNisMM(const uint8_t *x, uint64_t nx, const uint8_t *y, uint64_t ny, double pMM) {
    uint64_t nMatchBest = 0, nMMbest = 0, ixBest = nx;
    for (uint64_t ix = 0; ix < nx; ix++) {
        uint64_t nMatch = 0, nMM = 0;
        const uint64_t n = std::min(ny, nx - ix);
        for (uint64_t iy = 0; iy < n; iy++) { if (x[ix + iy] == y[iy] && y[iy] < 4) nMatch++; else nMM++; }
        if ((nMatch > nMatchBest || (nMatch == nMatchBest && nMM < nMMbest)) && double(nMM) / double(nMatch) <= pMM) { ixBest = ix; nMatchBest = nMatch; nMMbest = nMM; }
    }
    return ixBest;
}
}

void MergedBatch::build(const ReadBatch &b, const RunParams &P) {
    const uint32_t n = b.n;
    index.assign(n, -1); nOv.assign(n, 0); mateStart.assign(n, std::array<uint32_t, 2>{0, 0});
    reads.clear();
    std::vector<uint8_t> side(n, 0);                    // 1: mate 2 continues mate 1, 2: mate 1 continues mate 2
    const int T = (int)std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)std::min(std::max(P.runThreadN, 1), 64), n / 512));
    auto search = [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; i++) {
            const uint8_t *R = b.bases.data() + b.readOffset[i];
            const uint64_t Lread = b.readOffset[i + 1] - b.readOffset[i], len0 = b.mate1Length[i], len1 = Lread - len0 - 1;
            const uint64_t s1 = localSearchNisMM(R, len0, R + len0 + 1, len1, P.peOverlapMMp);
            const uint64_t s0 = localSearchNisMM(R + len0 + 1, len1, R, len0, P.peOverlapMMp);
            const uint64_t o1 = std::min(len1, len0 - s1), o0 = std::min(len0, len1 - s0);
            uint64_t ov = std::max(o0, o1);
            if (ov < P.peOverlapNbasesMin) continue;
            nOv[i] = (uint32_t)ov;
            if (o1 >= o0) { mateStart[i] = {0u, (uint32_t)s1}; side[i] = 1; }
            else { mateStart[i] = {(uint32_t)s0, 0u}; side[i] = 2; }
        }
    };
    if (T == 1) search(0, n);
    else {
        std::vector<std::thread> th; const uint32_t per = (n + T - 1) / T;
        for (int t = 0; t < T; t++) th.emplace_back(search, std::min(n, (uint32_t)t * per), std::min(n, (uint32_t)(t + 1) * per));
        for (auto &x : th) x.join();
    }
    for (uint32_t i = 0; i < n; i++) {
        if (!side[i]) continue;
        const uint8_t *R = b.bases.data() + b.readOffset[i];
        const uint64_t Lread = b.readOffset[i + 1] - b.readOffset[i], len0 = b.mate1Length[i], len1 = Lread - len0 - 1;
        index[i] = (int32_t)reads.n;
        if (side[i] == 1) {                              // mate 1, then what is left of mate 2 after the overlap
            const uint64_t o1 = std::min(len1, len0 - mateStart[i][1]);
            reads.bases.insert(reads.bases.end(), R, R + len0);
            if (o1 < len1) reads.bases.insert(reads.bases.end(), R + len0 + 1 + o1, R + len0 + 1 + len1);
        } else {                                         // mate 2, then what is left of mate 1
            const uint64_t o0 = std::min(len0, len1 - mateStart[i][0]);
            reads.bases.insert(reads.bases.end(), R + len0 + 1, R + len0 + 1 + len1);
            if (o0 < len0) reads.bases.insert(reads.bases.end(), R + o0, R + len0);
        }
        const uint64_t Lm = Lread - nOv[i] - 1;
        reads.readOffset.push_back(reads.readOffset.back() + Lm);
        reads.mate1Length.push_back((uint16_t)Lm);
        reads.mmMaxTotal.push_back(b.mmMaxTotal[i]);     // copyRead keeps the pair's mismatch budget (ReadAlign_waspMap.cpp:120)
        reads.n++;
    }
}
} // namespace staramd
