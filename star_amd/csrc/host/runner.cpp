// runner.cpp -- host run orchestration (the alignReads part of STAR.cpp main(), source/STAR.cpp:58-313)
// and its C interface `sah_*`, used by the CLI (main.cpp) and, through ctypes, by tests/ and bench.py.
// The engine (HIP library behind include/star_amd.h) is NOT linked here: callers pass result buffers in,
// so the same post-map code is exercised with results produced by the HIP engine (product) or,
// in tests only, by the CPU oracle.
#include "host.h"
#include <cstring>
#include <ctime>
#include <memory>

namespace staramd {

struct Runner {
    RunParams P;
    GenomeIndex gi;
    FastqReader reader;
    ReadBatch batch;
    staramd_batch batchView;
    std::unique_ptr<PostMap> post;
    OutSJ sj;
    Stats stats;
    FILE *samOut = nullptr;
    std::string samBuf;
    std::string error;

    bool init(int argc, char **argv) {
        time(&stats.timeStart);
        error = P.parse(argc, argv);
        if (!error.empty()) return false;
        error = gi.load(P.genomeDir);
        if (!error.empty()) return false;
        P.finalize(gi);
        error = reader.open(P.readFilesIn);
        if (!error.empty()) return false;
        post.reset(new PostMap(P, gi));
        std::string samPath = P.outFileNamePrefix + "Aligned.out.sam";
        samOut = fopen(samPath.c_str(), "wb");
        if (!samOut) { error = "EXITING because of fatal ERROR: could not create output file " + samPath; return false; }
        setvbuf(samOut, nullptr, _IOFBF, 1 << 22);
        std::string h = post->samHeader();
        fwrite(h.data(), 1, h.size(), samOut);
        time(&stats.timeStartMap);
        return true;
    }
    int nextBatch(uint64_t maxReads) {
        std::string err;
        bool ok = reader.nextBatch(batch, P, maxReads, err);
        if (!err.empty()) { error = err; return -1; }
        if (!ok) return 0;
        batchView = batch.view();
        return (int)batch.n;
    }
    bool emit(const staramd_results *r) {
        samBuf.clear();
        error = post->process(batch, *r, samBuf, sj, stats);
        if (!error.empty()) return false;
        fwrite(samBuf.data(), 1, samBuf.size(), samOut);
        if (sj.data.size() > 4000000) sj.collapse();     // ReadAlignChunk_mapChunk.cpp:66-86 (bounded memory)
        return true;
    }
    bool finish() {
        if (samOut) { fclose(samOut); samOut = nullptr; }
        error = sj.filterAndWrite(P, gi, P.outFileNamePrefix + "SJ.out.tab");
        if (!error.empty()) return false;
        stats.reportFinal(P.outFileNamePrefix + "Log.final.out");
        return true;
    }
    ~Runner() { if (samOut) fclose(samOut); }
};

} // namespace staramd

using staramd::Runner;

extern "C" {

void *sah_create(int argc, char **argv, char *errbuf, int errlen) {
    Runner *r = new Runner();
    if (!r->init(argc, argv)) {
        if (errbuf && errlen > 0) { strncpy(errbuf, r->error.c_str(), errlen - 1); errbuf[errlen - 1] = 0; }
        delete r;
        return nullptr;
    }
    return r;
}
const staramd_genome *sah_genome(void *h) { return &((Runner *)h)->gi.view; }
const staramd_params *sah_params(void *h) { return &((Runner *)h)->P.dev; }
uint64_t sah_batch_reads(void *h) { return ((Runner *)h)->P.gpuBatchReads; }
int sah_device(void *h) { return ((Runner *)h)->P.gpuDevice; }
double sah_genome_load_seconds(void *h) { return ((Runner *)h)->gi.loadSeconds; }
int sah_next_batch(void *h, uint64_t maxReads, staramd_batch *out) {
    Runner *r = (Runner *)h;
    int n = r->nextBatch(maxReads);
    if (n > 0 && out) *out = r->batchView;
    return n;
}
int sah_emit(void *h, const staramd_results *res) { return ((Runner *)h)->emit(res) ? 0 : -1; }
int sah_finish(void *h) { return ((Runner *)h)->finish() ? 0 : -1; }
const char *sah_error(void *h) { return ((Runner *)h)->error.c_str(); }
void sah_destroy(void *h) { delete (Runner *)h; }

}

// struct sizes of the C ABI, so that language bindings can verify their mirrors
extern "C" uint64_t sah_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(staramd_genome);
        case 1: return sizeof(staramd_params);
        case 2: return sizeof(staramd_batch);
        case 3: return sizeof(staramd_read_result);
        case 4: return sizeof(staramd_transcript);
        case 5: return sizeof(staramd_exon);
        case 6: return sizeof(staramd_results);
        default: return 0;
    }
}
