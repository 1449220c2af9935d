// main.cpp -- `star_amd`: command-line drop-in for `STAR --runMode alignReads`.
// Same flags (the subset that reaches the hot path or its outputs; anything else is rejected),
// same genomeDir, same Aligned.out.sam / SJ.out.tab / Log.final.out.  The per-read hot path runs on
// the MI355X through the C ABI of include/star_amd.h; there is no CPU path.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#include "../../../include/star_amd.h"

extern "C" {
void *sah_create(int argc, char **argv, char *errbuf, int errlen);
const staramd_genome *sah_genome(void *h);
const staramd_params *sah_params(void *h);
uint64_t sah_batch_reads(void *h);
int sah_device(void *h);
int sah_next_batch(void *h, uint64_t maxReads, staramd_batch *out);
int sah_emit(void *h, const staramd_results *res);
int sah_finish(void *h);
const char *sah_error(void *h);
void sah_destroy(void *h);
}

int main(int argc, char **argv) {
    char err[4096];
    void *h = sah_create(argc, argv, err, sizeof(err));
    if (!h) { fprintf(stderr, "\n%s\n", err); return 104; }
    uint64_t batchReads = sah_batch_reads(h);
    staramd_ctx *ctx = nullptr;
    int rc = staramd_create(&ctx, sah_device(h), sah_genome(h), sah_params(h), (uint32_t)batchReads, 0);
    if (rc) { fprintf(stderr, "\nEXITING because of FATAL ERROR: cannot initialise the MI355X engine: %s\n", staramd_last_error()); sah_destroy(h); return 105; }
    std::vector<staramd_read_result> reads(batchReads);
    std::vector<staramd_transcript> tr(batchReads * 64 + 4096);
    std::vector<staramd_exon> ex(tr.size() * 3);
    staramd_results res; memset(&res, 0, sizeof(res));
    res.reads = reads.data(); res.tr = tr.data(); res.trCapacity = tr.size(); res.ex = ex.data(); res.exCapacity = ex.size();
    uint64_t nReads = 0; double msDevice = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        staramd_batch b;
        int n = sah_next_batch(h, batchReads, &b);
        if (n < 0) { fprintf(stderr, "\n%s\n", sah_error(h)); return 104; }
        if (n == 0) break;
        rc = staramd_map_batch(ctx, &b, &res);
        if (rc) { fprintf(stderr, "\nEXITING because of FATAL ERROR in the MI355X engine (%d): %s\n", rc, staramd_last_error()); return 105; }
        if (sah_emit(h, &res)) { fprintf(stderr, "\n%s\n", sah_error(h)); return 104; }
        nReads += (uint64_t)n; msDevice += res.msTotalDevice;
    }
    if (sah_finish(h)) { fprintf(stderr, "\n%s\n", sah_error(h)); return 104; }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "star_amd: %llu reads, %.3f s wall in the mapping loop (%.3f s on the device)\n", (unsigned long long)nReads, sec, msDevice / 1e3);
    staramd_destroy(ctx);
    sah_destroy(h);
    return 0;
}
