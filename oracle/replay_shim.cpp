// replay_shim.cpp -- MEASUREMENT INFRASTRUCTURE for the HOST stages, never shipped, never linked by anything under star_amd/.
// The engine's C ABI (include/star_amd.h) as a recorder / player: the first time a batch is seen its results are computed by the CPU oracle
// (on STARAMD_REPLAY_THREADS threads, one oracle object each, pieces of the batch rebased to offset 0) and kept; every later batch with the
// same content gets a copy of the kept arrays, after the call has lasted STARAMD_REPLAY_DEVICE_MS milliseconds (a stand-in for the time
// staramd_map_batch blocks its mapper thread on the MI355X without using a host core).  With an input made of one block of reads repeated k
// times and --benchWarmupReads = that block, the timed region of staramd_cli_main runs the shipped reader / post-map / writer code at the rate
// the host allows, on a box without a GPU: tools/host_bench.py ("7 of them may replay recorded result buffers -- a
// bench-only tool, never a product path").  `make oracle` links it with cli_run.cpp into oracle/_build/libstaramd_cli_replay.so.
#include "../include/star_amd.h"
#include "../include/star_amd_index.h"
#include "../include/star_amd_async.h"
#include <string>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include <map>
#include <chrono>
#include <memory>

extern "C" {
void *oracle_create(const staramd_genome *g, const staramd_params *p);
void oracle_destroy(void *h);
int oracle_set_novel_junctions(void *h, const uint64_t *start, const uint64_t *end, uint64_t n, uint32_t stage);
int oracle_map_batch(void *h, const staramd_batch *b, staramd_results *r);
}

namespace {
struct Kept { std::vector<staramd_read_result> reads; std::vector<staramd_transcript> tr; std::vector<staramd_exon> ex; };
struct Shared {
    std::vector<void *> oracles;                     // one per recording thread
    std::mutex m;                                    // recording is one batch at a time
    std::map<uint64_t, std::shared_ptr<Kept>> kept;
    uint64_t nRecorded = 0, nPlayed = 0;
};
uint64_t batchKey(const staramd_batch *b) {          // content hash: every base of the first 64 KB, then every 61st, and the shape
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; };
    const uint64_t nb = b->readOffset[b->nReads];
    mix(b->nReads); mix(nb);
    for (uint64_t i = 0; i < nb && i < 65536; i++) mix(b->bases[i]);
    for (uint64_t i = 65536; i < nb; i += 61) mix(b->bases[i]);
    for (uint32_t i = 0; i < b->nReads; i += 97) { mix(b->mate1Length[i]); mix(b->mmMaxTotal[i]); }
    return h;
}
std::string lastError;
}

struct staramd_ctx { Shared *s; staramd_ctx *owner; };

extern "C" {
int staramd_create(staramd_ctx **out, int, const staramd_genome *g, const staramd_params *p, uint32_t, uint64_t) {
    staramd_ctx *c = new staramd_ctx(); c->owner = nullptr; c->s = new Shared();
    int T = getenv("STARAMD_REPLAY_THREADS") ? atoi(getenv("STARAMD_REPLAY_THREADS")) : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    for (int t = 0; t < T; t++) c->s->oracles.push_back(oracle_create(g, p));
    *out = c; return STARAMD_OK;
}
int staramd_create_shared(staramd_ctx **out, staramd_ctx *owner, uint32_t, uint64_t) { staramd_ctx *c = new staramd_ctx(); c->owner = owner->owner ? owner->owner : owner; c->s = c->owner->s; *out = c; return STARAMD_OK; }
int staramd_update_index(staramd_ctx *, const staramd_genome *, const staramd_params *) { lastError = "replay shim: one-phase runs only"; return STARAMD_ERR_DEVICE; }
int staramd_update_tables(staramd_ctx *, const staramd_genome *, const staramd_params *) { lastError = "replay shim: one-phase runs only"; return STARAMD_ERR_DEVICE; }
int staramd_set_novel_junctions(staramd_ctx *, const uint64_t *, const uint64_t *, uint64_t, uint32_t) { lastError = "replay shim: one-phase runs only"; return STARAMD_ERR_DEVICE; }

int staramd_map_batch(staramd_ctx *ctx, const staramd_batch *b, staramd_results *r) {
    const auto t0 = std::chrono::steady_clock::now();
    static const double deviceMs = getenv("STARAMD_REPLAY_DEVICE_MS") ? atof(getenv("STARAMD_REPLAY_DEVICE_MS")) : 0.0;
    Shared &S = *ctx->s;
    const uint64_t key = batchKey(b);
    std::shared_ptr<Kept> k;
    bool played = false;
    {
        std::lock_guard<std::mutex> lock(S.m);
        auto it = S.kept.find(key);
        if (it != S.kept.end()) { k = it->second; played = true; S.nPlayed++; }
        else {
            // record: pieces of the batch through the oracles, side by side
            const int T = (int)S.oracles.size();
            const uint32_t n = b->nReads, per = (n + T - 1) / T;
            std::vector<Kept> part(T); std::vector<int> rcs(T, 0);
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&, t] {
                const uint32_t lo = std::min(n, (uint32_t)t * per), hi = std::min(n, lo + per);
                if (hi <= lo) return;
                std::vector<uint64_t> off(hi - lo + 1);
                for (uint32_t i = lo; i <= hi; i++) off[i - lo] = b->readOffset[i] - b->readOffset[lo];
                staramd_batch pb; pb.nReads = hi - lo; pb.bases = b->bases + b->readOffset[lo]; pb.readOffset = off.data(); pb.mate1Length = b->mate1Length + lo; pb.mmMaxTotal = b->mmMaxTotal + lo;
                Kept &p = part[t];
                p.reads.resize(pb.nReads); p.tr.resize((size_t)pb.nReads * 8 + 1024); p.ex.resize(p.tr.size() * 4);
                for (;;) {
                    staramd_results pr; memset(&pr, 0, sizeof(pr));
                    pr.reads = p.reads.data(); pr.tr = p.tr.data(); pr.trCapacity = p.tr.size(); pr.ex = p.ex.data(); pr.exCapacity = p.ex.size();
                    rcs[t] = oracle_map_batch(S.oracles[t], &pb, &pr);
                    if (rcs[t] == STARAMD_ERR_RESULT_OVERFLOW) { p.tr.resize(pr.trCount + 1024); p.ex.resize(pr.exCount + 1024); continue; }
                    p.tr.resize(pr.trCount); p.ex.resize(pr.exCount);
                    break;
                }
            });
            for (auto &x : th) x.join();
            for (int t = 0; t < T; t++) if (rcs[t]) { lastError = "replay shim: the oracle failed on a piece"; return rcs[t]; }
            k = std::make_shared<Kept>();
            for (int t = 0; t < T; t++) {
                const uint64_t trN = k->tr.size(), exN = k->ex.size();
                for (auto x : part[t].reads) { x.trOffset += (uint32_t)trN; k->reads.push_back(x); }
                for (auto x : part[t].tr) { x.exonOffset += (uint32_t)exN; k->tr.push_back(x); }
                k->ex.insert(k->ex.end(), part[t].ex.begin(), part[t].ex.end());
            }
            S.kept[key] = k; S.nRecorded++;
        }
    }
    r->trCount = k->tr.size(); r->exCount = k->ex.size();
    r->msSeed = r->msWindows = r->msStitch = 0; r->msTotalDevice = played ? (float)deviceMs : 0;
    if (k->tr.size() > r->trCapacity || k->ex.size() > r->exCapacity) { lastError = "result buffers too small"; return STARAMD_ERR_RESULT_OVERFLOW; }
    memcpy(r->reads, k->reads.data(), k->reads.size() * sizeof(staramd_read_result));
    if (!k->tr.empty()) memcpy(r->tr, k->tr.data(), k->tr.size() * sizeof(staramd_transcript));
    if (!k->ex.empty()) memcpy(r->ex, k->ex.data(), k->ex.size() * sizeof(staramd_exon));
    if (played && deviceMs > 0) std::this_thread::sleep_until(t0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double, std::milli>(deviceMs)));
    return STARAMD_OK;
}
void staramd_destroy(staramd_ctx *ctx) {
    if (!ctx) return;
    if (!ctx->owner) {
        if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "replay shim: %llu batches recorded, %llu played\n", (unsigned long long)ctx->s->nRecorded, (unsigned long long)ctx->s->nPlayed);
        for (void *o : ctx->s->oracles) oracle_destroy(o);
        delete ctx->s;
    }
    delete ctx;
}
void *staramd_pinned_alloc(uint64_t bytes) { return malloc(bytes ? bytes : 1); }
void staramd_pinned_free(void *p) { free(p); }
const char *staramd_last_error(void) { return lastError.c_str(); }
int staramd_get_timings(staramd_ctx *, float *, int) { return 0; }
int staramd_insert_junctions_fits(staramd_ctx *, uint64_t, uint32_t) { return 0; }
int staramd_prefetch_batch(staramd_ctx *, const staramd_batch *) { return 0; }
int staramd_prefetch_cancel(staramd_ctx *) { return 0; }
// the two halves of staramd_map_batch: the stand-in maps when the batch is finished
static thread_local staramd_batch g_inFlight;          // (the descriptor by value: the caller's struct need not outlive the call, only the arrays it points at)
int staramd_map_begin(staramd_ctx *, const staramd_batch *b) { g_inFlight = *b; return 0; }
int staramd_map_wait(staramd_ctx *) { return 0; }
int staramd_map_end(staramd_ctx *ctx, staramd_results *r, const staramd_batch *next) { const int rc = staramd_map_batch(ctx, &g_inFlight, r); if (rc == STARAMD_ERR_RESULT_OVERFLOW) return rc; if (next) g_inFlight = *next; return rc; }
uint64_t staramd_overlapped_batches(staramd_ctx *) { return 0; }
uint64_t staramd_launch_count(staramd_ctx *) { return 0; }
uint32_t staramd_capabilities(void) { return 0; }      // (the restatement returns what staramd_params::resultSelect 0 / 1 ask for)
uint64_t staramd_prefetch_hits(staramd_ctx *) { return 0; }
int staramd_get_counters(staramd_ctx *, uint64_t *, int) { return 0; }
int staramd_index_build(int, const uint8_t *, const staramd_index_params *, uint8_t *, uint64_t, uint8_t *, uint64_t, staramd_index_result *) { lastError = "replay shim: no index build"; return STARAMD_ERR_DEVICE; }
int staramd_sjdb_insert(int, const staramd_sjdb_args *, staramd_sjdb_result *) { lastError = "replay shim: no junction insertion"; return STARAMD_ERR_DEVICE; }
const char *staramd_index_last_error(void) { return lastError.c_str(); }
}
