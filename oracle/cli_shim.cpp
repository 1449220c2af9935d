// cli_shim.cpp -- TEST INFRASTRUCTURE, never shipped, never linked by anything under star_amd/.
// The engine's C ABI (include/star_amd.h) implemented on top of the CPU oracle, so that the command-line front end (star_amd/csrc/host/main.cpp: the
// three-stage batch pipeline, the phases of 2-pass / BySJout, the second batch of merged mates) can be exercised by tests on a box without a GPU.
// `make oracle` links it with main.cpp into oracle/_build/star_amd_oracle_cli; tests/test_cli_pipeline.py is the only user.
#include "../include/star_amd.h"
#include "../include/star_amd_index.h"
#include "../include/star_amd_async.h"
#include <string>
#include <cstdlib>
#include <mutex>

extern "C" {
void *oracle_create(const staramd_genome *g, const staramd_params *p);
void oracle_destroy(void *h);
int oracle_set_novel_junctions(void *h, const uint64_t *start, const uint64_t *end, uint64_t n, uint32_t stage);
int oracle_map_batch(void *h, const staramd_batch *b, staramd_results *r);
int sjdb_emul_insert(int, const staramd_sjdb_args *a, staramd_sjdb_result *res);
int index_emul_build(const uint8_t *G, uint64_t nGenome, uint32_t GstrandBit, uint32_t saIndexNbases, uint8_t *SA, uint64_t saCap, uint8_t *SAi, uint64_t saiCap, uint64_t *out);
}

struct staramd_ctx { void *o; staramd_ctx *owner; std::mutex m; };     // a "shared" context forwards to its owner's oracle (one batch at a time)
static std::string lastError;

extern "C" {
int staramd_create(staramd_ctx **out, int, const staramd_genome *g, const staramd_params *p, uint32_t, uint64_t) { *out = new staramd_ctx(); (*out)->o = oracle_create(g, p); (*out)->owner = nullptr; return STARAMD_OK; }
int staramd_create_shared(staramd_ctx **out, staramd_ctx *owner, uint32_t, uint64_t) { *out = new staramd_ctx(); (*out)->o = nullptr; (*out)->owner = owner->owner ? owner->owner : owner; return STARAMD_OK; }
int staramd_update_index(staramd_ctx *ctx, const staramd_genome *g, const staramd_params *p) { oracle_destroy(ctx->o); ctx->o = oracle_create(g, p); return STARAMD_OK; }
int staramd_set_novel_junctions(staramd_ctx *ctx, const uint64_t *start, const uint64_t *end, uint64_t n, uint32_t stage) { return oracle_set_novel_junctions(ctx->o, start, end, n, stage); }
int staramd_map_batch(staramd_ctx *ctx, const staramd_batch *b, staramd_results *r) {
    staramd_ctx *own = ctx->owner ? ctx->owner : ctx;
    std::lock_guard<std::mutex> lock(own->m);
    int rc = oracle_map_batch(own->o, b, r);
    r->msSeed = r->msWindows = r->msStitch = r->msTotalDevice = 0;
    if (rc) lastError = "result buffers too small";
    return rc;
}
void staramd_destroy(staramd_ctx *ctx) { if (ctx) { if (ctx->o) oracle_destroy(ctx->o); delete ctx; } }
void *staramd_pinned_alloc(uint64_t bytes) { return malloc(bytes ? bytes : 1); }
void staramd_pinned_free(void *p) { free(p); }
const char *staramd_last_error(void) { return lastError.c_str(); }
int staramd_update_tables(staramd_ctx *ctx, const staramd_genome *g, const staramd_params *p) { return staramd_update_index(ctx, g, p); }
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
// index build: the same algorithm code as the device build (star_amd/csrc/index/index_core.h) on the plain-loop backend of oracle/index_emul.cpp
int staramd_index_build(int, const uint8_t *G, const staramd_index_params *p, uint8_t *SA, uint64_t saCap, uint8_t *SAi, uint64_t saiCap, staramd_index_result *res) {
    uint64_t out[32];
    int rc = index_emul_build(G, p->nGenome, p->GstrandBit, p->gSAindexNbases, SA, saCap, SAi, saiCap, out);
    *res = staramd_index_result();
    res->nSA = out[0]; res->nSAbyte = out[1]; res->nSAi = out[2]; res->nSAibyte = out[3]; res->doublingRounds = (uint32_t)out[4];
    for (int i = 0; i < 17; i++) res->genomeSAindexStart[i] = out[5 + i];
    if (rc) lastError = "index_emul_build failed";
    return rc;
}
int staramd_sjdb_insert(int d, const staramd_sjdb_args *a, staramd_sjdb_result *res) { return sjdb_emul_insert(d, a, res); }
const char *staramd_index_last_error(void) { return lastError.c_str(); }
}
