// main.cpp -- `star_amd`: command-line drop-in for `STAR --runMode alignReads`.
// Same flags (the subset that reaches the hot path or its outputs; anything else is rejected),
// same genomeDir, same Aligned.out.sam / SJ.out.tab / Log.final.out.  The per-read hot path runs on
// the MI355X through the C ABI of include/star_amd.h; there is no CPU path.
//
// Pipeline (the reference interleaves these per thread, ReadAlignChunk_processChunks.cpp / _mapChunk.cpp):
//   reader thread   FASTQ text -> numeric batch k+1            (sah_parse_slot)
//   main thread     batch k through the engine                 (staramd_map_batch)
//   writer thread   post-map + SAM text of batch k-1 on --runThreadN host threads, in input order (sah_emit_slot)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <string>
#include "../../../include/star_amd_host.h"
#include "../../../include/star_amd_index.h"

namespace {
struct Msg { int slot; int n; staramd_batch b; int resIdx; bool merged; };
struct Queue {                                   // bounded hand-off between two pipeline stages
    std::mutex m; std::condition_variable cv; std::deque<Msg> q; bool closed = false;
    void push(const Msg &x) { { std::lock_guard<std::mutex> l(m); q.push_back(x); } cv.notify_all(); }
    void close() { { std::lock_guard<std::mutex> l(m); closed = true; } cv.notify_all(); }
    bool pop(Msg &x) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; x = q.front(); q.pop_front(); return true; }
};
struct Tokens {                                  // counting semaphore over a small set of buffer indices
    std::mutex m; std::condition_variable cv; std::deque<int> free;
    int take() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !free.empty(); }); int v = free.front(); free.pop_front(); return v; }
    void give(int v) { { std::lock_guard<std::mutex> l(m); free.push_back(v); } cv.notify_all(); }
};
struct ResBuf { std::vector<staramd_read_result> reads; std::vector<staramd_transcript> tr; std::vector<staramd_exon> ex; staramd_results res; };
}

int main(int argc, char **argv) {
    for (int i = 1; i < argc; i++) if (std::string(argv[i]) == "--version") { printf("2.7.11b\n"); return 0; }      // the version whose behaviour is reproduced (Parameters.cpp:340-343)
    char err[4096];
    void *h = sah_create(argc, argv, err, sizeof(err));
    if (!h) { fprintf(stderr, "\n%s\n", err); return 104; }
    if (sah_tool_done(h)) { sah_destroy(h); return 0; }          // --runMode inputAlignmentsFromBAM: nothing to map
    if (sah_generate_mode(h)) {                                  // --runMode genomeGenerate: suffix array + SAindex on the device
        const uint8_t *G; uint64_t nGenome, saCap, saiCap; uint32_t gsb, nb; uint8_t *SA, *SAi;
        if (sah_generate_buffers(h, &G, &nGenome, &gsb, &nb, &SA, &saCap, &SAi, &saiCap)) { fprintf(stderr, "\n%s\n", sah_error(h)); return 104; }
        staramd_index_params ip; memset(&ip, 0, sizeof(ip));
        ip.nGenome = nGenome; ip.GstrandBit = gsb; ip.gSAindexNbases = nb; ip.gSAsparseD = 1;
        staramd_index_result ir;
        auto tg = std::chrono::steady_clock::now();
        int grc = staramd_index_build(sah_device(h), G, &ip, SA, saCap, SAi, saiCap, &ir);
        if (grc) { fprintf(stderr, "\nEXITING because of FATAL ERROR: index build on the MI355X failed: %s\n", staramd_index_last_error()); sah_destroy(h); return 105; }
        double sBuild = std::chrono::duration<double>(std::chrono::steady_clock::now() - tg).count();
        if (sah_generate_finish(h, ir.nSA, ir.nSAbyte, ir.nSAibyte)) { fprintf(stderr, "\n%s\n", sah_error(h)); sah_destroy(h); return 104; }
        double sAll = std::chrono::duration<double>(std::chrono::steady_clock::now() - tg).count();
        fprintf(stderr, "star_amd: genomeGenerate: %llu suffixes, %u doubling rounds, device build %.3f s (%.1f ms on the stream), junction insertion + files %.3f s\n",
                (unsigned long long)ir.nSA, ir.doublingRounds, sBuild, ir.msTotal, sAll - sBuild);
        sah_destroy(h);
        return 0;
    }
    const uint64_t batchReads = sah_batch_reads(h);
    staramd_ctx *ctx = nullptr;
    int rc = staramd_create(&ctx, sah_device(h), sah_genome(h), sah_params(h), (uint32_t)batchReads, 0);
    if (rc) { fprintf(stderr, "\nEXITING because of FATAL ERROR: cannot initialise the MI355X engine: %s\n", staramd_last_error()); sah_destroy(h); return 105; }
    ResBuf piecePart;                                // one piece of a WASP re-mapping batch
    ResBuf rb[6];                                    // [0..1] the batches in flight, [2..3] their merged mates (--peOverlapNbasesMin), [4..5] their WASP re-mapping
    for (ResBuf *rp : {&rb[0], &rb[1], &rb[2], &rb[3], &rb[4], &rb[5], &piecePart}) {
        ResBuf &r = *rp;
        r.reads.resize(batchReads); r.tr.resize(batchReads * 16 + 4096); r.ex.resize(r.tr.size() * 3);
        memset(&r.res, 0, sizeof(r.res));
        r.res.reads = r.reads.data(); r.res.tr = r.tr.data(); r.res.trCapacity = r.tr.size(); r.res.ex = r.ex.data(); r.res.exCapacity = r.ex.size();
    }
    std::string failure; std::mutex failM;
    auto fail = [&](const std::string &s) { std::lock_guard<std::mutex> l(failM); if (failure.empty()) failure = s; };
    uint64_t nReads = 0; double msDevice = 0;
    auto t0 = std::chrono::steady_clock::now();
    // one pass over the reads through the three-stage pipeline
    auto mapAllBatches = [&]() {
        Queue parsed, mapped; Tokens slots, results;
        for (int i = 0; i < 3; i++) slots.give(i);
        for (int i = 0; i < 2; i++) results.give(i);
        std::thread reader([&] {
            for (;;) {
                Msg m; m.slot = slots.take(); m.resIdx = -1;
                m.n = sah_parse_slot(h, m.slot, batchReads, &m.b);
                if (m.n < 0) { fail(sah_error(h)); break; }
                if (m.n == 0) break;
                parsed.push(m);
            }
            parsed.close();
        });
        std::thread writer([&] {
            Msg m;
            while (mapped.pop(m)) {
                if (failure.empty() && (m.merged ? sah_emit_slot_merged(h, m.slot, &rb[m.resIdx].res, &rb[2 + m.resIdx].res) : sah_emit_slot(h, m.slot, &rb[m.resIdx].res))) fail(sah_error(h));
                results.give(m.resIdx); slots.give(m.slot);
            }
        });
        Msg m;
        while (parsed.pop(m)) {
            if (!failure.empty()) { slots.give(m.slot); continue; }
            m.resIdx = results.take();
            auto mapInto = [&](const staramd_batch &bt, ResBuf &r) {
                staramd_results &res = r.res;
                int e = staramd_map_batch(ctx, &bt, &res);
                if (e == STARAMD_ERR_RESULT_OVERFLOW) {          // rare: more transcripts than the buffers hold -> grow and retry
                    r.tr.resize(res.trCount + res.trCount / 4 + 4096); r.ex.resize(res.exCount + res.exCount / 4 + 4096);
                    res.tr = r.tr.data(); res.trCapacity = r.tr.size(); res.ex = r.ex.data(); res.exCapacity = r.ex.size();
                    e = staramd_map_batch(ctx, &bt, &res);
                }
                if (!e) msDevice += res.msTotalDevice;
                return e;
            };
            rc = mapInto(m.b, rb[m.resIdx]);
            m.merged = false;
            if (!rc) {                                           // --peOverlapNbasesMin: the overlapping mates of the batch, merged into single reads, are a second batch
                staramd_batch mb;
                if (sah_merged_slot(h, m.slot, &mb) > 0) { m.merged = true; rc = mapInto(mb, rb[2 + m.resIdx]); }
            }
            if (!rc) {                                           // --waspOutputMode: allele-swapped copies of some reads, one more batch (can be larger than the batch itself)
                staramd_batch wb;
                int nw = sah_wasp_slot(h, m.slot, &rb[m.resIdx].res, &wb);
                if (nw > 0) {
                    // as many pieces as it takes (a read over a dense cluster of SNVs has up to 1023 copies); the results of the pieces are appended to one set
                    ResBuf &r = rb[4 + m.resIdx];
                    if (r.reads.size() < (size_t)nw) { r.reads.resize((size_t)nw); r.res.reads = r.reads.data(); }
                    uint64_t trN = 0, exN = 0;
                    for (uint32_t done = 0; done < (uint32_t)nw && !rc; ) {
                        staramd_batch piece = wb; piece.nReads = std::min<uint32_t>((uint32_t)batchReads, (uint32_t)nw - done);
                        piece.readOffset = wb.readOffset + done; piece.mate1Length = wb.mate1Length + done; piece.mmMaxTotal = wb.mmMaxTotal + done;
                        rc = mapInto(piece, piecePart);
                        if (rc) break;
                        const staramd_results &pr = piecePart.res;
                        if (r.tr.size() < trN + pr.trCount) r.tr.resize((trN + pr.trCount) * 3 / 2 + 1024);
                        if (r.ex.size() < exN + pr.exCount) r.ex.resize((exN + pr.exCount) * 3 / 2 + 1024);
                        for (uint32_t k = 0; k < piece.nReads; k++) { r.reads[done + k] = pr.reads[k]; r.reads[done + k].trOffset += (uint32_t)trN; }
                        for (uint64_t k = 0; k < pr.trCount; k++) { r.tr[trN + k] = pr.tr[k]; r.tr[trN + k].exonOffset += (uint32_t)exN; }
                        if (pr.exCount) memcpy(&r.ex[exN], pr.ex, pr.exCount * sizeof(staramd_exon));
                        trN += pr.trCount; exN += pr.exCount; done += piece.nReads;
                    }
                    r.res.reads = r.reads.data(); r.res.tr = r.tr.data(); r.res.trCapacity = r.tr.size(); r.res.trCount = trN; r.res.ex = r.ex.data(); r.res.exCapacity = r.ex.size(); r.res.exCount = exN;
                }
                if (!rc && sah_wasp_results_slot(h, m.slot, &rb[m.resIdx].res, nw > 0 ? &rb[4 + m.resIdx].res : nullptr)) { fail(sah_error(h)); results.give(m.resIdx); slots.give(m.slot); continue; }
            }
            if (rc) { fail(std::string("EXITING because of FATAL ERROR in the MI355X engine: ") + staramd_last_error()); results.give(m.resIdx); slots.give(m.slot); continue; }
            nReads += (uint64_t)m.n;
            mapped.push(m);
        }
        mapped.close();
        reader.join(); writer.join();
    };
    // phases (sah_next_phase): plain run = one; --twopassMode Basic adds a 1st pass without SAM, after which the junctions it found
    // are inserted into the index on the host (sjdb_insert.cpp) and the HBM copy is replaced (twoPassRunPass1.cpp:9-96);
    // --outFilterType BySJout adds a 2nd stage over the held reads with the filtered novel junctions as a whitelist (STAR.cpp:203-220)
    for (;;) {
        mapAllBatches();
        if (!failure.empty()) break;
        int phase = sah_next_phase(h);
        if (phase < 0) { fail(sah_error(h)); break; }
        if (phase == 0) break;
        if (phase == 1) {
            if (staramd_update_index(ctx, sah_genome(h), sah_params(h))) { fail(std::string("EXITING because of FATAL ERROR: index re-upload failed: ") + staramd_last_error()); break; }
            double s1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            fprintf(stderr, "star_amd: 1st pass + junction insertion + index re-upload: %.3f s (%llu reads)\n", s1, (unsigned long long)nReads);
        } else {
            const uint64_t *ns, *ne; uint64_t nn = sah_novel_junctions(h, &ns, &ne);
            if (staramd_set_novel_junctions(ctx, ns, ne, nn, 2)) { fail(std::string("EXITING because of FATAL ERROR: ") + staramd_last_error()); break; }
            fprintf(stderr, "star_amd: BySJout stage 1 done (%llu reads so far), %llu novel junctions passed filtering\n", (unsigned long long)nReads, (unsigned long long)nn);
        }
    }
    if (!failure.empty()) { fprintf(stderr, "\n%s\n", failure.c_str()); return 104; }
    if (sah_finish(h)) { fprintf(stderr, "\n%s\n", sah_error(h)); return 104; }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "star_amd: %llu reads, %.3f s wall in the mapping loop (%.3f s on the device) -> %.3f Mreads/s end to end\n",
            (unsigned long long)nReads, sec, msDevice / 1e3, sec > 0 ? (double)nReads / sec / 1e6 : 0.0);
    staramd_destroy(ctx);
    sah_destroy(h);
    return 0;
}
