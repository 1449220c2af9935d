// cli_run.cpp -- `star_amd`: command-line drop-in for `STAR --runMode alignReads` as a function
// (include/star_amd_cli.h).  Same flags (the subset that reaches the hot path or its outputs; anything else is rejected),
// same genomeDir, same Aligned.out.sam / SJ.out.tab / Log.final.out.  The per-read hot path runs on the MI355X(s) through
// the C ABI of include/star_amd.h; there is no CPU path.
//
// Pipeline (the reference interleaves these per thread, ReadAlignChunk_processChunks.cpp / _mapChunk.cpp):
//   reader thread    FASTQ text -> numeric batch            (sah_parse_slot)            one, batches numbered in input order
//   mapper threads   one per GPU (--gpuDevices a,b,...): a batch through that GPU's engine context (staramd_map_batch),
//                    plus its merged-mates / WASP re-mapping batches on the same context
//   writer thread    post-map + SAM text on --runThreadN host threads (sah_emit_slot), batches taken in INPUT order whatever
//                    order the GPUs finish in (the order-dependent host state: random multimapper order, vW carry, KeepInputOrder)
// Junction table, Stats and gene counts live in the one host object: nothing to merge inside a process (the reference's
// per-thread tables, outputSJ.cpp:39-83, collapse to one).  Between phases (2-pass, BySJout) every context gets the new
// index / whitelist.
#include <sys/file.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdio>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <array>
#include <algorithm>
#include <chrono>
#include <thread>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <string>
#include <memory>
#include "../../../include/star_amd_host.h"
#include "../../../include/star_amd_index.h"
#include "../../../include/star_amd_cli.h"
#include "../../../include/star_amd_async.h"

namespace {
typedef std::chrono::steady_clock Clock;
inline double since(Clock::time_point t) { return std::chrono::duration<double>(Clock::now() - t).count(); }

struct Msg { int slot = -1; int n = 0; uint64_t seq = 0; staramd_batch b; bool merged = false; };
struct Queue {                                   // hand-off between two pipeline stages
    std::mutex m; std::condition_variable cv; std::deque<Msg> q; bool closed = false;
    void push(const Msg &x) { { std::lock_guard<std::mutex> l(m); q.push_back(x); } cv.notify_all(); }
    void close() { { std::lock_guard<std::mutex> l(m); closed = true; } cv.notify_all(); }
    bool pop(Msg &x) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; x = q.front(); q.pop_front(); return true; }
    bool tryPop(Msg &x) { std::lock_guard<std::mutex> l(m); if (q.empty()) return false; x = q.front(); q.pop_front(); return true; }        // what is waiting, without waiting for it
    bool peek(Msg &x) { std::lock_guard<std::mutex> l(m); if (q.empty()) return false; x = q.front(); return true; }          // what the next pop would hand out, if anything is waiting
};
struct Tokens {                                  // counting semaphore over a small set of buffer indices
    std::mutex m; std::condition_variable cv; std::deque<int> free;
    // the slot that was given back LAST is handed out first: a batch then lands in buffers (line tables, page-locked numeric arrays, result arrays) that were in use a
    // moment ago, and a slot is used for the first time (page faults and hipHostMalloc inside the reader) only when the pipeline really is that deep.  Round-robin order
    // went round all slots whatever the depth: 4.94 / 5.56 M pairs/s against 6.84 / 6.19 with the host stages alone on a GPU box (profiles/r04_host_stages_on_the_gpu_box.txt)
    int take() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !free.empty(); }); static const bool fifo = getenv("STARAMD_SLOTS_FIFO") != nullptr; int v = fifo ? free.front() : free.back(); if (fifo) free.pop_front(); else free.pop_back(); return v; }
    void give(int v) { { std::lock_guard<std::mutex> l(m); free.push_back(v); } cv.notify_all(); }
};
// result arrays of a batch: page-locked (staramd_pinned_alloc), so that the engine's device-to-host copies are DMA transfers straight into them;
// resize() leaves new elements as they are (the engine overwrites what it reports)
template <class T> struct PinnedAlloc {
    typedef T value_type;
    PinnedAlloc() = default;
    template <class U> PinnedAlloc(const PinnedAlloc<U> &) {}
    // (page-locked memory is a limited resource: when the runtime has none left the array lives in ordinary memory -- the copies are then staged by the
    // runtime, nothing else changes; an exception thrown on a mapper thread would end the process)
    T *allocate(size_t n) {
        void *p = staramd_pinned_alloc((uint64_t)n * sizeof(T));
        if (p) return (T *)p;
        { const size_t bytes = n * sizeof(T); p = malloc(bytes > 0 ? bytes : 1); }
        if (!p) throw std::bad_alloc();
        { std::lock_guard<std::mutex> l(pageableMutex()); pageable().push_back(p); }
        return (T *)p;
    }
    void deallocate(T *p, size_t) {
        {
            std::lock_guard<std::mutex> l(pageableMutex());
            std::vector<void *> &v = pageable();
            for (size_t i = 0; i < v.size(); i++) if (v[i] == (void *)p) { v[i] = v.back(); v.pop_back(); free(p); return; }
        }
        staramd_pinned_free(p);
    }
    static std::vector<void *> &pageable() { static std::vector<void *> v; return v; }
    static std::mutex &pageableMutex() { static std::mutex m; return m; }
    template <class U> void construct(U *p) { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const PinnedAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const PinnedAlloc<U> &) const { return false; }
};
struct ResBuf {
    std::vector<staramd_read_result, PinnedAlloc<staramd_read_result>> reads; std::vector<staramd_transcript, PinnedAlloc<staramd_transcript>> tr; std::vector<staramd_exon, PinnedAlloc<staramd_exon>> ex; staramd_results res;
    // what the batches of this run have needed per read so far (x 1024): a slot that is used for the first time starts there instead of learning it by an overflow of its own
    // (--chimSegmentMin returns every transcript of every window: 32 per pair, 2 GB of page-locked memory per slot)
    static std::atomic<uint64_t> &seenTr() { static std::atomic<uint64_t> v(0); return v; }
    static std::atomic<uint64_t> &seenEx() { static std::atomic<uint64_t> v(0); return v; }
    static void learn(uint64_t nReads, uint64_t trCount, uint64_t exCount) {
        if (!nReads) return;
        const uint64_t t = trCount * 1024 / nReads + 1, e = exCount * 1024 / nReads + 1;
        for (uint64_t o = seenTr().load(); o < t && !seenTr().compare_exchange_weak(o, t); ) {}
        for (uint64_t o = seenEx().load(); o < e && !seenEx().compare_exchange_weak(o, e); ) {}
    }
    void size(uint64_t nReads) {
        if (reads.size() < nReads) reads.resize(nReads);
        const uint64_t wantTr = std::max<uint64_t>(nReads * 4 + 4096, nReads * seenTr().load() / 1024 * 5 / 4 + 4096);
        if (tr.size() < wantTr) { tr.resize(wantTr); ex.resize(std::max<uint64_t>(tr.size() * 3, nReads * seenEx().load() / 1024 * 5 / 4 + 4096)); }
        memset(&res, 0, sizeof(res));
        point();
    }
    void point() { res.reads = reads.data(); res.tr = tr.data(); res.trCapacity = tr.size(); res.ex = ex.data(); res.exCapacity = ex.size(); }
};

// CLI-level flags that never reach the host library: --gpuDevices a,b,c   --benchWarmupReads N
struct CliFlags { std::vector<int> devices; uint64_t warmupReads = 0; std::vector<char *> rest; };
CliFlags splitFlags(int argc, char **argv) {
    CliFlags f;
    for (int i = 0; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--gpuDevices" && i + 1 < argc) {
            std::string v = argv[++i]; size_t p = 0;
            while (p <= v.size()) { size_t q = v.find(',', p); if (q == std::string::npos) q = v.size(); if (q > p) f.devices.push_back(atoi(v.substr(p, q - p).c_str())); p = q + 1; }
        } else if (a == "--benchWarmupReads" && i + 1 < argc) f.warmupReads = strtoull(argv[++i], nullptr, 10);
        else f.rest.push_back(argv[i]);
    }
    return f;
}
} // namespace

extern "C" namespace {
// STARAMD_PIPELINE_LOG=<file>: one line per batch and stage (stage, batch number, start and end in ms since the run's first batch) -- where a batch waited and for what
struct PipeLog {
    std::mutex m; std::vector<std::array<double, 4> > ev; const char *path = getenv("STARAMD_PIPELINE_LOG"); std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double now() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
    void add(int stage, uint64_t seq, double a, double b) { if (!path) return; std::lock_guard<std::mutex> l(m); ev.push_back({(double)stage, (double)seq, a, b}); }
    void dump() {
        if (!path) return;
        FILE *f = fopen(path, "w"); if (!f) return;
        static const char *names[] = {"fill", "convert", "map", "emit"};
        std::sort(ev.begin(), ev.end(), [](const std::array<double, 4> &x, const std::array<double, 4> &y) { return x[2] < y[2]; });
        for (auto &e : ev) fprintf(f, "%-8s %4.0f %10.2f %10.2f %8.2f\n", names[(int)e[0]], e[1], e[2], e[3], e[3] - e[2]);
        fclose(f);
    }
};
struct StageCpu {           // thread CPU of one batch's work on a stage thread (sah_cpu_seconds; the helper threads of a stage count their own)
    int stage; timespec t0;
    explicit StageCpu(int st) : stage(st) { clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t0); }
    ~StageCpu() { timespec t1; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t1); sah_cpu_add(stage, (uint64_t)((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec))); }
};
}

int staramd_cli_main(int argc, char **argv, const staramd_cli_hooks *hooks, staramd_cli_report *report) {
    staramd_cli_report rep; memset(&rep, 0, sizeof(rep));
    auto publish = [&]() { if (report) *report = rep; };
    for (int i = 1; i < argc; i++) if (std::string(argv[i]) == "--version") { printf("2.7.11b\n"); return 0; }      // the version whose behaviour is reproduced (Parameters.cpp:340-343)
    CliFlags flags = splitFlags(argc, argv);
    if (!getenv("STARAMD_SJDB_HOST")) sah_set_sjdb_device_fn(staramd_sjdb_insert, flags.devices.empty() ? 0 : flags.devices[0]);   // junction insertion on the device
    char err[4096];
    // One process per GPU on a node: every rank reads the ~30 GB index into host memory before it uploads it, and keeps only ~5 GB of it afterwards
    // (sah_engines_ready).  STARAMD_INDEX_LOAD_LOCK=<file> makes the ranks take turns (an advisory file lock from here until the engine contexts hold
    // the index), so that the peak is one index + a few GB per rank instead of one index per rank -- which a container limit does not survive.
    struct LoadLock { int fd = -1; void take() { if (const char *p = getenv("STARAMD_INDEX_LOAD_LOCK")) { fd = open(p, O_CREAT | O_RDWR, 0666); if (fd >= 0 && flock(fd, LOCK_EX) != 0) { close(fd); fd = -1; } } }
                      void drop() { if (fd >= 0) { flock(fd, LOCK_UN); close(fd); fd = -1; } } ~LoadLock() { drop(); } } loadLock;
    loadLock.take();
    if (!getenv("STARAMD_PAGEABLE_BATCHES")) sah_set_batch_alloc(staramd_pinned_alloc, staramd_pinned_free);      // batch arrays in page-locked memory: the uploads are DMA transfers
    void *h = sah_create((int)flags.rest.size(), flags.rest.data(), err, sizeof(err));
    if (!h) { fprintf(stderr, "\n%s\n", err); return 104; }
    if (sah_tool_done(h)) { sah_destroy(h); return 0; }          // --runMode inputAlignmentsFromBAM: nothing to map
    if (sah_generate_mode(h)) {                                  // --runMode genomeGenerate: suffix array + SAindex on the device
        const uint8_t *G; uint64_t nGenome, saCap, saiCap; uint32_t gsb, nb; uint8_t *SA, *SAi;
        if (sah_generate_buffers(h, &G, &nGenome, &gsb, &nb, &SA, &saCap, &SAi, &saiCap)) { fprintf(stderr, "\n%s\n", sah_error(h)); return 104; }
        staramd_index_params ip; memset(&ip, 0, sizeof(ip));
        ip.nGenome = nGenome; ip.GstrandBit = gsb; ip.gSAindexNbases = nb; ip.gSAsparseD = 1;
        staramd_index_result ir;
        auto tg = Clock::now();
        int grc = staramd_index_build(flags.devices.empty() ? sah_device(h) : flags.devices[0], G, &ip, SA, saCap, SAi, saiCap, &ir);
        if (grc) { fprintf(stderr, "\nEXITING because of FATAL ERROR: index build on the MI355X failed: %s\n", staramd_index_last_error()); sah_destroy(h); return 105; }
        double sBuild = since(tg);
        if (sah_generate_finish(h, ir.nSA, ir.nSAbyte, ir.nSAibyte)) { fprintf(stderr, "\n%s\n", sah_error(h)); sah_destroy(h); return 104; }
        fprintf(stderr, "star_amd: genomeGenerate: %llu suffixes, %u doubling rounds, device build %.3f s (%.1f ms on the stream), junction insertion + files %.3f s\n",
                (unsigned long long)ir.nSA, ir.doublingRounds, sBuild, ir.msTotal, since(tg) - sBuild);
        sah_destroy(h);
        return 0;
    }
    const uint64_t batchReads = sah_batch_reads(h);
    std::vector<int> devices = flags.devices;
    if (devices.empty()) devices.push_back(sah_device(h));
    if (devices.size() > STARAMD_CLI_MAX_DEV) { fprintf(stderr, "\nEXITING because of fatal PARAMETERS error: --gpuDevices lists more than %d devices\n", STARAMD_CLI_MAX_DEV); sah_destroy(h); return 104; }
    // ---- engine contexts.  Every entry of --gpuDevices is an OWNER: a context with an index replica of its own, uploaded concurrently.  Every owner
    // gets STARAMD_CONTEXTS_PER_GPU - 1 (default: 1) further contexts that share its resident index (staramd_create_shared: work space and stream of
    // their own, no second replica); each context has a mapper thread, so a GPU always has a second batch in flight while the results of one are
    // copied out and handed on (the reference: runThreadN workers over one shared genome, STAR.cpp:194-201)
    const int nOwners = (int)devices.size();
    // engine contexts (= mapper threads) per GPU.  A second context over the same resident index was worth +7 % in round 3, when kernels were slower and the host had 64 threads;
    // with the kernels of round 4 and the 16 CPUs the GPU boxes really give a container, one context is as fast on a quiet box (6.8 M pairs/s either way) and faster on a
    // loaded one (5.7 vs 5.1): the launches of two contexts do not overlap, they stretch each other (profiles/r04_timeline_two_contexts.txt)
    int perGpu = 1; if (const char *e = getenv("STARAMD_CONTEXTS_PER_GPU")) perGpu = std::max(1, atoi(e));
    while (nOwners * perGpu > STARAMD_CLI_MAX_DEV) perGpu--;
    const int nDev = nOwners * perGpu;                       // contexts = mapper threads; context d belongs to owner d % nOwners
    rep.nDevices = nOwners; rep.nContexts = nDev; rep.genomeLoadSeconds = sah_genome_load_seconds(h);
    // chimeric detection: the partner loop on the device when the engine can (the stand-ins of the CPU tests cannot) -- STARAMD_CHIM_ON_DEVICE=0: every transcript of every
    // window comes back and the loop runs here, as in rounds 1 - 5
    if ((staramd_capabilities() & STARAMD_CAP_CHIM_SELECT) && !(getenv("STARAMD_CHIM_ON_DEVICE") && atoi(getenv("STARAMD_CHIM_ON_DEVICE")) == 0)) { if (sah_chim_select_on_device(h) && getenv("STARAMD_VERBOSE")) fprintf(stderr, "star_amd: partner of chimeric detection chosen on the device (resultSelect 2)\n"); }
    std::vector<staramd_ctx *> ctx(nDev, nullptr);
    auto destroyAll = [&]() { for (int d = nDev - 1; d >= 0; d--) if (ctx[d]) { staramd_destroy(ctx[d]); ctx[d] = nullptr; } };      // sharers before their owners
    {
        auto tu = Clock::now();
        std::vector<int> rcs(nDev, 0); std::vector<std::string> es(nDev);
        std::vector<std::thread> th;
        for (int d = 0; d < nOwners; d++) th.emplace_back([&, d] { rcs[d] = staramd_create(&ctx[d], devices[d], sah_genome(h), sah_params(h), (uint32_t)batchReads, 0); if (rcs[d]) es[d] = staramd_last_error(); });
        for (auto &t : th) t.join();
        for (int d = nOwners; d < nDev; d++) if (!rcs[d % nOwners]) { rcs[d] = staramd_create_shared(&ctx[d], ctx[d % nOwners], (uint32_t)batchReads, 0); if (rcs[d]) es[d] = staramd_last_error(); }
        for (int d = 0; d < nDev; d++) if (rcs[d]) {
            fprintf(stderr, "\nEXITING because of FATAL ERROR: cannot initialise the MI355X engine on device %d: %s\n", devices[d % nOwners], es[d].c_str());
            destroyAll();
            sah_destroy(h); return 105;
        }
        rep.indexUploadSeconds = since(tu);
    }
#ifndef STARAMD_NO_RESIDENT_SJDB
    // junction insertion between the passes runs on the arrays resident in HBM, on every context (no host copy of the new SA unless it is to be saved)
    std::vector<staramd_ctx *> owners(ctx.begin(), ctx.begin() + nOwners);      // index changes go to the owners; the contexts that share an index follow
    struct ResidentUser { std::vector<staramd_ctx *> *ctx; } residentUser{&owners};
    // ... provided every device has the room for it (the work space of the insertion is several times the suffix array): asked ONCE, here, because the host copy of
    // the suffix array is released below and nothing could insert without it afterwards.  Without the room: host buffers (staramd_sjdb_insert) + re-upload.
    bool residentFits = true;
    for (int d = 0; d < nOwners; d++) if (!staramd_insert_junctions_fits(ctx[d], (uint64_t)sah_limit_sjdb_insert(h), (uint32_t)sah_sjdb_length(h))) residentFits = false;
    if (!residentFits && (sah_in_pass1(h) || getenv("STARAMD_VERBOSE"))) fprintf(stderr, "star_amd: not enough free device memory to insert junctions into the resident index: the suffix array stays on the host, insertion goes through host buffers\n");
    if (residentFits && !getenv("STARAMD_SJDB_HOST") && !getenv("STARAMD_SJDB_NO_RESIDENT")) {
        sah_set_sjdb_resident_fn([](void *user, const staramd_sjdb_args *a, staramd_sjdb_result *res) -> int {
            std::vector<staramd_ctx *> &cx = *((ResidentUser *)user)->ctx;
            std::vector<int> rcs(cx.size(), 0); std::vector<staramd_sjdb_result> rs(cx.size()); std::vector<std::string> es(cx.size());
            std::vector<std::thread> th;
            // (the engine keeps its last error per calling thread: the text is taken on the thread that made the call)
            for (size_t d = 0; d < cx.size(); d++) th.emplace_back([&, d] { rcs[d] = staramd_insert_junctions(cx[d], a, d == 0 ? a->SAout : nullptr, a->saOutCapacity, d == 0 ? a->SAiOut : nullptr, a->saiOutCapacity, &rs[d]); if (rcs[d]) es[d] = staramd_last_error(); });
            for (auto &t : th) t.join();
            for (size_t d = 0; d < cx.size(); d++) if (rcs[d]) { fprintf(stderr, "star_amd: junction insertion on device context %zu failed: %s\n", d, es[d].c_str()); return rcs[d]; }
            *res = rs[0];
            return 0;
        }, &residentUser);
        sah_engines_ready(h);
    }
#endif
    loadLock.drop();
    // in flight: one batch in each half of the reader, one per mapper, one in the writer, the rest queued between them.  Three more than the stages hold: the reader's time per
    // batch scatters (25 - 80 ms for 400 k pairs on a shared host) around a mean well under the device's 52 ms, and with a queue of one a single slow block left the GPU
    // idle for 25 - 30 ms every few batches (rocprofv3 timeline, profiles/r04_timeline_*.txt)
    int extraSlots = 3; if (const char *e = getenv("STARAMD_EXTRA_SLOTS")) extraSlots = std::max(0, atoi(e));
    const int nSlots = std::min(24, 2 * nDev + 3 + extraSlots);
    ResBuf::seenTr().store(0); ResBuf::seenEx().store(0);          // (what an earlier run in this process learned -- another data set, other flags -- does not size this one's arrays)
    std::vector<ResBuf> rb(nSlots), rbMerged(nSlots), rbWasp(nSlots);
    std::vector<ResBuf> piecePart(nDev);
    for (auto &r : rb) r.size(batchReads);
    std::string failure; std::mutex failM; std::atomic<bool> failed(false);
    auto fail = [&](const std::string &s) { std::lock_guard<std::mutex> l(failM); if (failure.empty()) failure = s; failed = true; };
    // ---- timing state
    PipeLog plog;
    std::mutex statM;
    uint64_t nReads = 0; double msDeviceAll = 0;
    bool timedOn = flags.warmupReads == 0; bool warmupPending = flags.warmupReads > 0;
    Clock::time_point tTimed = Clock::now();
    const auto t0 = Clock::now();
    // warm-up pause: the reader stops after >= warmupReads reads, waits until every batch handed out so far is written
    std::mutex drainM; std::condition_variable drainCv; uint64_t seqParsed = 0, seqEmitted = 0;

    auto mapAllBatches = [&]() {
        Queue parsed; Tokens slots;
        std::mutex doneM; std::condition_variable doneCv; std::map<uint64_t, Msg> done; bool mappersClosed = false;
        for (int i = 0; i < nSlots; i++) slots.give(i);
        { std::lock_guard<std::mutex> l(drainM); seqParsed = seqEmitted = 0; }
        // the reader is two stages on two threads: `filler` reads the text of batch k+1 and finds its lines (everything that touches the input) while
        // `reader` turns the text of batch k into the numeric batch (sah_fill_slot / sah_convert_slot; tools/host_bench.py: the two halves take about
        // the same time, and their sum was the longest stage of the pipeline at the thread budget of an 8-GPU node)
        Queue filled;
        std::thread filler([&] {
            uint64_t seq = 0, readsOut = 0;
            for (;;) {
                if (warmupPending && readsOut >= flags.warmupReads) {            // drain, barrier, start the clock
                    { std::unique_lock<std::mutex> l(drainM); drainCv.wait(l, [&] { return seqEmitted == seqParsed || failed.load(); }); }
                    if (hooks && hooks->warmup_done) hooks->warmup_done(hooks->user);
                    std::lock_guard<std::mutex> l(statM);
                    warmupPending = false; timedOn = true; tTimed = Clock::now();
                    { double z[8]; sah_cpu_seconds(z, 1); }
                }
                Msg m; m.slot = slots.take();
                if (failed.load()) { slots.give(m.slot); break; }
                auto tp = Clock::now();
                { StageCpu sc(0); const double a = plog.now(); m.n = sah_fill_slot(h, m.slot, batchReads); plog.add(0, seq, a, plog.now()); }
                if (m.n < 0) { fail(sah_error(h)); slots.give(m.slot); break; }
                if (m.n == 0) { slots.give(m.slot); break; }
                { std::lock_guard<std::mutex> l(statM); if (timedOn) rep.parseBusy += since(tp); }
                m.seq = seq++; readsOut += (uint64_t)m.n;
                { std::lock_guard<std::mutex> l(drainM); seqParsed = seq; }
                filled.push(m);
            }
            filled.close();
        });
        std::thread reader([&] {
            Msg m;
            while (filled.pop(m)) {
                auto tp = Clock::now();
                { StageCpu sc(1); const double a = plog.now(); if (!failed.load() && sah_convert_slot(h, m.slot, &m.b) < 0) fail(sah_error(h)); plog.add(1, m.seq, a, plog.now()); }
                if (failed.load()) m.n = 0;                                      // still goes down the pipeline so that the slot and the sequence number are released
                // a slot on its first way down the pipeline gets result arrays of the size the batches before it needed, here, beside the kernels of the batch ahead -- not on
                // the mapper thread between two launches (page-locked memory: ~0.1 s per GB)
                if (m.n > 0 && rb[m.slot].tr.size() < (uint64_t)m.n * ResBuf::seenTr().load() / 1024) rb[m.slot].size((uint64_t)m.n);
                { std::lock_guard<std::mutex> l(statM); if (timedOn && m.n > 0) rep.convertBusy += since(tp); }
                parsed.push(m);
            }
            parsed.close();
        });
        std::thread writer([&] {
            uint64_t next = 0;
            for (;;) {
                Msg m;
                {
                    std::unique_lock<std::mutex> l(doneM);
                    doneCv.wait(l, [&] { return done.count(next) || mappersClosed; });
                    auto it = done.find(next);
                    if (it == done.end()) { if (mappersClosed) break; continue; }
                    m = it->second; done.erase(it);
                }
                next++;
                auto te = Clock::now();
                StageCpu sc(3); const double ea = plog.now();
                if (!failed.load() && m.n > 0 && (m.merged ? sah_emit_slot_merged(h, m.slot, &rb[m.slot].res, &rbMerged[m.slot].res) : sah_emit_slot(h, m.slot, &rb[m.slot].res))) fail(sah_error(h));
                { std::lock_guard<std::mutex> l(statM); if (timedOn && m.n > 0) { rep.emitBusy += since(te); rep.timedReads += (uint64_t)m.n; rep.batches++; } if (m.n > 0) nReads += (uint64_t)m.n; }
                plog.add(3, m.seq, ea, plog.now());
                slots.give(m.slot);
                { std::lock_guard<std::mutex> l(drainM); seqEmitted = next; }
                drainCv.notify_all();
            }
        });
        std::vector<std::thread> mappers;
        // STARAMD_OVERLAP_COPIES=1 (one context per device, no second batch per batch -- merged mates, allele-swapped reads): the mapper keeps TWO batches going -- when the kernels
        // of batch k are done it begins batch k+1 before the results of k leave the device (staramd_map_begin / _wait / _end, include/star_amd_async.h), so the ~100 MB copy runs
        // beside kernels.  Measured (profiles/r05_e2e_session8_*): a batch every 46.2 ms instead of 47.7 -- but the copy is a shader ("blit") kernel of the runtime, the persistent
        // kernels of batch k+1 leave it no CU, and it completes when THEY do: the results of every batch arrive one batch late, the post-map stage and the writer run one
        // batch behind, and a run of 20 batches ends 40 ms later than with blocking calls.  Worth 3 % on a long run, a loss on a short one: off by default.
        const bool overlapCopies = nDev == nOwners && !sah_needs_second_batch(h) && getenv("STARAMD_OVERLAP_COPIES") && atoi(getenv("STARAMD_OVERLAP_COPIES")) > 0;
        for (int d = 0; d < nDev && overlapCopies; d++) mappers.emplace_back([&, d] {
            auto pushDone = [&](const Msg &m) { { std::lock_guard<std::mutex> l(doneM); done[m.seq] = m; } doneCv.notify_all(); };
            auto begin = [&](Msg &m) -> bool {            // false: the batch goes down the pipeline unmapped (an error is recorded, or it carries no reads)
                if (failed.load() || m.n <= 0) { if (failed.load()) m.n = 0; return false; }
                ResBuf &r = rb[m.slot];
                if (r.reads.size() < m.b.nReads || r.tr.empty()) r.size(std::max<uint64_t>(m.b.nReads, 1024));
                if (staramd_map_begin(ctx[d], &m.b)) { fail(std::string("EXITING because of FATAL ERROR in the MI355X engine: ") + staramd_last_error()); m.n = 0; return false; }
                return true;
            };
            Msg cur; bool have = false; double curStart = 0;
            for (;;) {
                if (!have) {
                    if (!parsed.pop(cur)) break;
                    curStart = plog.now();
                    StageCpu sc(2);
                    if (!begin(cur)) { cur.merged = false; plog.add(2, cur.seq, curStart, plog.now()); pushDone(cur); continue; }
                    have = true;
                }
                StageCpu sc(2);
                auto tm = Clock::now();
                // the upload of the batch behind it, beside the kernels of this one.  Only with ONE mapper: with several, another mapper may pop the batch that was peeked
                // here, and this context would keep a stale upload (the blocking loop below has the same guard)
                uint64_t peekedSeq = ~0ull; bool peeked = false;
                { Msg pk; if (nDev == 1 && parsed.peek(pk) && pk.n > 0) { (void)staramd_prefetch_batch(ctx[d], &pk.b); peeked = true; peekedSeq = pk.seq; } }
                int rc = staramd_map_wait(ctx[d]);
                Msg nx; bool haveNx = false, begunNx = false;
                ResBuf &r = rb[cur.slot];
                if (!rc) {
                    haveNx = parsed.tryPop(nx);
                    if (peeked && (!haveNx || nx.seq != peekedSeq)) (void)staramd_prefetch_cancel(ctx[d]);      // the batch shown to the engine is not the one that follows: its upload is forgotten
                    const bool passNx = haveNx && !failed.load() && nx.n > 0;
                    if (passNx) { ResBuf &rn = rb[nx.slot]; if (rn.reads.size() < nx.b.nReads || rn.tr.empty()) rn.size(std::max<uint64_t>(nx.b.nReads, 1024)); }
                    rc = staramd_map_end(ctx[d], &r.res, passNx ? &nx.b : nullptr);
                    if (rc == STARAMD_ERR_RESULT_OVERFLOW) {             // more transcripts than the buffers hold -> grow and ask again (the batch is still in flight)
                        r.tr.resize(r.res.trCount + r.res.trCount / 4 + 4096); r.ex.resize(r.res.exCount + r.res.exCount / 4 + 4096); r.point();
                        rc = staramd_map_end(ctx[d], &r.res, passNx ? &nx.b : nullptr);
                    }
                    begunNx = passNx && !rc;
                }
                if (rc) { fail(std::string("EXITING because of FATAL ERROR in the MI355X engine: ") + staramd_last_error()); cur.n = 0; }
                else {
                    float st8[8] = {0}; uint64_t cnt[64] = {0};
                    const int k = staramd_get_timings(ctx[d], st8, 8); const int kc = staramd_get_counters(ctx[d], cnt, 64);
                    std::lock_guard<std::mutex> l(statM);
                    msDeviceAll += r.res.msTotalDevice;
                    if (timedOn) { rep.deviceBusy[d] += since(tm); rep.deviceMs[d] += r.res.msTotalDevice; for (int i = 0; i < k && i < 8; i++) rep.stageMs[i] += st8[i]; for (int i = 0; i < kc && i < 64; i++) rep.counters[i] += cnt[i]; }
                }
                cur.merged = false;
                if (failed.load()) cur.n = 0;
                plog.add(2, cur.seq, curStart, plog.now());
                pushDone(cur);
                have = false;
                if (haveNx) {
                    curStart = plog.now();
                    if (begunNx) { cur = nx; have = true; }
                    else if (begin(nx)) { cur = nx; have = true; }       // (it was not handed to staramd_map_end, or that call failed before it began it)
                    else { nx.merged = false; pushDone(nx); }
                }
            }
        });
        for (int d = 0; d < nDev && !overlapCopies; d++) mappers.emplace_back([&, d] {
            Msg m;
            while (parsed.pop(m)) {
                int rc = 0;
                StageCpu sc(2); const double ma = plog.now();
                if (!failed.load()) {
                    auto tm = Clock::now(); double msDev = 0; float stage[8] = {0}; uint64_t cnt[64] = {0};
                    auto mapInto = [&](const staramd_batch &bt, ResBuf &r, bool main) {
                        if (r.reads.size() < bt.nReads || r.tr.empty() || r.tr.size() < bt.nReads * ResBuf::seenTr().load() / 1024) r.size(std::max<uint64_t>(bt.nReads, 1024));
                        staramd_results &res = r.res;
                        int e = staramd_map_batch(ctx[d], &bt, &res);
                        if (e == STARAMD_ERR_RESULT_OVERFLOW) {          // more transcripts than the buffers hold -> grow and ask again (the results are resident: the engine copies them, nothing is mapped twice)
                            ResBuf::learn(bt.nReads, res.trCount, res.exCount);
                            r.tr.resize(res.trCount + res.trCount / 4 + 4096); r.ex.resize(res.exCount + res.exCount / 4 + 4096); r.point();
                            e = staramd_map_batch(ctx[d], &bt, &res);
                        }
                        if (!e) {
                            ResBuf::learn(bt.nReads, res.trCount, res.exCount);
                            msDev += res.msTotalDevice;
                            if (main) { float s[8] = {0}; int k = staramd_get_timings(ctx[d], s, 8); for (int i = 0; i < k; i++) stage[i] += s[i]; uint64_t c[64] = {0}; int kc = staramd_get_counters(ctx[d], c, 64); for (int i = 0; i < kc; i++) cnt[i] += c[i]; }
                        }
                        return e;
                    };
                    // the batch that is next in line (if the reader is ahead, as it normally is) starts its upload now, beside the kernels of this one
                    { Msg nx; if (nDev == 1 && parsed.peek(nx) && nx.n > 0) (void)staramd_prefetch_batch(ctx[d], &nx.b); }
                    rc = mapInto(m.b, rb[m.slot], true);
                    m.merged = false;
                    if (!rc) {                                           // --peOverlapNbasesMin: the overlapping mates of the batch, merged into single reads, are a second batch
                        staramd_batch mb;
                        if (sah_merged_slot(h, m.slot, &mb) > 0) { m.merged = true; rc = mapInto(mb, rbMerged[m.slot], false); }
                    }
                    if (!rc) {                                           // --waspOutputMode: allele-swapped copies of some reads, one more batch (can be larger than the batch itself)
                        staramd_batch wb;
                        int nw = sah_wasp_slot(h, m.slot, &rb[m.slot].res, &wb);
                        if (nw > 0) {
                            // as many pieces as it takes (a read over a dense cluster of SNVs has up to 1023 copies); every piece is rebased to offset 0
                            // (the engine sizes and uploads a batch by readOffset[nReads]); the results of the pieces are appended to one set
                            ResBuf &r = rbWasp[m.slot]; ResBuf &part = piecePart[d];
                            if (r.reads.size() < (size_t)nw) { r.reads.resize((size_t)nw); }
                            uint64_t trN = 0, exN = 0; std::vector<uint64_t> off;
                            for (uint32_t doneN = 0; doneN < (uint32_t)nw && !rc; ) {
                                staramd_batch piece = wb; piece.nReads = std::min<uint32_t>((uint32_t)batchReads, (uint32_t)nw - doneN);
                                const uint64_t base = wb.readOffset[doneN];
                                off.resize(piece.nReads + 1);
                                for (uint32_t k = 0; k <= piece.nReads; k++) off[k] = wb.readOffset[doneN + k] - base;
                                piece.bases = wb.bases + base; piece.readOffset = off.data(); piece.mate1Length = wb.mate1Length + doneN; piece.mmMaxTotal = wb.mmMaxTotal + doneN;
                                rc = mapInto(piece, part, false);
                                if (rc) break;
                                const staramd_results &pr = part.res;
                                if (r.tr.size() < trN + pr.trCount) r.tr.resize((trN + pr.trCount) * 3 / 2 + 1024);
                                if (r.ex.size() < exN + pr.exCount) r.ex.resize((exN + pr.exCount) * 3 / 2 + 1024);
                                for (uint32_t k = 0; k < piece.nReads; k++) { r.reads[doneN + k] = pr.reads[k]; r.reads[doneN + k].trOffset += (uint32_t)trN; }
                                for (uint64_t k = 0; k < pr.trCount; k++) { r.tr[trN + k] = pr.tr[k]; r.tr[trN + k].exonOffset += (uint32_t)exN; }
                                if (pr.exCount) memcpy(&r.ex[exN], pr.ex, pr.exCount * sizeof(staramd_exon));
                                trN += pr.trCount; exN += pr.exCount; doneN += piece.nReads;
                            }
                            r.point(); r.res.trCount = trN; r.res.exCount = exN;
                        }
                        if (!rc && sah_wasp_results_slot(h, m.slot, &rb[m.slot].res, nw > 0 ? &rbWasp[m.slot].res : nullptr)) { fail(sah_error(h)); rc = 0; }
                    }
                    if (rc) fail(std::string("EXITING because of FATAL ERROR in the MI355X engine: ") + staramd_last_error());
                    std::lock_guard<std::mutex> l(statM);
                    msDeviceAll += msDev;
                    if (timedOn && !rc) { rep.deviceBusy[d] += since(tm); rep.deviceMs[d] += msDev; for (int i = 0; i < 8; i++) rep.stageMs[i] += stage[i]; for (int i = 0; i < 64; i++) rep.counters[i] += cnt[i]; }
                }
                if (failed.load()) m.n = 0;                              // still goes through the writer so that the slot and the sequence number are released
                plog.add(2, m.seq, ma, plog.now());
                { std::lock_guard<std::mutex> l(doneM); done[m.seq] = m; }
                doneCv.notify_all();
            }
        });
        for (auto &t : mappers) t.join();
        { std::lock_guard<std::mutex> l(doneM); mappersClosed = true; }
        doneCv.notify_all();
        drainCv.notify_all();
        filler.join(); reader.join(); writer.join();
    };
    // phases (sah_next_phase): plain run = one; --twopassMode Basic adds a 1st pass without SAM, after which the junctions it found
    // are inserted into the index (sjdb_insert.cpp) and every HBM copy is replaced (twoPassRunPass1.cpp:9-96);
    // --outFilterType BySJout adds a 2nd stage over the held reads with the filtered novel junctions as a whitelist (STAR.cpp:203-220)
    for (;;) {
        mapAllBatches();
        if (failed.load()) break;
        // the phase that follows is known only after sah_next_phase; ranks exchange their tables before it
        if (hooks && hooks->exchange && hooks->exchange(hooks->user, h, 0)) { fail("cross-rank exchange failed"); break; }
        int phase = sah_next_phase(h);
        if (phase < 0) { fail(sah_error(h)); break; }
        if (phase == 0) break;
        std::vector<int> rcs(nDev, 0); std::vector<std::string> es(nDev);
        std::vector<std::thread> th;
        if (phase == 1) {
            const bool inEngine = sah_index_in_engine(h) != 0;       // the insertion ran on the resident arrays: only tables and parameters are new
            for (int d = 0; d < nOwners; d++) th.emplace_back([&, d] {
                rcs[d] = inEngine ? staramd_update_tables(ctx[d], sah_genome(h), sah_params(h)) : staramd_update_index(ctx[d], sah_genome(h), sah_params(h));
                if (rcs[d]) es[d] = staramd_last_error(); });
            for (auto &t : th) t.join();
            for (int d = 0; d < nOwners; d++) if (rcs[d]) fail(std::string("EXITING because of FATAL ERROR: index re-upload failed: ") + es[d]);
            if (failed.load()) break;
            rep.pass1Seconds = since(t0);
            fprintf(stderr, "star_amd: 1st pass + junction insertion + index re-upload: %.3f s (%llu reads)\n", rep.pass1Seconds, (unsigned long long)nReads);
        } else {
            const uint64_t *ns, *ne; uint64_t nn = sah_novel_junctions(h, &ns, &ne);
            for (int d = 0; d < nOwners; d++) if (staramd_set_novel_junctions(ctx[d], ns, ne, nn, 2)) fail(std::string("EXITING because of FATAL ERROR: ") + staramd_last_error());
            if (failed.load()) break;
            fprintf(stderr, "star_amd: BySJout stage 1 done (%llu reads so far), %llu novel junctions passed filtering\n", (unsigned long long)nReads, (unsigned long long)nn);
        }
    }
    int exitCode = 0;
    if (failed.load()) { fprintf(stderr, "\n%s\n", failure.c_str()); exitCode = 104; }
    else if (hooks && hooks->exchange && hooks->exchange(hooks->user, h, 1)) { fprintf(stderr, "\ncross-rank exchange failed\n"); exitCode = 104; }
    else { const auto tf = Clock::now(); if (sah_finish(h)) { fprintf(stderr, "\n%s\n", sah_error(h)); exitCode = 104; } rep.finishSeconds = since(tf); }
    sah_emit_seconds(h, rep.emitParts);
    plog.dump();
    sah_fast_path_counts(h, rep.fastPaths);
    sah_cpu_seconds(rep.cpuSeconds, 0);
    for (int d = 0; d < nDev; d++) if (ctx[d]) { rep.fastPaths[2] += staramd_prefetch_hits(ctx[d]); rep.fastPaths[3] += staramd_overlapped_batches(ctx[d]); }
    double sec = since(t0);
    rep.reads = nReads; rep.wallMapping = sec; rep.timedWall = since(tTimed);
    if (!exitCode) fprintf(stderr, "star_amd: %llu reads, %.3f s wall in the mapping loop (%.3f s on the device) -> %.3f Mreads/s end to end, %d GPU(s)\n",
                           (unsigned long long)nReads, sec, msDeviceAll / 1e3 / nDev, sec > 0 ? (double)nReads / sec / 1e6 : 0.0, nOwners);
    if (!exitCode) fprintf(stderr, "star_amd: fast paths: output through a file mapping %llu batches, input from file mappings %llu, uploads prefetched %llu, kernels begun beside the copy of the results before them %llu\n",
                           (unsigned long long)rep.fastPaths[0], (unsigned long long)rep.fastPaths[1], (unsigned long long)rep.fastPaths[2], (unsigned long long)rep.fastPaths[3]);
    sah_set_sjdb_resident_fn(nullptr, nullptr);
    destroyAll();
    sah_destroy(h);
    publish();
    return exitCode;
}
