// engine.hip -- C ABI of the MI355X engine (include/star_amd.h): index upload, work-space, launches, timing.
// The hot path has no CPU fallback: every entry point fails with an error code when the GPU or the
// kernels are not usable.
#include "dev.h"
#include "../index/hip_backend.h"
#include "../index/sjdb_core.h"
#include "../../../include/star_amd_index.h"
#include "../../../include/star_amd_async.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <mutex>

extern "C" __global__ void k_seed_search(const DevIndex *X, DevBatch B, DSeed *scratch, u32 scratchPerLane, const u32 *inList);
extern "C" __global__ void k_seed_plan(const DevIndex *X, DevBatch B, SeedWork W);
extern "C" __global__ void k_seed_units(const DevIndex *X, DevBatch B, SeedWork W);
extern "C" __global__ void k_seed_merge(const DevIndex *X, DevBatch B, SeedWork W);
extern "C" __global__ void k_pack_reads(DevBatch B, u32 *packed, u32 packWords);
extern "C" __global__ void k_windows(const DevIndex *X, DevBatch B, u8 *scratch, u32 capW, u32 capBlocks, u32 mode, u32 lightEst, u32 useMid, u32 hashBits);
extern "C" __global__ void k_windows_big(const DevIndex *X, DevBatch B, u8 *scratch, u32 capW, u32 capBlocks, u32 lightEst, u32 useMid);
extern "C" __global__ void k_order_hist(DevBatch B);
extern "C" __global__ void k_order_offsets(DevBatch B);
extern "C" __global__ void k_order_scatter(DevBatch B);
extern "C" __global__ void k_stitch_win(const DevIndex *X, DevBatch B, u8 *bigArena, u32 capDepth, u32 capRank, u32 arenaBytes, u32 bigArenaBytes, u32 ldsWords, u32 mode, u32 pruneEnable);
extern "C" __global__ void k_stitch_lane(const DevIndex *X, DevBatch B, u8 *laneArena, u32 laneArenaBytes, u32 ldsWords, u32 pruneEnable, u32 maxClass);
extern "C" __global__ void k_stitch_replay(const DevIndex *X, DevBatch B, u8 *bigArena, u32 capDepth, u32 capRank, u32 arenaBytes, u32 bigArenaBytes, u32 ldsWords);
extern "C" __global__ void k_stitch_verify(const DevIndex *X, DevBatch B);
extern "C" __global__ void k_stitch_finish(const DevIndex *X, DevBatch B);
extern "C" __global__ void k_scan_local(DevBatch B, u32 *trBase, u32 *exBase, u32 *blockTot);
extern "C" __global__ void k_scan_offsets(DevBatch B, u32 *blockTot, u32 nBlocks, u32 *totals);
extern "C" __global__ void k_gather(DevBatch B, const u32 *trBase, const u32 *exBase, const u32 *blockBase, staramd_read_result *outReads,
                                    staramd_transcript *outTr, u32 outTrCap, staramd_exon *outEx, u32 outExCap);

// per-lane / per-wave work-space sizes (same formulas as the kernels)
static inline u32 waRowsH(u32 capDepth) { return capDepth == 0 ? (u32)WA_MAX : std::min<u32>(capDepth - 1u, (u32)WA_MAX); }
static inline u32 stitchStateBytesH(u32 capDepth, u32 capRank, u32 arenaBytes) {
    u32 b = capDepth * 112u + 2u * STARAMD_MAX_N_EXONS * 32u + ((capRank * 2u + 31u) & ~31u) + waRowsH(capDepth) * 32u + 96u + arenaBytes;
    return (b + 127u) & ~127u;
}
static inline u64 winWaveBytesH(u32 capW, u32 capBlocks, u32 big) {
    u64 b = (u64)capBlocks * WA_MAX * sizeof(DWA);
    if (big) b += (u64)capW * 8 * sizeof(u32) + 4096 / 8;
    return (b + 255) & ~255ull;
}

static thread_local std::string g_err;
extern "C" const char *staramd_last_error(void) { return g_err.c_str(); }

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); return STARAMD_ERR_DEVICE; } } while (0)


struct staramd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    DevIndex X;                 // host copy
    DevIndex *dX = nullptr;     // device copy read by the kernels through scalar loads
    std::vector<void *> indexAllocs, workAllocs;
    u32 maxReads = 0; u64 maxBases = 0;
    DevBatch B;
    u8 *dBases = nullptr; u64 *dReadOffset = nullptr; u16 *dMate1 = nullptr, *dMM = nullptr;
    // two sets of input buffers + a copy stream: the batch that follows is uploaded while this one is on the device (staramd_prefetch_batch, include/star_amd_async.h).
    // in[k].pending: the set holds an uploaded batch that staramd_map_batch has not consumed yet; cur: the set the last mapped batch used (dBases ... above point into it)
    struct InSet { u8 *bases = nullptr; u64 *readOffset = nullptr; u16 *mate1 = nullptr, *mm = nullptr; hipEvent_t up = nullptr;
                   const uint8_t *hBases = nullptr; const uint64_t *hReadOffset = nullptr; u32 nReads = 0; bool pending = false; } in[2];
    // staramd_map_begin / staramd_map_end (include/star_amd_async.h): a batch whose kernels are enqueued; its flags and totals looked at; the copy of the results of the
    // batch before still running on the copy stream (the next k_gather waits for it)
    bool inFlight = false, collected = false, downloadPending = false; hipEvent_t evDownload = nullptr; u64 nOverlapped = 0; staramd_results msRes = {};
    u64 nPrefetchHits = 0;      // staramd_map_batch calls that found their upload done ahead (staramd_prefetch_hits)
    int cur = 0; hipStream_t copyStream = nullptr;
    u32 *dPacked = nullptr; u32 packWordsCap = 0;
    int nCU = 256;
    // seed kernel: one lane per read
    u32 seedLanes = 0; DSeed *scrSeed = nullptr; u32 seedPerLane = 0;
    SeedWork seedWork = {}; u32 seedUnits = 1, seedUnitLanes = 0;      // lane = unit mapping of the seed stage (STARAMD_SEED_UNITS=0: lane = read, k_seed_search over every read)
    // window kernel: one wave per read; fast pass (table in LDS) + big pass (reference limits, table in global memory)
    // the batch whose results did not fit the caller's arrays (STARAMD_ERR_RESULT_OVERFLOW): they stay resident; the same batch handed in again is copied out, not mapped again
    const void *ovfBases = nullptr, *ovfOffsets = nullptr; u32 ovfReads = 0; u64 ovfMark = 0; float ovfMs[4] = {0, 0, 0, 0};
    bool residentInsertKeepsKeys = false;          // staramd_insert_junctions_fits said yes with the keys resident
    u64 *sakBuf = nullptr; u64 sakCapBytes = 0;   // the allocation behind DevIndex::SAK (kept across a junction insertion: freeing and allocating 100 GB costs seconds)
    u64 nLaunches = 0;                    // times the kernels of a batch were enqueued (staramd_launch_count: tests)
    u32 winBlocks = 0, winBlocksBig = 0; u8 *scrWin = nullptr, *scrWinBig = nullptr; u32 capW = 0, capBlocks = 0, capWBig = 0, capBlocksBig = 0;
    // middle pass of k_windows: the few reads with more windows than the first pass has LDS rows for get a larger LDS table, one wavefront per block
    u32 winBlocksMid = 0, capWMid = 0, capBlocksMid = 0, hashBitsMid = 65536; u8 *scrWinMid = nullptr;
    u32 hashBits = 4096, winOwnerMap = 1;         // first launch: bits of the covered-bins filter = 32 x slots of the owner map (k_window.hip)
    // stitch kernel: one lane per read; fast pass (compact arena) + big pass (worst-case arena)
    u32 lightEst = 65536;                 // reads whose walk-size estimate is at most this are ONE stitch work item
    u32 stBlocks = 0, stBlocksBig = 0, replayBlocks = 0; u8 *scrStitch = nullptr, *scrStitchBig = nullptr;
    u32 capDepth = 0, capRank = 0, arenaFast = 0, arenaBig = 0, ldsWordsCap = 0;
    // lean pass-0 launch: windows of at most leanDepth-1 seeds (almost all) walked with a small LDS slice per wavefront, so that more
    // wavefronts are resident per CU; the others are handed to a second launch with the full-size slice (0 = one full-size launch)
    u32 leanDepth = 0, leanArena = 0, stBlocksLean = 0;
    // main cooperative launch behind the lane kernel: a walk stack of mainDepth frames (windows of up to mainDepth - 1 seeds: all but a handful) makes the wavefront's LDS
    // slice 10 KB instead of 12.5 = a FOURTH block per CU; what holds more seeds goes on to the full-depth launch
    u32 mainDepth = 0, stBlocksMain = 0;
    // lane-per-read stitcher (k_stitch_lane.hip): takes the light reads whose windows hold few seeds; the cooperative kernel gets the rest
    u32 laneBlocks = 0, laneArenaBytes = 0, laneClass = 3; u8 *scrLane = nullptr;
    u32 prune = 15;                       // STARAMD_PRUNE: bit 0 = window pruning, bit 1 = two-mate windows of a light read first, bit 2 = single-mate leaves of two-mate windows skipped (5.6),
                                          // bit 3 = pruning under resultSelect 2 as well: reads whose best alignment cannot be the main segment of a chimera (5.8)
    u32 ldsLimit = 65536;                 // dynamic LDS a block may ask for
    u32 kernelTurns = 0;                  // STARAMD_KERNEL_TURNS=1: the kernel phase of a batch is serialised over the contexts of a device (runDevice); off: measured no gain
    u32 *dTrBase = nullptr, *dExBase = nullptr, *dTotals = nullptr, *dBlockTot = nullptr;
    staramd_read_result *dOutReads = nullptr; staramd_transcript *dOutTr = nullptr; staramd_exon *dOutEx = nullptr;
    hipEvent_t ev[10];
    hipEvent_t evWait = nullptr;          // blocking-sync event: the mapper thread SLEEPS while its batch is on the device (hipStreamSynchronize spins on a core; the
                                          // front end runs one mapper thread per context beside its parser and formatter threads, on hosts with a CPU quota)
    float ms[8] = {0, 0, 0, 0, 0, 0, 0, 0}; float ms8 = 0;   // per-stage HIP-event times of the last batch (staramd_get_timings)
    u64 counters[DC_N];
    u32 residentReads = 0; u32 residentMaxLread = 0;
    u32 *hostScratch = nullptr;         // pinned: totals + cursors + counters read-back
    std::vector<u64> rebased;           // read offsets of a batch that does not start at base 0
    // a context created with staramd_create_shared maps against the resident index of its OWNER (same device): work space, stream and
    // events are its own, X / dX are copies of the owner's, refreshed whenever the owner's index changes
    staramd_ctx *owner = nullptr; std::vector<staramd_ctx *> sharers;
};
static void refreshSharers(staramd_ctx *c) { for (staramd_ctx *s : c->sharers) { s->X = c->X; s->dX = c->dX; } }
#define OWNER_ONLY(c) do { if ((c)->owner) { g_err = "this context shares the index of another one (staramd_create_shared): change the index through its owner"; return STARAMD_ERR_ARG; } } while (0)

static u32 envU32(const char *name, u32 dflt) { const char *s = getenv(name); return s ? (u32)strtoul(s, nullptr, 10) : dflt; }
// Streams.  The copies of this runtime are shader kernels (rocprofv3: __amd_rocclr_copyBuffer; no SDMA engine is used on this pool), so a copy that is to run BESIDE the
// persistent kernels of the next batch needs compute units those kernels do not hold: with STARAMD_COPY_CUS = k > 0 the copy stream of a context is confined to the last k
// CUs of the device and its kernel stream to the others (hipExtStreamCreateWithCUMask).  The kernels lose k / nCU of the chip; the result copy of batch i then completes
// ~2 ms after its kernels while batch i + 1 runs (staramd_map_end), instead of behind them.  0 (or a runtime that refuses the mask): plain streams.
static hipError_t makeStream(staramd_ctx *c, hipStream_t *s, bool copySide) {
    const u32 k = envU32("STARAMD_COPY_CUS", 0), n = (u32)c->nCU;
    if (k > 0 && k < n) {
        std::vector<uint32_t> mask((n + 31) / 32, 0u);
        for (u32 i = 0; i < n; i++) if ((i >= n - k) == copySide) mask[i >> 5] |= 1u << (i & 31u);
        if (hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data()) == hipSuccess) {
            if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "staramd: %s stream confined to %u of %u CUs\n", copySide ? "copy" : "kernel", copySide ? k : n - k, n);
            return hipSuccess;
        }
        (void)hipGetLastError();
        if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "staramd: hipExtStreamCreateWithCUMask refused, plain stream\n");
    }
    return hipStreamCreate(s);
}

template <class T> static int devAlloc(std::vector<void *> &reg, T **p, u64 n) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<u64>(n * sizeof(T), 16));
    if (e != hipSuccess) { g_err = std::string("hipMalloc(") + std::to_string(n * sizeof(T)) + " bytes): " + hipGetErrorString(e); return STARAMD_ERR_DEVICE; }
    reg.push_back(q); *p = (T *)q; return 0;
}
template <class T> static int devUpload(std::vector<void *> &reg, const T **dst, const T *src, u64 n, u64 padElems = 0) {
    T *q = nullptr;
    int rc = devAlloc(reg, &q, n + padElems); if (rc) return rc;
    if (padElems) { hipError_t e = hipMemset(q, 0, (n + padElems) * sizeof(T)); if (e != hipSuccess) { g_err = hipGetErrorString(e); return STARAMD_ERR_DEVICE; } }
    if (n) { hipError_t e = hipMemcpy(q, src, n * sizeof(T), hipMemcpyHostToDevice); if (e != hipSuccess) { g_err = std::string("hipMemcpy H2D: ") + hipGetErrorString(e); return STARAMD_ERR_DEVICE; } }
    *dst = q; return 0;
}

static void freeAll(std::vector<void *> &v) { for (void *p : v) (void)hipFree(p); v.clear(); }

// genomic-length score term int(ceil(log2(double(gLen))*scale-0.5)) (stitchWindowAligns.cpp:221-225) as integer
// break points: evaluated with the host's libm exactly as the reference does, the device only compares integers.
static void buildGlBreaks(DevIndex &X, double scale) {
    X.nBreak = 0; X.glScoreAt1 = 0; X.glStep = 0;
    if (scale == 0) return;
    auto f = [&](u64 g) { return (i32)std::ceil(std::log2((double)g) * scale - 0.5); };
    X.glScoreAt1 = f(1);
    X.glStep = scale < 0 ? -1 : +1;
    i32 cur = X.glScoreAt1; u64 g = 1; const u64 GMAX = 1ull << 40;
    while (X.nBreak < NBREAK_MAX) {
        // smallest g' > g with f(g') != cur (f is monotone)
        if (f(GMAX) == cur) break;
        u64 lo = g, hi = GMAX;              // f(lo)==cur, f(hi)!=cur
        while (hi - lo > 1) { u64 mid = lo + (hi - lo) / 2; if (f(mid) == cur) lo = mid; else hi = mid; }
        i32 nv = f(hi);
        i32 steps = std::abs(nv - cur);
        for (i32 k = 0; k < steps && X.nBreak < NBREAK_MAX; k++) X.glBreak[X.nBreak++] = hi;
        cur = nv; g = hi;
    }
}

// DevIndex::sjdbHash (dev.h): slots = a power of two >= 2 x junctions, >= 128 (the cooperative look-up probes 64 consecutive slots per step)
static int buildSjdbHash(staramd_ctx *c, const staramd_genome *g);
// chromosome / junction tables, index geometry, parameters and the DevIndex block (everything but G, SA, SAindex)
static int uploadTables(staramd_ctx *c, const staramd_genome *g, const staramd_params *p) {
    DevIndex &X = c->X;
    int rc;
    if ((rc = devUpload(c->indexAllocs, &X.chrBin, g->chrBin, g->chrBinN, 4))) return rc;
    if ((rc = devUpload(c->indexAllocs, &X.chrStart, g->chrStart, (u64)g->nChrReal + 1))) return rc;
    if ((rc = devUpload(c->indexAllocs, &X.chrLength, g->chrLength, (u64)g->nChrReal))) return rc;
    if ((rc = devUpload(c->indexAllocs, &X.sjDstart, g->sjDstart, (u64)g->sjdbN))) return rc;
    if ((rc = devUpload(c->indexAllocs, &X.sjAstart, g->sjAstart, (u64)g->sjdbN))) return rc;
    if ((rc = devUpload(c->indexAllocs, &X.sjdbStart, g->sjdbStart, (u64)g->sjdbN))) return rc;
    if ((rc = devUpload(c->indexAllocs, &X.sjdbEnd, g->sjdbEnd, (u64)g->sjdbN))) return rc;
    if ((rc = devUpload(c->indexAllocs, &X.sjdbMotif, g->sjdbMotif, (u64)g->sjdbN))) return rc;
    if ((rc = devUpload(c->indexAllocs, &X.sjdbShiftLeft, g->sjdbShiftLeft, (u64)g->sjdbN))) return rc;
    if ((rc = devUpload(c->indexAllocs, &X.sjdbShiftRight, g->sjdbShiftRight, (u64)g->sjdbN))) return rc;
    if ((rc = devUpload(c->indexAllocs, &X.sjdbStrand, g->sjdbStrand, (u64)g->sjdbN))) return rc;
    {
        std::vector<u32> info((size_t)g->sjdbN);
        for (u32 i = 0; i < g->sjdbN; i++) info[i] = SJ_INFO(g->sjdbMotif[i] & 7u, g->sjdbStrand[i] & 3u, g->sjdbShiftLeft[i], g->sjdbShiftRight[i]);
        if ((rc = devUpload(c->indexAllocs, &X.sjdbInfo, (const u32 *)info.data(), (u64)g->sjdbN))) return rc;
    }
    X.nGenome = g->nGenome; X.nSA = g->nSA; X.sjGstart = g->sjGstart;
    for (int i = 0; i < 17; i++) X.saiStart[i] = g->genomeSAindexStart[i];
    X.strandBit = g->GstrandBit; X.saBits = g->GstrandBit + 1; X.saiBits = g->GstrandBit + 3;
    X.saMask = X.saBits >= 64 ? ~0ull : ((1ull << X.saBits) - 1); X.saiMask = (1ull << X.saiBits) - 1;
    X.strandMask = ~(1ull << g->GstrandBit);               // Genome_genomeLoad.cpp:157
    X.saiNbit = 1ull << (g->GstrandBit + 1); X.saiAbsentBit = 1ull << (g->GstrandBit + 2);   // :161-166
    X.saiNbases = g->gSAindexNbases; X.sparseD = g->gSAsparseD;
    X.sjdbOverhang = g->sjdbOverhang; X.sjdbLength = g->sjdbLength ? g->sjdbLength : 1; X.sjdbN = g->sjdbN; X.nChrReal = g->nChrReal;
    X.P = *p;
    X.sjNovelStart = X.sjNovelEnd = nullptr; X.sjNovelN = 0;
    if ((rc = buildSjdbHash(c, g))) return rc;
    buildGlBreaks(X, p->scoreGenomicLengthLog2scale);
    { int rc2 = devAlloc(c->indexAllocs, &c->dX, (u64)1); if (rc2) return rc2; }
    HIPCHK(hipMemcpy(c->dX, &X, sizeof(DevIndex), hipMemcpyHostToDevice));
    return 0;
}


static int buildSjdbHash(staramd_ctx *c, const staramd_genome *g) {
    DevIndex &X = c->X;
    X.sjdbHash = nullptr; X.sjdbHashMask = 0; X.padHash = 0;
    if (g->sjdbN == 0 || g->sjdbN >= (1u << 23) || (g->nGenome >> SJH_START_BITS) != 0 || getenv("STARAMD_NO_SJDB_HASH")) return 0;
    u32 slots = 128; while (slots < 2u * g->sjdbN) slots <<= 1;
    std::vector<u64> tab((size_t)slots * 2, 0);
    const u32 mask = slots - 1;
    for (u32 i = 0; i < g->sjdbN; i++) {
        const u64 st = g->sjdbStart[i];
        u32 h = (u32)((st * 0x9E3779B97F4A7C15ull) >> 40) & mask;
        while (tab[2 * (size_t)h]) h = (h + 1) & mask;
        tab[2 * (size_t)h] = ((u64)(i + 1) << SJH_START_BITS) | st;
        tab[2 * (size_t)h + 1] = g->sjdbEnd[i] | ((u64)SJ_INFO(g->sjdbMotif[i] & 7u, g->sjdbStrand[i] & 3u, g->sjdbShiftLeft[i], g->sjdbShiftRight[i]) << SJH_START_BITS);
    }
    int rc = devUpload(c->indexAllocs, &X.sjdbHash, (const u64 *)tab.data(), (u64)slots * 2);
    if (!rc) X.sjdbHashMask = mask;
    return rc;
}

// The keys beside the suffix array (dev.h DevIndex::SAK, k_seed.hip): 16 bytes per suffix -- 99 GB for a human-size index, out of HBM that is otherwise empty.  Built on the
// device from the resident genome and suffix array (~1 s at 6.3e9 suffixes); skipped (the seed stage then probes the packed array and the genome, as before) when
// STARAMD_SA_KEYS=0, with a sparse suffix array, or when the array would not leave `reserve` bytes of HBM for the work space.
extern "C" __global__ void k_sak_build(const DevIndex *Xp, u64 *out, u64 n0, u64 n1);
// keepAllocation: the keys are void (the suffix array is about to change) but their memory stays for the ones that follow -- hipFree + hipMalloc of 100 GB between the
// passes of a 2-pass run cost 6.4 s (profiles/r06_two_pass_leg_*), the rebuild itself 0.46 s
static void dropSak(staramd_ctx *c, bool keepAllocation = false) {
    DevIndex &X = c->X;
    if (c->sakBuf && !keepAllocation) {
        for (size_t i = 0; i < c->indexAllocs.size(); i++) if (c->indexAllocs[i] == (void *)c->sakBuf) { (void)hipFree(c->indexAllocs[i]); c->indexAllocs.erase(c->indexAllocs.begin() + i); break; }
        c->sakBuf = nullptr; c->sakCapBytes = 0;
    }
    X.SAK = nullptr; X.sakBases = 0;
}
static int buildSak(staramd_ctx *c) {
    DevIndex &X = c->X;
    const u64 needNow = X.nSA * 16ull;
    dropSak(c, c->sakBuf && needNow <= c->sakCapBytes);
    HIPCHK(hipMemcpy(c->dX, &X, sizeof(DevIndex), hipMemcpyHostToDevice));
    if (!envU32("STARAMD_SA_KEYS", 1) || X.sparseD != 1 || X.saBits > 58 || X.saiNbases == 0 || X.nSA == 0) { dropSak(c); return 0; }
    const u64 need = X.nSA * 16ull;
    u64 *out = c->sakBuf;
    if (!out) {
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return 0;
        const u64 reserve = (u64)envU32("STARAMD_SA_KEYS_RESERVE_GB", 48) << 30;
        if ((u64)freeB < need + std::min<u64>(reserve, (u64)totalB / 4)) {
            if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "staramd: no keys beside the suffix array: %.1f GB needed, %.1f GB free\n", need / 1e9, freeB / 1e9);
            return 0;
        }
        // room for the suffixes a junction insertion adds (2 x junctions x sjdbLength: 0.4e9 for a million junctions of 2 x 100 bases), so that the rebuild behind it fits
        u64 cap = need + std::min<u64>(std::max<u64>(need / 8, 64ull << 20), 8ull << 30);
        if ((u64)freeB < cap + std::min<u64>(reserve, (u64)totalB / 4)) cap = need;
        if (hipMalloc((void **)&out, cap) != hipSuccess) { (void)hipGetLastError(); return 0; }
        c->indexAllocs.push_back(out); c->sakBuf = out; c->sakCapBytes = cap;
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, 0);
    const u64 chunk = 1ull << 30;                      // (grids of at most 2^30 lanes)
    for (u64 n0 = 0; n0 < X.nSA; n0 += chunk) {
        const u64 n1 = std::min<u64>(X.nSA, n0 + chunk);
        hipLaunchKernelGGL(k_sak_build, dim3((u32)((n1 - n0 + 255) / 256)), dim3(256), 0, 0, (const DevIndex *)c->dX, out, n0, n1);
    }
    (void)hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { dropSak(c); g_err = "k_sak_build failed"; return STARAMD_ERR_DEVICE; }
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    X.SAK = out; X.sakBases = X.saiNbases;
    HIPCHK(hipMemcpy(c->dX, &X, sizeof(DevIndex), hipMemcpyHostToDevice));
    if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "staramd: keys beside the suffix array: %.1f GB, built in %.0f ms\n", need / 1e9, ms);
    return 0;
}

static int uploadIndex(staramd_ctx *c, const staramd_genome *g, const staramd_params *p) {
    DevIndex &X = c->X;
    memset(&X, 0, sizeof(X));
    c->sakBuf = nullptr; c->sakCapBytes = 0; c->residentInsertKeepsKeys = false;          // (a new index: whatever held the keys of the old one was freed with it)
    if (g->gSAsparseD < 1 || g->gSAsparseD > 8) { g_err = "genomeSAsparseD must be in 1..8"; return STARAMD_ERR_ARG; }
    if (g->gSAindexNbases > 16 || g->GstrandBit + 3 > 63) { g_err = "unsupported index geometry"; return STARAMD_ERR_ARG; }
    if (p->seedPerWindowNmax > WA_MAX || p->seedPerWindowNmax < 1) { g_err = "seedPerWindowNmax must be in 1..64 on the device (one lane per window seed)"; return STARAMD_ERR_ARG; }
    if (p->alignTranscriptsPerWindowNmax > 2000 || p->alignTranscriptsPerWindowNmax < 1) { g_err = "alignTranscriptsPerWindowNmax must be in 1..2000 on the device"; return STARAMD_ERR_ARG; }
    if (p->winAnchorMultimapNmax > 64) { /* anchors are enumerated in chunks of 64 loci: any value works */ }
    // genome with padding
    {
        u8 *dG = nullptr;
        int rc = devAlloc(c->indexAllocs, &dG, g->nGenome + 2 * GPAD); if (rc) return rc;
        HIPCHK(hipMemset(dG, 5, g->nGenome + 2 * GPAD));
        HIPCHK(hipMemcpy(dG + GPAD, g->G, g->nGenome, hipMemcpyHostToDevice));
        X.G = dG + GPAD;
    }
    {
        u64 nW = (g->nSAbyte + 7) / 8 + 2; u64 *d = nullptr;
        int rc = devAlloc(c->indexAllocs, &d, nW); if (rc) return rc;
        HIPCHK(hipMemset(d, 0, nW * 8)); HIPCHK(hipMemcpy(d, g->SA, g->nSAbyte, hipMemcpyHostToDevice)); X.SA = d;
        nW = (g->nSAibyte + 7) / 8 + 2;
        rc = devAlloc(c->indexAllocs, &d, nW); if (rc) return rc;
        HIPCHK(hipMemset(d, 0, nW * 8)); HIPCHK(hipMemcpy(d, g->SAi, g->nSAibyte, hipMemcpyHostToDevice)); X.SAi = d;
    }
    const int rc = uploadTables(c, g, p);
    return rc ? rc : buildSak(c);
}


template <class T> static int devRealloc(std::vector<void *> &reg, T **p, u64 n) {
    for (size_t i = 0; i < reg.size(); i++) if (reg[i] == (void *)*p) { (void)hipFree(reg[i]); reg.erase(reg.begin() + i); break; }
    *p = nullptr;
    return devAlloc(reg, p, n);
}

static int allocWork(staramd_ctx *c) {
    std::vector<void *> &R = c->workAllocs;
    u32 N = c->maxReads; int rc;
    hipDeviceProp_t prop; memset(&prop, 0, sizeof(prop));
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->nCU = prop.multiProcessorCount;
    // 64 bytes of padding either side: the seed search compares 8 bases per step and may read a few bytes past a read
    { u8 *raw = nullptr; if ((rc = devAlloc(R, &raw, c->maxBases + 192))) return rc; if (hipMemset(raw, 4, c->maxBases + 192) != hipSuccess) { g_err = "hipMemset failed"; return STARAMD_ERR_DEVICE; } c->dBases = raw + 64; }
    if ((rc = devAlloc(R, &c->dReadOffset, (u64)N + 1))) return rc;
    if ((rc = devAlloc(R, &c->dMate1, (u64)N))) return rc;
    if ((rc = devAlloc(R, &c->dMM, (u64)N))) return rc;
    c->in[0].bases = c->dBases; c->in[0].readOffset = c->dReadOffset; c->in[0].mate1 = c->dMate1; c->in[0].mm = c->dMM; c->cur = 0;
    if (envU32("STARAMD_PREFETCH", 1)) {
        u8 *raw = nullptr; if ((rc = devAlloc(R, &raw, c->maxBases + 192))) return rc; if (hipMemset(raw, 4, c->maxBases + 192) != hipSuccess) { g_err = "hipMemset failed"; return STARAMD_ERR_DEVICE; } c->in[1].bases = raw + 64;
        if ((rc = devAlloc(R, &c->in[1].readOffset, (u64)N + 1))) return rc;
        if ((rc = devAlloc(R, &c->in[1].mate1, (u64)N))) return rc;
        if ((rc = devAlloc(R, &c->in[1].mm, (u64)N))) return rc;
        if (makeStream(c, &c->copyStream, true) != hipSuccess) { g_err = "hipStreamCreate failed"; return STARAMD_ERR_DEVICE; }
        for (int k = 0; k < 2; k++) if (hipEventCreateWithFlags(&c->in[k].up, hipEventDisableTiming) != hipSuccess) { g_err = "hipEventCreate failed"; return STARAMD_ERR_DEVICE; }
    }
    c->packWordsCap = 0;
    DevBatch &B = c->B; memset(&B, 0, sizeof(B));
    B.bases = c->dBases; B.readOffset = c->dReadOffset; B.mate1Length = c->dMate1; B.mmMaxTotal = c->dMM;
    if ((rc = devAlloc(R, &B.reads, (u64)N))) return rc;
    const u64 slack = envU32("STARAMD_POOL_SLACK", 65536);
    B.seedCap = (u32)std::min<u64>((u64)N * envU32("STARAMD_SEEDS_PER_READ", 32) + slack, 0xFFFFFFF0ull);
    B.winCap = (u32)std::min<u64>((u64)N * envU32("STARAMD_WINDOWS_PER_READ", 24) + slack, 0xFFFFFFF0ull);
    B.waCap = (u32)std::min<u64>((u64)N * envU32("STARAMD_WA_PER_READ", 96) + slack, 0xFFFFFFF0ull);
    B.trCap = (u32)std::min<u64>((u64)N * envU32("STARAMD_TR_PER_READ", 48) + slack, 0xFFFFFFF0ull);
    B.exCap = (u32)std::min<u64>((u64)B.trCap * 3, 0xFFFFFFF0ull);
    if ((rc = devAlloc(R, &B.seedPool, (u64)B.seedCap))) return rc;
    if ((rc = devAlloc(R, &B.winPool, (u64)B.winCap))) return rc;
    if ((rc = devAlloc(R, &B.waPool, (u64)B.waCap))) return rc;
    if ((rc = devAlloc(R, &B.wout, (u64)B.winCap))) return rc;
    if ((rc = devAlloc(R, &B.items, (u64)B.winCap))) return rc;
    if ((rc = devAlloc(R, &B.itemClass, (u64)B.winCap))) return rc;
    if ((rc = devAlloc(R, &B.order, (u64)B.winCap + 64))) return rc;
    if ((rc = devAlloc(R, &B.redoList, (u64)B.winCap))) return rc;
    if ((rc = devAlloc(R, &B.replayList, (u64)B.winCap))) return rc;
    if ((rc = devAlloc(R, &B.heavyList, (u64)B.winCap))) return rc;
    if ((rc = devAlloc(R, &B.heavyList2, (u64)B.winCap))) return rc;
    if ((rc = devAlloc(R, &B.trPool, (u64)B.trCap))) return rc;
    if ((rc = devAlloc(R, &B.exPool, (u64)B.exCap))) return rc;
    if ((rc = devAlloc(R, &B.costHist, (u64)64))) return rc;
    if ((rc = devAlloc(R, &B.ovfWin, (u64)N))) return rc;
    if ((rc = devAlloc(R, &B.ovfWin2, (u64)N))) return rc;
    if ((rc = devAlloc(R, &B.cursors, (u64)CUR_N))) return rc;
    if ((rc = devAlloc(R, &B.counters, (u64)DC_N))) return rc;
    if ((rc = devAlloc(R, &c->dTrBase, (u64)N))) return rc;
    if ((rc = devAlloc(R, &c->dExBase, (u64)N))) return rc;
    if ((rc = devAlloc(R, &c->dTotals, (u64)4))) return rc;
    if ((rc = devAlloc(R, &c->dBlockTot, (u64)2 * ((N + 255) / 256) + 2))) return rc;
    if ((rc = devAlloc(R, &c->dOutReads, (u64)N))) return rc;
    if ((rc = devAlloc(R, &c->dOutTr, (u64)B.trCap))) return rc;
    if ((rc = devAlloc(R, &c->dOutEx, (u64)B.exCap))) return rc;
    if (hipHostMalloc((void **)&c->hostScratch, (64 + CUR_N) * sizeof(u32) + DC_N * sizeof(u64)) != hipSuccess) { g_err = "hipHostMalloc failed"; return STARAMD_ERR_DEVICE; }
    const staramd_params &P = c->X.P;
    // ---- seed kernel: one lane per read, PC table per lane sized by the reference's seedPerReadNmax
    int seedPerCU = 2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&seedPerCU, k_seed_search, 256, 0) != hipSuccess || seedPerCU < 1) seedPerCU = 2;
    u32 lanes = envU32("STARAMD_SEED_LANES", (u32)c->nCU * (u32)seedPerCU * 256u);
    lanes = std::max<u32>(256, std::min<u32>(lanes, ((N + 255) / 256) * 256));
    c->seedLanes = (lanes / 256) * 256;
    c->seedPerLane = P.seedPerReadNmax + 1;
    if ((rc = devAlloc(R, &c->scrSeed, (u64)c->seedLanes * c->seedPerLane))) return rc;
    // 0: a lane per read (k_seed_search over every read); 1: a lane per unit of the search schedule (k_seed_plan / k_seed_units / k_seed_merge)
    c->seedUnits = envU32("STARAMD_SEED_UNITS", 1);
    if (c->seedUnits) {
        // 2x101 has 12 groups / 10 units per pair, 2x150 16 / 14; a read that does not fit what is left of the pools takes k_seed_search (no regrowth, no re-run)
        SeedWork &W = c->seedWork;
        W.groupCap = (u32)std::min<u64>((u64)N * envU32("STARAMD_SEED_GROUPS_PER_READ", 20) + 256, 0xFFFFFFF0ull);
        W.unitCap = (u32)std::min<u64>((u64)N * envU32("STARAMD_SEED_GROUPS_PER_READ", 20) + 256, 0xFFFFFFF0ull);
        W.slotLimit = std::min<u32>(SEED_SLOTS, std::max<u32>(1, envU32("STARAMD_SEED_SLOT_LIMIT", SEED_SLOTS)));
        if ((rc = devAlloc(R, &W.units, (u64)W.unitCap))) return rc;
        if ((rc = devAlloc(R, &W.slots, (u64)W.groupCap * SEED_SLOTS))) return rc;
        if ((rc = devAlloc(R, &W.groupHead, (u64)W.groupCap))) return rc;
        if ((rc = devAlloc(R, &W.plan, (u64)N))) return rc;
        if ((rc = devAlloc(R, &W.handOn, (u64)N))) return rc;
        int upCU = seedPerCU;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&upCU, k_seed_units, 256, 0) != hipSuccess || upCU < 1) upCU = seedPerCU;
        c->seedUnitLanes = envU32("STARAMD_SEED_UNIT_LANES", (u32)c->nCU * (u32)upCU * 256u) / 256u * 256u;
        if (c->seedUnitLanes < 256) c->seedUnitLanes = 256;
    }
    // ---- window kernel
    c->lightEst = envU32("STARAMD_LIGHT_EST", 65536);
    c->prune = envU32("STARAMD_PRUNE", 15); c->kernelTurns = envU32("STARAMD_KERNEL_TURNS", 0); c->laneClass = envU32("STARAMD_LANE_CLASS", 0);          // (knobs are read here, once: not on the launch path)
    if (prop.sharedMemPerBlock >= 16384) c->ldsLimit = (u32)std::min<size_t>(prop.sharedMemPerBlock, 65536);
    // first launch: 128 table rows + 512 owner-map slots = 6 KB of LDS per wavefront, 6 blocks of 4 wavefronts per CU (k_windows is held to 6 waves per SIMD)
    c->capW = envU32("STARAMD_CAP_WINDOWS", 128); c->capBlocks = envU32("STARAMD_CAP_WA_BLOCKS", 128);
    int winPerCU = 3;
    c->winOwnerMap = envU32("STARAMD_WIN_OWNER_MAP", 1);
    { u32 hb = envU32("STARAMD_WIN_HASH_BITS", c->winOwnerMap ? 16384 : 4096); c->hashBits = 1024; while (c->hashBits < hb && c->hashBits < (1u << 18)) c->hashBits <<= 1; }      // a power of two
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&winPerCU, k_windows, 256, 4 * (c->capW * 8 + c->hashBits / 32) * sizeof(u32)) != hipSuccess || winPerCU < 1) winPerCU = 3;
    c->winBlocks = (u32)c->nCU * envU32("STARAMD_WIN_BLOCKS_PER_CU", (u32)winPerCU);
    c->winBlocks = std::max<u32>(1, std::min<u32>(c->winBlocks, (N + 3) / 4));
    if ((rc = devAlloc(R, &c->scrWin, (u64)c->winBlocks * 4 * winWaveBytesH(c->capW, c->capBlocks, 0)))) return rc;
    c->capWMid = envU32("STARAMD_CAP_WINDOWS_MID", 1024); c->capBlocksMid = envU32("STARAMD_CAP_WA_BLOCKS_MID", 1024);
    if (c->capWMid <= c->capW || c->capWMid >= P.alignWindowsPerReadNmax) c->capWMid = 0;
    if (c->capWMid) {
        u32 hb = envU32("STARAMD_WIN_HASH_BITS_MID", c->winOwnerMap ? 262144 : 65536);
        c->hashBitsMid = 4096; while (c->hashBitsMid < hb && c->hashBitsMid < (1u << 18)) c->hashBitsMid <<= 1;       // a power of two
        while (c->hashBitsMid > 4096 && ((u64)c->capWMid * 8 + c->hashBitsMid / 32) * 4 > 65536) c->hashBitsMid >>= 1;   // table + map within 64 KB of dynamic LDS
        if (((u64)c->capWMid * 8 + c->hashBitsMid / 32) * 4 > 65536) c->capWMid = (65536 / 4 - c->hashBitsMid / 32) / 8;
        c->winBlocksMid = envU32("STARAMD_WIN_BLOCKS_MID", (u32)c->nCU * 4u);
        if ((rc = devAlloc(R, &c->scrWinMid, (u64)c->winBlocksMid * winWaveBytesH(c->capWMid, c->capBlocksMid, 0)))) return rc;
    }
    c->capWBig = P.alignWindowsPerReadNmax; c->capBlocksBig = P.alignWindowsPerReadNmax;
    c->winBlocksBig = envU32("STARAMD_WIN_BLOCKS_BIG", 64);
    if ((rc = devAlloc(R, &c->scrWinBig, (u64)c->winBlocksBig * 4 * winWaveBytesH(c->capWBig, c->capBlocksBig, 1)))) return rc;
    // ---- stitch kernel
    c->capDepth = P.seedPerWindowNmax + 1; c->capRank = P.alignTranscriptsPerWindowNmax + 1;
    c->arenaFast = envU32("STARAMD_STITCH_ARENA", 6144) & ~31u;
    c->arenaBig = 2u * (P.alignTranscriptsPerWindowNmax + 2) * (96u + 32u * STARAMD_MAX_N_EXONS);      // twice the largest live set
    if (c->arenaBig > 2000000u) { g_err = "alignTranscriptsPerWindowNmax too large for the device record arena"; return STARAMD_ERR_ARG; }
    // one wavefront per window, walk state in LDS; blocks of 4 wavefronts; one worst-case record arena per wavefront in HBM
    c->arenaFast = envU32("STARAMD_STITCH_ARENA", 3584) & ~31u;
    int stPerCU = 2;
    size_t ldsFast = 4 * (size_t)(stitchStateBytesH(c->capDepth, c->capRank, c->arenaFast) + 27 * 4 + 16);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&stPerCU, k_stitch_win, 256, ldsFast) != hipSuccess || stPerCU < 1) stPerCU = 2;
    c->stBlocks = (u32)c->nCU * envU32("STARAMD_STITCH_BLOCKS_PER_CU", (u32)stPerCU);
    if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "staramd: k_stitch_win %d blocks/CU (LDS %zu B/block), k_windows %d blocks/CU, k_seed_search %d blocks/CU\n", stPerCU, ldsFast, winPerCU, seedPerCU);
    // the replay kernel needs no walk stack and no read in LDS: more blocks per CU hide the latency of its candidate-log reads
    {
        int rpPerCU = stPerCU;
        size_t ldsReplay = 4 * (size_t)stitchStateBytesH(0, c->capRank, c->arenaFast);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&rpPerCU, k_stitch_replay, 256, ldsReplay) != hipSuccess || rpPerCU < 1) rpPerCU = stPerCU;
        c->replayBlocks = (u32)c->nCU * envU32("STARAMD_REPLAY_BLOCKS_PER_CU", (u32)std::min(rpPerCU, 8));
        if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "staramd: k_stitch_replay %d blocks/CU (LDS %zu B/block)\n", rpPerCU, ldsReplay);
    }
    // lean launch geometry (used when the lane kernel is off or does not fit: with the lane kernel in front a lean middle launch -- 20 frames, 7.6 KB per wavefront, 5 wavefronts
    // per SIMD -- was measured and removed in round 5: stitch stage 24.6 ms against 22.7, profiles/r05_ab_session1_*.txt)
    c->leanDepth = envU32("STARAMD_LEAN_DEPTH", 9); c->leanArena = envU32("STARAMD_LEAN_ARENA", 2048) & ~31u;
    if (c->leanDepth >= c->capDepth) c->leanDepth = 0;
    c->mainDepth = envU32("STARAMD_MAIN_DEPTH", 33);
    if (c->mainDepth < 3 || c->mainDepth >= c->capDepth) c->mainDepth = 0;
    if (c->mainDepth) {
        int mpCU = stPerCU;
        size_t ldsMain = 4 * (size_t)(stitchStateBytesH(c->mainDepth, c->capRank, c->arenaFast) + 27 * 4 + 16);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&mpCU, k_stitch_win, 256, ldsMain) != hipSuccess || mpCU < 1) mpCU = stPerCU;
        c->stBlocksMain = (u32)c->nCU * envU32("STARAMD_MAIN_BLOCKS_PER_CU", (u32)mpCU);
        if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "staramd: k_stitch_win main launch: depth %u, %d blocks/CU (LDS %zu B/block)\n", c->mainDepth, mpCU, ldsMain);
    }
    c->stBlocksLean = 0;
    if (c->leanDepth) {
        int lpCU = stPerCU;
        size_t ldsLean = 4 * (size_t)(stitchStateBytesH(c->leanDepth, c->capRank, c->leanArena) + 27 * 4 + 16);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&lpCU, k_stitch_win, 256, ldsLean) != hipSuccess || lpCU < 1) lpCU = stPerCU;
        c->stBlocksLean = (u32)c->nCU * envU32("STARAMD_LEAN_BLOCKS_PER_CU", (u32)lpCU);
        if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "staramd: k_stitch_win lean launch: depth %u, arena %u B, %d blocks/CU (LDS %zu B/block)\n", c->leanDepth, c->leanArena, lpCU, ldsLean);
    }
    // lane-per-read launch: 256 lanes per block, each with an LDS slot for its packed read and a record arena in HBM
    c->laneBlocks = 0;
    if (envU32("STARAMD_LANE", 1)) {
        int lnPerCU = 4;
        const size_t ldsLane = 256 * (size_t)(27 * 4);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&lnPerCU, k_stitch_lane, 256, ldsLane) != hipSuccess || lnPerCU < 1) lnPerCU = 2;
        c->laneBlocks = (u32)c->nCU * envU32("STARAMD_LANE_BLOCKS_PER_CU", (u32)lnPerCU);
        c->laneBlocks = std::max<u32>(1, std::min<u32>(c->laneBlocks, (N + 255) / 256));
        c->laneArenaBytes = envU32("STARAMD_LANE_ARENA", 2048) & ~31u;
        if ((rc = devAlloc(R, &c->scrLane, (u64)c->laneBlocks * 256 * c->laneArenaBytes))) return rc;
        if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "staramd: k_stitch_lane %d blocks/CU (LDS %zu B/block), %u blocks, %u B of record arena per lane\n", lnPerCU, ldsLane, c->laneBlocks, c->laneArenaBytes);
    }
    const u32 maxStBlocks = std::max(std::max(std::max(c->stBlocks, c->stBlocksMain), c->stBlocksLean), c->replayBlocks);
    if ((rc = devAlloc(R, &c->scrStitchBig, (u64)maxStBlocks * 4 * c->arenaBig))) return rc;
    // candidate logs: one private region per wavefront; sized so that a wavefront's share of a full batch fits
    B.candWaveBytes = ((u64)envU32("STARAMD_CAND_KB_PER_WAVE", 0) * 1024) & ~31ull;
    if (B.candWaveBytes == 0) { u64 per = (u64)N * 6144 / ((u64)c->stBlocks * 4) + 262144; B.candWaveBytes = std::min<u64>(per, 0xFFFF0000ull) & ~31ull; }
    if ((rc = devAlloc(R, &B.candPool, (u64)std::max(std::max(c->stBlocks, c->stBlocksMain), c->stBlocksLean) * 4 * B.candWaveBytes))) return rc;
    if ((rc = devAlloc(R, &B.candTops, (u64)maxStBlocks * 4 + 64))) return rc;
    return 0;
}

extern "C" int staramd_create(staramd_ctx **out, int device, const staramd_genome *g, const staramd_params *p, uint32_t maxBatchReads, uint64_t maxBatchBases) {
    if (!out || !g || !p || maxBatchReads == 0) { g_err = "bad arguments"; return STARAMD_ERR_ARG; }
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev == 0) { g_err = "no HIP device visible: the seed-search-and-stitch engine needs an MI355X (no CPU fallback)"; return STARAMD_ERR_DEVICE; }
    HIPCHK(hipSetDevice(device));
    staramd_ctx *c = new staramd_ctx();
    c->device = device; c->maxReads = maxBatchReads; c->maxBases = maxBatchBases ? maxBatchBases : (u64)maxBatchReads * (STARAMD_READ_LEN_MAX + 1);
    int rc = uploadIndex(c, g, p);
    if (!rc) rc = allocWork(c);
    if (!rc) { if (makeStream(c, &c->stream, false) != hipSuccess) { g_err = "hipStreamCreate failed"; rc = STARAMD_ERR_DEVICE; } }
    if (!rc) for (int i = 0; i < 10; i++) if (hipEventCreate(&c->ev[i]) != hipSuccess) { g_err = "hipEventCreate failed"; rc = STARAMD_ERR_DEVICE; }
    if (!rc && !getenv("STARAMD_SPIN_WAIT") && hipEventCreateWithFlags(&c->evWait, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) c->evWait = nullptr;
    if (!rc && hipEventCreateWithFlags(&c->evDownload, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { g_err = "hipEventCreate failed"; rc = STARAMD_ERR_DEVICE; }
    if (rc) { freeAll(c->indexAllocs); freeAll(c->workAllocs); delete c; return rc; }
    memset(c->counters, 0, sizeof(c->counters));
    *out = c;
    return STARAMD_OK;
}

extern "C" int staramd_create_shared(staramd_ctx **out, staramd_ctx *owner, uint32_t maxBatchReads, uint64_t maxBatchBases) {
    if (!out || !owner || maxBatchReads == 0) { g_err = "bad arguments"; return STARAMD_ERR_ARG; }
    if (owner->owner) owner = owner->owner;
    HIPCHK(hipSetDevice(owner->device));
    staramd_ctx *c = new staramd_ctx();
    c->device = owner->device; c->maxReads = maxBatchReads; c->maxBases = maxBatchBases ? maxBatchBases : (u64)maxBatchReads * (STARAMD_READ_LEN_MAX + 1);
    c->owner = owner; c->X = owner->X; c->dX = owner->dX;
    int rc = allocWork(c);
    if (!rc) { if (makeStream(c, &c->stream, false) != hipSuccess) { g_err = "hipStreamCreate failed"; rc = STARAMD_ERR_DEVICE; } }
    if (!rc) for (int i = 0; i < 10; i++) if (hipEventCreate(&c->ev[i]) != hipSuccess) { g_err = "hipEventCreate failed"; rc = STARAMD_ERR_DEVICE; }
    if (!rc && !getenv("STARAMD_SPIN_WAIT") && hipEventCreateWithFlags(&c->evWait, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) c->evWait = nullptr;
    if (!rc && hipEventCreateWithFlags(&c->evDownload, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { g_err = "hipEventCreate failed"; rc = STARAMD_ERR_DEVICE; }
    if (rc) { freeAll(c->workAllocs); delete c; return rc; }
    memset(c->counters, 0, sizeof(c->counters));
    owner->sharers.push_back(c);
    *out = c;
    return STARAMD_OK;
}

// page-locked host memory for the caller's batch / result arrays (include/star_amd.h: the copies of staramd_map_batch then run as DMA
// transfers straight from / into them instead of being staged through the runtime's own pinned buffer)
extern "C" void *staramd_pinned_alloc(uint64_t bytes) { void *p = nullptr; if (hipHostMalloc(&p, bytes ? bytes : 1) != hipSuccess) { g_err = "hipHostMalloc failed"; return nullptr; } return p; }
extern "C" void staramd_pinned_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int staramd_update_index(staramd_ctx *c, const staramd_genome *g, const staramd_params *p) {
    if (!c) { g_err = "null context"; return STARAMD_ERR_ARG; }
    OWNER_ONLY(c);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    freeAll(c->indexAllocs);
    int rc = uploadIndex(c, g, p);
    refreshSharers(c);
    return rc;
}


// ---- junction insertion into the resident index (include/star_amd.h) ---------------------------------------------------------------
static void dropAlloc(std::vector<void *> &reg, const void *p) {
    for (size_t i = 0; i < reg.size(); i++) if (reg[i] == p) { (void)hipFree(reg[i]); reg.erase(reg.begin() + i); return; }
}

extern "C" int staramd_insert_junctions(staramd_ctx *c, const staramd_sjdb_args *a, uint8_t *SAout, uint64_t saOutCapacity, uint8_t *SAiOut, uint64_t saiOutCapacity,
                                        staramd_sjdb_result *res) {
    if (!c || !a || !res || !a->Gsj || !a->isOld || (a->oldSjdbN && !a->oldSJind)) { g_err = "staramd_insert_junctions: null argument"; return STARAMD_ERR_ARG; }
    if (a->sjdbN == 0 || a->sjdbLength < 3) { g_err = "staramd_insert_junctions: no junctions"; return STARAMD_ERR_ARG; }
    OWNER_ONLY(c);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    using namespace staridx;
    DevIndex &X = c->X;
    dropSak(c, c->residentInsertKeepsKeys);            // (the keys describe the old suffix array: rebuilt below -- in the same memory when staramd_insert_junctions_fits found room beside them)
    HipBackend be; be.s = c->stream;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, be.s);
    SjdbParams P; P.nGenomeOld = X.nGenome; P.nGenomeReal = a->nGenomeReal; P.nSAold = X.nSA; P.GstrandBit = X.strandBit;
    P.sjdbN = a->sjdbN; P.sjdbLength = a->sjdbLength; P.oldSjdbN = a->oldSjdbN; P.sjNew = a->sjNew; P.saIndexNbases = X.saiNbases;
    SjdbDeviceResult R; memset(&R, 0, sizeof(R));
    sjdbInsertDevice(be, P, X.G, X.SA, a->Gsj, a->isOld, a->oldSJind, X.saiStart, R, (u64)GPAD);
    memset(res, 0, sizeof(*res));
    res->nInd = R.nInd; res->nSAnew = R.nSAnew;
    res->nSAbyteNew = packedBytes(R.nSAnew, X.saBits); res->nSAibyte = packedBytes(X.saiStart[X.saiNbases], X.saiBits);
    int rc = STARAMD_OK;
    if (be.err != hipSuccess) { g_err = std::string("staramd_insert_junctions: ") + be.where + ": " + hipGetErrorString(be.err); rc = STARAMD_ERR_DEVICE; }
    else if (R.badFirstSuffix) { g_err = "staramd_insert_junctions: the first suffix has a non-ACGT base inside the SAindex prefix"; rc = STARAMD_ERR_ARG; }
    else if ((SAout && saOutCapacity < res->nSAbyteNew) || (SAiOut && saiOutCapacity < res->nSAibyte)) { g_err = "staramd_insert_junctions: output buffers too small"; rc = STARAMD_ERR_RESULT_OVERFLOW; }
    if (!rc) {
        if (SAout) be.copyToHost(SAout, (const u8 *)R.dSApacked, res->nSAbyteNew);
        if (SAiOut) be.copyToHost(SAiOut, (const u8 *)R.dSAiPacked, res->nSAibyte);
        if (be.err != hipSuccess) { g_err = std::string("staramd_insert_junctions: ") + be.where + ": " + hipGetErrorString(be.err); rc = STARAMD_ERR_DEVICE; }
    }
    (void)hipEventRecord(e1, be.s); (void)hipStreamSynchronize(be.s);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); res->msTotal = ms;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (be.tmp) (void)hipFree(be.tmp);
    if (rc) { if (R.dSApacked) (void)hipFree(R.dSApacked); if (R.dGnew) (void)hipFree(R.dGnew); if (R.dSAiPacked) (void)hipFree(R.dSAiPacked); (void)buildSak(c); refreshSharers(c); return rc; }
    // the new arrays take the place of the old ones
    dropAlloc(c->indexAllocs, X.G - GPAD); dropAlloc(c->indexAllocs, X.SA); dropAlloc(c->indexAllocs, X.SAi);
    c->indexAllocs.push_back(R.dGnew); c->indexAllocs.push_back(R.dSApacked); c->indexAllocs.push_back(R.dSAiPacked);
    X.G = R.dGnew + GPAD; X.SA = R.dSApacked; X.SAi = R.dSAiPacked;
    X.nGenome = R.nGenomeNew; X.nSA = R.nSAnew;
    HIPCHK(hipMemcpy(c->dX, &X, sizeof(DevIndex), hipMemcpyHostToDevice));
    { const int rcK = buildSak(c); if (rcK) return rcK; }
    refreshSharers(c);
    return STARAMD_OK;
}

// work space of sjdbInsertDevice (sjdb_core.h) beside the resident index: per new suffix ~7 words (offsets, positions, sort keys and permutations, double buffered);
// the new packed suffix array + genome + SAindex; and for the SAindex rebuild the text of both strands, one 64-bit word per suffix, one per SAindex entry
extern "C" int staramd_insert_junctions_fits(staramd_ctx *c, uint64_t maxJunctions, uint32_t sjdbLength) {
    if (!c || c->owner) return 0;
    if (hipSetDevice(c->device) != hipSuccess) return 0;
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return 0;
    const DevIndex &X = c->X;
    const u64 nInd = 2ull * maxJunctions * sjdbLength;
    const u64 nSAnew = X.nSA + nInd, nGnew = X.nGenome + maxJunctions * sjdbLength;
    const u64 nSAi = X.saiStart[X.saiNbases];
    u64 need = nInd * 8ull * 8ull                                               // per new suffix
             + (nSAnew * X.saBits + 7) / 8 + nGnew + 2 * GPAD + (nSAi * X.saiBits + 7) / 8      // the new arrays
             + 2 * nGnew + nSAnew * 8ull + nSAi * 8ull                          // SAindex rebuild
             + (2ull << 30);                                                    // sort temporaries, slack
    if (const char *e = getenv("STARAMD_SJDB_FITS_FREE_GB")) freeB = (size_t)(strtod(e, nullptr) * 1e9);      // (tests: pretend this much is free)
    if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "staramd: resident junction insertion needs up to %.1f GB, %.1f GB free\n", need / 1e9, freeB / 1e9);
    c->residentInsertKeepsKeys = (u64)freeB >= need;            // (free memory was asked for with the keys in place)
    return (u64)freeB >= need ? 1 : 0;
}

extern "C" int staramd_update_tables(staramd_ctx *c, const staramd_genome *g, const staramd_params *p) {
    if (!c || !g || !p) { g_err = "staramd_update_tables: null argument"; return STARAMD_ERR_ARG; }
    OWNER_ONLY(c);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    DevIndex &X = c->X;
    if (g->nGenome != X.nGenome || g->nSA != X.nSA) { g_err = "staramd_update_tables: the resident arrays belong to another index (nGenome / nSA differ)"; return STARAMD_ERR_ARG; }
    const void *keep[4] = {X.G - GPAD, X.SA, X.SAi, X.SAK};
    std::vector<void *> kept;
    for (void *q : c->indexAllocs) { if (q == keep[0] || q == keep[1] || q == keep[2] || (keep[3] && q == keep[3])) kept.push_back(q); else (void)hipFree(q); }
    c->indexAllocs.swap(kept);
    const int rc = uploadTables(c, g, p);          // (X.SAK / sakBases stay: same genome, same suffix array; uploadTables copies X to the device)
    refreshSharers(c);
    return rc;
}

extern "C" int staramd_set_novel_junctions(staramd_ctx *c, const uint64_t *start, const uint64_t *end, uint64_t n, uint32_t stage) {
    if (!c || (n && (!start || !end))) { g_err = "staramd_set_novel_junctions: null argument"; return STARAMD_ERR_ARG; }
    if (n > 0xFFFFFFF0ull) { g_err = "staramd_set_novel_junctions: too many junctions"; return STARAMD_ERR_ARG; }
    OWNER_ONLY(c);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    DevIndex &X = c->X; int rc;
    for (const u64 *old : {X.sjNovelStart, X.sjNovelEnd})
        for (size_t i = 0; old && i < c->indexAllocs.size(); i++) if (c->indexAllocs[i] == (void *)old) { (void)hipFree(c->indexAllocs[i]); c->indexAllocs.erase(c->indexAllocs.begin() + i); break; }
    X.sjNovelStart = X.sjNovelEnd = nullptr; X.sjNovelN = 0;
    if ((rc = devUpload(c->indexAllocs, &X.sjNovelStart, (const u64 *)start, (u64)n, 1))) return rc;
    if ((rc = devUpload(c->indexAllocs, &X.sjNovelEnd, (const u64 *)end, (u64)n, 1))) return rc;
    X.sjNovelN = n;
    X.P.outFilterBySJoutStage = (uint8_t)stage;
    HIPCHK(hipMemcpy(c->dX, &X, sizeof(DevIndex), hipMemcpyHostToDevice));
    refreshSharers(c);
    return 0;
}

extern "C" void staramd_destroy(staramd_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->owner) { auto &v = c->owner->sharers; for (size_t i = 0; i < v.size(); i++) if (v[i] == c) { v.erase(v.begin() + i); break; } }
    for (staramd_ctx *s : c->sharers) s->owner = nullptr;         // (their index is gone with this context: destroy the sharers first)
    if (!c->owner) freeAll(c->indexAllocs);
    freeAll(c->workAllocs);
    for (int i = 0; i < 10; i++) (void)hipEventDestroy(c->ev[i]);
    if (c->evWait) (void)hipEventDestroy(c->evWait);
    if (c->evDownload) (void)hipEventDestroy(c->evDownload);
    for (int k = 0; k < 2; k++) if (c->in[k].up) (void)hipEventDestroy(c->in[k].up);
    if (c->copyStream) (void)hipStreamDestroy(c->copyStream);
    if (c->hostScratch) (void)hipHostFree(c->hostScratch);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// grow a pool after an overflow of the bump allocator (the batch is then simply run again: inputs are resident)
// grow a pool after an overflow of the bump allocator (the batch is then simply run again: inputs are resident).
// The cursors keep counting past the capacity, so they tell the demand of the stage that overflowed.
static int growPools(staramd_ctx *c, u32 flags, const u32 *cur) {
    DevBatch &B = c->B; std::vector<void *> &R = c->workAllocs; int rc;
    auto grow = [](u32 cap, u32 demand) { return (u32)std::min<u64>(std::max<u64>((u64)cap * 2, (u64)demand + demand / 4 + 1024), 0xFFFFFFF0ull); };
    if (flags & OVF_SEEDPOOL) { B.seedCap = grow(B.seedCap, cur[CUR_SEED]); if ((rc = devRealloc(R, &B.seedPool, (u64)B.seedCap))) return rc; }
    if (flags & OVF_WINPOOL) {
        B.winCap = grow(B.winCap, std::max(cur[CUR_WIN], cur[CUR_ITEM])); B.waCap = grow(B.waCap, cur[CUR_WA]);
        if ((rc = devRealloc(R, &B.winPool, (u64)B.winCap))) return rc;
        if ((rc = devRealloc(R, &B.waPool, (u64)B.waCap))) return rc;
        if ((rc = devRealloc(R, &B.wout, (u64)B.winCap))) return rc;
        if ((rc = devRealloc(R, &B.items, (u64)B.winCap))) return rc;
        if ((rc = devRealloc(R, &B.itemClass, (u64)B.winCap))) return rc;
        if ((rc = devRealloc(R, &B.order, (u64)B.winCap + 64))) return rc;
        if ((rc = devRealloc(R, &B.redoList, (u64)B.winCap))) return rc;
        if ((rc = devRealloc(R, &B.replayList, (u64)B.winCap))) return rc;
        if ((rc = devRealloc(R, &B.heavyList, (u64)B.winCap))) return rc;
        if ((rc = devRealloc(R, &B.heavyList2, (u64)B.winCap))) return rc;
    }
    if (flags & OVF_TRPOOL) {
        B.trCap = grow(B.trCap, cur[CUR_TR]); B.exCap = grow(B.exCap, cur[CUR_EX]);
        if ((rc = devRealloc(R, &B.trPool, (u64)B.trCap))) return rc;
        if ((rc = devRealloc(R, &B.exPool, (u64)B.exCap))) return rc;
        if ((rc = devRealloc(R, &c->dOutTr, (u64)B.trCap))) return rc;
        if ((rc = devRealloc(R, &c->dOutEx, (u64)B.exCap))) return rc;
    }
    return 0;
}

static hipError_t waitStream(staramd_ctx *c) {
    if (!c->evWait) return hipStreamSynchronize(c->stream);
    hipError_t e = hipEventRecord(c->evWait, c->stream);
    return e != hipSuccess ? e : hipEventSynchronize(c->evWait);
}

// every kernel of a batch and the read-back of its totals, cursors and counters, enqueued; nothing is waited for
static int enqueueAll(staramd_ctx *c) {
    c->nLaunches++;
    DevBatch &B = c->B; hipStream_t s = c->stream;
    u32 n = B.nReads;
    HIPCHK(hipMemsetAsync(B.cursors, 0, CUR_N * sizeof(u32), s));
    HIPCHK(hipMemsetAsync(B.counters, 0, DC_N * sizeof(u64), s));
    HIPCHK(hipMemsetAsync(B.costHist, 0, 64 * sizeof(u32), s));
    HIPCHK(hipMemsetAsync(B.candTops, 0, ((size_t)std::max(std::max(std::max(c->stBlocks, c->stBlocksMain), c->stBlocksLean), c->replayBlocks) * 4 + 64) * sizeof(u32), s));
    dim3 block(256);
    u32 ldsWords = ((c->residentMaxLread + 7) / 8) | 1u;              // odd stride: conflict-free LDS staging
    HIPCHK(hipEventRecord(c->ev[0], s));
    {
        u32 lanes = std::min<u32>(c->seedLanes, ((n + 255) / 256) * 256);
        if (c->seedUnits) {
            hipLaunchKernelGGL(k_seed_plan, dim3((n + 255) / 256), block, 0, s, c->dX, B, c->seedWork);
            hipLaunchKernelGGL(k_seed_units, dim3(c->seedUnitLanes / 256), block, 0, s, c->dX, B, c->seedWork);
            hipLaunchKernelGGL(k_seed_merge, dim3(std::min<u32>((n + 3) / 4, (u32)c->nCU * 8u)), block, 0, s, c->dX, B, c->seedWork);
            hipLaunchKernelGGL(k_seed_search, dim3(std::min<u32>(lanes / 256, 64u)), block, 0, s, c->dX, B, c->scrSeed, c->seedPerLane, (const u32 *)c->seedWork.handOn);      // what the units handed on (rarely anything)
        } else
        hipLaunchKernelGGL(k_seed_search, dim3(lanes / 256), block, 0, s, c->dX, B, c->scrSeed, c->seedPerLane, (const u32 *)nullptr);
    }
    HIPCHK(hipEventRecord(c->ev[1], s));
    {
        u32 blocks = std::max<u32>(1, std::min<u32>(c->winBlocks, (n + 3) / 4));
        const u32 useMid = (c->capWMid ? 1u : 0u) | (c->winOwnerMap ? 2u : 0u);
        hipLaunchKernelGGL(k_windows, dim3(blocks), block, 4 * (c->capW * 8 + c->hashBits / 32) * sizeof(u32), s, c->dX, B, c->scrWin, c->capW, c->capBlocks, 0u, c->lightEst, useMid, c->hashBits);
        HIPCHK(hipEventRecord(c->ev[7], s));
        if (c->capWMid) hipLaunchKernelGGL(k_windows, dim3(c->winBlocksMid), dim3(64), (c->capWMid * 8 + c->hashBitsMid / 32) * sizeof(u32), s, c->dX, B, c->scrWinMid, c->capWMid, c->capBlocksMid, 2u, c->lightEst, useMid, c->hashBitsMid);
        hipLaunchKernelGGL(k_windows_big, dim3(c->winBlocksBig), block, 0, s, c->dX, B, c->scrWinBig, c->capWBig, c->capBlocksBig, c->lightEst, useMid);
        HIPCHK(hipEventRecord(c->ev[5], s));
        hipLaunchKernelGGL(k_order_hist, dim3(1024), block, 0, s, B);
        hipLaunchKernelGGL(k_order_offsets, dim3(1), dim3(1), 0, s, B);
        hipLaunchKernelGGL(k_order_scatter, dim3(1024), block, 0, s, B);
    }
    HIPCHK(hipEventRecord(c->ev[2], s));
    {
        size_t readBytes = (ldsWords * 4u + 15u) & ~15u;
        size_t ldsFast = 4 * (readBytes + stitchStateBytesH(c->capDepth, c->capRank, c->arenaFast));
        const u32 prune = c->prune;
        // the lane kernel keeps the packed read of each of its 256 lanes in LDS: reads beyond ~512 bases (2x250 and longer) do not fit into what a block
        // may ask for, and the batch takes the cooperative launches alone (same results: the class cap only picks the kernel)
        const bool laneFits = 256 * (size_t)ldsWords * 4 <= c->ldsLimit;
        if (!(c->laneBlocks && laneFits)) HIPCHK(hipEventRecord(c->ev[8], s));
        size_t ldsLean = c->leanDepth ? 4 * (readBytes + stitchStateBytesH(c->leanDepth, c->capRank, c->leanArena)) : 0;
        for (u32 mode = 0; mode < 2; mode++) {
            if (mode == 0 && c->laneBlocks && laneFits) {      // pass 0 in two launches: one LANE per read for the light reads of few seeds per window, the cooperative walk for the rest
                // highest cost class the lane kernel takes (STARAMD_LANE_CLASS; 0 = by the kind of reads): 3 for paired-end reads, 5 for single-end ones, whose windows are cheap enough for a lane up
                // to there (same box, ms of the stitch stage per 400 k: 2x101 at 3.1 Gb 14.5 / 15.7 / 16.5 / 18.2 at 3 / 4 / 5 / 6; 1x50 at 12 Mb 37.3 / 32.3 / 33.3 / 35.0 at 3 / 5 / 6 / 7: profiles/r06_ab_session7_*)
                const u32 laneClass = c->laneClass ? c->laneClass : (c->X.P.readNmates == 2 ? 3u : 5u);
                hipLaunchKernelGGL(k_stitch_lane, dim3(c->laneBlocks), block, 256 * (size_t)ldsWords * 4, s, c->dX, B, c->scrLane, c->laneArenaBytes, ldsWords, prune, laneClass);
                HIPCHK(hipEventRecord(c->ev[8], s));
                if (c->mainDepth) {     // the cooperative walk in two launches: windows of up to mainDepth - 1 seeds at four blocks per CU, the few that hold more at full depth
                    const size_t ldsMain = 4 * (readBytes + stitchStateBytesH(c->mainDepth, c->capRank, c->arenaFast));
                    hipLaunchKernelGGL(k_stitch_win, dim3(c->stBlocksMain), block, ldsMain, s, c->dX, B, c->scrStitchBig, c->mainDepth, c->capRank, c->arenaFast, c->arenaBig, ldsWords, 3u, prune);
                    hipLaunchKernelGGL(k_stitch_win, dim3(c->stBlocks), block, ldsFast, s, c->dX, B, c->scrStitchBig, c->capDepth, c->capRank, c->arenaFast, c->arenaBig, ldsWords, 4u, prune);
                } else
                hipLaunchKernelGGL(k_stitch_win, dim3(c->stBlocks), block, ldsFast, s, c->dX, B, c->scrStitchBig, c->capDepth, c->capRank, c->arenaFast, c->arenaBig, ldsWords, 2u, prune);
            } else
            if (mode == 0 && c->leanDepth) {       // pass 0 in two launches: lean LDS slices for the windows of few seeds, full-size slices for the rest
                hipLaunchKernelGGL(k_stitch_win, dim3(c->stBlocksLean), block, ldsLean, s, c->dX, B, c->scrStitchBig, c->leanDepth, c->capRank, c->leanArena, c->arenaBig, ldsWords, 0u, prune);
                hipLaunchKernelGGL(k_stitch_win, dim3(c->stBlocks), block, ldsFast, s, c->dX, B, c->scrStitchBig, c->capDepth, c->capRank, c->arenaFast, c->arenaBig, ldsWords, 2u, prune);
            } else
            hipLaunchKernelGGL(k_stitch_win, dim3(c->stBlocks), block, ldsFast, s, c->dX, B, c->scrStitchBig, c->capDepth, c->capRank, c->arenaFast, c->arenaBig, ldsWords, mode, prune);
            if (mode == 0) HIPCHK(hipEventRecord(c->ev[6], s));
            if (mode == 0) {
                hipLaunchKernelGGL(k_stitch_verify, dim3((n + 255) / 256), block, 0, s, c->dX, B);
                hipLaunchKernelGGL(k_stitch_replay, dim3(c->replayBlocks), block, 4 * (size_t)stitchStateBytesH(0, c->capRank, c->arenaFast), s, c->dX, B, c->scrStitchBig, 0u, c->capRank, c->arenaFast, c->arenaBig, 0u);
            }
        }
        hipLaunchKernelGGL(k_stitch_finish, dim3((n + 255) / 256), block, 0, s, c->dX, B);
    }
    HIPCHK(hipEventRecord(c->ev[3], s));
    hipLaunchKernelGGL(k_scan_local, dim3((n + 255) / 256), block, 0, s, B, c->dTrBase, c->dExBase, c->dBlockTot);
    hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(1024), 0, s, B, c->dBlockTot, (n + 255) / 256, c->dTotals);
    if (c->downloadPending) { HIPCHK(hipStreamWaitEvent(s, c->evDownload, 0)); c->downloadPending = false; }      // the results of the batch before may still be on their way out of dOut* (staramd_map_end)
    hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), block, 0, s, B, c->dTrBase, c->dExBase, c->dBlockTot, c->dOutReads, c->dOutTr, B.trCap, c->dOutEx, B.exCap);
    HIPCHK(hipEventRecord(c->ev[4], s));
    HIPCHK(hipGetLastError());
    u32 *hs = c->hostScratch;
    HIPCHK(hipMemcpyAsync(hs, c->dTotals, 2 * sizeof(u32), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hs + 8, B.cursors, CUR_N * sizeof(u32), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hs + 8 + CUR_N, B.counters, DC_N * sizeof(u64), hipMemcpyDeviceToHost, s));      // pinned; collectAll copies them into c->counters: staramd_get_counters
    return STARAMD_OK;                                                                                     // then always describes the last COLLECTED batch, whatever is in flight
}
// ... waited for: stage times, overflow flags
static int collectAll(staramd_ctx *c, staramd_results *r, u32 *flagsOut) {
    u32 *hs = c->hostScratch;
    HIPCHK(waitStream(c));
    if (getenv("STARAMD_VERBOSE")) fprintf(stderr, "staramd: seed units %u in %u groups, reads handed on to k_seed_search %u; stitch work items %u, handed on by the lane kernel %u, by the main cooperative launch to the full-depth one %u\n", hs[8 + CUR_SEED_UNITS], hs[8 + CUR_SEED_GROUPS], hs[8 + CUR_OVF_SEED], hs[8 + CUR_ITEM], hs[8 + CUR_ST_HEAVY], hs[8 + CUR_ST_HEAVY2]);
    HIPCHK(hipEventElapsedTime(&r->msSeed, c->ev[0], c->ev[1]));
    HIPCHK(hipEventElapsedTime(&r->msWindows, c->ev[1], c->ev[2]));
    HIPCHK(hipEventElapsedTime(&r->msStitch, c->ev[2], c->ev[3]));
    HIPCHK(hipEventElapsedTime(&r->msTotalDevice, c->ev[0], c->ev[4]));
    // [0] k_seed_search  [1] k_windows (both passes)  [2] k_order_*  [3] k_stitch_win pass 0  [4] verify + replay + re-walk + finish
    // [5] scan + gather  [6] total
    c->ms[0] = r->msSeed;
    HIPCHK(hipEventElapsedTime(&c->ms[1], c->ev[1], c->ev[5]));
    HIPCHK(hipEventElapsedTime(&c->ms[2], c->ev[5], c->ev[2]));
    HIPCHK(hipEventElapsedTime(&c->ms[3], c->ev[2], c->ev[6]));
    HIPCHK(hipEventElapsedTime(&c->ms[4], c->ev[6], c->ev[3]));
    HIPCHK(hipEventElapsedTime(&c->ms[5], c->ev[3], c->ev[4]));
    HIPCHK(hipEventElapsedTime(&c->ms[7], c->ev[7], c->ev[5]));        // [7] the middle + last k_windows launches alone (part of [1])
    { float t6 = 0; HIPCHK(hipEventElapsedTime(&t6, c->ev[2], c->ev[8])); c->ms8 = t6; }      // [8] the k_stitch_lane launch alone (part of [3])
    c->ms[6] = r->msTotalDevice;
    *flagsOut = hs[8 + CUR_FLAGS];
    memcpy(c->counters, hs + 8 + CUR_N, DC_N * sizeof(u64));
    return STARAMD_OK;
}
static int launchAll(staramd_ctx *c, staramd_results *r, u32 *flagsOut) {
    int rc = enqueueAll(c);
    return rc ? rc : collectAll(c, r, flagsOut);
}

// The kernels of the engine are persistent launches sized to fill the GPU: when two contexts of one device (the front end runs two, so that the copies of one
// batch overlap with the kernels of the other) have their launches in flight at the same time they do not run side by side, they take each other's CUs -- every
// kernel stretches, and the short ones (k_stitch_verify: 1 ms alone) wait 5-7 ms for a CU behind the other context's persistent blocks (rocprofv3 timeline,
// profiles/r04_timeline_two_contexts.txt).  STARAMD_KERNEL_TURNS=1 takes the KERNEL phase of a batch in turns per device (uploads before it and result copies after it
// still overlap with the other context's kernels).  Measured, alternating runs on one box: 6.65 M pairs/s with turns, 6.82 without, 6.78 with ONE context -- the front end
// runs one context per GPU by default now, and the knob stays off.
// the result arrays of the batch that was mapped last, into the caller's: totals first -- arrays that are too small are an error return, and the results stay where they are
static int copyResults(staramd_ctx *c, staramd_results *r) {
    DevBatch &B = c->B; hipStream_t s = c->stream; const u32 n = B.nReads;
    const u32 *totals = c->hostScratch;
    r->trCount = totals[0]; r->exCount = totals[1];
    if (totals[0] > r->trCapacity || totals[1] > r->exCapacity) { g_err = "result arrays too small: need " + std::to_string(totals[0]) + " transcripts, " + std::to_string(totals[1]) + " exons"; return STARAMD_ERR_RESULT_OVERFLOW; }
    HIPCHK(hipMemcpyAsync(r->reads, c->dOutReads, (u64)n * sizeof(staramd_read_result), hipMemcpyDeviceToHost, s));
    if (totals[0]) HIPCHK(hipMemcpyAsync(r->tr, c->dOutTr, (u64)totals[0] * sizeof(staramd_transcript), hipMemcpyDeviceToHost, s));
    if (totals[1]) HIPCHK(hipMemcpyAsync(r->ex, c->dOutEx, (u64)totals[1] * sizeof(staramd_exon), hipMemcpyDeviceToHost, s));
    HIPCHK(waitStream(c));
    return STARAMD_OK;
}
static std::mutex g_kernelTurn[64];
static int runDevice(staramd_ctx *c, staramd_results *r) {
    DevBatch &B = c->B; hipStream_t s = c->stream;
    u32 n = B.nReads;
    u32 flags = 0;
    for (int attempt = 0;; attempt++) {
        int rc;
        if (c->kernelTurns) { std::lock_guard<std::mutex> turn(g_kernelTurn[c->device & 63]); rc = launchAll(c, r, &flags); }
        else rc = launchAll(c, r, &flags);
        if (rc) return rc;
        if (flags == 0) break;
        const u32 *cur = c->hostScratch + 8;
        if ((flags & OVF_HARD) || attempt >= 12) {
            char buf[320];
            snprintf(buf, sizeof(buf), "device work-space overflow (flags 0x%x): seeds %u/%u windows %u/%u WA %u/%u tr %u/%u ex %u/%u",
                     flags, cur[CUR_SEED], B.seedCap, cur[CUR_WIN], B.winCap, cur[CUR_WA], B.waCap, cur[CUR_TR], B.trCap, cur[CUR_EX], B.exCap);
            g_err = buf; return STARAMD_ERR_SCRATCH_OVERFLOW;
        }
        rc = growPools(c, flags, cur);
        if (rc) return rc;
    }
    return copyResults(c, r);
}

// an upload that was started for a batch which will not be mapped: waited for and forgotten (the sets are free again)
static void dropPrefetched(staramd_ctx *c) {
    if (!c->copyStream || (!c->in[0].pending && !c->in[1].pending)) return;
    (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->copyStream);
    c->in[0].pending = c->in[1].pending = false;
}

static int mapBatchImpl(staramd_ctx *c, const staramd_batch *b, staramd_results *r);
static int runDevice(staramd_ctx *c, staramd_results *r);
extern "C" int staramd_map_batch(staramd_ctx *c, const staramd_batch *b, staramd_results *r) {
    if (!c || !b || !r || !r->reads) { g_err = "bad arguments"; return STARAMD_ERR_ARG; }
    const int rc = mapBatchImpl(c, b, r);
    if (rc != STARAMD_OK) dropPrefetched(c);        // after an error nothing uploaded ahead is trusted: the caller's next batch is uploaded by its own call
    return rc;
}

// checks, upload (unless staramd_prefetch_batch has done it) and the packed copy of the reads: the batch is resident, nothing is waited for
static int stageBatch(staramd_ctx *c, const staramd_batch *b) {
    // a batch may be a slice of a larger one (readOffset[0] > 0: the pieces of a WASP re-mapping batch): sized and uploaded from its own first base
    const u64 base0 = b->readOffset[0], nBases = b->readOffset[b->nReads] - base0;
    if (b->nReads > c->maxReads || nBases > c->maxBases) { g_err = "batch larger than the context's work space"; return STARAMD_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream; u32 n = b->nReads;
    u32 maxL = 0;
    for (u32 i = 0; i < n; i++) { u64 L = b->readOffset[i + 1] - b->readOffset[i]; if (L > maxL) maxL = (u32)L; }
    const u64 *offs = b->readOffset;
    if (base0) { c->rebased.resize((size_t)n + 1); for (u32 i = 0; i <= n; i++) c->rebased[i] = b->readOffset[i] - base0; offs = c->rebased.data(); }
    if (maxL > 2 * STARAMD_READ_LEN_MAX + 1) { g_err = "read longer than DEF_readSeqLengthMax"; return STARAMD_ERR_ARG; }
    u32 packWords = (maxL + 7) / 8;
    if ((u64)packWords * n > c->packWordsCap) {
        int rc = devRealloc(c->workAllocs, &c->dPacked, (u64)packWords * c->maxReads); if (rc) return rc;
        c->packWordsCap = (u32)std::min<u64>((u64)packWords * c->maxReads, 0xFFFFFFFFull);
    }
    {
        int use = -1;
        for (int k = 0; k < 2; k++) if (c->in[k].pending && c->in[k].hBases == b->bases && c->in[k].hReadOffset == b->readOffset && c->in[k].nReads == n && base0 == 0) use = k;
        if (use >= 0) {             // this batch was shown to staramd_prefetch_batch: its upload is in flight (or done) on the copy stream
            HIPCHK(hipStreamWaitEvent(s, c->in[use].up, 0));
            c->nPrefetchHits++;
        } else {                    // not prefetched: into the set that holds nothing pending (both pending: the older one is given up)
            use = !c->in[c->cur].pending ? c->cur : (c->copyStream && !c->in[1 - c->cur].pending ? 1 - c->cur : c->cur);
            staramd_ctx::InSet &I = c->in[use];
            if (I.pending) HIPCHK(hipStreamSynchronize(c->copyStream));          // (an upload nobody asked for any more may still be writing into the set)
            HIPCHK(hipMemcpyAsync(I.bases, b->bases + base0, nBases, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(I.readOffset, offs, (u64)(n + 1) * 8, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(I.mate1, b->mate1Length, (u64)n * 2, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(I.mm, b->mmMaxTotal, (u64)n * 2, hipMemcpyHostToDevice, s));
        }
        c->in[use].pending = false; c->cur = use;
        c->dBases = c->in[use].bases; c->dReadOffset = c->in[use].readOffset; c->dMate1 = c->in[use].mate1; c->dMM = c->in[use].mm;
        c->B.bases = c->dBases; c->B.readOffset = c->dReadOffset; c->B.mate1Length = c->dMate1; c->B.mmMaxTotal = c->dMM;
    }
    c->B.nReads = n; c->residentReads = n; c->residentMaxLread = maxL;
    c->B.packed = c->dPacked; c->B.packWords = packWords;
    hipLaunchKernelGGL(k_pack_reads, dim3(n), dim3(64), 0, s, c->B, c->dPacked, packWords);
    return STARAMD_OK;
}
// what tells one batch from another in the same host arrays: its size and the bases at its two ends
static u64 batchMark(const staramd_batch *b) {
    const u64 lo = b->readOffset[0], hi = b->readOffset[b->nReads]; u64 h = 0x9E3779B97F4A7C15ull ^ hi;
    for (u64 i = lo; i < hi && i < lo + 64; i++) h = (h ^ b->bases[i]) * 0x100000001B3ull;
    for (u64 i = hi > lo + 64 ? hi - 64 : lo; i < hi; i++) h = (h ^ b->bases[i]) * 0x100000001B3ull;
    return h;
}
static int mapBatchImpl(staramd_ctx *c, const staramd_batch *b, staramd_results *r) {
    if (b->nReads == 0) { r->trCount = r->exCount = 0; return STARAMD_OK; }
    if (c->inFlight) { g_err = "a batch begun with staramd_map_begin is in flight: staramd_map_end first"; return STARAMD_ERR_ARG; }
    if (c->ovfBases == (const void *)b->bases && c->ovfOffsets == (const void *)b->readOffset && c->ovfReads == b->nReads && c->B.nReads == b->nReads && c->ovfMark == batchMark(b)) {
        // the call before this one mapped this very batch and could not hand the results over (STARAMD_ERR_RESULT_OVERFLOW); they are resident: copied out, nothing runs again
        c->ovfBases = c->ovfOffsets = nullptr; c->ovfReads = 0;
        HIPCHK(hipSetDevice(c->device));
        r->msSeed = c->ovfMs[0]; r->msWindows = c->ovfMs[1]; r->msStitch = c->ovfMs[2]; r->msTotalDevice = c->ovfMs[3];
        const int rc2 = copyResults(c, r);
        if (rc2 == STARAMD_ERR_RESULT_OVERFLOW) { c->ovfBases = b->bases; c->ovfOffsets = b->readOffset; c->ovfReads = b->nReads; c->ovfMark = batchMark(b); }
        return rc2;
    }
    c->ovfBases = c->ovfOffsets = nullptr; c->ovfReads = 0;
    int rc = stageBatch(c, b);
    if (!rc) rc = runDevice(c, r);
    if (rc == STARAMD_ERR_RESULT_OVERFLOW) { c->ovfBases = b->bases; c->ovfOffsets = b->readOffset; c->ovfReads = b->nReads; c->ovfMark = batchMark(b); c->ovfMs[0] = r->msSeed; c->ovfMs[1] = r->msWindows; c->ovfMs[2] = r->msStitch; c->ovfMs[3] = r->msTotalDevice; }
    return rc;
}

// ---- the two halves of staramd_map_batch (include/star_amd_async.h) ----
extern "C" int staramd_map_begin(staramd_ctx *c, const staramd_batch *b) {
    if (!c || !b || b->nReads == 0) { g_err = "bad arguments"; return STARAMD_ERR_ARG; }
    if (c->inFlight) { g_err = "a batch is in flight already: staramd_map_end first"; return STARAMD_ERR_ARG; }
    int rc = stageBatch(c, b);
    if (!rc) rc = enqueueAll(c);
    if (rc) { dropPrefetched(c); return rc; }
    c->inFlight = true; c->collected = false;
    return STARAMD_OK;
}
extern "C" int staramd_map_wait(staramd_ctx *c) {
    if (!c) { g_err = "bad arguments"; return STARAMD_ERR_ARG; }
    if (!c->inFlight) { g_err = "no batch in flight: staramd_map_begin first"; return STARAMD_ERR_ARG; }
    if (c->collected) return STARAMD_OK;
    HIPCHK(hipSetDevice(c->device));
    u32 flags = 0;
    int rc = collectAll(c, &c->msRes, &flags);
    for (int attempt = 0; !rc && flags; attempt++) {          // a pool overflowed: grown, and the batch (its inputs are resident) run again -- as staramd_map_batch does
        const u32 *cur = c->hostScratch + 8;
        if ((flags & OVF_HARD) || attempt >= 12) { g_err = "device work-space overflow (flags " + std::to_string(flags) + ")"; rc = STARAMD_ERR_SCRATCH_OVERFLOW; break; }
        rc = growPools(c, flags, cur);
        if (!rc) rc = launchAll(c, &c->msRes, &flags);
    }
    if (rc) { c->inFlight = false; dropPrefetched(c); return rc; }
    c->collected = true;
    return STARAMD_OK;
}
extern "C" int staramd_map_end(staramd_ctx *c, staramd_results *r, const staramd_batch *next) {
    if (!c || !r || !r->reads) { g_err = "bad arguments"; return STARAMD_ERR_ARG; }
    if (!c->inFlight) { g_err = "no batch in flight: staramd_map_begin first"; return STARAMD_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    DevBatch &B = c->B; const u32 n = B.nReads;
    { const int rc = staramd_map_wait(c); if (rc) return rc; }
    r->msSeed = c->msRes.msSeed; r->msWindows = c->msRes.msWindows; r->msStitch = c->msRes.msStitch; r->msTotalDevice = c->msRes.msTotalDevice;
    const u32 *totals = c->hostScratch;
    r->trCount = totals[0]; r->exCount = totals[1];
    if (totals[0] > r->trCapacity || totals[1] > r->exCapacity) { g_err = "result arrays too small: need " + std::to_string(totals[0]) + " transcripts, " + std::to_string(totals[1]) + " exons"; return STARAMD_ERR_RESULT_OVERFLOW; }   // (the batch stays in flight: call again with larger arrays)
    // the results leave on the copy stream; the kernels of `next` start beside them (its k_gather, the only writer of dOut*, waits for the copy)
    hipStream_t cs = c->copyStream ? c->copyStream : c->stream;
    HIPCHK(hipMemcpyAsync(r->reads, c->dOutReads, (u64)n * sizeof(staramd_read_result), hipMemcpyDeviceToHost, cs));
    if (totals[0]) HIPCHK(hipMemcpyAsync(r->tr, c->dOutTr, (u64)totals[0] * sizeof(staramd_transcript), hipMemcpyDeviceToHost, cs));
    if (totals[1]) HIPCHK(hipMemcpyAsync(r->ex, c->dOutEx, (u64)totals[1] * sizeof(staramd_exon), hipMemcpyDeviceToHost, cs));
    HIPCHK(hipEventRecord(c->evDownload, cs));
    c->downloadPending = c->copyStream != nullptr;
    c->inFlight = false; c->collected = false;
    int rcNext = STARAMD_OK;
    if (next && next->nReads) { rcNext = staramd_map_begin(c, next); if (!rcNext) c->nOverlapped++; }
    HIPCHK(hipEventSynchronize(c->evDownload));
    return rcNext;
}
extern "C" uint64_t staramd_overlapped_batches(staramd_ctx *c) { return c ? c->nOverlapped : 0; }
extern "C" uint64_t staramd_launch_count(staramd_ctx *c) { return c ? c->nLaunches : 0; }
extern "C" uint32_t staramd_capabilities(void) { return STARAMD_CAP_CHIM_SELECT; }

extern "C" int staramd_prefetch_batch(staramd_ctx *c, const staramd_batch *b) {
    if (!c || !b) { g_err = "bad arguments"; return STARAMD_ERR_ARG; }
    if (!c->copyStream || b->nReads == 0 || b->readOffset[0] != 0) return STARAMD_OK;      // prefetch off, or a batch that map_batch rebases: uploaded there
    const u64 nBases = b->readOffset[b->nReads];
    if (b->nReads > c->maxReads || nBases > c->maxBases) return STARAMD_OK;                // (map_batch reports it)
    // the set that holds nothing pending: the one the last mapped batch used (that call has returned), unless the other one is free too
    // (while a batch begun with staramd_map_begin is in flight its kernels read in[cur]: only the other set may be written)
    const int k = !c->in[1 - c->cur].pending ? 1 - c->cur : (!c->in[c->cur].pending && !c->inFlight ? c->cur : -1);
    if (k < 0) return STARAMD_OK;                                                          // two batches waiting already, or the free set is being read by kernels
    HIPCHK(hipSetDevice(c->device));
    staramd_ctx::InSet &I = c->in[k];
    const u32 n = b->nReads; hipStream_t cs = c->copyStream;
    HIPCHK(hipMemcpyAsync(I.bases, b->bases, nBases, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(I.readOffset, b->readOffset, (u64)(n + 1) * 8, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(I.mate1, b->mate1Length, (u64)n * 2, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(I.mm, b->mmMaxTotal, (u64)n * 2, hipMemcpyHostToDevice, cs));
    HIPCHK(hipEventRecord(I.up, cs));
    I.hBases = b->bases; I.hReadOffset = b->readOffset; I.nReads = n; I.pending = true;
    return STARAMD_OK;
}

extern "C" int staramd_prefetch_cancel(staramd_ctx *c) {
    if (!c) { g_err = "bad arguments"; return STARAMD_ERR_ARG; }
    dropPrefetched(c);
    return STARAMD_OK;
}

extern "C" uint64_t staramd_prefetch_hits(staramd_ctx *c) { return c ? c->nPrefetchHits : 0; }

extern "C" int staramd_map_resident(staramd_ctx *c, staramd_results *r) {
    if (!c || !r || !r->reads) { g_err = "bad arguments"; return STARAMD_ERR_ARG; }
    if (c->residentReads == 0) { g_err = "no batch resident in HBM: call staramd_map_batch first"; return STARAMD_ERR_ARG; }
    if (c->in[c->cur].pending) { g_err = "the resident batch was overwritten by staramd_prefetch_batch"; return STARAMD_ERR_ARG; }
    if (c->inFlight) { g_err = "a batch begun with staramd_map_begin is in flight: staramd_map_end first"; return STARAMD_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    c->B.nReads = c->residentReads;
    return runDevice(c, r);
}

extern "C" int staramd_get_timings(staramd_ctx *c, float *out, int n) {
    if (!c || !out) return 0;
    int k = n < 9 ? n : 9;
    for (int i = 0; i < k; i++) out[i] = i < 8 ? c->ms[i] : c->ms8;
    return k;
}

extern "C" int staramd_get_counters(staramd_ctx *c, uint64_t *out, int n) {
    if (!c || !out) return 0;
    int k = n < (int)DC_N ? n : (int)DC_N;
    for (int i = 0; i < k; i++) out[i] = c->counters[i];
    return k;
}
