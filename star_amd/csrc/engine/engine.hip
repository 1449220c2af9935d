// engine.hip -- C ABI of the MI355X engine (include/star_amd.h): index upload, work-space, launches, timing.
// The hot path has no CPU fallback: every entry point fails with an error code when the GPU or the
// kernels are not usable.
#include "dev.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>

extern "C" __global__ void k_seed_search(DevIndex X, DevBatch B, DSeed *scratch, u32 scratchPerLane);
extern "C" __global__ void k_windows(DevIndex X, DevBatch B, u8 *scratch, u32 capW, u32 capBlocks);
extern "C" __global__ void k_stitch(DevIndex X, DevBatch B, u8 *scratch, u32 capDepth, u32 capTr);
extern "C" __global__ void k_scan_offsets(DevBatch B, u32 *trBase, u32 *exBase, u32 *totals);
extern "C" __global__ void k_gather(DevBatch B, const u32 *trBase, const u32 *exBase, staramd_read_result *outReads,
                                    staramd_transcript *outTr, u32 outTrCap, staramd_exon *outEx, u32 outExCap);

static thread_local std::string g_err;
extern "C" const char *staramd_last_error(void) { return g_err.c_str(); }

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); return STARAMD_ERR_DEVICE; } } while (0)


struct staramd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    DevIndex X;
    std::vector<void *> indexAllocs, workAllocs;
    u32 maxReads = 0; u64 maxBases = 0;
    DevBatch B;
    u8 *dBases = nullptr; u64 *dReadOffset = nullptr; u16 *dMate1 = nullptr, *dMM = nullptr;
    u32 lanes = 0;
    DSeed *scrSeed = nullptr; u32 seedPerLane = 0;
    u8 *scrWin = nullptr; u32 capW = 0, capBlocks = 0;
    u8 *scrStitch = nullptr; u32 capDepth = 0, capTr = 0;
    u32 *dTrBase = nullptr, *dExBase = nullptr, *dTotals = nullptr;
    staramd_read_result *dOutReads = nullptr; staramd_transcript *dOutTr = nullptr; staramd_exon *dOutEx = nullptr;
    hipEvent_t ev[6];
    u64 counters[DC_N];
    u32 residentReads = 0;
};

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

static int uploadIndex(staramd_ctx *c, const staramd_genome *g, const staramd_params *p) {
    DevIndex &X = c->X;
    memset(&X, 0, sizeof(X));
    if (g->gSAsparseD < 1 || g->gSAsparseD > 8) { g_err = "genomeSAsparseD must be in 1..8"; return STARAMD_ERR_ARG; }
    if (g->gSAindexNbases > 16 || g->GstrandBit + 3 > 63) { g_err = "unsupported index geometry"; return STARAMD_ERR_ARG; }
    if (p->seedPerWindowNmax > 4096 || p->alignTranscriptsPerWindowNmax > 60000) { g_err = "seedPerWindowNmax/alignTranscriptsPerWindowNmax too large for the device work space"; return STARAMD_ERR_ARG; }
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
    X.nGenome = g->nGenome; X.nSA = g->nSA; X.sjGstart = g->sjGstart;
    for (int i = 0; i < 17; i++) X.saiStart[i] = g->genomeSAindexStart[i];
    X.strandBit = g->GstrandBit; X.saBits = g->GstrandBit + 1; X.saiBits = g->GstrandBit + 3;
    X.saMask = X.saBits >= 64 ? ~0ull : ((1ull << X.saBits) - 1); X.saiMask = (1ull << X.saiBits) - 1;
    X.strandMask = ~(1ull << g->GstrandBit);               // Genome_genomeLoad.cpp:157
    X.saiNbit = 1ull << (g->GstrandBit + 1); X.saiAbsentBit = 1ull << (g->GstrandBit + 2);   // :161-166
    X.saiNbases = g->gSAindexNbases; X.sparseD = g->gSAsparseD;
    X.sjdbOverhang = g->sjdbOverhang; X.sjdbLength = g->sjdbLength ? g->sjdbLength : 1; X.sjdbN = g->sjdbN; X.nChrReal = g->nChrReal;
    X.P = *p;
    buildGlBreaks(X, p->scoreGenomicLengthLog2scale);
    return 0;
}

static u32 envU32(const char *name, u32 dflt) { const char *s = getenv(name); return s ? (u32)strtoul(s, nullptr, 10) : dflt; }

static int allocWork(staramd_ctx *c) {
    std::vector<void *> &R = c->workAllocs;
    u32 N = c->maxReads; int rc;
    if ((rc = devAlloc(R, &c->dBases, c->maxBases + 64))) return rc;
    if ((rc = devAlloc(R, &c->dReadOffset, (u64)N + 1))) return rc;
    if ((rc = devAlloc(R, &c->dMate1, (u64)N))) return rc;
    if ((rc = devAlloc(R, &c->dMM, (u64)N))) return rc;
    DevBatch &B = c->B; memset(&B, 0, sizeof(B));
    B.bases = c->dBases; B.readOffset = c->dReadOffset; B.mate1Length = c->dMate1; B.mmMaxTotal = c->dMM;
    if ((rc = devAlloc(R, &B.reads, (u64)N))) return rc;
    B.seedCap = (u32)std::min<u64>((u64)N * envU32("STARAMD_SEEDS_PER_READ", 64) + 4096, 0xFFFFFFF0ull);
    B.winCap = (u32)std::min<u64>((u64)N * envU32("STARAMD_WINDOWS_PER_READ", 40) + 4096, 0xFFFFFFF0ull);
    B.waCap = (u32)std::min<u64>((u64)N * envU32("STARAMD_WA_PER_READ", 256) + 4096, 0xFFFFFFF0ull);
    B.wtCap = B.winCap;
    B.trCap = (u32)std::min<u64>((u64)N * envU32("STARAMD_TR_PER_READ", 64) + 4096, 0xFFFFFFF0ull);
    B.exCap = (u32)std::min<u64>((u64)B.trCap * 3, 0xFFFFFFF0ull);
    if ((rc = devAlloc(R, &B.seedPool, (u64)B.seedCap))) return rc;
    if ((rc = devAlloc(R, &B.winPool, (u64)B.winCap))) return rc;
    if ((rc = devAlloc(R, &B.waPool, (u64)B.waCap))) return rc;
    if ((rc = devAlloc(R, &B.wtPool, (u64)B.wtCap))) return rc;
    if ((rc = devAlloc(R, &B.trPool, (u64)B.trCap))) return rc;
    if ((rc = devAlloc(R, &B.exPool, (u64)B.exCap))) return rc;
    if ((rc = devAlloc(R, &B.cursors, (u64)16))) return rc;
    if ((rc = devAlloc(R, &B.counters, (u64)DC_N))) return rc;
    if ((rc = devAlloc(R, &c->dTrBase, (u64)N))) return rc;
    if ((rc = devAlloc(R, &c->dExBase, (u64)N))) return rc;
    if ((rc = devAlloc(R, &c->dTotals, (u64)4))) return rc;
    if ((rc = devAlloc(R, &c->dOutReads, (u64)N))) return rc;
    if ((rc = devAlloc(R, &c->dOutTr, (u64)B.trCap))) return rc;
    if ((rc = devAlloc(R, &c->dOutEx, (u64)B.exCap))) return rc;
    // per-lane scratch sized by the reference's own per-read limits
    const staramd_params &P = c->X.P;
    u32 lanes = envU32("STARAMD_LANES", 65536);
    lanes = std::max<u32>(256, std::min<u32>(lanes, ((N + 255) / 256) * 256));
    lanes = (lanes / 256) * 256;
    c->lanes = lanes;
    c->seedPerLane = P.seedPerReadNmax + 1;
    if ((rc = devAlloc(R, &c->scrSeed, (u64)lanes * c->seedPerLane))) return rc;
    c->capW = envU32("STARAMD_CAP_WINDOWS", 512); c->capBlocks = envU32("STARAMD_CAP_WA_BLOCKS", 96);
    u64 perLaneW = (u64)c->capW * sizeof(WScr) + (u64)c->capBlocks * P.seedPerWindowNmax * sizeof(DWA);
    if ((rc = devAlloc(R, &c->scrWin, (u64)lanes * perLaneW))) return rc;
    c->capDepth = P.seedPerWindowNmax + 1; c->capTr = P.alignTranscriptsPerWindowNmax + 1;
    u64 perLaneS = (u64)c->capDepth * sizeof(Frame) + (u64)c->capTr * sizeof(DTr) + (u64)c->capTr * sizeof(u16);
    perLaneS = (perLaneS + 15) & ~15ull;
    if ((rc = devAlloc(R, &c->scrStitch, (u64)lanes * perLaneS))) return rc;
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
    if (!rc) { if (hipStreamCreate(&c->stream) != hipSuccess) { g_err = "hipStreamCreate failed"; rc = STARAMD_ERR_DEVICE; } }
    if (!rc) for (int i = 0; i < 6; i++) if (hipEventCreate(&c->ev[i]) != hipSuccess) { g_err = "hipEventCreate failed"; rc = STARAMD_ERR_DEVICE; }
    if (rc) { freeAll(c->indexAllocs); freeAll(c->workAllocs); delete c; return rc; }
    memset(c->counters, 0, sizeof(c->counters));
    *out = c;
    return STARAMD_OK;
}

extern "C" int staramd_update_index(staramd_ctx *c, const staramd_genome *g, const staramd_params *p) {
    if (!c) { g_err = "null context"; return STARAMD_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    freeAll(c->indexAllocs);
    return uploadIndex(c, g, p);
}

extern "C" void staramd_destroy(staramd_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    freeAll(c->indexAllocs); freeAll(c->workAllocs);
    for (int i = 0; i < 6; i++) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int runDevice(staramd_ctx *c, staramd_results *r) {
    DevBatch &B = c->B; hipStream_t s = c->stream;
    u32 n = B.nReads;
    HIPCHK(hipMemsetAsync(B.cursors, 0, 16 * sizeof(u32), s));
    HIPCHK(hipMemsetAsync(B.counters, 0, DC_N * sizeof(u64), s));
    u32 lanes = std::min<u32>(c->lanes, ((n + 255) / 256) * 256);
    dim3 grid(lanes / 256), block(256);
    HIPCHK(hipEventRecord(c->ev[0], s));
    hipLaunchKernelGGL(k_seed_search, grid, block, 0, s, c->X, B, c->scrSeed, c->seedPerLane);
    HIPCHK(hipEventRecord(c->ev[1], s));
    hipLaunchKernelGGL(k_windows, grid, block, 0, s, c->X, B, c->scrWin, c->capW, c->capBlocks);
    HIPCHK(hipEventRecord(c->ev[2], s));
    hipLaunchKernelGGL(k_stitch, grid, block, 0, s, c->X, B, c->scrStitch, c->capDepth, c->capTr);
    HIPCHK(hipEventRecord(c->ev[3], s));
    hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(1024), 0, s, B, c->dTrBase, c->dExBase, c->dTotals);
    hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), block, 0, s, B, c->dTrBase, c->dExBase, c->dOutReads, c->dOutTr, B.trCap, c->dOutEx, B.exCap);
    HIPCHK(hipEventRecord(c->ev[4], s));
    HIPCHK(hipGetLastError());
    u32 totals[4] = {0, 0, 0, 0}; u32 cursors[16];
    HIPCHK(hipMemcpyAsync(totals, c->dTotals, 2 * sizeof(u32), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(cursors, B.cursors, 16 * sizeof(u32), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(c->counters, B.counters, DC_N * sizeof(u64), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipEventElapsedTime(&r->msSeed, c->ev[0], c->ev[1]));
    HIPCHK(hipEventElapsedTime(&r->msWindows, c->ev[1], c->ev[2]));
    HIPCHK(hipEventElapsedTime(&r->msStitch, c->ev[2], c->ev[3]));
    HIPCHK(hipEventElapsedTime(&r->msTotalDevice, c->ev[0], c->ev[4]));
    if (cursors[6] != 0) {
        char buf[256];
        snprintf(buf, sizeof(buf), "device work-space overflow (flags 0x%x): seeds %u/%u windows %u/%u WA %u/%u tr %u/%u ex %u/%u; raise STARAMD_*_PER_READ / STARAMD_CAP_* or lower the batch size",
                 cursors[6], cursors[0], B.seedCap, cursors[1], B.winCap, cursors[2], B.waCap, cursors[4], B.trCap, cursors[5], B.exCap);
        g_err = buf; return STARAMD_ERR_SCRATCH_OVERFLOW;
    }
    r->trCount = totals[0]; r->exCount = totals[1];
    if (totals[0] > r->trCapacity || totals[1] > r->exCapacity) { g_err = "result arrays too small: need " + std::to_string(totals[0]) + " transcripts, " + std::to_string(totals[1]) + " exons"; return STARAMD_ERR_RESULT_OVERFLOW; }
    HIPCHK(hipMemcpyAsync(r->reads, c->dOutReads, (u64)n * sizeof(staramd_read_result), hipMemcpyDeviceToHost, s));
    if (totals[0]) HIPCHK(hipMemcpyAsync(r->tr, c->dOutTr, (u64)totals[0] * sizeof(staramd_transcript), hipMemcpyDeviceToHost, s));
    if (totals[1]) HIPCHK(hipMemcpyAsync(r->ex, c->dOutEx, (u64)totals[1] * sizeof(staramd_exon), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return STARAMD_OK;
}

extern "C" int staramd_map_batch(staramd_ctx *c, const staramd_batch *b, staramd_results *r) {
    if (!c || !b || !r || !r->reads) { g_err = "bad arguments"; return STARAMD_ERR_ARG; }
    if (b->nReads == 0) { r->trCount = r->exCount = 0; return STARAMD_OK; }
    if (b->nReads > c->maxReads || b->readOffset[b->nReads] > c->maxBases) { g_err = "batch larger than the context's work space"; return STARAMD_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream; u32 n = b->nReads;
    HIPCHK(hipMemcpyAsync(c->dBases, b->bases, b->readOffset[n], hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->dReadOffset, b->readOffset, (u64)(n + 1) * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->dMate1, b->mate1Length, (u64)n * 2, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->dMM, b->mmMaxTotal, (u64)n * 2, hipMemcpyHostToDevice, s));
    c->B.nReads = n; c->residentReads = n;
    return runDevice(c, r);
}

extern "C" int staramd_map_resident(staramd_ctx *c, staramd_results *r) {
    if (!c || !r || !r->reads) { g_err = "bad arguments"; return STARAMD_ERR_ARG; }
    if (c->residentReads == 0) { g_err = "no batch resident in HBM: call staramd_map_batch first"; return STARAMD_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    c->B.nReads = c->residentReads;
    return runDevice(c, r);
}

extern "C" int staramd_get_counters(staramd_ctx *c, uint64_t *out, int n) {
    if (!c || !out) return 0;
    int k = n < (int)DC_N ? n : (int)DC_N;
    for (int i = 0; i < k; i++) out[i] = c->counters[i];
    return k;
}
