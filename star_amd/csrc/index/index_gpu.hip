// index_gpu.hip -- HIP backend of index_core.h + the C ABI of include/star_amd_index.h.
// Element-wise passes are one thread per element (grid-stride free: the grids are >> 256 CUs); sorts and scans are
// rocPRIM's device-wide primitives (onesweep radix sort: every pass streams keys+values once through HBM).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <string>
#include <algorithm>
#include "index_core.h"
#include "sjdb_core.h"
#include "../../../include/star_amd.h"
#include "../../../include/star_amd_index.h"

using namespace staridx;

namespace {

template <class F> __global__ void __launch_bounds__(256) k_forEach(u64 n, F f) {
    u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n) f(i);
}

struct HipBackend {
    hipStream_t s = nullptr;
    hipError_t err = hipSuccess;
    const char *where = "";
    void *tmp = nullptr; size_t tmpBytes = 0;
    u64 liveBytes = 0, peakBytes = 0;

    void chk(hipError_t e, const char *w) { if (e != hipSuccess && err == hipSuccess) { err = e; where = w; } }
    template <class T> T *alloc(u64 n) {
        void *p = nullptr; size_t b = std::max<u64>(n * sizeof(T), 256);
        chk(hipMalloc(&p, b), "hipMalloc");
        return (T *)p;
    }
    void free(void *p) { if (p) chk(hipFree(p), "hipFree"); }
    template <class F> void forEach(u64 n, F f) {
        if (n == 0 || err != hipSuccess) return;
        u64 blocks = (n + 255) / 256;
        hipLaunchKernelGGL(k_forEach<F>, dim3((unsigned)blocks), dim3(256), 0, s, n, f);
        chk(hipGetLastError(), "k_forEach launch");
    }
    void needTmp(size_t b) {
        if (b <= tmpBytes) return;
        if (tmp) chk(hipFree(tmp), "hipFree(tmp)");
        tmpBytes = b + b / 4 + 4096; tmp = nullptr;
        chk(hipMalloc(&tmp, tmpBytes), "hipMalloc(tmp)");
    }
    void sortPairs(u64 *&k, u64 *&kAlt, u64 *&v, u64 *&vAlt, u64 n, int b0, int b1) {
        if (err != hipSuccess) return;
        rocprim::double_buffer<u64> dk(k, kAlt), dv(v, vAlt);
        size_t need = 0;
        chk(rocprim::radix_sort_pairs(nullptr, need, dk, dv, (size_t)n, (unsigned)b0, (unsigned)b1, s), "radix_sort_pairs(size)");
        needTmp(need);
        chk(rocprim::radix_sort_pairs(tmp, need, dk, dv, (size_t)n, (unsigned)b0, (unsigned)b1, s), "radix_sort_pairs");
        k = dk.current(); kAlt = dk.alternate(); v = dv.current(); vAlt = dv.alternate();
    }
    void exclusiveSum(u64 *a, u64 n) {
        if (err != hipSuccess || n == 0) return;
        size_t need = 0;
        chk(rocprim::exclusive_scan(nullptr, need, a, a, (u64)0, (size_t)n, rocprim::plus<u64>(), s), "exclusive_scan(size)");
        needTmp(need);
        chk(rocprim::exclusive_scan(tmp, need, a, a, (u64)0, (size_t)n, rocprim::plus<u64>(), s), "exclusive_scan");
    }
    void inclusiveMax(u64 *a, u64 n) {
        if (err != hipSuccess || n == 0) return;
        size_t need = 0;
        chk(rocprim::inclusive_scan(nullptr, need, a, a, (size_t)n, rocprim::maximum<u64>(), s), "inclusive_scan(size)");
        needTmp(need);
        chk(rocprim::inclusive_scan(tmp, need, a, a, (size_t)n, rocprim::maximum<u64>(), s), "inclusive_scan");
    }
    u64 readOne(const u64 *p) {
        u64 v = 0;
        if (err != hipSuccess) return 0;
        chk(hipMemcpyAsync(&v, p, 8, hipMemcpyDeviceToHost, s), "readOne");
        chk(hipStreamSynchronize(s), "readOne sync");
        return v;
    }
    template <class T> void copyToHost(T *dst, const T *src, u64 n) {
        if (err != hipSuccess) return;
        chk(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, s), "copyToHost");
        chk(hipStreamSynchronize(s), "copyToHost sync");
    }
    template <class T> void copyToDevice(T *dst, const T *src, u64 n) {
        if (err != hipSuccess) return;
        chk(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, s), "copyToDevice");
        chk(hipStreamSynchronize(s), "copyToDevice sync");
    }
};

thread_local std::string g_idxErr;

} // namespace

extern "C" const char *staramd_index_last_error(void) { return g_idxErr.c_str(); }

extern "C" int staramd_index_build(int device, const uint8_t *G, const staramd_index_params *p,
                                   uint8_t *SA, uint64_t saCapacity, uint8_t *SAi, uint64_t saiCapacity, staramd_index_result *res) {
    if (!G || !p || !res) { g_idxErr = "staramd_index_build: null argument"; return STARAMD_ERR_ARG; }
    if (p->gSAsparseD != 1) { g_idxErr = "staramd_index_build: only --genomeSAsparseD 1 is built on the device"; return STARAMD_ERR_ARG; }
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev == 0) { g_idxErr = "no HIP device visible: the index builder runs on the GPU only (no CPU fallback)"; return STARAMD_ERR_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { g_idxErr = "hipSetDevice failed"; return STARAMD_ERR_DEVICE; }
    HipBackend be;
    if (hipStreamCreate(&be.s) != hipSuccess) { g_idxErr = "hipStreamCreate failed"; return STARAMD_ERR_DEVICE; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, be.s);
    BuildParams P; P.nGenome = p->nGenome; P.GstrandBit = p->GstrandBit; P.saIndexNbases = p->gSAindexNbases;
    BuildResult R; memset(&R, 0, sizeof(R));
    int rc = buildAll(be, G, P, SA, saCapacity, SAi, saiCapacity, R);
    (void)hipEventRecord(e1, be.s);
    (void)hipStreamSynchronize(be.s);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (be.tmp) (void)hipFree(be.tmp);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(be.s);
    memset(res, 0, sizeof(*res));
    res->nSA = R.nSA; res->nSAbyte = R.nSAbyte; res->nSAi = R.nSAi; res->nSAibyte = R.nSAibyte; res->doublingRounds = (uint32_t)R.rounds;
    for (int i = 0; i < 17; i++) res->genomeSAindexStart[i] = R.saiStart[i];
    res->msTotal = ms;
    if (be.err != hipSuccess) { g_idxErr = std::string("staramd_index_build: ") + be.where + ": " + hipGetErrorString(be.err); return STARAMD_ERR_DEVICE; }
    if (rc == -1) { g_idxErr = "staramd_index_build: bad parameters"; return STARAMD_ERR_ARG; }
    if (rc == -2) { g_idxErr = "staramd_index_build: output buffers too small"; return STARAMD_ERR_RESULT_OVERFLOW; }
    if (rc == -3) { g_idxErr = "staramd_index_build: the first suffix of the genome has a non-ACGT base inside the SAindex prefix (the reference cannot index such a genome either)"; return STARAMD_ERR_ARG; }
    return STARAMD_OK;
}

extern "C" int staramd_sjdb_insert(int device, const staramd_sjdb_args *a, staramd_sjdb_result *res) {
    if (!a || !res || !a->G || !a->SA || !a->Gsj || !a->isOld || !a->SAout || !a->SAiOut || (a->oldSjdbN && !a->oldSJind)) { g_idxErr = "staramd_sjdb_insert: null argument"; return STARAMD_ERR_ARG; }
    if (a->sjdbN == 0 || a->sjdbLength < 3 || a->gSAindexNbases < 1 || a->gSAindexNbases > 16 || a->nSAold == 0) { g_idxErr = "staramd_sjdb_insert: bad parameters"; return STARAMD_ERR_ARG; }
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev == 0) { g_idxErr = "no HIP device visible: junction insertion runs on the GPU only (no CPU fallback in this library)"; return STARAMD_ERR_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { g_idxErr = "hipSetDevice failed"; return STARAMD_ERR_DEVICE; }
    HipBackend be;
    if (hipStreamCreate(&be.s) != hipSuccess) { g_idxErr = "hipStreamCreate failed"; return STARAMD_ERR_DEVICE; }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, be.s);
    SjdbParams P; P.nGenomeOld = a->nGenomeOld; P.nGenomeReal = a->nGenomeReal; P.nSAold = a->nSAold; P.GstrandBit = a->GstrandBit;
    P.sjdbN = a->sjdbN; P.sjdbLength = a->sjdbLength; P.oldSjdbN = a->oldSjdbN; P.sjNew = a->sjNew; P.saIndexNbases = a->gSAindexNbases;
    SjdbHostArgs A{a->G, a->SA, a->nSAbyteOld, a->Gsj, a->isOld, a->oldSJind, a->SAout, a->saOutCapacity, a->SAiOut, a->saiOutCapacity};
    u64 nInd = 0, nb = 0, nbi = 0;
    int rc = sjdbInsertHost(be, P, A, nInd, nb, nbi);
    (void)hipEventRecord(e1, be.s); (void)hipStreamSynchronize(be.s);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (be.tmp) (void)hipFree(be.tmp);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(be.s);
    memset(res, 0, sizeof(*res));
    res->nInd = nInd; res->nSAnew = a->nSAold + nInd; res->nSAbyteNew = nb; res->nSAibyte = nbi; res->msTotal = ms;
    if (be.err != hipSuccess) { g_idxErr = std::string("staramd_sjdb_insert: ") + be.where + ": " + hipGetErrorString(be.err); return STARAMD_ERR_DEVICE; }
    if (rc == -2) { g_idxErr = "staramd_sjdb_insert: output buffers too small"; return STARAMD_ERR_RESULT_OVERFLOW; }
    if (rc == -3) { g_idxErr = "staramd_sjdb_insert: the first suffix has a non-ACGT base inside the SAindex prefix"; return STARAMD_ERR_ARG; }
    return STARAMD_OK;
}
