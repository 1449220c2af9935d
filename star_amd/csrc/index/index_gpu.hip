// index_gpu.hip -- HIP backend of index_core.h + the C ABI of include/star_amd_index.h.
// Element-wise passes are one thread per element (grid-stride free: the grids are >> 256 CUs); sorts and scans are
// rocPRIM's device-wide primitives (onesweep radix sort: every pass streams keys+values once through HBM).
#include <cstdio>
#include <string>
#include "hip_backend.h"
#include "sjdb_core.h"
#include "../../../include/star_amd.h"
#include "../../../include/star_amd_index.h"

using namespace staridx;

namespace {

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
