// index_emul.cpp -- TEST INFRASTRUCTURE ONLY.  The index-building algorithm of star_amd/csrc/index/index_core.h
// instantiated with a plain-loop backend, so that its LOGIC can be checked against the reference's genomeGenerate on a
// machine without a GPU (tests/test_index_build.py, -m "not gpu").  The product is the HIP instantiation in
// star_amd/csrc/index/index_gpu.hip; nothing under star_amd/ links this file.
#include "../star_amd/csrc/index/index_core.h"
#include "../star_amd/csrc/index/sjdb_core.h"
#include "../include/star_amd_index.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

using namespace staridx;

struct LoopBackend {
    template <class T> T *alloc(u64 n) { return (T *)malloc(std::max<u64>(n * sizeof(T), 16)); }
    void free(void *p) { ::free(p); }
    void stage(const char *) {}
    template <class F> void forEach(u64 n, F f) {
#pragma omp parallel for schedule(static)
        for (u64 i = 0; i < n; i++) f(i);
    }
    void sortPairs(u64 *&k, u64 *&kAlt, u64 *&v, u64 *&vAlt, u64 n, int b0, int b1) {
        u64 mask = (b1 >= 64 ? ~0ull : ((1ull << b1) - 1)) & ~((1ull << b0) - 1);
        std::vector<u64> idx(n);
        std::iota(idx.begin(), idx.end(), 0);
        const u64 *kk = k;
        std::stable_sort(idx.begin(), idx.end(), [=](u64 a, u64 b) { return (kk[a] & mask) < (kk[b] & mask); });
        for (u64 i = 0; i < n; i++) { kAlt[i] = k[idx[i]]; vAlt[i] = v[idx[i]]; }
        std::swap(k, kAlt); std::swap(v, vAlt);
    }
    void exclusiveSum(u64 *a, u64 n) { u64 s = 0; for (u64 i = 0; i < n; i++) { u64 t = a[i]; a[i] = s; s += t; } }
    void inclusiveMax(u64 *a, u64 n) { u64 m = 0; for (u64 i = 0; i < n; i++) { m = std::max(m, a[i]); a[i] = m; } }
    u64 readOne(const u64 *p) { return *p; }
    template <class T> void copyToHost(T *dst, const T *src, u64 n) { memcpy(dst, src, n * sizeof(T)); }
    template <class T> void copyToDevice(T *dst, const T *src, u64 n) { memcpy(dst, src, n * sizeof(T)); }
};

extern "C" int index_emul_build(const uint8_t *G, uint64_t nGenome, uint32_t GstrandBit, uint32_t saIndexNbases,
                                uint8_t *SA, uint64_t saCap, uint8_t *SAi, uint64_t saiCap, uint64_t *out /* nSA nSAbyte nSAi nSAibyte rounds saiStart[17] */) {
    LoopBackend be;
    BuildParams P; P.nGenome = nGenome; P.GstrandBit = GstrandBit; P.saIndexNbases = saIndexNbases;
    BuildResult R;
    int rc = buildAll(be, G, P, SA, saCap, SAi, saiCap, R);
    out[0] = R.nSA; out[1] = R.nSAbyte; out[2] = R.nSAi; out[3] = R.nSAibyte; out[4] = R.rounds;
    for (int i = 0; i < 17; i++) out[5 + i] = R.saiStart[i];
    return rc;
}

// junction insertion: the device algorithm (star_amd/csrc/index/sjdb_core.h) on the plain-loop backend; same signature as staramd_sjdb_insert
extern "C" int sjdb_emul_insert(int, const staramd_sjdb_args *a, staramd_sjdb_result *res) {
    LoopBackend be;
    SjdbParams P; P.nGenomeOld = a->nGenomeOld; P.nGenomeReal = a->nGenomeReal; P.nSAold = a->nSAold; P.GstrandBit = a->GstrandBit;
    P.sjdbN = a->sjdbN; P.sjdbLength = a->sjdbLength; P.oldSjdbN = a->oldSjdbN; P.sjNew = a->sjNew; P.saIndexNbases = a->gSAindexNbases;
    SjdbHostArgs A{a->G, a->SA, a->nSAbyteOld, a->Gsj, a->isOld, a->oldSJind, a->SAout, a->saOutCapacity, a->SAiOut, a->saiOutCapacity};
    u64 nInd = 0, nb = 0, nbi = 0;
    int rc = sjdbInsertHost(be, P, A, nInd, nb, nbi);
    memset(res, 0, sizeof(*res));
    res->nInd = nInd; res->nSAnew = a->nSAold + nInd; res->nSAbyteNew = nb; res->nSAibyte = nbi;
    return rc;
}
