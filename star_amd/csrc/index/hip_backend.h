// hip_backend.h -- the HIP backend of index_core.h / sjdb_core.h: element-wise passes are one thread per element (the grids are >> 256 CUs),
// sorts and scans are rocPRIM's device-wide primitives (onesweep radix sort: every pass streams keys + values once through HBM).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include "index_core.h"

namespace staridx {

// one thread per element of [base, base + n); a launch carries at most 2^31 work-items (the dispatch packet counts them in 32 bits)
template <class F> __global__ void __launch_bounds__(256) k_forEach(u64 base, u64 n, F f) {
    u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n) f(base + i);
}

struct HipBackend {
    hipStream_t s = nullptr;
    hipError_t err = hipSuccess;
    const char *where = "";
    void *tmp = nullptr; size_t tmpBytes = 0;
    u64 liveBytes = 0, peakBytes = 0;

    void chk(hipError_t e, const char *w) { if (e != hipSuccess && err == hipSuccess) { err = e; where = w; } }
    // STARAMD_VERBOSE: wall time of every stage of an index operation (a stage ends with a stream synchronisation then)
    bool verbose = getenv("STARAMD_VERBOSE") != nullptr; double tStage = 0;
    static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
    void stage(const char *name) {
        if (!verbose) return;
        chk(hipStreamSynchronize(s), "stage sync");
        const double t = now();
        if (name && tStage > 0) fprintf(stderr, "staramd index stage %-28s %9.1f ms\n", name, (t - tStage) * 1e3);
        tStage = t;
    }
    template <class T> T *alloc(u64 n) {
        void *p = nullptr; size_t b = std::max<u64>(n * sizeof(T), 256);
        chk(hipMalloc(&p, b), "hipMalloc");
        return (T *)p;
    }
    void free(void *p) { if (p) chk(hipFree(p), "hipFree"); }
    template <class F> void forEach(u64 n, F f) {
        if (n == 0 || err != hipSuccess) return;
        const u64 SLICE = 1ull << 31;
        for (u64 base = 0; base < n; base += SLICE) {
            const u64 m = n - base < SLICE ? n - base : SLICE;
            hipLaunchKernelGGL(k_forEach<F>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, base, m, f);
        }
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


} // namespace staridx
