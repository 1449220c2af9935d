// Stand-in for <hip/hip_runtime.h> when the engine's kernels are compiled for the HOST by the wavefront emulator (oracle/wave_emul/emu.h).
// TEST INFRASTRUCTURE: found only with -I oracle/wave_emul; the product is compiled by hipcc against the real header.
#pragma once
#include "../emu.h"
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <cmath>
using std::fmin; using std::fmax;

#define STARAMD_WAVE_EMUL 1

// ---- language ----------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ thread_local          // one OS thread runs the blocks of a launch one after the other: block-shared == thread-local storage
typedef emu::Dim3 dim3;
#define threadIdx (emu::cur->tIdx)
#define blockIdx (emu::cur->bIdx)
#define blockDim (emu::cur->bDim)
#define gridDim (emu::cur->gDim)

template <class A, class B, class = std::enable_if_t<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>>
constexpr std::common_type_t<A, B> min(A a, B b) { using C = std::common_type_t<A, B>; return (C)a < (C)b ? (C)a : (C)b; }
template <class A, class B, class = std::enable_if_t<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>>
constexpr std::common_type_t<A, B> max(A a, B b) { using C = std::common_type_t<A, B>; return (C)a > (C)b ? (C)a : (C)b; }

// ---- wave operations -----------------------------------------------------------------------------------------------------------
static __forceinline__ unsigned long long __ballot(int p) {
    emu::Exchange e = emu::exchange(p ? 1u : 0u, emu::K_BALLOT);
    unsigned long long m = 0; for (int l = 0; l < 64; l++) if (((e.active >> l) & 1) && e.val[l]) m |= 1ull << l;
    return m;
}
static __forceinline__ int __any(int p) { return __ballot(p) != 0; }
static __forceinline__ int emu_readlane(int v, int l) { emu::Exchange e = emu::exchange((uint32_t)v, emu::K_READLANE); return (int)e.val[l & 63]; }
static __forceinline__ int emu_readfirstlane(int v) { emu::Exchange e = emu::exchange((uint32_t)v, emu::K_READFIRST); return (int)e.val[__builtin_ctzll(e.active)]; }
static __forceinline__ int __shfl(int v, int src, int width = 64) { (void)width; emu::Exchange e = emu::exchange((uint32_t)v, emu::K_SHFL); return (int)e.val[src & 63]; }
static __forceinline__ int __shfl_xor(int v, int mask, int width = 64) { (void)width; emu::Exchange e = emu::exchange((uint32_t)v, emu::K_SHFL); return (int)e.val[(emu::cur->lane ^ (unsigned)mask) & 63]; }
// v_mov_b32_dpp: the control codes the kernels use -- row_shr:n (0x111..0x11f), wave_shl:1 (0x130), wave_shr:1 (0x138), row_bcast:15 (0x142), row_bcast:31 (0x143)
static __forceinline__ int emu_update_dpp(int old, int src, int ctrl, int rowMask, int bankMask, bool boundCtrl) {
    emu::Exchange e = emu::exchange((uint32_t)src, emu::K_DPP);
    const int lane = (int)emu::cur->lane, row = lane >> 4, inRow = lane & 15;
    if (!((rowMask >> row) & 1) || !((bankMask >> (inRow >> 2)) & 1)) return old;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11f) { int n = ctrl - 0x110; if (inRow >= n) from = lane - n; }
    else if (ctrl == 0x130) { if (lane < 63) from = lane + 1; }
    else if (ctrl == 0x138) { if (lane > 0) from = lane - 1; }
    else if (ctrl == 0x142) { if (row > 0) from = row * 16 - 1; }
    else if (ctrl == 0x143) { if (lane >= 32) from = 31; }
    else { fprintf(stderr, "wave emulator: DPP control 0x%x is not modelled\n", ctrl); abort(); }
    if (from < 0 || !((e.active >> from) & 1)) return boundCtrl ? 0 : old;
    return (int)e.val[from];
}
#define __builtin_amdgcn_readlane(v, l) emu_readlane((v), (l))
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane((v))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
static inline unsigned emu_mbcnt_lo(unsigned mask, unsigned add) { unsigned l = emu::cur->lane; return add + (unsigned)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u))); }
static inline unsigned emu_mbcnt_hi(unsigned mask, unsigned add) { unsigned l = emu::cur->lane; return add + (l <= 32 ? 0u : (unsigned)__builtin_popcount(mask & ((1u << (l - 32)) - 1u))); }
#define __builtin_amdgcn_mbcnt_lo(m, a) emu_mbcnt_lo((m), (a))
#define __builtin_amdgcn_mbcnt_hi(m, a) emu_mbcnt_hi((m), (a))
// On the device the lanes of a wavefront advance in lock step, so a store of one lane is ordered against later loads of the others by a
// fence alone.  Here the lanes run one after the other between two rendezvous: every fence is made a rendezvous of the wavefront, which
// gives the same order (all lanes finish what precedes the fence before any lane goes on).
static __forceinline__ void emu_wave_sync() { (void)emu::exchange(0u, emu::K_FENCE); }
#define __builtin_amdgcn_fence(order, scope) emu_wave_sync()
#define __builtin_amdgcn_wave_barrier() emu_wave_sync()
static __forceinline__ void __threadfence_block() { emu_wave_sync(); }
static __forceinline__ void __threadfence() { emu_wave_sync(); }
static inline void __syncthreads() { emu::blockBarrier(); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }

// ---- atomics: the work-items of a launch run on one OS thread; two engine contexts on two OS threads own disjoint buffers ----------
template <class T, class V> static inline T atomicAdd(T *p, V v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class V> static inline T atomicOr(T *p, V v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class V> static inline T atomicMax(T *p, V v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class V> static inline T atomicMin(T *p, V v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class P, class V> static inline auto emu_fetch_or(P p, V v) { auto o = *p; *p = o | v; return o; }
#define __HIP_MEMORY_SCOPE_WAVEFRONT 1
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __HIP_MEMORY_SCOPE_SYSTEM 4
#define __hip_atomic_fetch_or(p, v, order, scope) emu_fetch_or((p), (v))
#define __hip_atomic_load(p, order, scope) (*(p))
template <class P, class V> static inline auto emu_fetch_max(P p, V v) { auto o = *p; if ((decltype(o))v > o) *p = (decltype(o))v; return o; }
template <class P, class E, class V> static inline bool emu_cas(P p, E *expected, V desired) { if (*p == *expected) { *p = desired; return true; } *expected = *p; return false; }
#define __hip_atomic_fetch_max(p, v, order, scope) emu_fetch_max((p), (v))
#define __hip_atomic_compare_exchange_strong(p, e, d, so, fo, scope) emu_cas((p), (e), (d))

// ---- runtime -----------------------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
struct emuStream { int unused; };
typedef emuStream *hipStream_t;
struct emuEvent { std::chrono::steady_clock::time_point t; };
typedef emuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
struct hipDeviceProp_t { char name[256]; size_t totalGlobalMem; int multiProcessorCount; size_t sharedMemPerBlock; };
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (wave emulator)" : "error (wave emulator)"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "wave emulator"); p->totalGlobalMem = (size_t)64 << 30; p->multiProcessorCount = 2; p->sharedMemPerBlock = 65536; return hipSuccess; }
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) { *n = 1; return hipSuccess; }
// device memory is not cleared by hipMalloc: a recognisable pattern instead of zeros, so that a kernel that reads what nobody wrote shows up
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); if (*p) memset(*p, 0xA5, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)48 << 30; *t = (size_t)64 << 30; return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { static emuStream one; *s = &one; return hipSuccess; }      // (everything runs in order anyway)
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *, uint32_t, const uint32_t *) { return hipErrorInvalidValue; }      // (the engine falls back to a plain stream)
static inline hipError_t hipStreamWaitEvent(hipStream_t, struct emuEvent *, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emuEvent(); return hipSuccess; }
enum { hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emuEvent(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu::launch(#kernel, dim3(grid), dim3(block), (size_t)(shmem), [=]() { kernel(__VA_ARGS__); })
