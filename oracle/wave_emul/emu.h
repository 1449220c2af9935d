// emu.h -- wavefront emulator: TEST INFRASTRUCTURE, never part of the product.
//
// The HIP kernels of star_amd/csrc/engine are compiled as plain C++ for the host (clang++, -I oracle/wave_emul puts the stand-in
// <hip/hip_runtime.h> of this directory in front of the real one) and linked into oracle/_build/libstaramd_emul.so, which exports the
// engine's C ABI.  The `-m "not gpu"` tests run data sets through it and compare the result buffers with the oracle: the logic of the
// kernel SOURCE that ships is exercised on a machine without a GPU.  What it cannot show: memory ordering, address spaces, occupancy,
// speed -- those are the `-m gpu` tests and the bench.
//
// Model: one work-item = one fiber (user-level context, own stack); the fibers of a block are scheduled round-robin on the calling OS
// thread, blocks of a grid run one after the other (the kernels are persistent and ticket-driven, or independent per element, so a
// grid of any size produces the same results).  A cross-lane operation (ballot, readlane, DPP, shuffle) is a rendezvous of the 64
// fibers of a wavefront: every lane deposits its operand, the last arrival releases them, each computes its own result from the 64
// operands.  All lanes of a wavefront must reach the same sequence of cross-lane operations -- the kernels keep control flow around
// them wave-uniform; the emulator checks the kind of every operation and aborts on a mismatch or on a rendezvous that cannot complete.
#pragma once
#include <cstdint>
#include <cstddef>
#include <functional>

namespace emu {

struct Dim3 { unsigned x, y, z; Dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

struct Wave;
struct Block;
struct Fiber {
    Dim3 tIdx, bIdx, bDim, gDim;
    unsigned lane = 0;
    Wave *wave = nullptr; Block *block = nullptr;
    void *sp = nullptr; void *stack = nullptr;
    bool done = false;
    uint64_t seq = 0;            // cross-lane operations performed so far
    uint64_t bseq = 0;           // block barriers passed so far
    const char *waitingFor = nullptr;
};
extern thread_local Fiber *cur;

struct Exchange { const uint32_t *val; uint64_t active; };
enum Kind { K_BALLOT = 1, K_READLANE, K_READFIRST, K_SHFL, K_DPP, K_FENCE };
Exchange exchange(uint32_t v, int kind);
void blockBarrier();
void launch(const char *name, Dim3 grid, Dim3 block, size_t shmemBytes, const std::function<void()> &body);     // STARAMD_EMUL_TRACE=1: one line per launch

}  // namespace emu
