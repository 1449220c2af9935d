// emu.cpp -- fibers, rendezvous and launch of the wavefront emulator (see emu.h; test infrastructure only)
#include "emu.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <vector>
#include <algorithm>
#include <chrono>
#include <csignal>
#include <dlfcn.h>
#include <ucontext.h>
#include <unistd.h>

// void emu_switch(void **saveSp, void *loadSp): callee-saved registers of the System V x86-64 ABI
extern "C" void emu_switch(void **saveSp, void *loadSp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

extern thread_local uint32_t ldsTab[], ldsReads[];       // emu_lds.cpp

// AddressSanitizer build (oracle/wave_emul/build.sh asan): the sanitizer has to be told about every change of stack
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void **fakeStackSave, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fakeStackSave, const void **bottomOld, size_t *sizeOld);
extern "C" void __asan_unpoison_memory_region(void const volatile *addr, size_t size);
#endif
#endif

namespace emu {

thread_local Fiber *cur = nullptr;

struct Wave {
    unsigned nLanes = 0; uint64_t active = 0;
    struct Rec { uint64_t seq = ~0ull; unsigned arrived = 0; int kind = 0; void *site = nullptr; unsigned need = 0; uint64_t active = 0; uint32_t val[64]; } ring[8];
};
struct Block { unsigned nThreads = 0, alive = 0; uint64_t barSeq = ~0ull; unsigned barArrived = 0; uint64_t barDone = 0; };

static const size_t STACK_BYTES = 1u << 20;
struct Sched {
    void *mainSp = nullptr;
    std::vector<Fiber> fibers; std::vector<Wave> waves; Block block;
    std::vector<void *> stacks;
    const std::function<void()> *body = nullptr;
    // STARAMD_EMUL_ORDER=desc: the lanes run in descending order between two rendezvous.  Results must not depend on the order: a difference
    // means that lanes exchange data through memory without a fence / wave barrier in between (fine in lock step on the device, wrong here).
    // STARAMD_EMUL_HASHLOG=<file>: one line per completed rendezvous with a hash of the block's LDS -- diff the logs of two orders to find the place.
    bool descending = getenv("STARAMD_EMUL_ORDER") && !strcmp(getenv("STARAMD_EMUL_ORDER"), "desc");
    FILE *hashLog = getenv("STARAMD_EMUL_HASHLOG") ? fopen(getenv("STARAMD_EMUL_HASHLOG"), "w") : nullptr;
    size_t ldsBytes = 0; const char *kernel = "";
    bool strictSites = getenv("STARAMD_EMUL_STRICT_SITES") != nullptr;
    unsigned left = 0;
    uint64_t progress = 0, switches = 0, limit = getenv("STARAMD_EMUL_MAX_SWITCHES") ? strtoull(getenv("STARAMD_EMUL_MAX_SWITCHES"), nullptr, 10) : 0;
};
static thread_local Sched *S = nullptr;
static Sched &sched() { if (!S) S = new Sched(); return *S; }

// A lane that has to wait hands over to the next live lane of its wavefront directly; the last one of the pass returns to the scheduler
// (which looks for progress, other wavefronts, the end of the block).
#ifdef EMU_ASAN
static thread_local const void *mainStackBottom = nullptr; static thread_local size_t mainStackSize = 0;
static void switchStacks(void **saveSp, void *loadSp, Fiber *to) {           // to == nullptr: back to the scheduler's (the OS thread's) stack
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, to ? (const char *)to->stack : (const char *)mainStackBottom, to ? STACK_BYTES : mainStackSize);
    emu_switch(saveSp, loadSp);
    const void *b; size_t n; __sanitizer_finish_switch_fiber(fake, &b, &n);
    if (!mainStackBottom && b) { mainStackBottom = b; mainStackSize = n; }
}
#else
static inline void switchStacks(void **saveSp, void *loadSp, Fiber *) { emu_switch(saveSp, loadSp); }
#endif

static void yield() {
    Sched &s = *S; Fiber *f = cur;
    const unsigned t = f->tIdx.x, lo = t & ~63u, hi = std::min<unsigned>((unsigned)s.fibers.size(), lo + 64);
    Fiber *nxt = nullptr;
    if (!s.descending) { for (unsigned k = t + 1; k < hi; k++) if (!s.fibers[k].done) { nxt = &s.fibers[k]; break; } }
    else { for (unsigned k = t; k-- > lo;) if (!s.fibers[k].done) { nxt = &s.fibers[k]; break; } }
    s.switches++;
    if (nxt) { cur = nxt; switchStacks(&f->sp, nxt->sp, nxt); }
    else { cur = nullptr; switchStacks(&f->sp, s.mainSp, nullptr); }
}

static void fiberEntry() {
#ifdef EMU_ASAN
    { const void *b; size_t n; __sanitizer_finish_switch_fiber(nullptr, &b, &n); if (!mainStackBottom && b) { mainStackBottom = b; mainStackSize = n; } }   // first time on this stack
#endif
    Sched &s = *S; Fiber *f = cur;
    (*s.body)();
    f->done = true;
    if (f->wave) f->wave->active &= ~(1ull << f->lane);
    f->block->alive--; s.left--;
    s.progress++;
    cur = nullptr;
    switchStacks(&f->sp, s.mainSp, nullptr);
    abort();                                   // a finished fiber is never resumed
}

static void die(const char *what) {
    Sched &s = *S;
    if (s.hashLog) fflush(s.hashLog);
    fprintf(stderr, "wave emulator: %s\n", what);
    for (size_t i = 0; i < s.fibers.size() && i < 256; i++) { const Fiber &f = s.fibers[i]; if (!f.done) fprintf(stderr, "  thread %3zu (block %u): seq %llu waiting for %s\n", i, f.bIdx.x, (unsigned long long)f.seq, f.waitingFor ? f.waitingFor : "-"); }
    abort();
}

static const char *kindName(int k) { static const char *n[] = {"?", "ballot", "readlane", "readfirstlane", "shuffle", "dpp", "fence"}; return k >= 1 && k <= 6 ? n[k] : "?"; }

__attribute__((noinline)) Exchange exchange(uint32_t v, int kind) {
    Fiber *f = cur; Wave &w = *f->wave; Sched &s = *S;
    const uint64_t q = f->seq++;
    Wave::Rec &r = w.ring[q & 7];
    void *site = __builtin_return_address(0);            // the operation's place in the kernel code: the same for every lane of a wavefront
    if (r.seq != q) {
        if (r.seq != ~0ull && (r.seq > q || r.arrived < r.need)) {
            fprintf(stderr, "wave emulator: lane %u is at operation %llu, the slot holds operation %llu with %u of %u arrivals\n", f->lane, (unsigned long long)q, (unsigned long long)r.seq, r.arrived, r.need);
            die("the lanes of a wavefront have lost step");
        }
        r.seq = q; r.arrived = 0; r.kind = kind; r.site = site; r.need = (unsigned)__builtin_popcountll(w.active); r.active = 0; memset(r.val, 0, sizeof(r.val));
    }
    else if (r.site != site && s.strictSites) {          // (off by default: the compiler may duplicate a call site into both arms of a lane-dependent branch)
        Dl_info a, b; memset(&a, 0, sizeof(a)); memset(&b, 0, sizeof(b)); dladdr(r.site, &a); dladdr(site, &b);
        fprintf(stderr, "wave emulator: operation %llu of the wavefront: lane %u is at %s (lib+0x%lx), an earlier lane at %s (lib+0x%lx)\n", (unsigned long long)q, f->lane,
                kindName(kind), (unsigned long)((uintptr_t)site - (uintptr_t)b.dli_fbase), kindName(r.kind), (unsigned long)((uintptr_t)r.site - (uintptr_t)a.dli_fbase));
        f->waitingFor = kindName(kind); die("lanes of one wavefront reached different cross-lane operations (divergent control flow around a wave operation)");
    }
    else if (r.kind != kind) {
        Dl_info a, b; memset(&a, 0, sizeof(a)); memset(&b, 0, sizeof(b)); dladdr(r.site, &a); dladdr(site, &b);
        fprintf(stderr, "wave emulator: %s, operation %llu of wavefront %u: lane %u is at a %s (lib+0x%lx), an earlier lane at a %s (lib+0x%lx)\n", s.kernel, (unsigned long long)q, f->tIdx.x >> 6, f->lane,
                kindName(kind), (unsigned long)((uintptr_t)site - (uintptr_t)b.dli_fbase), kindName(r.kind), (unsigned long)((uintptr_t)r.site - (uintptr_t)a.dli_fbase));
        f->waitingFor = kindName(kind); die("lanes of one wavefront reached different cross-lane operations (divergent control flow around a wave operation)");
    }
    r.val[f->lane] = v; r.arrived++; s.progress++;
    f->waitingFor = kindName(kind);
    while (r.arrived < (unsigned)__builtin_popcountll(w.active)) yield();
    if (!r.active && s.hashLog) {
        uint64_t h = 1469598103934665603ull;
        const unsigned char *a = (const unsigned char *)ldsTab, *b = (const unsigned char *)ldsReads;
        for (size_t i = 0; i < s.ldsBytes; i++) { h = (h ^ a[i]) * 1099511628211ull; h = (h ^ b[i]) * 1099511628211ull; }
        Dl_info di; memset(&di, 0, sizeof(di)); dladdr(site, &di);
        fprintf(s.hashLog, "%s block %u wave %u op %llu %s lib+0x%lx lds %016llx\n", s.kernel, f->bIdx.x, f->tIdx.x >> 6, (unsigned long long)q, kindName(kind), (unsigned long)((uintptr_t)site - (uintptr_t)di.dli_fbase), (unsigned long long)h);
    }
    if (!r.active) r.active = w.active;            // the lanes that took part: fixed by the first lane released (later ones may find that earlier ones have left the kernel)
    f->waitingFor = nullptr;
    return Exchange{r.val, r.active};
}

void blockBarrier() {
    Fiber *f = cur; Block &b = *f->block; Sched &s = *S;
    const uint64_t q = f->bseq++;
    if (b.barSeq != q) { b.barSeq = q; b.barArrived = 0; }
    b.barArrived++; s.progress++;
    f->waitingFor = "__syncthreads";
    while (b.barDone <= q) { if (b.barSeq == q && b.barArrived >= b.alive) { b.barDone = q + 1; s.progress++; break; } yield(); }
    f->waitingFor = nullptr;
}

static void onAlarm(int sig, siginfo_t *si, void *uc) {
    if (sig == SIGSEGV) { char b[96]; int n = snprintf(b, sizeof(b), "wave emulator: SIGSEGV at address %p\n", si->si_addr); (void)!write(2, b, (size_t)n); }          // STARAMD_EMUL_ALARM=<seconds>: where is the work-item that never reaches a rendezvous?
    const uintptr_t rip = (uintptr_t)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
    Dl_info di; memset(&di, 0, sizeof(di)); dladdr((void *)rip, &di);
    char buf[512];
    int n = snprintf(buf, sizeof(buf), "wave emulator: alarm -- running work-item at %s+0x%lx (addr2line -f -e %s 0x%lx), thread %u of block %u, %llu cross-lane operations so far\n",
                     di.dli_sname ? di.dli_sname : "?", (unsigned long)(rip - (uintptr_t)di.dli_saddr), di.dli_fname ? di.dli_fname : "?", (unsigned long)(rip - (uintptr_t)di.dli_fbase),
                     cur ? cur->tIdx.x : 0u, cur ? cur->bIdx.x : 0u, cur ? (unsigned long long)cur->seq : 0ull);
    (void)!write(2, buf, (size_t)n);
    _exit(3);
}

void launch(const char *name, Dim3 grid, Dim3 block, size_t shmemBytes, const std::function<void()> &body) {
    Sched &s = sched();
    static const int alarmS = getenv("STARAMD_EMUL_ALARM") ? atoi(getenv("STARAMD_EMUL_ALARM")) : 0;
    if (alarmS > 0) {
        static thread_local char altStack[1 << 16];
        stack_t ss; ss.ss_sp = altStack; ss.ss_size = sizeof(altStack); ss.ss_flags = 0; sigaltstack(&ss, nullptr);
        struct sigaction sa; memset(&sa, 0, sizeof(sa)); sa.sa_sigaction = onAlarm; sa.sa_flags = SA_SIGINFO | SA_ONSTACK; sigaction(SIGALRM, &sa, nullptr); sigaction(SIGSEGV, &sa, nullptr);
        alarm((unsigned)alarmS);
    }
    static const bool trace = getenv("STARAMD_EMUL_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t switches0 = s.switches;
    if (trace) fprintf(stderr, "emu: > %s\n", name);
    if (cur) { fprintf(stderr, "wave emulator: nested launch\n"); abort(); }
    if (shmemBytes > 160 * 1024) { fprintf(stderr, "wave emulator: %zu bytes of dynamic LDS requested, 160 KB exist\n", shmemBytes); abort(); }
    const unsigned nT = block.x * block.y * block.z;
    if (block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) { fprintf(stderr, "wave emulator: one-dimensional launches only\n"); abort(); }
    while (s.stacks.size() < nT) {
        void *p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { perror("wave emulator: mmap"); abort(); }
        mprotect(p, 4096, PROT_NONE);                                          // guard page under the stack
        s.stacks.push_back(p);
    }
    s.body = &body; s.ldsBytes = shmemBytes; s.kernel = name;
    for (unsigned bx = 0; bx < grid.x; bx++) {
        memset(ldsTab, 0xCD, shmemBytes); memset(ldsReads, 0xCD, shmemBytes);      // LDS is not cleared between blocks on the device either: a recognisable pattern
        s.fibers.assign(nT, Fiber()); s.waves.assign((nT + 63) / 64, Wave());
        s.block = Block(); s.block.nThreads = s.block.alive = nT;
        for (unsigned t = 0; t < nT; t++) {
            Fiber &f = s.fibers[t];
            f.tIdx = Dim3(t); f.bIdx = Dim3(bx); f.bDim = block; f.gDim = grid;
            f.lane = t & 63; f.wave = &s.waves[t >> 6]; f.block = &s.block;
            f.wave->nLanes++; f.wave->active |= 1ull << f.lane;
            f.stack = s.stacks[t];
#ifdef EMU_ASAN
            __asan_unpoison_memory_region((char *)f.stack + 4096, STACK_BYTES - 4096);      // a work-item that left its kernel never unwound its frames: their red zones are stale
#endif
            // initial frame: six callee-saved registers, then the address `ret` jumps to; at fiberEntry the stack is 8 mod 16 as after a call
            uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
            uint64_t *slot = (uint64_t *)(top - 16);
            slot[0] = (uint64_t)(uintptr_t)&fiberEntry; slot[1] = 0;
            uint64_t *sp = slot - 6;
            for (int k = 0; k < 6; k++) sp[k] = 0;
            f.sp = sp;
        }
        // a wavefront runs until it has left the kernel or all its lanes wait for the other wavefronts of the block (__syncthreads), then the next
        // one: what a wavefront finds in memory does not depend on how far the others happen to be (nor does the LDS hash of STARAMD_EMUL_HASHLOG)
        s.left = nT;
        const unsigned nW = (nT + 63) / 64;
        while (s.left) {
            const uint64_t before = s.progress;
            for (unsigned w = 0; w < nW; w++) {
                const unsigned lo = w * 64, hi = std::min(nT, lo + 64);
                for (;;) {
                    const uint64_t b2 = s.progress;
                    Fiber *first = nullptr;                                   // the first live lane in scheduling order; the others follow through yield()
                    if (!s.descending) { for (unsigned k = lo; k < hi; k++) if (!s.fibers[k].done) { first = &s.fibers[k]; break; } }
                    else { for (unsigned k = hi; k-- > lo;) if (!s.fibers[k].done) { first = &s.fibers[k]; break; } }
                    if (!first) break;
                    cur = first; s.switches++;
                    switchStacks(&s.mainSp, first->sp, first);
                    cur = nullptr;
                    if (s.limit && s.switches - switches0 > s.limit) die("STARAMD_EMUL_MAX_SWITCHES exceeded in one launch");
                    if (s.progress == b2) break;
                }
            }
            if (s.left && s.progress == before) die("no work-item can make progress (a rendezvous that not all lanes reach, or a spin-wait on another block)");
        }
    }
    s.body = nullptr;
    if (trace) fprintf(stderr, "emu: %-28s grid %6u x %4u  lds %6zu B  %9.1f ms  %llu fiber switches\n", name, grid.x, block.x, shmemBytes,
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), (unsigned long long)(s.switches - switches0));
}

}  // namespace emu
