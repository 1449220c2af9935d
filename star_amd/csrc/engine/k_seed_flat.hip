// k_seed_flat.hip -- kernel 1, experimental forms: the seed search with ONE memory site (STARAMD_SEED_FLAT=1..6; k_seed.hip is the measured default; nothing here
// has run on hardware yet: bench.py's leg `variants` times all of them, DESIGN.md 5.1.1 has the trace model that motivates them).
// Second form (below): a state machine per search (1 / 2 / 3 = 8 / 6 / 4 waves per SIMD).  Third form (end of the file): a state machine over the whole read
// (4 / 5 = 4 / 6 waves per SIMD; 6 = with the read staged in LDS).
//
// Same functions of the reference as k_seed.hip (ReadAlign_maxMappableLength2strands.cpp:5-115, SuffixArrayFuns.cpp:10-207,
// ReadAlign_storeAligns.cpp:10-160), same mapping (one lane = one read), same results bit for bit.  What differs is the shape of the
// control flow below one search.  In k_seed.hip a search is the reference's call tree: SAindex look-ups, then three bisections one after the
// other (maxMappableLength's main loop, findMultRange twice), each of them calling compareSeqToGenome, which loops over 8-byte words.  A
// wavefront executes a nest of data-dependent loops as "every level waits for its slowest lane": per search the 64 lanes pay
// (max bisection steps) x (max words per compare) round trips, three times over, where each lane needs only the SUM of its own -- rocprofv3
// counted 23 k vector-memory instructions per wavefront of reads where ~1.5 k per lane are needed, at 31 % VALU-busy and half
// of the measured gather ceiling of the memory system.
// Here the compares and bisections of a lane's search are a state machine around a single load site: every trip of the one loop issues, for all
// lanes at once, one funnel load (two aligned 8-byte words) from wherever the lane's state points -- a packed suffix-array entry or 8 genome
// bases -- plus the 8 read bases that go with it, waits once, and then lets every lane take its own transition (extract the entry, compare
// 8 bases, deliver a compare result to the bisection it belongs to, pick the next probe).  Lanes never wait for each other inside a search:
// the wavefront makes as many trips as its longest lane has dependent loads.  The transitions are straight-line code in phase order
// (L1 -> L2 -> MAIN -> F1 -> F2), so a lane falls through several of them in one trip when no load lies between.  The SAindex look-ups before
// it (one to three per search) are a loop of their own with one load site.
// Suffix-array intervals of 2^32 entries or more (a look-up whose upper neighbour is absent can span the array) keep the call-tree code.
//
// Parity: result buffers and the counters nSAi / nSAprobe / nGcmp / nSeeds identical to k_seed_search in the emulator (battery of 14 data set x flag cases, 290
// fuzz combinations, AddressSanitizer build); forced cases `seed_*` in tests/test_wave_emul.py (emulator) and tests/test_gpu_parity.py (hardware).
#define k_seed_search k_seed_search_calltree_copy      // the device functions of k_seed.hip are reused as they are; its kernel entry is compiled under another name in this object
#include "k_seed.hip"
#undef k_seed_search

// 64 bits from bit `sh` (0..63) of the 16 bytes at the 8-byte aligned address `a`; both words are always loaded (index arrays and read buffers are padded)
__device__ __forceinline__ u64 funnel64(u64 a, u32 sh) {
    const __attribute__((address_space(1))) u64 *q = GLOBAL(u64, a);
    const u64 w0 = q[0], w1 = q[1];
    return sh ? ((w0 >> sh) | (w1 << (64u - sh))) : w0;
}

// an interval of 2^32 suffix-array entries or more: the call-tree search of k_seed.hip with 64-bit indices, kept out of line (rare; its registers and
// code stay out of the loop of the state machine)
__device__ __attribute__((noinline)) static u64 searchWide(const DevIndex &X, const u8 *R, u32 S, u32 N, u64 i1, u64 i2, bool dirR, u32 &L, u64 &ind0, SeedCnt &cn) {
    u64 ind1;
    return maxMappableLengthT<u64>(X, R, S, N, i1, i2, dirR, L, ind0, ind1, cn);
}

enum { PH_DONE = 0, PH_SINGLE, PH_L1, PH_L2, PH_MAIN, PH_F1, PH_F2 };

// Emulated builds only (oracle/wave_emul): STARAMD_SEED_TRACE=<file> logs, per search, the loads it makes -- "S read piece dir start step dist" opens a search,
// "a" is a SAindex look-up, "c phase words" a compare (1 suffix-array load + `words` 8-base steps).  tools/seed_divergence.py turns the log into the number of
// load round trips a wavefront makes under the call-tree control flow, under this state machine, and under a state machine over the whole read.
#ifdef STARAMD_WAVE_EMUL
#include <cstdio>
#include <cstdlib>
#include <string>
struct SeedTrace { FILE *f; u32 ir, piece, dir, start, step; SeedTrace() : f(getenv("STARAMD_SEED_TRACE") ? fopen(getenv("STARAMD_SEED_TRACE"), "w") : nullptr), ir(0), piece(0), dir(0), start(0), step(0) {} ~SeedTrace() { if (f) fclose(f); } };
static SeedTrace g_seedTrace;
#define SEED_TRACE(...) do { if (g_seedTrace.f) fprintf(g_seedTrace.f, __VA_ARGS__); } while (0)
#define SEED_TRACE_SET(field, v) (g_seedTrace.field = (v))
#else
#define SEED_TRACE(...) ((void)0)
#define SEED_TRACE_SET(field, v) ((void)0)
#endif

// one start offset of one seed (searchOneDist of k_seed.hip); see the head of the file.  Two loops with one load site each: the SAindex look-ups
// (one to three trips), then the compare machine.  The state that a trip carries is kept small on purpose (the kernel is held to 64 VGPRs for 8
// waves per SIMD): what can be derived from the phase is derived (the target length of a compare, min(L1, L2)), the results are computed from the
// machine's own variables at the end.
__device__ __forceinline__ void searchOneDistFlat(const DevIndex &X, const u8 *R, u32 pieceStartIn, u32 pieceLengthIn, bool dirR, u32 iDist, u64 &Nrep, u64 &i0, u32 &maxL, SeedCnt &cn) {
    const u32 pieceLength = pieceLengthIn - iDist;
    const u32 pieceStart = dirR ? pieceStartIn + iDist : pieceStartIn - iDist;
    const u32 Lmax = min(X.saiNbases, pieceLength);
    u64 ind1 = 0;
    for (u32 ii = 0; ii < Lmax; ii += 8) {               // L-mer prefix, as in k_seed.hip
        const u64 raw = dirR ? load8(R + pieceStart + ii) : load8rev(R + pieceStart - ii);
        const u32 nb = min(8u, Lmax - ii);
        const u64 used = nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
        if ((raw & used & 0xFCFCFCFCFCFCFCFCull) == 0) {
            u64 x = dirR ? raw : (raw ^ 0x0303030303030303ull);
            u64 z = __builtin_bswap64(x & 0x0303030303030303ull);
            z = (z | (z >> 6)) & 0x000F000F000F000Full;
            z = (z | (z >> 12)) & 0x000000FF000000FFull;
            z = (z | (z >> 24)) & 0xFFFFull;
            ind1 = (ind1 << (2 * nb)) | (z >> (2 * (8 - nb)));
        } else {
            for (u32 k = 0; k < nb; k++) { const u64 cde = (raw >> (8 * k)) & 0xFFull; ind1 = (ind1 << 2) + (dirR ? cde : 3ull - cde); }
        }
    }
    SEED_TRACE("S %u %u %u %u %u %u\n", g_seedTrace.ir, g_seedTrace.piece, g_seedTrace.dir, g_seedTrace.start, g_seedTrace.step, iDist);
    // ---- loop A: the SAindex look-ups (ReadAlign_maxMappableLength2strands.cpp:39-75).  sa: 1 = first entry wanted, 2 = its upper neighbour, 0 = done
    u32 Lind = Lmax; u64 iSA1 = 0, iSA2 = 0; bool iSA2good = true;
    const u64 saiEnd = X.saiStart[X.saiNbases];
    while (Lind > 0 && X.saiStart[Lind - 1] + ind1 >= saiEnd) { --Lind; ind1 >>= 2; }     // outside the table: shortened without a load (treated as absent)
    u32 sa = Lind > 0 ? 1u : 0u;
    while (sa != 0) {
        const u64 b = (X.saiStart[Lind - 1] + ind1 + (sa == 2u ? 1ull : 0ull)) * X.saiBits;
        const u64 v = funnel64((u64)X.SAi + (b >> 6) * 8ull, (u32)(b & 63ull)) & X.saiMask;
        cn.nSAi++;
        SEED_TRACE("a\n");
        if (sa == 1u) {
            iSA1 = v;
            if (iSA1 & X.saiAbsentBit) {
                --Lind; ind1 >>= 2;
                while (Lind > 0 && X.saiStart[Lind - 1] + ind1 >= saiEnd) { --Lind; ind1 >>= 2; }
                if (Lind == 0) sa = 0;
            } else if (X.saiStart[Lind - 1] + ind1 + 1 < X.saiStart[Lind]) sa = 2;
            else { iSA2 = X.nSA - 1; iSA2good = false; sa = 0; }
        } else {
            if ((v & X.saiAbsentBit) == 0) iSA2 = (v & ~X.saiNbit) - 1;
            else { iSA2 = X.nSA - 1; iSA2good = false; }
            sa = 0;
        }
    }
    Nrep = 0; i0 = 0; maxL = 0;
    if (Lind == 0) return;                               // base absent from the genome (reference: out-of-bounds)
    const bool iSA1noN = (iSA1 & X.saiNbit) == 0;
    if (Lind < X.saiNbases && iSA1noN && iSA2good) { i0 = iSA1; Nrep = iSA2 - iSA1 + 1; maxL = Lind; return; }
    // ---- loop B: the compares and the three bisections around them
    u32 ph; u32 cL;                                      // cL: bases known to be equal when the compare in flight starts (its L)
    u32 i1 = 0, i2 = 0;
    const u64 base = iSA1 & ~X.saiNbit;                  // suffix-array indices of the machine are offsets from it
    if (iSA1 == iSA2 && iSA1noN && iSA2good) { ph = PH_SINGLE; cL = Lind; }
    else {
        maxL = (iSA2good && iSA1noN) ? Lind : 0;
        if (iSA2 - base >= 0xFFFFFFFFull) { Nrep = searchWide(X, R, pieceStart, pieceLength, base, iSA2, dirR, maxL, i0, cn); return; }   // (rare)
        i2 = (u32)(iSA2 - base); ph = PH_L1; cL = maxL;
    }
    u32 L1 = 0, L2 = 0, L3 = 0, L1a = 0, L1b = 0, L2a = 0, L2b = 0;
    u32 i3 = 0, i1a = 0, i1b = 0, i2a = 0, i2b = 0;
    u32 cI = 0, ii = 0; bool haveSA = false, dirG = true, compRes = false;      // compare in flight: suffix cI; haveSA: its position is known, ii bases of it are compared
    u64 gAddr = 0;
    const u64 sBase = (u64)R + pieceStart;
    while (ph != PH_DONE) {
        const u32 cN = ph >= PH_F1 ? L3 : pieceLength;   // length the compare in flight runs to
        // ---- the one memory site: 64 bits of the suffix array or of the genome, and the 8 read bases of the compare step
        u64 a; u32 sh;
        if (!haveSA) { const u64 b = (base + cI) * X.saBits; a = (u64)X.SA + (b >> 6) * 8ull; sh = (u32)(b & 63ull); }
        else { const u64 p = dirG ? gAddr + ii : gAddr - ii - 7ull; a = p & ~7ull; sh = (u32)(p & 7ull) * 8u; }
        const u64 sp = !haveSA ? sBase : dirR ? sBase + cL + ii : sBase - cL - ii - 7ull;      // (no compare step in this trip: any address inside the read; `ii` is the previous compare's)
        const u64 v = funnel64(a, sh);
        u64 s8 = funnel64(sp & ~7ull, (u32)(sp & 7ull) * 8u);
        // ---- transitions
        bool fin = false; u32 Lc = cN;                   // fin: the compare ended, Lc bases are equal
        if (!haveSA) {
            cn.nSAprobe++;
            u64 SAstr = v & X.saMask;
            dirG = (SAstr >> X.strandBit) == 0;
            SAstr &= X.strandMask;
            gAddr = dirG ? (u64)X.G + SAstr + cL : (u64)X.G + (X.nGenome - 1 - SAstr) - cL;
            haveSA = true; ii = 0;
            fin = cN == cL;
        } else {
            const u32 n = cN - cL;
            if (!dirR) s8 = __builtin_bswap64(s8);
            const u64 g8 = dirG ? v : __builtin_bswap64(v);
            if (dirR != dirG) s8 = comp8(s8);
            u64 d = s8 ^ g8;
            const u32 rem = n - ii;
            if (rem < 8) d &= (1ull << (rem * 8)) - 1ull;
            if (d) {
                const u32 k = (u32)__builtin_ctzll(d) >> 3;
                const u8 sc = (u8)(s8 >> (k * 8)), gc = (u8)(g8 >> (k * 8));
                cn.nGcmp += ii + k + 1;
                compRes = dirG ? (sc > gc) : !(sc > gc || gc > 3);
                fin = true; Lc = ii + k + cL;
            } else { ii += 8; if (ii >= n) { cn.nGcmp += n; fin = true; } }
        }
        if (fin) {
            haveSA = false;
            bool brk = false;
            SEED_TRACE("c %u %u\n", ph, (cN - cL) == 0 ? 0u : (Lc == cN ? (cN - cL + 7) / 8 : (Lc - cL) / 8 + 1));
            // the compare belongs to ...
            if (ph == PH_SINGLE) { L3 = Lc; ph = PH_DONE; }                                             // (i1 = i2 = 0: one suffix)
            else if (ph == PH_L1) { L1 = Lc; cI = i2; ph = PH_L2; }                                    // (cL stays)
            else if (ph == PH_L2) {
                L2 = Lc;
                L1a = L1; L1b = L1; i1a = i1; i1b = i1; L2a = L2; L2b = L2; i2a = i2; i2b = i2;
                i3 = i1; L3 = L1; ph = PH_MAIN;
            } else if (ph == PH_MAIN) {
                L3 = Lc;
                if (L3 == pieceLength) brk = true;
                else if (compRes) { if (L3 > L1) { L1b = L1a; L1a = L1; i1b = i1a; i1a = i1; } i1 = i3; L1 = L3; }
                else { if (L3 > L2) { L2b = L2a; L2a = L2; i2b = i2a; i2a = i2; } i2 = i3; L2 = L3; }
            } else if (ph == PH_F1) { if (Lc == L3) i1a = cI; else { i1b = cI; L1b = Lc; } }
            else { if (Lc == L3) i2a = cI; else { i2b = cI; L2b = Lc; } }
            // ... and the next probe, in phase order
            if (ph == PH_MAIN) {
                if (!brk && (u64)i1 + 1 < (u64)i2) { i3 = (u32)medianUint2(i1, i2); cI = i3; cL = min(L1, L2); }
                else {
                    if (L3 < pieceLength) { if (L1 > L2) { i3 = i1; L3 = L1; } else { i3 = i2; L3 = L2; } }
                    if (L1 < L3) { L1b = L1; i1b = i1; i1a = i3; }                                       // findMultRange (SuffixArrayFuns.cpp:106-131), lower end
                    else if (L1a < L1) { L1b = L1a; i1b = i1a; i1a = i1; }
                    ph = PH_F1;
                }
            }
            if (ph == PH_F1) {
                if (((u64)i1b + 1 < (u64)i1a) | ((u64)i1b > (u64)i1a + 1)) { cI = (u32)medianUint2(i1a, i1b); cL = L1b; }
                else {
                    i1 = i1a;
                    if (L2 < L3) { L2b = L2; i2b = i2; i2a = i3; }                                       // upper end
                    else if (L2a < L2) { L2b = L2a; i2b = i2a; i2a = i2; }
                    ph = PH_F2;
                }
            }
            if (ph == PH_F2) {
                if (((u64)i2b + 1 < (u64)i2a) | ((u64)i2b > (u64)i2a + 1)) { cI = (u32)medianUint2(i2a, i2b); cL = L2b; }
                else { i2 = i2a; ph = PH_DONE; }
            }
        }
    }
    maxL = L3; i0 = base + i1; Nrep = (u64)i2 - (u64)i1 + 1;
}

// maxMappableLength2strands of k_seed.hip over the flat search
__device__ static void maxMappableLength2strandsFlat(const DevIndex &X, const u8 *R, SeedState &st, u32 pieceStartIn, u32 pieceLengthIn, u32 iDir, u32 &maxLbest, u32 iFrag, SeedCnt &cn) {
    const bool dirR = iDir == 0;
    const u32 nD = min(pieceLengthIn, X.sparseD);
    maxLbest = 0;
    for (u32 it = 0; it < 2 * nD; it++) {
        const u32 phase = it >= nD ? 1u : 0u, iDist = phase ? it - nD : it;
        u64 Nrep, i0; u32 maxL;
        searchOneDistFlat(X, R, pieceStartIn, pieceLengthIn, dirR, iDist, Nrep, i0, maxL, cn);
        if (phase == 0) {
            if (maxL + iDist > maxLbest) maxLbest = maxL + iDist;
            if (nD > 1) continue;
        }
        if (maxL + iDist == maxLbest && Nrep > 0)
            storeAligns(X, st, iDir, dirR ? pieceStartIn + iDist : pieceStartIn - iDist, Nrep, maxL, i0, iFrag);
        if (nD == 1) break;
    }
}

// the read loop, qualitySplit and the seed schedule are those of k_seed_search (k_seed.hip), statement for statement.  Three register budgets of the same
// body (STARAMD_SEED_FLAT = 1 / 2 / 3): 8 waves per SIMD (64 VGPRs; the compare loop keeps ~20 spill accesses per trip), 6 (80 VGPRs), 4 (128 VGPRs: no
// scratch access inside the loops of a search) -- which of them wins is a question for the hardware (more chains in flight against fewer instructions per step)
__device__ __forceinline__ void seedSearchFlatBody(const DevIndex *__restrict__ Xp, DevBatch B, DSeed *scratch, u32 scratchPerLane) {
    const DevIndex &X = *Xp;
    u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    SeedState st; st.PC = scratch + (u64)lane * scratchPerLane; st.cap = scratchPerLane;
    SeedCnt cn = {0, 0, 0}; u64 nSeedsTot = 0;
    const staramd_params &P = X.P;
    for (;;) {
        u32 ir = atomicAdd(&B.cursors[CUR_TICKET_SEED], 1u);
        if (ir >= B.nReads) break;
        const u8 *R = B.bases + B.readOffset[ir];
        u32 Lread = (u32)(B.readOffset[ir + 1] - B.readOffset[ir]);
        st.nP = 0; st.nA = 0; st.multNmin = 0; st.multNminL = 0; st.fatal = false;
        u32 Nsplit = 0, LgoodMin = 0;
        const u32 seedSearchStartLmax = min(P.seedSearchStartLmax, (u32)(u64)(P.seedSearchStartLmaxOverLread * (double)(u64)(Lread - 1)));
        {
            u32 iR = 0, iFrag = 0;
            while ((iR < Lread) & (Nsplit < P.maxNsplit)) {
                while (iR < Lread && R[iR] > 3) { if (R[iR] == STARAMD_SPACER_BASE) iFrag++; iR++; }
                if (iR == Lread) break;
                const u32 pS = iR;
                for (;;) {
                    const u64 bad = load8(R + iR) & 0xFCFCFCFCFCFCFCFCull;
                    const u32 k = bad ? ((u32)__builtin_ctzll(bad) >> 3) : 8u;
                    iR += k;
                    if (iR >= Lread) { iR = Lread; break; }
                    if (k < 8u) break;
                }
                const u32 pL = iR - pS;
                if (pL > LgoodMin) LgoodMin = pL;
                if (pL < P.seedSplitMin) continue;
                Nsplit++;
                SEED_TRACE_SET(ir, ir); SEED_TRACE_SET(piece, Nsplit);
                const u32 Nstart = (P.seedSearchStartLmax > 0 && seedSearchStartLmax < pL) ? pL / seedSearchStartLmax + 1 : 1;
                const u32 Lstart = pL / Nstart;
                bool flagDirMap = true;
                for (u32 iDir = 0; iDir < 2; iDir++) {
                    for (u32 istart = 0; istart < Nstart; istart++) {
                        SEED_TRACE_SET(dir, iDir); SEED_TRACE_SET(start, istart); SEED_TRACE_SET(step, 0);
                        // the maximal-mappable-prefix walk over the piece (ReadAlign_mapOneRead.cpp:57-79) and, with --seedSearchLmax, one more search of a
                        // fixed length (:81-86): ONE call site, so that the state machine exists once in the kernel
                        u32 Lmapped = 0;
                        bool walk = flagDirMap || istart > 0, lmaxTodo = P.seedSearchLmax > 0;
                        for (;;) {
                            const bool isWalk = walk && (istart * Lstart + Lmapped + P.seedMapMin < pL);
                            u32 Shift, seedLength, Lm;
                            if (isWalk) {
                                Shift = iDir == 0 ? (pS + istart * Lstart + Lmapped) : (pS + pL - istart * Lstart - 1 - Lmapped);
                                seedLength = pL - Lmapped - istart * Lstart;
                            } else if (lmaxTodo) {
                                lmaxTodo = false; walk = false;
                                Shift = iDir == 0 ? (pS + istart * Lstart) : (pS + pL - istart * Lstart - 1);
                                seedLength = min(P.seedSearchLmax, iDir == 0 ? (pS + pL - Shift) : (Shift + 1));
                            } else break;
                            maxMappableLength2strandsFlat(X, R, st, Shift, seedLength, iDir, Lm, iFrag, cn);
                            SEED_TRACE_SET(step, g_seedTrace.step + 1);
                            if (isWalk) {
                                if (iDir == 0 && istart == 0 && Lmapped == 0 && Shift + Lm == pL) flagDirMap = false;
                                Lmapped += Lm;
                                if (Lm == 0) walk = false;
                            }
                        }
                    }
                }
            }
        }
        DRead rd;
        rd.status = 0; rd.seedOffset = 0; rd.nSeeds = 0; rd.unmappedLength = 0; rd.winOffset = 0; rd.nWin = 0; rd.wtOffset = 0; rd.nWt = 0; rd.pruneBest = 0; rd.pad0 = 0;
        rd.maxScoreMate[0] = rd.maxScoreMate[1] = 0; rd.bestW = -1; rd.nTr = 0; rd.nEx = 0;
        nSeedsTot += st.nP;
        if (st.fatal) rd.status |= STARAMD_ST_FATAL_SEEDS_PER_READ;
        else if (Lread < P.outFilterMatchNmin) { rd.status |= STARAMD_ST_READ_TOO_SHORT; rd.unmappedLength = 0; }
        else if (Nsplit == 0) { rd.status |= STARAMD_ST_NO_GOOD_PIECES; rd.unmappedLength = LgoodMin; }
        else if (st.nA == 0) { rd.status |= STARAMD_ST_ALL_PIECES_MULTI; rd.unmappedLength = st.multNminL; }
        else {
            u32 off = atomicAdd(&B.cursors[CUR_SEED], st.nP);
            if (off + st.nP > B.seedCap) { atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_SEEDPOOL); }
            else {
                rd.seedOffset = off; rd.nSeeds = st.nP;
                for (u32 k = 0; k < st.nP; k++) B.seedPool[off + k] = st.PC[k];
            }
        }
        B.reads[ir] = rd;
    }
    atomicAdd((unsigned long long *)&B.counters[DC_nSAi], (unsigned long long)cn.nSAi);
    atomicAdd((unsigned long long *)&B.counters[DC_nSAprobe], (unsigned long long)cn.nSAprobe);
    atomicAdd((unsigned long long *)&B.counters[DC_nGcmp], (unsigned long long)cn.nGcmp);
    atomicAdd((unsigned long long *)&B.counters[DC_nSeeds], (unsigned long long)nSeedsTot);
}
extern "C" __global__ void __launch_bounds__(256, 8) k_seed_search_flat(const DevIndex *__restrict__ Xp, DevBatch B, DSeed *scratch, u32 scratchPerLane) { seedSearchFlatBody(Xp, B, scratch, scratchPerLane); }
extern "C" __global__ void __launch_bounds__(256, 6) k_seed_search_flat6(const DevIndex *__restrict__ Xp, DevBatch B, DSeed *scratch, u32 scratchPerLane) { seedSearchFlatBody(Xp, B, scratch, scratchPerLane); }
extern "C" __global__ void __launch_bounds__(256, 4) k_seed_search_flat4(const DevIndex *__restrict__ Xp, DevBatch B, DSeed *scratch, u32 scratchPerLane) { seedSearchFlatBody(Xp, B, scratch, scratchPerLane); }

// =====================================================================================================================================================
// Third form (STARAMD_SEED_FLAT = 4 / 5): the WHOLE READ as one state machine.  tools/seed_divergence.py (trace of the emulated engine, 2 048 pairs on a 60 Mb
// genome whose per-read counters equal those of the 3.1 Gb bench) counts the load round trips a wavefront makes: call tree 4 208, flat search 2 191, one state
// machine over the whole read 822, mean lane 470 -- the searches of a read are aligned across lanes by their position in the schedule, and reads differ in how
// many seeds they need, so most of what the flat search leaves on the table is the divergence ABOVE a search.  Here a lane never waits for another lane
// except in the short loops that touch only its own read or its own seed table (quality split, L-mer prefix, storeAligns): every trip of the one loop is
//   TICKET  take the next read                                   (a lane without a read)
//   SCHED   advance the reference's schedule to the next search  (piece -> direction -> start -> walk / --seedSearchLmax; ReadAlign_mapOneRead.cpp:40-93)
//   SETUP   L-mer prefix of the search                           (ReadAlign_maxMappableLength2strands.cpp:23-37)
//   LOAD    the one index load site: SAindex entry / suffix-array entry / 8 genome bases (+ 8 read bases), then the lane's transition
//   POST    a finished search: sparse-SA bookkeeping, storeAligns, the walk's Lmapped                    (:77-109, ReadAlign_mapOneRead.cpp:66-79)
// in this order, so a lane that finishes a search in one trip issues the first load of its next search in the following one.
enum { M_TICKET = 0, M_SCHED, M_SETUP, M_SAI1, M_SAI2, M_CMP, M_POST, M_EXIT };

// STAGED form (STARAMD_SEED_FLAT = 6): the lane's read lives in LDS, 4 bits per base (the packed copy k_pack_reads makes for the stitch kernels), in a slot of its own
// (odd stride in words: lanes that read the same word of their reads hit different banks).  That takes the read out of the load site, and with it two things out of
// the trips: the L-mer prefix of a new search is computed without a memory round trip (no SETUP trip), and the second stream of the load site is free to fetch the
// upper neighbour of a SAindex entry together with the entry itself (no SAI2 trip): about two trips less per search of about eight.
extern __shared__ u32 ldsReads[];
// codes (one per byte) of the 8 bases pos .. pos + 7 of the read in the lane's slot; positions outside the slot read as 15 (never used: masked by every caller)
__device__ __forceinline__ u64 ldsBases8(u32 slotWord0, u32 nWords, i32 pos) {
    const __attribute__((address_space(3))) u32 *W = (const __attribute__((address_space(3))) u32 *)ldsReads + slotWord0;
    const i32 w = pos >> 3; const u32 off = (u32)(pos & 7) * 4u;
    const u32 lo = (w >= 0 && (u32)w < nWords) ? W[w] : 0xFFFFFFFFu;
    const u32 hi = (w + 1 >= 0 && (u32)(w + 1) < nWords) ? W[w + 1] : 0xFFFFFFFFu;
    const u32 x = off ? ((lo >> off) | (hi << (32u - off))) : lo;
    u64 t = x;
    t = (t | (t << 16)) & 0x0000FFFF0000FFFFull;
    t = (t | (t << 8)) & 0x00FF00FF00FF00FFull;
    t = (t | (t << 4)) & 0x0F0F0F0F0F0F0F0Full;
    return t;
}

template <bool STAGED> __device__ __forceinline__ void seedSearchReadBody(const DevIndex *__restrict__ Xp, DevBatch B, DSeed *scratch, u32 scratchPerLane) {
    const DevIndex &X = *Xp;
    const staramd_params &P = X.P;
    const u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    SeedState st; st.PC = scratch + (u64)lane * scratchPerLane; st.cap = scratchPerLane;
    st.nP = 0; st.nA = 0; st.multNmin = 0; st.multNminL = 0; st.fatal = false;
    SeedCnt cn = {0, 0, 0}; u64 nSeedsTot = 0;
    const u64 saiEnd = X.saiStart[X.saiNbases];
    u32 mode = M_TICKET;
    // read
    u32 ir = 0, Lread = 0, iR = 0, iFrag = 0, Nsplit = 0, LgoodMin = 0, sssLmax = 0; const u8 *R = nullptr;
    // piece and the loops below it
    bool pieceActive = false, startInner = false, flagDirMap = true, walk = false, lmaxTodo = false, isWalk = false;
    u32 pS = 0, pL = 0, Nstart = 1, Lstart = 0, iDir = 0, istart = 0, Lmapped = 0, Shift = 0, seedLength = 0;
    // maxMappableLength2strands
    u32 it = 0, nD = 1, maxLbest = 0;
    u32 lastR = 0, lastL = 0;             // (rStart, L) of the last row of the lane's seed table (valid while st.nP > 0)
    const u32 slotWords = B.packWords | 1u, slotWord0 = threadIdx.x * slotWords;      // STAGED: this lane's slot in ldsReads
    // one search
    u32 pieceStart = 0, pieceLength = 0, Lind = 0; u64 ind1 = 0, iSA1 = 0, iSA2 = 0; bool iSA2good = true;
    u64 Nrep = 0, i0 = 0; u32 maxL = 0;
    u32 ph = PH_DONE, cL = 0, i1 = 0, i2 = 0; u64 base = 0;
    u32 L1 = 0, L2 = 0, L3 = 0, L1a = 0, L1b = 0, L2a = 0, L2b = 0, i3 = 0, i1a = 0, i1b = 0, i2a = 0, i2b = 0;
    u32 cI = 0, ii = 0; bool haveSA = false, dirG = true, compRes = false; u64 gAddr = 0;
#ifdef STARAMD_WAVE_EMUL
    // STARAMD_SEED_TRACE (emulated builds): one line per read, "M <read> <one letter per trip>": which blocks of the loop the lane went through in the trip
    // (bit 0 TICKET, 1 SCHED, 2 the load site, 3 POST, as 'A' + bits) -- tools/seed_divergence.py --trips: how often a WAVEFRONT executes each block
    std::string tripLog;
#endif
    while (mode != M_EXIT) {
#ifdef STARAMD_WAVE_EMUL
        const u32 tripBits = (mode == M_TICKET ? 1u : 0u);
        u32 tb = tripBits;
        struct TripEnd { std::string &s; u32 &b; ~TripEnd() { if (g_seedTrace.f) s.push_back((char)('A' + b)); } } tripEnd{tripLog, tb};
#define TRIP_MARK(bit) (tb |= (bit))
#else
#define TRIP_MARK(bit) ((void)0)
#endif
        if (mode == M_TICKET) {
#ifdef STARAMD_WAVE_EMUL
            if (g_seedTrace.f && !tripLog.empty()) { fprintf(g_seedTrace.f, "M %u %s\n", ir, tripLog.c_str()); tripLog.clear(); }     // (one line per READ: the emulator runs the lanes one after the other, lane 0 takes every ticket)
#endif
            ir = atomicAdd(&B.cursors[CUR_TICKET_SEED], 1u);
            if (ir >= B.nReads) { mode = M_EXIT; continue; }
            R = B.bases + B.readOffset[ir];
            Lread = (u32)(B.readOffset[ir + 1] - B.readOffset[ir]);
            st.nP = 0; st.nA = 0; st.multNmin = 0; st.multNminL = 0; st.fatal = false;
            Nsplit = 0; LgoodMin = 0; iR = 0; iFrag = 0; pieceActive = false;
            sssLmax = min(P.seedSearchStartLmax, (u32)(u64)(P.seedSearchStartLmaxOverLread * (double)(u64)(Lread - 1)));
            if constexpr (STAGED) {
                const u32 *src = B.packed + (u64)ir * B.packWords;
                __attribute__((address_space(3))) u32 *dst = (__attribute__((address_space(3))) u32 *)ldsReads + slotWord0;
                for (u32 k = 0; k < B.packWords; k++) dst[k] = GLOBAL(u32, src)[k];
            }
            mode = M_SCHED;
        }
        if (mode == M_SCHED) {
            TRIP_MARK(2u);
            for (;;) {
                if (!pieceActive) {
                    // qualitySplit (SequenceFuns.cpp:411-444): the next piece, or the end of the read
                    bool done = !((iR < Lread) & (Nsplit < P.maxNsplit));
                    if (!done) {
                        while (iR < Lread && R[iR] > 3) { if (R[iR] == STARAMD_SPACER_BASE) iFrag++; iR++; }
                        done = iR == Lread;
                    }
                    if (done) {
                        // classification and output of the read (ReadAlign_mapOneRead.cpp:100-115), as in k_seed_search
                        DRead rd;
                        rd.status = 0; rd.seedOffset = 0; rd.nSeeds = 0; rd.unmappedLength = 0; rd.winOffset = 0; rd.nWin = 0; rd.wtOffset = 0; rd.nWt = 0; rd.pruneBest = 0; rd.pad0 = 0;
                        rd.maxScoreMate[0] = rd.maxScoreMate[1] = 0; rd.bestW = -1; rd.nTr = 0; rd.nEx = 0;
                        nSeedsTot += st.nP;
                        if (st.fatal) rd.status |= STARAMD_ST_FATAL_SEEDS_PER_READ;
                        else if (Lread < P.outFilterMatchNmin) { rd.status |= STARAMD_ST_READ_TOO_SHORT; rd.unmappedLength = 0; }
                        else if (Nsplit == 0) { rd.status |= STARAMD_ST_NO_GOOD_PIECES; rd.unmappedLength = LgoodMin; }
                        else if (st.nA == 0) { rd.status |= STARAMD_ST_ALL_PIECES_MULTI; rd.unmappedLength = st.multNminL; }
                        else {
                            u32 off = atomicAdd(&B.cursors[CUR_SEED], st.nP);
                            if (off + st.nP > B.seedCap) { atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_SEEDPOOL); }
                            else {
                                rd.seedOffset = off; rd.nSeeds = st.nP;
                                for (u32 k = 0; k < st.nP; k++) B.seedPool[off + k] = st.PC[k];
                            }
                        }
                        B.reads[ir] = rd;
                        mode = M_TICKET;
                        break;
                    }
                    pS = iR;
                    for (;;) {
                        const u64 bad = load8(R + iR) & 0xFCFCFCFCFCFCFCFCull;
                        const u32 k = bad ? ((u32)__builtin_ctzll(bad) >> 3) : 8u;
                        iR += k;
                        if (iR >= Lread) { iR = Lread; break; }
                        if (k < 8u) break;
                    }
                    pL = iR - pS;
                    if (pL > LgoodMin) LgoodMin = pL;
                    if (pL < P.seedSplitMin) continue;
                    Nsplit++;
                    Nstart = (P.seedSearchStartLmax > 0 && sssLmax < pL) ? pL / sssLmax + 1 : 1;
                    Lstart = pL / Nstart;
                    flagDirMap = true; iDir = 0; istart = 0; pieceActive = true; startInner = true;
                }
                if (startInner) { Lmapped = 0; walk = flagDirMap || istart > 0; lmaxTodo = P.seedSearchLmax > 0; startInner = false; }
                isWalk = walk && (istart * Lstart + Lmapped + P.seedMapMin < pL);
                if (isWalk) {
                    Shift = iDir == 0 ? (pS + istart * Lstart + Lmapped) : (pS + pL - istart * Lstart - 1 - Lmapped);
                    seedLength = pL - Lmapped - istart * Lstart;
                } else if (lmaxTodo) {
                    lmaxTodo = false; walk = false;
                    Shift = iDir == 0 ? (pS + istart * Lstart) : (pS + pL - istart * Lstart - 1);
                    seedLength = min(P.seedSearchLmax, iDir == 0 ? (pS + pL - Shift) : (Shift + 1));
                } else {                                         // this start is done: next start / direction / piece
                    istart++;
                    if (istart >= Nstart) { istart = 0; iDir++; if (iDir >= 2) { pieceActive = false; continue; } }
                    startInner = true;
                    continue;
                }
                nD = min(seedLength, X.sparseD); maxLbest = 0; it = 0;     // maxMappableLength2strands starts
                if (nD == 0) {                                   // (a search of no bases: the reference's loop over the start offsets does not run, Lm = 0)
                    if (isWalk) { if (iDir == 0 && istart == 0 && Lmapped == 0 && Shift == pL) flagDirMap = false; walk = false; }
                    continue;
                }
                mode = M_SETUP;
                break;
            }
        }
        const bool dirR = iDir == 0;
        if constexpr (STAGED) {
            if (mode == M_SETUP) {                               // a search begins: its L-mer prefix from the read in LDS, no trip of its own
                const u32 iDist = it >= nD ? it - nD : it;
                pieceLength = seedLength - iDist;
                pieceStart = dirR ? Shift + iDist : Shift - iDist;
                const u32 Lmax = min(X.saiNbases, pieceLength);
                ind1 = 0;
                for (u32 k8 = 0; k8 < Lmax; k8 += 8) {
                    const u64 raw = dirR ? ldsBases8(slotWord0, B.packWords, (i32)(pieceStart + k8)) : __builtin_bswap64(ldsBases8(slotWord0, B.packWords, (i32)pieceStart - (i32)k8 - 7));
                    const u32 nb = min(8u, Lmax - k8);
                    const u64 used = nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
                    if ((raw & used & 0xFCFCFCFCFCFCFCFCull) == 0) {
                        u64 x = dirR ? raw : (raw ^ 0x0303030303030303ull);
                        u64 z = __builtin_bswap64(x & 0x0303030303030303ull);
                        z = (z | (z >> 6)) & 0x000F000F000F000Full;
                        z = (z | (z >> 12)) & 0x000000FF000000FFull;
                        z = (z | (z >> 24)) & 0xFFFFull;
                        ind1 = (ind1 << (2 * nb)) | (z >> (2 * (8 - nb)));
                    } else {
                        for (u32 k = 0; k < nb; k++) { const u64 cde = (raw >> (8 * k)) & 0xFFull; ind1 = (ind1 << 2) + (dirR ? cde : 3ull - cde); }
                    }
                }
                Lind = Lmax; iSA1 = 0; iSA2 = 0; iSA2good = true;
                Nrep = 0; i0 = 0; maxL = 0;
                while (Lind > 0 && X.saiStart[Lind - 1] + ind1 >= saiEnd) { --Lind; ind1 >>= 2; }
                mode = Lind > 0 ? M_SAI1 : M_POST;
            }
        }
        if (mode == M_SETUP || mode == M_SAI1 || mode == M_SAI2 || mode == M_CMP) {
            TRIP_MARK(4u);
            if (mode == M_SETUP) {                               // a search begins (one start offset of maxMappableLength2strands)
                const u32 iDist = it >= nD ? it - nD : it;
                pieceLength = seedLength - iDist;
                pieceStart = dirR ? Shift + iDist : Shift - iDist;
            }
            // ---- the one load site: two funnel loads.  SETUP: the first 16 bases of the search (L-mer prefix); SAI: a SAindex entry; CMP: a suffix-array entry, or
            // 8 genome bases and the 8 read bases that go with them
            const u32 cN = ph >= PH_F1 ? L3 : pieceLength;
            const u64 sBase = (u64)R + pieceStart;
            u64 a; u32 sh;
            if (mode == M_SETUP) { const u64 p = dirR ? sBase + 8ull : sBase - 15ull; a = p & ~7ull; sh = (u32)(p & 7ull) * 8u; }
            else if (mode != M_CMP) { const u64 b = (X.saiStart[Lind - 1] + ind1 + (mode == M_SAI2 ? 1ull : 0ull)) * X.saiBits; a = (u64)X.SAi + (b >> 6) * 8ull; sh = (u32)(b & 63ull); }
            else if (!haveSA) { const u64 b = (base + cI) * X.saBits; a = (u64)X.SA + (b >> 6) * 8ull; sh = (u32)(b & 63ull); }
            else { const u64 p = dirG ? gAddr + ii : gAddr - ii - 7ull; a = p & ~7ull; sh = (u32)(p & 7ull) * 8u; }
            const u64 v = funnel64(a, sh);
            u64 s8;
            if constexpr (STAGED) {
                // second stream: the upper neighbour of the SAindex entry (SAI1 trips); the read bases of a compare step come from LDS
                if (mode == M_SAI1) { const u64 b = (X.saiStart[Lind - 1] + ind1 + 1ull) * X.saiBits; s8 = funnel64((u64)X.SAi + (b >> 6) * 8ull, (u32)(b & 63ull)); }
                else if (mode == M_CMP && haveSA) s8 = dirR ? ldsBases8(slotWord0, B.packWords, (i32)(pieceStart + cL + ii)) : ldsBases8(slotWord0, B.packWords, (i32)pieceStart - (i32)cL - (i32)ii - 7);
                else s8 = 0;
            } else {
                const u64 sp = mode == M_SETUP ? (dirR ? sBase : sBase - 7ull)
                             : (mode != M_CMP || !haveSA) ? sBase : dirR ? sBase + cL + ii : sBase - cL - ii - 7ull;      // (no compare step in this trip: any address inside the read)
                s8 = funnel64(sp & ~7ull, (u32)(sp & 7ull) * 8u);
            }
            const u32 m0 = mode;                                 // what this trip's loads were for
            bool start = false, fin = false; u32 Lc = cN;
            if (m0 == M_SETUP) {
                // L-mer prefix (ReadAlign_maxMappableLength2strands.cpp:23-37), as in k_seed.hip: word 0 = bases 0..7 of the scan, word 1 = bases 8..15
                const u32 Lmax = min(X.saiNbases, pieceLength);
                ind1 = 0;
                for (u32 k8 = 0; k8 < Lmax; k8 += 8) {
                    const u64 w = k8 == 0 ? s8 : v;
                    const u64 raw = dirR ? w : __builtin_bswap64(w);
                    const u32 nb = min(8u, Lmax - k8);
                    const u64 used = nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
                    if ((raw & used & 0xFCFCFCFCFCFCFCFCull) == 0) {
                        u64 x = dirR ? raw : (raw ^ 0x0303030303030303ull);
                        u64 z = __builtin_bswap64(x & 0x0303030303030303ull);
                        z = (z | (z >> 6)) & 0x000F000F000F000Full;
                        z = (z | (z >> 12)) & 0x000000FF000000FFull;
                        z = (z | (z >> 24)) & 0xFFFFull;
                        ind1 = (ind1 << (2 * nb)) | (z >> (2 * (8 - nb)));
                    } else {
                        for (u32 k = 0; k < nb; k++) { const u64 cde = (raw >> (8 * k)) & 0xFFull; ind1 = (ind1 << 2) + (dirR ? cde : 3ull - cde); }
                    }
                }
                Lind = Lmax; iSA1 = 0; iSA2 = 0; iSA2good = true;
                Nrep = 0; i0 = 0; maxL = 0;
                while (Lind > 0 && X.saiStart[Lind - 1] + ind1 >= saiEnd) { --Lind; ind1 >>= 2; }
                mode = Lind > 0 ? M_SAI1 : M_POST;
            } else if (m0 == M_SAI1) {
                iSA1 = v & X.saiMask; cn.nSAi++;
                if (iSA1 & X.saiAbsentBit) {
                    --Lind; ind1 >>= 2;
                    while (Lind > 0 && X.saiStart[Lind - 1] + ind1 >= saiEnd) { --Lind; ind1 >>= 2; }
                    if (Lind == 0) mode = M_POST;                // base absent from the genome: Nrep = 0
                } else if (X.saiStart[Lind - 1] + ind1 + 1 < X.saiStart[Lind]) {
                    if constexpr (STAGED) {                      // the neighbour came with this trip's second stream
                        iSA2 = s8 & X.saiMask; cn.nSAi++;
                        if ((iSA2 & X.saiAbsentBit) == 0) iSA2 = (iSA2 & ~X.saiNbit) - 1;
                        else { iSA2 = X.nSA - 1; iSA2good = false; }
                        start = true;
                    } else mode = M_SAI2;
                } else { iSA2 = X.nSA - 1; iSA2good = false; start = true; }
            } else if (m0 == M_SAI2) {
                iSA2 = v & X.saiMask; cn.nSAi++;
                if ((iSA2 & X.saiAbsentBit) == 0) iSA2 = (iSA2 & ~X.saiNbit) - 1;
                else { iSA2 = X.nSA - 1; iSA2good = false; }
                start = true;
            } else if (!haveSA) {
                cn.nSAprobe++;
                u64 SAstr = v & X.saMask;
                dirG = (SAstr >> X.strandBit) == 0;
                SAstr &= X.strandMask;
                gAddr = dirG ? (u64)X.G + SAstr + cL : (u64)X.G + (X.nGenome - 1 - SAstr) - cL;
                haveSA = true; ii = 0;
                fin = cN == cL;
            } else {
                const u32 n = cN - cL;
                if (!dirR) s8 = __builtin_bswap64(s8);
                const u64 g8 = dirG ? v : __builtin_bswap64(v);
                if (dirR != dirG) s8 = comp8(s8);
                u64 d = s8 ^ g8;
                const u32 rem = n - ii;
                if (rem < 8) d &= (1ull << (rem * 8)) - 1ull;
                if (d) {
                    const u32 k = (u32)__builtin_ctzll(d) >> 3;
                    const u8 sc = (u8)(s8 >> (k * 8)), gc = (u8)(g8 >> (k * 8));
                    cn.nGcmp += ii + k + 1;
                    compRes = dirG ? (sc > gc) : !(sc > gc || gc > 3);
                    fin = true; Lc = ii + k + cL;
                } else { ii += 8; if (ii >= n) { cn.nGcmp += n; fin = true; } }
            }
            if (start) {                                         // searchOneDist of k_seed.hip after the two look-ups
                const bool iSA1noN = (iSA1 & X.saiNbit) == 0;
                haveSA = false; cI = 0; i1 = 0; i2 = 0; L3 = 0;
                base = iSA1 & ~X.saiNbit;
                if (Lind < X.saiNbases && iSA1noN && iSA2good) { i0 = iSA1; Nrep = iSA2 - iSA1 + 1; maxL = Lind; mode = M_POST; }
                else if (iSA1 == iSA2 && iSA1noN && iSA2good) { ph = PH_SINGLE; cL = Lind; mode = M_CMP; }
                else {
                    maxL = (iSA2good && iSA1noN) ? Lind : 0;
                    if (iSA2 - base >= 0xFFFFFFFFull) { Nrep = searchWide(X, R, pieceStart, pieceLength, base, iSA2, dirR, maxL, i0, cn); mode = M_POST; }   // (rare)
                    else { i2 = (u32)(iSA2 - base); ph = PH_L1; cL = maxL; mode = M_CMP; }
                }
            }
            if (fin) {
                haveSA = false;
                bool brk = false;
                if (ph == PH_SINGLE) { L3 = Lc; ph = PH_DONE; }
                else if (ph == PH_L1) { L1 = Lc; cI = i2; ph = PH_L2; }
                else if (ph == PH_L2) {
                    L2 = Lc;
                    L1a = L1; L1b = L1; i1a = i1; i1b = i1; L2a = L2; L2b = L2; i2a = i2; i2b = i2;
                    i3 = i1; L3 = L1; ph = PH_MAIN;
                } else if (ph == PH_MAIN) {
                    L3 = Lc;
                    if (L3 == pieceLength) brk = true;
                    else if (compRes) { if (L3 > L1) { L1b = L1a; L1a = L1; i1b = i1a; i1a = i1; } i1 = i3; L1 = L3; }
                    else { if (L3 > L2) { L2b = L2a; L2a = L2; i2b = i2a; i2a = i2; } i2 = i3; L2 = L3; }
                } else if (ph == PH_F1) { if (Lc == L3) i1a = cI; else { i1b = cI; L1b = Lc; } }
                else { if (Lc == L3) i2a = cI; else { i2b = cI; L2b = Lc; } }
                if (ph == PH_MAIN) {
                    if (!brk && (u64)i1 + 1 < (u64)i2) { i3 = (u32)medianUint2(i1, i2); cI = i3; cL = min(L1, L2); }
                    else {
                        if (L3 < pieceLength) { if (L1 > L2) { i3 = i1; L3 = L1; } else { i3 = i2; L3 = L2; } }
                        if (L1 < L3) { L1b = L1; i1b = i1; i1a = i3; }
                        else if (L1a < L1) { L1b = L1a; i1b = i1a; i1a = i1; }
                        ph = PH_F1;
                    }
                }
                if (ph == PH_F1) {
                    if (((u64)i1b + 1 < (u64)i1a) | ((u64)i1b > (u64)i1a + 1)) { cI = (u32)medianUint2(i1a, i1b); cL = L1b; }
                    else {
                        i1 = i1a;
                        if (L2 < L3) { L2b = L2; i2b = i2; i2a = i3; }
                        else if (L2a < L2) { L2b = L2a; i2b = i2a; i2a = i2; }
                        ph = PH_F2;
                    }
                }
                if (ph == PH_F2) {
                    if (((u64)i2b + 1 < (u64)i2a) | ((u64)i2b > (u64)i2a + 1)) { cI = (u32)medianUint2(i2a, i2b); cL = L2b; }
                    else { i2 = i2a; ph = PH_DONE; }
                }
                if (ph == PH_DONE) { maxL = L3; i0 = base + i1; Nrep = (u64)i2 - (u64)i1 + 1; mode = M_POST; }
            }
        }
        if (mode == M_POST) {
            TRIP_MARK(8u);
            // one start offset of maxMappableLength2strands is searched (ReadAlign_maxMappableLength2strands.cpp:77-109)
            const u32 phase = it >= nD ? 1u : 0u, iDist = phase ? it - nD : it;
            bool more = true;
            if (phase == 0) { if (maxL + iDist > maxLbest) maxLbest = maxL + iDist; }
            if (!(phase == 0 && nD > 1)) {
                if (maxL + iDist == maxLbest && Nrep > 0) {
                    // storeAligns (ReadAlign_storeAligns.cpp:10-160) looks at the table from its end; the two commonest outcomes are decided by the LAST row alone, which is
                    // kept in registers: the new seed starts behind it (a forward walk finds its seeds in that order: append) or equals it (found again: nothing to do).
                    // Everything else goes through the table in the lane's scratch memory.  Same table, same counters, no load in the common case.
                    const u32 sShift = dirR ? Shift + iDist : Shift - iDist;
                    const u32 rStart = iDir == 0 ? sShift : sShift + 1 - maxL;
                    const bool fits = st.nP + 1 <= X.P.seedPerReadNmax && st.nP + 1 <= st.cap;
                    if (Nrep <= X.P.seedMultimapNmax && st.nP > 0 && lastR == rStart && lastL == maxL) st.nA |= 1u;                    // the duplicate: (:30-33) return before anything is written
                    else if (Nrep <= X.P.seedMultimapNmax && fits && (st.nP == 0 || lastR < rStart)) {                                  // the append
                        st.nA |= 1u;
                        DSeed sd; sd.saStart = i0; sd.nrep = (u32)Nrep; sd.rStart = (u16)rStart; sd.L = (u16)maxL; sd.dir = (u8)iDir; sd.iFrag = (u8)iFrag;
                        for (int k = 0; k < 6; k++) sd.pad[k] = 0;
                        st.PC[st.nP] = sd; st.nP++;
                        lastR = rStart; lastL = maxL;
                        if (Nrep != 1) { if (Nrep < st.multNmin || st.multNmin == 0) { st.multNmin = (u32)Nrep; st.multNminL = maxL; } }
                    } else {
                        const u32 nBefore = st.nP;
                        storeAligns(X, st, iDir, sShift, Nrep, maxL, i0, iFrag);
                        if (st.nP != nBefore) { const DSeed t = st.PC[st.nP - 1]; lastR = t.rStart; lastL = t.L; }                      // (an insertion may have moved the last row)
                    }
                }
                if (nD == 1) more = false;
            }
            it++;
            if (more && it < 2 * nD) mode = M_SETUP;
            else {
                // the search of this seed is done with Lm = maxLbest (ReadAlign_mapOneRead.cpp:66-79)
                if (isWalk) {
                    if (iDir == 0 && istart == 0 && Lmapped == 0 && Shift + maxLbest == pL) flagDirMap = false;
                    Lmapped += maxLbest;
                    if (maxLbest == 0) walk = false;
                }
                mode = M_SCHED;
            }
        }
    }

    atomicAdd((unsigned long long *)&B.counters[DC_nSAi], (unsigned long long)cn.nSAi);
    atomicAdd((unsigned long long *)&B.counters[DC_nSAprobe], (unsigned long long)cn.nSAprobe);
    atomicAdd((unsigned long long *)&B.counters[DC_nGcmp], (unsigned long long)cn.nGcmp);
    atomicAdd((unsigned long long *)&B.counters[DC_nSeeds], (unsigned long long)nSeedsTot);
}
extern "C" __global__ void __launch_bounds__(256, 4) k_seed_search_read4(const DevIndex *__restrict__ Xp, DevBatch B, DSeed *scratch, u32 scratchPerLane) { seedSearchReadBody<false>(Xp, B, scratch, scratchPerLane); }
extern "C" __global__ void __launch_bounds__(256, 6) k_seed_search_read6(const DevIndex *__restrict__ Xp, DevBatch B, DSeed *scratch, u32 scratchPerLane) { seedSearchReadBody<false>(Xp, B, scratch, scratchPerLane); }
// launched with 256 * (B.packWords | 1) * 4 bytes of dynamic LDS
extern "C" __global__ void __launch_bounds__(256, 4) k_seed_search_staged4(const DevIndex *__restrict__ Xp, DevBatch B, DSeed *scratch, u32 scratchPerLane) { seedSearchReadBody<true>(Xp, B, scratch, scratchPerLane); }
