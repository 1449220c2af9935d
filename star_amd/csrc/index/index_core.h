// index_core.h -- suffix array + SAindex of a STAR genomeDir, built as data-parallel passes (radix sorts, scans,
// element-wise kernels) over arrays resident in HBM.
//
// What it replaces: the SA sort and SAindex stage of Genome::genomeGenerate
//   suffix order      source/Genome_genomeGenerate.cpp:29-91 (funCompareSuffixes), :191-305 (prefix chunks, qsort, packing)
//   SAindex           source/genomeSAindex.cpp:6-217, funCalcSAiFromSA source/SuffixArrayFuns.cpp:354-395
//   packed arrays     source/PackedArray.cpp:8-26
// The result is byte-identical to the reference's `SA` and `SAindex` files (tests/test_index_build.py).
//
// The algorithm is written once against a small backend interface (element-wise forEach, stable LSD radix sort of
// (key,value) pairs, prefix scans): `HipBackend` (index_gpu.hip, rocPRIM primitives + HIP kernels) is the product;
// `oracle/index_emul.cpp` instantiates the same code with plain loops so that the LOGIC can be checked against the
// reference on a machine without a GPU (test infrastructure only).
//
// Order being reproduced.  Text T[0,2N): T[i] = G[i], T[2N-1-i] = complement(G[i]); codes 0..3 ACGT, 4 N, 5 padding.
// Suffix p < suffix q iff T[p..] < T[q..] lexicographically by code, where a comparison that reaches padding in both
// suffixes at the same offset stops there and the smaller position wins (funCompareSuffixes, "anti-stable" on reversed
// indices = ascending text position).  That is the ordinary suffix order of the text in which every padding byte is a
// distinct symbol ordered by position, so prefix doubling applies:
//   round 0   positions bucketed by their first two codes (36 buckets, positions stay in text order), each bucket
//             radix-sorted (stable) by the next 21 codes (3 bits each); a key that contains padding is its own group
//   round k   the still ambiguous groups are sorted by (group, rank of the suffix h positions further on), h doubling
// Positions that start with N or padding are ranked too (doubling needs their ranks) and dropped at the end: they are
// exactly the tail of the sorted array.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef __HIPCC__
#define IDX_HD __host__ __device__ __forceinline__
#define IDX_L __host__ __device__            /* lambdas handed to Backend::forEach */
#else
#define IDX_HD inline
#define IDX_L
#endif

namespace staridx {

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;
typedef uint8_t u8;

enum { TPAD = 64 };                  // bytes of code 5 either side of the text (keys read up to 23 codes past a position)
enum { KEY_CODES = 21, BUCKET_CODES = 2, NBUCKET = 36, CHUNK = 256 };
static const u64 KEY_HASPAD = 1ull << 63;

struct BuildParams {
    u64 nGenome;                     // N (with chromosome padding), text length 2N
    u32 GstrandBit;                  // Genome_genomeGenerate.cpp:177-178
    u32 saIndexNbases;               // --genomeSAindexNbases
};

// ---------------------------------------------------------------------------------------------------------------------
// element-wise pieces (host + device)

IDX_HD u8 compCode(u8 c) { return c < 4 ? (u8)(3 - c) : c; }

// 21-code key of the suffix at p starting `skip` codes in; codes after the first padding byte (counted from offset 0)
// are zero, bit 63 tells that padding was met (the suffix is then unique: ties are broken by position = stable order)
IDX_HD u64 prefixKey(const u8 *T, u64 p, u32 skip) {
    u64 key = 0; bool pad = false;
    for (u32 k = 0; k < skip + KEY_CODES; k++) {
        u8 c = pad ? (u8)0 : T[p + k];
        if (k >= skip) key = (key << 3) | c;
        if (c == 5) pad = true;
    }
    return pad ? (key | KEY_HASPAD) : key;
}
IDX_HD u32 bucketOf(const u8 *T, u64 p) { u8 a = T[p], b = a == 5 ? (u8)0 : T[p + 1]; return (u32)a * 6u + b; }

// rank of the suffix at text position q (positions past the end behave as distinct padding in position order)
IDX_HD u64 rankAt(const u64 *ISA, u64 n2, u64 q) { return q < n2 ? ISA[q] : n2 + (q - n2); }

// SAindex class of a suffix: funCalcSAiFromSA (SuffixArrayFuns.cpp:354-395) on the 2N text
// read from two 8-byte words of text (L <= 16; T is padded): one pair of loads instead of up to L dependent byte loads -- the class of
// EVERY suffix is needed when the table is built (6.3 * 10^9 suffixes for a human genome), at a random text position each
IDX_HD u32 saiClassFast(const u8 *T, u64 pos, u32 L, int &iL4) {
    u64 w0, w1;
    __builtin_memcpy(&w0, T + pos, 8); __builtin_memcpy(&w1, T + pos + 8, 8);
    u32 ind = 0; iL4 = -1;
    for (u32 ii = 0; ii < L; ii++) {
        const u32 g = (u32)((ii < 8 ? w0 >> (8 * ii) : w1 >> (8 * (ii - 8))) & 0xFFu);
        if (g > 3) { iL4 = (int)ii; ind <<= 2 * (L - ii); return ind; }
        ind = (ind << 2) + g;
    }
    return ind;
}

IDX_HD void packedSet(u64 *words, u64 i, u32 bits, u64 v) {       // single-threaded use only (host side helpers)
    u64 b = i * bits, w = b >> 6; u32 s = (u32)(b & 63);
    u64 mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    words[w] = (words[w] & ~(mask << s)) | (v << s);
    if (s + bits > 64) { u32 r = 64 - s; words[w + 1] = (words[w + 1] & ~(mask >> r)) | (v >> r); }
}
IDX_HD u64 packedGetW(const u64 *a, u64 i, u32 bits) {
    u64 b = i * bits, w = b >> 6; u32 s = (u32)(b & 63);
    u64 mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    u64 v = a[w] >> s;
    if (s + bits > 64) v |= a[w + 1] << (64 - s);
    return v & mask;
}

IDX_HD void be_atomicOr(u64 *p, u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr((unsigned long long *)p, (unsigned long long)v);
#else
    __atomic_fetch_or(p, v, __ATOMIC_RELAXED);
#endif
}

static inline u32 bitsFor(u64 maxValue) { u32 b = 1; while (b < 64 && (maxValue >> b)) b++; return b; }

// ---------------------------------------------------------------------------------------------------------------------
// generic helpers on a backend

// stable compaction: emit(i, outIndex) for every i in [0,n) with pred(i), in order; returns the count
template <class BE, class P, class E> u64 compactIf(BE &be, u64 n, P pred, E emit) {
    if (n == 0) return 0;
    u64 nCh = (n + CHUNK - 1) / CHUNK;
    u64 *cnt = be.template alloc<u64>(nCh + 1);
    be.forEach(nCh + 1, [=] IDX_L (u64 c) {
        u64 k = 0;
        if (c < nCh) { u64 e = (c + 1) * CHUNK < n ? (c + 1) * CHUNK : n; for (u64 i = c * CHUNK; i < e; i++) k += pred(i) ? 1 : 0; }
        cnt[c] = k;
    });
    be.exclusiveSum(cnt, nCh + 1);
    u64 total = be.readOne(cnt + nCh);
    be.forEach(nCh, [=] IDX_L (u64 c) {
        u64 o = cnt[c]; u64 e = (c + 1) * CHUNK < n ? (c + 1) * CHUNK : n;
        for (u64 i = c * CHUNK; i < e; i++) if (pred(i)) emit(i, o++);
    });
    be.free(cnt);
    return total;
}

// ---------------------------------------------------------------------------------------------------------------------
// 1. text

// dT: 2N + 2*TPAD bytes; dG: N bytes (genome as in genomeDir).  Returns pointer to T[0].
template <class BE> u8 *buildText(BE &be, const u8 *dG, u64 N, u8 *dTraw) {
    u8 *T = dTraw + TPAD;
    be.forEach(2 * TPAD, [=] IDX_L (u64 i) { if (i < TPAD) dTraw[i] = 5; else dTraw[2 * N + i] = 5; });
    be.forEach(N, [=] IDX_L (u64 i) { u8 c = dG[i]; T[i] = c; T[2 * N - 1 - i] = compCode(c); });
    return T;
}

// ---------------------------------------------------------------------------------------------------------------------
// 2. suffix sort of ALL 2N positions.  dSA, dISA: 2N entries each.  Returns the number of suffixes that start with ACGT
//    (they are dSA[0 .. nSA)).
template <class BE> u64 suffixSort(BE &be, const u8 *T, u64 N, u64 *dSA, u64 *dISA, u64 *roundsOut) {
    const u64 n2 = 2 * N;
    // ---- bucket positions by their first two codes, positions in text order inside a bucket
    u64 nCh = (n2 + CHUNK - 1) / CHUNK;
    u64 *cnt = be.template alloc<u64>(NBUCKET * nCh + 1);
    be.forEach(nCh, [=] IDX_L (u64 c) {
        u32 k[NBUCKET];
        for (u32 b = 0; b < NBUCKET; b++) k[b] = 0;
        u64 e = (c + 1) * CHUNK < n2 ? (c + 1) * CHUNK : n2;
        for (u64 p = c * CHUNK; p < e; p++) k[bucketOf(T, p)]++;
        for (u32 b = 0; b < NBUCKET; b++) cnt[b * nCh + c] = k[b];
    });
    be.forEach(1, [=] IDX_L (u64) { cnt[NBUCKET * nCh] = 0; });
    be.exclusiveSum(cnt, NBUCKET * nCh + 1);
    u64 bStart[NBUCKET + 1];
    for (u32 b = 0; b < NBUCKET; b++) bStart[b] = be.readOne(cnt + b * nCh);
    bStart[NBUCKET] = n2;
    be.forEach(nCh, [=] IDX_L (u64 c) {
        u64 o[NBUCKET];
        for (u32 b = 0; b < NBUCKET; b++) o[b] = cnt[b * nCh + c];
        u64 e = (c + 1) * CHUNK < n2 ? (c + 1) * CHUNK : n2;
        for (u64 p = c * CHUNK; p < e; p++) { u32 b = bucketOf(T, p); dSA[o[b]++] = p; }
    });
    be.free(cnt);
    u64 nSA = bStart[4 * 6];            // buckets 0..23 start with ACGT
    u64 maxB = 0;
    for (u32 b = 0; b < NBUCKET; b++) if (bStart[b + 1] - bStart[b] > maxB) maxB = bStart[b + 1] - bStart[b];
    // ---- round 0: every bucket sorted by the next 21 codes
    {
        u64 *k0 = be.template alloc<u64>(maxB), *k1 = be.template alloc<u64>(maxB), *v0 = be.template alloc<u64>(maxB), *v1 = be.template alloc<u64>(maxB);
        for (u32 b = 0; b < NBUCKET; b++) {
            u64 s = bStart[b], nb = bStart[b + 1] - s;
            if (nb == 0) continue;
            u64 *keys = k0, *keysAlt = k1, *vals = v0, *valsAlt = v1;
            u64 *saB = dSA + s;
            be.forEach(nb, [=] IDX_L (u64 i) { u64 p = saB[i]; keys[i] = prefixKey(T, p, BUCKET_CODES); vals[i] = p; });
            if (nb > 1) be.sortPairs(keys, keysAlt, vals, valsAlt, nb, 0, 63);      // bit 63 (padding flag) is not a sort bit
            // group heads -> rank = index of the head of the group
            u64 *hd = keysAlt;
            {
                const u64 *ks = keys;
                be.forEach(nb, [=] IDX_L (u64 i) { bool h = i == 0 || ks[i] != ks[i - 1] || (ks[i] & KEY_HASPAD); hd[i] = h ? i : 0; });
            }
            be.inclusiveMax(hd, nb);
            {
                const u64 *vs = vals; const u64 *hh = hd;
                be.forEach(nb, [=] IDX_L (u64 i) { u64 p = vs[i]; saB[i] = p; dISA[p] = s + hh[i]; });
            }
        }
        be.free(k0); be.free(k1); be.free(v0); be.free(v1);
    }
    // ---- doubling rounds on the groups that are still ambiguous
    u64 h = BUCKET_CODES + KEY_CODES;
    // unresolved slots: not (head(i) && head(i+1))
    u64 U;
    u64 *p = nullptr, *slot = nullptr;
    {
        auto unresolved = [=] IDX_L (u64 i) {
            bool hi = dISA[dSA[i]] == i;
            bool hn = i + 1 >= n2 || dISA[dSA[i + 1]] == i + 1;
            return !(hi && hn);
        };
        U = compactIf(be, n2, unresolved, [=] IDX_L (u64, u64) {});      // count first to size the buffers
        if (U) {
            p = be.template alloc<u64>(U); slot = be.template alloc<u64>(U);
            u64 *pp = p, *ss = slot;
            compactIf(be, n2, unresolved, [=] IDX_L (u64 i, u64 o) { pp[o] = dSA[i]; ss[o] = i; });
        }
    }
    u64 rounds = 0;
    if (U) {
        u64 cap = U;
        u64 *pAlt = be.template alloc<u64>(cap), *k = be.template alloc<u64>(cap), *kAlt = be.template alloc<u64>(cap), *slotAlt = be.template alloc<u64>(cap);
        const u32 rbits = bitsFor(4 * n2);
        while (U) {
            rounds++;
            { u64 *pp = p, *kk = k; be.forEach(U, [=] IDX_L (u64 j) { kk[j] = rankAt(dISA, n2, pp[j] + h); }); }
            if (U > 1) be.sortPairs(k, kAlt, p, pAlt, U, 0, rbits);
            { u64 *pp = p, *kk = k; be.forEach(U, [=] IDX_L (u64 j) { kk[j] = dISA[pp[j]]; }); }
            if (U > 1) be.sortPairs(k, kAlt, p, pAlt, U, 0, rbits);
            // p is now ordered by (group, rank h further on): the j-th element belongs into the j-th ambiguous slot
            { u64 *pp = p, *k2 = kAlt; be.forEach(U, [=] IDX_L (u64 j) { k2[j] = rankAt(dISA, n2, pp[j] + h); }); }
            {
                const u64 *r = k, *k2 = kAlt; u64 *hd = pAlt;
                be.forEach(U, [=] IDX_L (u64 j) { bool hh = j == 0 || r[j] != r[j - 1] || k2[j] != k2[j - 1]; hd[j] = hh ? j : 0; });
            }
            be.inclusiveMax(pAlt, U);
            {
                const u64 *pp = p, *ss = slot, *hd = pAlt;
                be.forEach(U, [=] IDX_L (u64 j) { u64 q = pp[j]; dSA[ss[j]] = q; dISA[q] = ss[hd[j]]; });
            }
            // survivors: groups of more than one element
            u64 Ucur = U;
            const u64 *hd = pAlt; const u64 *pp = p, *ss = slot; u64 *pn = k, *sn = slotAlt;
            u64 Un = compactIf(be, Ucur, [=] IDX_L (u64 j) { bool single = hd[j] == j && (j + 1 >= Ucur || hd[j + 1] == j + 1); return !single; },
                               [=] IDX_L (u64 j, u64 o) { pn[o] = pp[j]; sn[o] = ss[j]; });
            // rotate buffers: new p = k, new slot = slotAlt
            u64 *t = p; p = k; k = t;
            t = slot; slot = slotAlt; slotAlt = t;
            U = Un;
            h *= 2;
        }
        be.free(pAlt); be.free(k); be.free(kAlt); be.free(slotAlt);
    }
    if (p) be.free(p);
    if (slot) be.free(slot);
    if (roundsOut) *roundsOut = rounds;
    return nSA;
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. packing: text positions -> the reference's SA values, (GstrandBit+1) bits each, PackedArray layout.
//    One work item per 64 entries = exactly `bits` 64-bit words.  dOut needs ceil(n/64)*bits words.
template <class BE, class V> void packArray(BE &be, u64 n, u32 bits, u64 *dOut, V valueOf) {
    u64 nGroups = (n + 63) / 64;
    be.forEach(nGroups, [=] IDX_L (u64 g) {
        u64 *w = dOut + g * bits;
        u64 acc = 0; u32 fill = 0; u32 wi = 0;
        for (u32 e = 0; e < 64; e++) {
            u64 i = g * 64 + e;
            u64 v = i < n ? valueOf(i) : 0;
            acc |= fill < 64 ? (v << fill) : 0;
            if (fill + bits >= 64) {
                w[wi++] = acc;
                u32 used = 64 - fill;                   // bits of v already stored
                acc = used < 64 ? (v >> used) : 0;
                fill = bits - used;
            } else fill += bits;
        }
    });
}
IDX_HD u64 saValueOfPos(u64 pos, u64 N, u32 GstrandBit) { return pos < N ? pos : ((pos - N) | (1ull << GstrandBit)); }

// ---------------------------------------------------------------------------------------------------------------------
// 4. SAindex.  dSApos: nSA text positions in suffix order.  dSAiU: nSAi unpacked entries (output).
//    The reference walks the suffix array run by run (runs of equal (L-mer prefix, position of the first non-ACGT code))
//    and keeps one cursor per prefix length; every prefix length is independent of the others, and within one the
//    cursor is the running maximum of the prefixes seen, which turns the walk into a max-scan over the run list:
//      present prefix   first run that shows it: entry = isa of the run                      genomeSAindex.cpp:170-178
//      absent prefix    entry = isa of the next present prefix | absent flag                 :174-176
//      after the last   nSA | absent flag                                                    :192-196
//      N mark           a run whose first non-ACGT code is at offset <= iL marks the entry of the latest present prefix
//                       of every length iL1 >= that offset                                   :161-166
//    Returns 0, or 1 if the first suffix has a non-ACGT code inside the index prefix (the reference runs off its tables then).
template <class BE> int buildSAindex(BE &be, const u8 *T, const u64 *dSApos, u64 nSA, u32 L, u32 GstrandBit, const u64 *saiStart, u64 *dSAiU) {
    const u64 absentBit = 1ull << (GstrandBit + 2), nBit = 1ull << (GstrandBit + 1);
    // ---- run list.  The head flags are computed once, one thread per suffix (two classes = four 8-byte gathers each), and kept as bytes: the two
    // passes of the compaction then stream over them instead of classifying every suffix again, byte by byte, inside a serial loop per chunk
    // (8.7 s of a 9 s junction insertion at 3.1 Gb before)
    u8 *head = be.template alloc<u8>(nSA);
    be.forEach(nSA, [=] IDX_L (u64 i) {
        if (i == 0) { head[i] = 1; return; }
        int a4, b4; u32 a = saiClassFast(T, dSApos[i], L, a4), b = saiClassFast(T, dSApos[i - 1], L, b4);
        head[i] = (a != b || a4 != b4) ? 1 : 0;
    });
    auto isHead = [=] IDX_L (u64 i) { return head[i] != 0; };
    u64 R = compactIf(be, nSA, isHead, [=] IDX_L (u64, u64) {});
    u64 *runIsa = be.template alloc<u64>(R); u32 *runInd = be.template alloc<u32>(R); int8_t *runL4 = be.template alloc<int8_t>(R);
    compactIf(be, nSA, isHead, [=] IDX_L (u64 i, u64 o) { int l4; u32 c = saiClassFast(T, dSApos[i], L, l4); runIsa[o] = i; runInd[o] = c; runL4[o] = (int8_t)l4; });
    be.free(head);
    int bad = 0;
    { int8_t first; be.copyToHost(&first, runL4, 1); if (first != -1) bad = 1; }
    u64 *M = be.template alloc<u64>(R);
    for (u32 iL = 0; iL < L && !bad; iL++) {
        const u32 shift = 2 * (L - 1 - iL);
        const u64 base = saiStart[iL], levelN = saiStart[iL + 1] - saiStart[iL];
        // M[j] = 1 + largest prefix among the valid runs 0..j (0: none)
        be.forEach(R, [=] IDX_L (u64 j) { int l4 = runL4[j]; bool valid = l4 < 0 || (u32)l4 > iL; M[j] = valid ? (u64)(runInd[j] >> shift) + 1 : 0; });
        be.inclusiveMax(M, R);
        // present entries + the absent entries in front of them
        be.forEach(R, [=] IDX_L (u64 j) {
            int l4 = runL4[j]; bool valid = l4 < 0 || (u32)l4 > iL;
            if (!valid) return;
            u64 pref = runInd[j] >> shift; u64 prevM = j ? M[j - 1] : 0;
            if (pref + 1 > prevM) {
                u64 isa = runIsa[j];
                dSAiU[base + pref] = isa;
                for (u64 e = prevM; e < pref; e++) dSAiU[base + e] = isa | absentBit;
            }
        });
        // tail
        {
            u64 lastM = be.readOne(M + (R - 1));
            be.forEach(levelN - lastM, [=] IDX_L (u64 e) { dSAiU[base + lastM + e] = nSA | absentBit; });
        }
        // N marks (after all values of this level are in place)
        be.forEach(R, [=] IDX_L (u64 j) {
            int l4 = runL4[j];
            if (l4 < 0 || (u32)l4 > iL) return;
            u64 m = M[j];
            if (m == 0) return;
            be_atomicOr(dSAiU + base + (m - 1), nBit);
        });
    }
    be.free(M); be.free(runIsa); be.free(runInd); be.free(runL4);
    return bad;
}

} // namespace staridx

// ---------------------------------------------------------------------------------------------------------------------
// 5. the whole build: genome bytes in, packed SA + packed SAindex out (host buffers sized by the caller).
namespace staridx {

struct BuildResult {
    u64 nSA, nSAbyte, nSAi, nSAibyte, rounds;
    u64 saiStart[17];
};

static inline u64 packedBytes(u64 n, u32 bits) { return n == 0 ? 8 : (n - 1) * bits / 8 + 8; }      // PackedArray::defineBits
static inline u64 packedWords(u64 n, u32 bits) { return ((n + 63) / 64) * bits + 2; }

// hSA / hSAi may be null (sizes only are reported, e.g. to validate capacities); returns 0 or a negative error
//   -1 bad arguments   -2 capacity too small   -3 the first suffix has N inside the index prefix
template <class BE> int buildAll(BE &be, const u8 *hG, const BuildParams &P, u8 *hSA, u64 saCap, u8 *hSAi, u64 saiCap, BuildResult &R) {
    const u64 N = P.nGenome; const u32 L = P.saIndexNbases;
    if (N == 0 || L < 1 || L > 16 || P.GstrandBit + 3 > 63) return -1;
    R.saiStart[0] = 0;
    for (u32 i = 1; i <= L; i++) R.saiStart[i] = R.saiStart[i - 1] + (1ull << (2 * i));
    for (u32 i = L + 1; i < 17; i++) R.saiStart[i] = 0;
    R.nSAi = R.saiStart[L];
    u8 *dG = be.template alloc<u8>(N);
    be.copyToDevice(dG, hG, N);
    u8 *dTraw = be.template alloc<u8>(2 * N + 2 * TPAD);
    u8 *T = buildText(be, dG, N, dTraw);
    be.free(dG);
    u64 *dSA = be.template alloc<u64>(2 * N), *dISA = be.template alloc<u64>(2 * N);
    R.nSA = suffixSort(be, T, N, dSA, dISA, &R.rounds);
    be.free(dISA);
    const u32 saBits = P.GstrandBit + 1, saiBits = P.GstrandBit + 3;
    R.nSAbyte = packedBytes(R.nSA, saBits); R.nSAibyte = packedBytes(R.nSAi, saiBits);
    int rc = 0;
    if ((hSA && saCap < R.nSAbyte) || (hSAi && saiCap < R.nSAibyte)) rc = -2;
    if (!rc && R.nSA) {
        u64 nW = packedWords(R.nSA, saBits);
        u64 *dPacked = be.template alloc<u64>(nW);
        const u64 *sa = dSA; const u32 gsb = P.GstrandBit;
        packArray(be, R.nSA, saBits, dPacked, [=] IDX_L (u64 i) { return saValueOfPos(sa[i], N, gsb); });
        if (hSA) be.copyToHost(hSA, (const u8 *)dPacked, R.nSAbyte);
        be.free(dPacked);
        u64 *dSAiU = be.template alloc<u64>(R.nSAi);
        u64 *dStart = nullptr; (void)dStart;
        if (buildSAindex(be, T, dSA, R.nSA, L, P.GstrandBit, R.saiStart, dSAiU)) rc = -3;
        if (!rc) {
            u64 nWi = packedWords(R.nSAi, saiBits);
            u64 *dPi = be.template alloc<u64>(nWi);
            const u64 *su = dSAiU;
            packArray(be, R.nSAi, saiBits, dPi, [=] IDX_L (u64 i) { return su[i]; });
            if (hSAi) be.copyToHost(hSAi, (const u8 *)dPi, R.nSAibyte);
            be.free(dPi);
        }
        be.free(dSAiU);
    }
    be.free(dSA); be.free(dTraw);
    return rc;
}

} // namespace staridx
