// k_seed.hip -- kernel 1: maximal-mappable-prefix seed search.
//
// Replaces, per read, the seed phase of ReadAlign::mapOneRead (source/ReadAlign_mapOneRead.cpp:17-93):
// qualitySplit (SequenceFuns.cpp:411-444), maxMappableLength2strands
// (ReadAlign_maxMappableLength2strands.cpp:5-115), maxMappableLength / findMultRange (here: mmpRun) /
// compareSeqToGenome (SuffixArrayFuns.cpp:10-207) and storeAligns (ReadAlign_storeAligns.cpp:10-160).
//
// Mapping (round 5): one lane = one UNIT of a read's search schedule.  The schedule of mapOneRead is a nest -- pieces x directions x start points x the loop that
// restarts a search behind the last maximal mappable prefix -- around a search that is itself a chain of dependent gathers (SAindex entry -> packed SA word ->
// genome bytes, ~18 round trips).  With a read per lane (rounds 1-4) the 64 lanes of a wavefront sat at 64 different places of that nest and the wavefront
// paid every level at the pace of its slowest lane: ~4 200 serialised load round trips per wavefront for 470 in the mean lane, 84 % of the wave cycles waiting.
// The searches of different (piece, direction, start point) triples do not depend on each other -- only the ORDER in which their results reach storeAligns
// matters (its de-duplication keeps the first (rStart, Length) in schedule order), and one quirk: the backward search from the first start point is skipped when
// the forward one mapped the whole piece (ReadAlign_mapOneRead.cpp:74).  So:
//   k_seed_plan   lane = read: qualitySplit, the start points of every piece; one unit per (piece, direction, start point), the forward and the backward
//                 search of start point 0 in one unit (the quirk stays inside a lane); slots for what each unit finds, in schedule order
//   k_seed_units  lane = unit (persistent lanes, ticket): the restart loop of one start point -- one to three searches -- into its slots
//   k_seed_merge  lane = read: the slots in schedule order through storeAligns, classification of the read, seeds into the pool
// A wavefront of k_seed_units holds 64 chains that are all at the same level of the same (shallow) nest.  Measured (profiles/r05_ab_session3_*, r05_ab_sessions4-6_*): on
// par with the lane-per-read form (k_seed_units 8.9 + k_seed_merge 1.45 + k_seed_plan 0.25 ms against 10.3): the stage does not wait for its divergence.  A third form -- the
// searches of all units in ROUNDS, a lookup kernel and a bisection kernel per round with the bisections sorted by interval length -- was built, measured (11.4 - 19.8 ms:
// every kernel boundary waits for the slowest chain, a search in a repeat) and removed the same day.  Reads whose units or slots do not fit the pools, or a
// unit that finds more than SEED_SLOTS seeds, take k_seed_search (lane = read, the whole nest: the general form, no limits of its own) through a list.
#include "dev.h"

struct SeedCnt { u32 nSAi, nSAprobe, nGcmp; };     // per lane and kernel: a lane compares < 2^32 bases

// 8 bytes starting at an arbitrary address, little endian, through aligned 8-byte loads (the arrays are padded)
__device__ __forceinline__ u64 load8(const u8 *p) {
    const u64 a = (u64)p; const __attribute__((address_space(1))) u64 *q = GLOBAL(u64, a & ~7ull); const u32 sh = (u32)(a & 7ull) * 8u;
    const u64 w0 = q[0];
    if (sh == 0) return w0;
    return (w0 >> sh) | (q[1] << (64u - sh));
}
// bytes p[0], p[-1], ..., p[-7] as a little-endian sequence (a scan that walks backwards)
__device__ __forceinline__ u64 load8rev(const u8 *p) { return __builtin_bswap64(load8(p - 7)); }
// complementSeqNumbers on 8 codes at once: c < 4 -> 3 - c (== 3 ^ c), others unchanged
__device__ __forceinline__ u64 comp8(u64 x) {
    const u64 ge4 = ((x + 0x7C7C7C7C7C7C7C7Cull) & 0x8080808080808080ull) >> 7;       // 1 in bytes >= 4 (codes <= 11: no carry between bytes)
    return x ^ (0x0303030303030303ull & ~(ge4 * 0xFFull));
}

// ---- keys beside the suffix array (DevIndex::SAK, built at upload by k_sak_build below) ----------------------------------------------------------------------------
// A probe of the bisection reads a suffix-array entry and then the genome AT that entry: two dependent random gathers, two 64-byte sectors for 8 + a few bytes.  The
// companion array holds, per entry, the entry itself and beside it -- in the same 16 bytes -- the 32 bases of its suffix that follow the SAindex prefix (all suffixes of a
// search share that prefix), 2 bits per base in the order the compare walks them, with the number of bases before the first non-ACGT code in the spare top bits of the
// entry word.  The piece's own bases at the same offsets are packed once per search (QKey).  A probe then is ONE gather: XOR of the two keys, first differing pair = the
// match length and the order -- unless all 32 bases agree (then the genome decides, from offset prefix + 32 on).  Suffixes that differ from the piece within 46 bases are
// most probes of most searches.  Results are unchanged: the key is the genome's own text (k_sak_build), compared by the same rule.
struct alignas(16) SakRec { u64 w0, key; };
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
struct QKey { u64 key; u32 valid; };       // bases P0 .. P0+31 of the piece in scan order (complemented for a backward scan); valid = how many before the piece ends / a non-ACGT code
__device__ __forceinline__ u64 pack8to16(u64 x) {       // 8 codes 0..3, one per byte -> 16 bits, first byte lowest
    u64 z = x & 0x0303030303030303ull;
    z = (z | (z >> 6)) & 0x000F000F000F000Full;
    z = (z | (z >> 12)) & 0x000000FF000000FFull;
    return (z | (z >> 24)) & 0xFFFFull;
}
__device__ __forceinline__ QKey makeQKey(const DevIndex &X, const u8 *R, u32 S, u32 N, bool dirR) {
    QKey q; q.key = 0; q.valid = 0;
    const u32 P0 = X.sakBases;
    if (X.SAK == nullptr || N <= P0) return q;
    u32 valid = 32;
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
        u64 raw = dirR ? load8(R + S + P0 + 8u * j) : load8rev(R + S - P0 - 8u * j);
        const u64 bad = raw & 0xFCFCFCFCFCFCFCFCull;
        if (bad && valid == 32) valid = 8u * j + ((u32)__builtin_ctzll(bad) >> 3);
        if (!dirR) raw ^= 0x0303030303030303ull;                                          // 3 - base (codes above 3 lie behind `valid`)
        q.key |= pack8to16(raw) << (16u * j);
    }
    q.valid = min(valid, N - P0);
    return q;
}

// SuffixArrayFuns.cpp:10-104 -- the four read/genome direction variants folded into one loop, 8 bases per step:
// read and genome are both 1 byte/base, so a step is two 8-byte words, one XOR and a count-trailing-zeros
__device__ static u32 compareSeqToGenome(const DevIndex &X, const u8 *R, u32 S, u32 N, u32 L, u64 iSA, bool dirR, bool &compRes, SeedCnt &cn, const QKey &qk) {
    cn.nSAprobe++;
    u64 SAstr;
    if (X.SAK) {
        const u64x2 rec = GLOBAL(u64x2, X.SAK)[iSA];                    // one 16-byte gather
        const u64 w0 = rec.x, tkey = rec.y;
        SAstr = w0 & X.saMask;
        const u32 P0 = X.sakBases;
        if (L >= P0 && L < P0 + 32u && N > P0) {
            const u32 klen = (u32)(w0 >> 58), sh = L - P0, lim = min(min(klen, qk.valid), N - P0);        // (N: the caller's bound, which may be shorter than the piece)
            u64 d = tkey ^ qk.key;
            d = (d | (d >> 1)) & 0x5555555555555555ull & ~((1ull << (2u * sh)) - 1ull);
            const u32 k = d ? ((u32)__builtin_ctzll(d) >> 1) : 32u;
            if (k < lim) {                                                    // the first base that differs lies inside both keys
                cn.nGcmp += k + 1u - sh;
                compRes = ((qk.key >> (2u * k)) & 3ull) > ((tkey >> (2u * k)) & 3ull);
                return P0 + k;
            }
            if (lim >= sh) cn.nGcmp += lim - sh;
            if (lim == N - P0) return N;                                      // the piece ends inside the key: all of it matches
            if (!(lim == qk.valid && lim < 32u)) {                            // (a non-ACGT code of the READ inside the key -- only the --seedSearchLmax leg: the genome path below decides)
                if (lim < 32u) { compRes = false; return P0 + lim; }           // the suffix meets a non-ACGT code of the genome: a mismatch there, the piece never sorts above it
            }
            L = P0 + lim;                                                     // all bases of the keys agree: on in the genome
        }
    } else
    SAstr = packedGet(X.SA, iSA, X.saBits, X.saMask);
    bool dirG = (SAstr >> X.strandBit) == 0;
    SAstr &= X.strandMask;
    bool useComp = dirR != dirG;
    const u8 *g = dirG ? X.G + SAstr + L : X.G + (X.nGenome - 1 - SAstr) - L;
    const u8 *s = dirR ? R + S + L : R + S - L;
    const u32 n = N - L;
    for (u32 ii = 0; ii < n; ii += 8) {
        u64 s8 = dirR ? load8(s + ii) : load8rev(s - ii);
        const u64 g8 = dirG ? load8(g + ii) : load8rev(g - ii);
        if (useComp) s8 = comp8(s8);
        u64 d = s8 ^ g8;
        const u32 rem = n - ii;
        if (rem < 8) d &= (1ull << (rem * 8)) - 1ull;                   // bases behind the end of the piece do not count
        if (d) {
            const u32 k = (u32)__builtin_ctzll(d) >> 3;
            const u8 sc = (u8)(s8 >> (k * 8)), gc = (u8)(g8 >> (k * 8));
            cn.nGcmp += ii + k + 1;
            compRes = dirG ? (sc > gc) : !(sc > gc || gc > 3);
            return ii + k + L;
        }
    }
    cn.nGcmp += n;
    return N;
}

// The maximal mappable prefix of the piece among the suffixes SA[first .. last] and the run of entries that reach it -- what SuffixArrayFuns.cpp:133-207 (maxMappableLength) and
// :106-131 (findMultRange, twice) compute, in this engine's own shape.  The entries are sorted, so the match length of the piece rises towards the place where the piece itself
// would be inserted and falls behind it; its maximum Lmax is reached by one contiguous run.  Three bisections over ONE kind of state, a pair (entry known to match less, entry
// known to match at least as much):
//   1. the insertion place: a bracket (lo, hi), lo sorts before the piece, hi behind it.  While it narrows, each side remembers the nearest entry outside that matched LESS than
//      the bracket's end does (`loLess`: set when a step raised lo's match length) and the first entry that reached the end's length (`loSame`); an entry that matches all N
//      bases ends the search at once.  Lmax = the longer of the two ends (the right one on a tie, as the reference decides), or N.
//   2. the run's first entry: between the nearest entry known to match less than Lmax on the left and the nearest one known to reach it; compares stop at Lmax.
//   3. its last entry, mirrored.
// Which entries are probed on the way is an implementation detail (the result is a function of the interval and the piece); the probes here start every compare at the length
// both ends of the current pair are known to share, as the reference's do.  IDX: offsets from `first` (u32 when the interval is shorter than 2^32 entries: half the registers).
template <class IDX> struct LessSame { IDX less, same; u32 Lless; bool haveLess; };        // `less` matches Lless < the length `same` reaches; haveLess = false: nothing known, the interval's end
template <class IDX> __device__ static IDX runEnd(const DevIndex &X, const u8 *R, u64 first, LessSame<IDX> p, u32 Lmax, u32 S, bool dirR, SeedCnt &cn, const QKey &qk) {
    bool above;
    if (!p.haveLess) return p.less;                             // (`less` then names the interval's own end, which reaches Lmax like everything up to `same`)
    while (((u64)p.less + 1 < (u64)p.same) | ((u64)p.less > (u64)p.same + 1)) {
        const IDX mid = (IDX)((u64)p.less / 2 + (u64)p.same / 2 + ((u64)p.less % 2 + (u64)p.same % 2) / 2);
        const u32 Lm = compareSeqToGenome(X, R, S, Lmax, p.Lless, first + mid, dirR, above, cn, qk);
        if (Lm == Lmax) p.same = mid; else { p.less = mid; p.Lless = Lm; }
    }
    return p.same;
}
template <class IDX> __device__ static u64 mmpRunT(const DevIndex &X, const u8 *R, u32 S, u32 N, u64 first, u64 last, bool dirR, u32 &L, u64 &ind0, u64 &ind1, SeedCnt &cn, const QKey &qk) {
    bool above;                                                   // the piece sorts behind the entry just compared
    IDX lo = 0, hi = (IDX)(last - first);
    u32 Llo = compareSeqToGenome(X, R, S, N, L, first + lo, dirR, above, cn, qk);
    u32 Lhi = compareSeqToGenome(X, R, S, N, L, first + hi, dirR, above, cn, qk);
    LessSame<IDX> left = {lo, lo, Llo, false}, right = {hi, hi, Lhi, false};
    IDX top = lo; u32 Lmax = 0; bool full = false;
    while ((u64)lo + 1 < (u64)hi) {
        const IDX mid = (IDX)((u64)lo / 2 + (u64)hi / 2 + ((u64)lo % 2 + (u64)hi % 2) / 2);
        const u32 Lm = compareSeqToGenome(X, R, S, N, min(Llo, Lhi), first + mid, dirR, above, cn, qk);
        if (Lm == N) { top = mid; full = true; break; }
        if (above) { if (Lm > Llo) { left.less = lo; left.Lless = Llo; left.haveLess = true; left.same = mid; } lo = mid; Llo = Lm; }
        else       { if (Lm > Lhi) { right.less = hi; right.Lless = Lhi; right.haveLess = true; right.same = mid; } hi = mid; Lhi = Lm; }
    }
    if (full) Lmax = N; else if (Llo > Lhi) { top = lo; Lmax = Llo; } else { top = hi; Lmax = Lhi; }
    // a side whose bracket end falls short of Lmax: the run begins between that end and `top`
    if (Llo < Lmax) { left.less = lo; left.Lless = Llo; left.haveLess = true; left.same = top; }
    if (Lhi < Lmax) { right.less = hi; right.Lless = Lhi; right.haveLess = true; right.same = top; }
    const IDX r0 = runEnd<IDX>(X, R, first, left, Lmax, S, dirR, cn, qk), r1 = runEnd<IDX>(X, R, first, right, Lmax, S, dirR, cn, qk);
    L = Lmax; ind0 = first + r0; ind1 = first + r1;
    return (u64)r1 - (u64)r0 + 1;
}
__device__ __forceinline__ u64 mmpRun(const DevIndex &X, const u8 *R, u32 S, u32 N, u64 i1, u64 i2, bool dirR, u32 &L, u64 &ind0, u64 &ind1, SeedCnt &cn, const QKey &qk) {
    if (i2 - i1 < 0xFFFFFFFFull) return mmpRunT<u32>(X, R, S, N, i1, i2, dirR, L, ind0, ind1, cn, qk);
    return mmpRunT<u64>(X, R, S, N, i1, i2, dirR, L, ind0, ind1, cn, qk);
}

struct SeedState {
    DSeed *PC; u32 nP; u32 cap;
    u32 nA; u32 multNmin, multNminL;        // nA: only ever compared with 0 (ReadAlign_mapOneRead.cpp:106), saturating
    bool fatal;
};

// ReadAlign_storeAligns.cpp:10-160 (OPTIM_STOREaligns_SIMPLE branch)
__device__ static void storeAligns(const DevIndex &X, SeedState &st, u32 iDir, u32 Shift, u64 Nrep, u32 L, u64 ind0, u32 iFrag) {
    if (Nrep > X.P.seedMultimapNmax) { if (Nrep < st.multNmin || st.multNmin == 0) { st.multNmin = (u32)min(Nrep, (u64)0xFFFFFFFFu); st.multNminL = L; } return; }
    st.nA |= 1u;
    u32 rStart = iDir == 0 ? Shift : Shift + 1 - L;
    int iP;
    for (iP = (int)st.nP - 1; iP >= 0; iP--) {
        if (st.PC[iP].rStart <= rStart) {
            if (st.PC[iP].rStart == rStart && st.PC[iP].L < L) continue;
            if (st.PC[iP].rStart == rStart && st.PC[iP].L == L) return;
            break;
        }
    }
    iP++;
    if (st.nP + 1 > X.P.seedPerReadNmax || st.nP + 1 > st.cap) { st.fatal = true; return; }    // reference: exitWithError :46-51
    for (int ii = (int)st.nP - 1; ii >= iP; ii--) st.PC[ii + 1] = st.PC[ii];
    st.nP++;
    DSeed s; s.saStart = ind0; s.nrep = (u32)Nrep; s.rStart = (u16)rStart; s.L = (u16)L; s.dir = (u8)iDir; s.iFrag = (u8)iFrag;
    for (int k = 0; k < 6; k++) s.pad[k] = 0;
    st.PC[iP] = s;
    if (Nrep != 1) { if (Nrep < st.multNmin || st.multNmin == 0) { st.multNmin = (u32)Nrep; st.multNminL = L; } }
}

// ReadAlign_maxMappableLength2strands.cpp:23-84: the L-mer prefix of the piece, its SAindex entry (shortened while absent), the entry behind it, and what they say:
//   kind 0  the first base is absent from the genome: nothing           kind 1  a prefix shorter than the table's L-mers pins the interval: [i1, i2], length maxL, no search
//   kind 2  one suffix: its length is one compare away                   kind 3  bisection over [i1, i2] from a common length of maxL
struct SeedLook { u64 i1, i2; u32 maxL, kind; };
__device__ __forceinline__ SeedLook seedLookup(const DevIndex &X, const u8 *R, u32 pieceStart, u32 pieceLength, bool dirR, SeedCnt &cn) {
    SeedLook k; k.i1 = 0; k.i2 = 0; k.maxL = 0; k.kind = 0;
    u32 Lmax = min(X.saiNbases, pieceLength);
    u64 ind1 = 0;
    // L-mer prefix (:23-37): 2 bits per base, first base most significant; the bases of a piece are all 0..3, so 8 of them are packed from one 8-byte word with shifts and masks
    for (u32 ii = 0; ii < Lmax; ii += 8) {
        const u64 raw = dirR ? load8(R + pieceStart + ii) : load8rev(R + pieceStart - ii);     // byte k = code of the (ii+k)-th base of the scan
        const u32 nb = min(8u, Lmax - ii);
        const u64 used = nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
        if ((raw & used & 0xFCFCFCFCFCFCFCFCull) == 0) {
            u64 x = dirR ? raw : (raw ^ 0x0303030303030303ull);                                   // 3 - base == 3 ^ base
            u64 z = __builtin_bswap64(x & 0x0303030303030303ull); // first base in the top byte (bytes behind the prefix may hold N / spacer codes)
            z = (z | (z >> 6)) & 0x000F000F000F000Full;
            z = (z | (z >> 12)) & 0x000000FF000000FFull;
            z = (z | (z >> 24)) & 0xFFFFull;                    // 8 bases -> 16 bits, first base most significant
            ind1 = (ind1 << (2 * nb)) | (z >> (2 * (8 - nb)));
        } else {
            // a code above 3 inside the prefix: only with --seedSearchLmax, whose backward search is given Shift + 1 bases (:from ReadAlign_mapOneRead.cpp:81-86)
            // and so runs over the start of its piece into an N, the mate spacer or the other mate.  The reference adds the code as it is
            // (index*4 + code, index*4 + (3 - code) in unsigned 64-bit arithmetic): the carries and borrows are part of its result
            for (u32 k2 = 0; k2 < nb; k2++) { const u64 cde = (raw >> (8 * k2)) & 0xFFull; ind1 = (ind1 << 2) + (dirR ? cde : 3ull - cde); }
        }
    }
    u32 Lind = Lmax; u64 iSA1 = 0, iSA2;
    while (Lind > 0) {
        const u64 flat = X.saiStart[Lind - 1] + ind1;                                    // (wraps like the reference's flat packed-array index)
        if (flat >= X.saiStart[X.saiNbases]) { --Lind; ind1 >>= 2; continue; }          // outside the table: the reference reads whatever lies there; treated as absent
        iSA1 = packedGet(X.SAi, flat, X.saiBits, X.saiMask); cn.nSAi++;
        if ((iSA1 & X.saiAbsentBit) == 0) break;
        --Lind; ind1 >>= 2;
    }
    if (Lind == 0) return k;                                  // base absent from the genome (reference: out-of-bounds)
    bool iSA2good = true;
    if (X.saiStart[Lind - 1] + ind1 + 1 < X.saiStart[Lind]) {
        iSA2 = packedGet(X.SAi, X.saiStart[Lind - 1] + ind1 + 1, X.saiBits, X.saiMask); cn.nSAi++;
        if ((iSA2 & X.saiAbsentBit) == 0) iSA2 = (iSA2 & ~X.saiNbit) - 1;
        else { iSA2 = X.nSA - 1; iSA2good = false; }
    } else { iSA2 = X.nSA - 1; iSA2good = false; }
    const bool iSA1noN = (iSA1 & X.saiNbit) == 0;
    if (Lind < X.saiNbases && iSA1noN && iSA2good) { k.i1 = iSA1; k.i2 = iSA2; k.maxL = Lind; k.kind = 1; }
    else if (iSA1 == iSA2 && iSA1noN && iSA2good) { k.i1 = k.i2 = iSA1; k.maxL = Lind; k.kind = 2; }
    else { k.i1 = iSA1 & ~X.saiNbit; k.i2 = iSA2; k.maxL = (iSA2good && iSA1noN) ? Lind : 0; k.kind = 3; }
    return k;
}

// ReadAlign_maxMappableLength2strands.cpp:12-109: one start offset iDist of the sparse-SA loop
__device__ __forceinline__ void searchOneDist(const DevIndex &X, const u8 *R, u32 pieceStartIn, u32 pieceLengthIn, bool dirR, u32 iDist, u64 &Nrep, u64 &i0, u32 &maxL, SeedCnt &cn) {
    const u32 pieceLength = pieceLengthIn - iDist, pieceStart = dirR ? pieceStartIn + iDist : pieceStartIn - iDist;
    const SeedLook k = seedLookup(X, R, pieceStart, pieceLength, dirR, cn);
    u64 i1;
    if (k.kind == 0) { Nrep = 0; i0 = 0; maxL = 0; }
    else if (k.kind == 1) { i0 = k.i1; Nrep = k.i2 - k.i1 + 1; maxL = k.maxL; }
    else if (k.kind == 2) { i0 = k.i1; Nrep = 1; bool cr; const QKey qk = makeQKey(X, R, pieceStart, pieceLength, dirR); maxL = compareSeqToGenome(X, R, pieceStart, pieceLength, k.maxL, k.i1, dirR, cr, cn, qk); }
    else { maxL = k.maxL; const QKey qk = makeQKey(X, R, pieceStart, pieceLength, dirR); Nrep = mmpRun(X, R, pieceStart, pieceLength, k.i1, k.i2, dirR, maxL, i0, i1, cn, qk); }
}

// ReadAlign_maxMappableLength2strands.cpp:5-115.  The reference keeps (Nrep, ind0, maxL) of every start offset of a sparse suffix
// array in small tables and stores the best ones afterwards; indexed local tables would live in scratch memory here, so with
// genomeSAsparseD > 1 the offsets are simply searched twice (first for the best length, then to store) -- with the default
// full suffix array there is one offset and one search.
// SINK: where a seed goes -- storeAligns at once (lane = read), or the slots of the unit (lane = unit; storeAligns runs later, in schedule order)
struct StoreNow {
    const DevIndex &X; SeedState &st;
    __device__ __forceinline__ void operator()(u32 iDir, u32 Shift, u64 Nrep, u32 L, u64 ind0, u32 iFrag) { storeAligns(X, st, iDir, Shift, Nrep, L, ind0, iFrag); }
};
template <class SINK> __device__ static void maxMappableLength2strands(const DevIndex &X, const u8 *R, SINK &sink, u32 pieceStartIn, u32 pieceLengthIn, u32 iDir, u32 &maxLbest, u32 iFrag, SeedCnt &cn) {
    const bool dirR = iDir == 0;
    const u32 nD = min(pieceLengthIn, X.sparseD);
    maxLbest = 0;
    for (u32 it = 0; it < 2 * nD; it++) {
        const u32 phase = it >= nD ? 1u : 0u, iDist = phase ? it - nD : it;
        u64 Nrep, i0; u32 maxL;
        searchOneDist(X, R, pieceStartIn, pieceLengthIn, dirR, iDist, Nrep, i0, maxL, cn);
        if (phase == 0) {
            if (maxL + iDist > maxLbest) maxLbest = maxL + iDist;
            if (nD > 1) continue;
        }
        if (maxL + iDist == maxLbest && Nrep > 0)
            sink(iDir, dirR ? pieceStartIn + iDist : pieceStartIn - iDist, Nrep, maxL, i0, iFrag);
        if (nD == 1) break;
    }
}

// ReadAlign_mapOneRead.cpp:57-92, the body of the loop over start points for one direction: the search restarts behind the last maximal mappable prefix until the
// piece is used up, then the extra search of --seedSearchLmax.  flagDirMap: :62 / :74
template <class SINK> __device__ __forceinline__ void searchFromStart(const DevIndex &X, const u8 *R, SINK &sink, u32 pS, u32 pL, u32 iDir, u32 istart, u32 Lstart, u32 iFrag, bool &flagDirMap, SeedCnt &cn) {
    const staramd_params &P = X.P;
    u32 Lm;
    if (flagDirMap || istart > 0) {
        u32 Lmapped = 0;
        while (istart * Lstart + Lmapped + P.seedMapMin < pL) {
            u32 Shift = iDir == 0 ? (pS + istart * Lstart + Lmapped) : (pS + pL - istart * Lstart - 1 - Lmapped);
            u32 seedLength = pL - Lmapped - istart * Lstart;
            maxMappableLength2strands(X, R, sink, Shift, seedLength, iDir, Lm, iFrag, cn);
            if (iDir == 0 && istart == 0 && Lmapped == 0 && Shift + Lm == pL) flagDirMap = false;
            Lmapped += Lm;
            if (Lm == 0) break;
        }
    }
    if (P.seedSearchLmax > 0) {
        u32 Shift = iDir == 0 ? (pS + istart * Lstart) : (pS + pL - istart * Lstart - 1);
        u32 seedLength = min(P.seedSearchLmax, iDir == 0 ? (pS + pL - Shift) : (Shift + 1));
        maxMappableLength2strands(X, R, sink, Shift, seedLength, iDir, Lm, iFrag, cn);
    }
}

// qualitySplit (SequenceFuns.cpp:411-444), one piece at a time: the next run of codes <= 3 from iR on; false at the end of the read.  iFrag counts the mate spacers passed
__device__ __forceinline__ bool nextPiece(const u8 *R, u32 Lread, u32 &iR, u32 &iFrag, u32 &pS, u32 &pL) {
    while (iR < Lread && R[iR] > 3) { if (R[iR] == STARAMD_SPACER_BASE) iFrag++; iR++; }
    if (iR == Lread) return false;
    pS = iR;
    for (;;) {                                                  // end of the run of good bases, 8 bases per step
        const u64 bad = load8(R + iR) & 0xFCFCFCFCFCFCFCFCull;
        const u32 k = bad ? ((u32)__builtin_ctzll(bad) >> 3) : 8u;
        iR += k;
        if (iR >= Lread) { iR = Lread; break; }
        if (k < 8u) break;
    }
    pL = iR - pS;
    return true;
}
__device__ __forceinline__ u32 startLmaxOf(const staramd_params &P, u32 Lread) { return min(P.seedSearchStartLmax, (u32)(u64)(P.seedSearchStartLmaxOverLread * (double)(u64)(Lread - 1))); }
__device__ __forceinline__ u32 nStartOf(const staramd_params &P, u32 startLmax, u32 pL) { return (P.seedSearchStartLmax > 0 && startLmax < pL) ? pL / startLmax + 1 : 1; }

// classification of a read whose searches are done (ReadAlign_mapOneRead.cpp:100-115); true: its st.nP seeds want a place in the pool
__device__ __forceinline__ bool classifyRead(const staramd_params &P, DRead &rd, u32 Lread, const SeedState &st, u32 Nsplit, u32 LgoodMin) {
    rd.status = 0; rd.seedOffset = 0; rd.nSeeds = 0; rd.unmappedLength = 0; rd.winOffset = 0; rd.nWin = 0; rd.wtOffset = 0; rd.nWt = 0; rd.pruneBest = 0; rd.pad0 = 0;
    rd.maxScoreMate[0] = rd.maxScoreMate[1] = 0; rd.bestW = -1; rd.nTr = 0; rd.nEx = 0;
    if (st.fatal) rd.status |= STARAMD_ST_FATAL_SEEDS_PER_READ;
    else if (Lread < P.outFilterMatchNmin) { rd.status |= STARAMD_ST_READ_TOO_SHORT; rd.unmappedLength = 0; }
    else if (Nsplit == 0) { rd.status |= STARAMD_ST_NO_GOOD_PIECES; rd.unmappedLength = LgoodMin; }
    else if (st.nA == 0) { rd.status |= STARAMD_ST_ALL_PIECES_MULTI; rd.unmappedLength = st.multNminL; }
    else return true;
    return false;
}
__device__ __forceinline__ void placeSeeds(DevBatch &B, DRead &rd, const SeedState &st, u32 off) {
    if (off + st.nP > B.seedCap) { atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_SEEDPOOL); return; }
    rd.seedOffset = off; rd.nSeeds = st.nP;
    for (u32 k = 0; k < st.nP; k++) B.seedPool[off + k] = st.PC[k];
}
__device__ static void finishRead(const DevIndex &X, DevBatch &B, u32 ir, u32 Lread, const SeedState &st, u32 Nsplit, u32 LgoodMin) {
    DRead rd;
    if (classifyRead(X.P, rd, Lread, st, Nsplit, LgoodMin)) placeSeeds(B, rd, st, atomicAdd(&B.cursors[CUR_SEED], st.nP));
    B.reads[ir] = rd;
}

// (a counter in global memory serves ~0.5 G atomics/s whoever asks -- profiles/r05_ab_sessions4-6_*: what every lane of a kernel adds at its end is summed per wavefront first)
__device__ __forceinline__ void addCounters(DevBatch &B, const SeedCnt &cn) {
    const u32 a = waveSumU32(cn.nSAi), b = waveSumU32(cn.nSAprobe), c = waveSumU32(cn.nGcmp);
    if (laneId() == 0) {
        atomicAdd((unsigned long long *)&B.counters[DC_nSAi], (unsigned long long)a);
        atomicAdd((unsigned long long *)&B.counters[DC_nSAprobe], (unsigned long long)b);
        atomicAdd((unsigned long long *)&B.counters[DC_nGcmp], (unsigned long long)c);
    }
}
#ifndef SEED_WAVES
#define SEED_WAVES 7        // minimum waves per SIMD the register allocation is held to.  With the keys beside the suffix array (round 6) a probe is one gather and the stage needs
                            // fewer chains in flight than it needs registers: same box, 3.1 Gb, ms per 400 k pairs: 8 (64 VGPRs, 67 spilled) 7.4, 7 (72, 33 spilled) 6.8, 6 (80, 28) 6.85
                            // (profiles/r06_ab_session6_*); without the keys 8 was 10 % faster than 4 (round 3)
#endif
// the whole nest of one read on one lane: the general form (any number of pieces, start points and seeds per search).  Runs over the reads of `inList`
// (B.cursors[CUR_OVF_SEED] of them: what the unit mapping below handed on), or over every read when inList is null (STARAMD_SEED_UNITS=0)
extern "C" __global__ void __launch_bounds__(256, SEED_WAVES) k_seed_search(const DevIndex *__restrict__ Xp, DevBatch B, DSeed *scratch, u32 scratchPerLane, const u32 *inList) {
    const DevIndex &X = *Xp;
    u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    SeedState st; st.PC = scratch + (u64)lane * scratchPerLane; st.cap = scratchPerLane;
    StoreNow sink{X, st};
    SeedCnt cn = {0, 0, 0}; u64 nSeedsTot = 0;
    const staramd_params &P = X.P;
    const u32 nItems = inList ? B.cursors[CUR_OVF_SEED] : B.nReads;
    for (;;) {
        if (*(volatile u32 *)&B.cursors[CUR_TICKET_SEED] >= nItems) break;          // (a lane that comes for nothing does not queue for the counter)
        u32 it = atomicAdd(&B.cursors[CUR_TICKET_SEED], 1u);
        if (it >= nItems) break;
        const u32 ir = inList ? inList[it] : it;
        const u8 *R = B.bases + B.readOffset[ir];
        u32 Lread = (u32)(B.readOffset[ir + 1] - B.readOffset[ir]);
        st.nP = 0; st.nA = 0; st.multNmin = 0; st.multNminL = 0; st.fatal = false;
        // qualitySplit and the loop over its pieces (ReadAlign_mapOneRead.cpp:40-93) fused: a piece is searched as soon as its end is found
        u32 Nsplit = 0, LgoodMin = 0;
        const u32 startLmax = startLmaxOf(P, Lread);
        u32 iR = 0, iFrag = 0, pS = 0, pL = 0;
        while (Nsplit < P.maxNsplit && nextPiece(R, Lread, iR, iFrag, pS, pL)) {
            if (pL > LgoodMin) LgoodMin = pL;
            if (pL < P.seedSplitMin) continue;
            Nsplit++;
            const u32 Nstart = nStartOf(P, startLmax, pL), Lstart = pL / Nstart;
            bool flagDirMap = true;
            for (u32 iDir = 0; iDir < 2; iDir++)
                for (u32 istart = 0; istart < Nstart; istart++) searchFromStart(X, R, sink, pS, pL, iDir, istart, Lstart, iFrag, flagDirMap, cn);
        }
        nSeedsTot += st.nP;
        finishRead(X, B, ir, Lread, st, Nsplit, LgoodMin);
    }
    addCounters(B, cn);
    { const u32 ns = waveSumU32((u32)nSeedsTot); if (laneId() == 0) atomicAdd((unsigned long long *)&B.counters[DC_nSeeds], (unsigned long long)ns); }
}

// ---- lane = unit -------------------------------------------------------------------------------------------------------------------------------------------
// group = one (piece, direction, start point) of a read, in schedule order: piece by piece, all start points forward, then all start points backward.
// A group owns SEED_SLOTS slots; its header word says how many are filled and what storeAligns needs beside them.
struct SlotSink {
    SeedSlot *slot; u32 n, limit; bool over;
    __device__ __forceinline__ void operator()(u32, u32 Shift, u64 Nrep, u32 L, u64 ind0, u32) {
        if (n >= limit) { over = true; return; }
        SeedSlot c; c.i0 = ind0; c.nrep = (u32)min(Nrep, (u64)0xFFFFFFFFu); c.shift = (u16)Shift; c.L = (u16)(L | (Nrep > 0xFFFFFFFFull ? 0x8000u : 0u));
        slot[n++] = c;
    }
};

extern "C" __global__ void __launch_bounds__(256) k_seed_plan(const DevIndex *__restrict__ Xp, DevBatch B, SeedWork W) {
    const DevIndex &X = *Xp; const staramd_params &P = X.P;
    const u32 ir0 = blockIdx.x * blockDim.x + threadIdx.x, lane = laneId();
    const bool live = ir0 < B.nReads;
    const u32 ir = live ? ir0 : 0u;                  // (the lanes behind the last read stay for the wavefront's scan and write nothing)
    const u8 *R = B.bases + B.readOffset[ir];
    const u32 Lread = live ? (u32)(B.readOffset[ir + 1] - B.readOffset[ir]) : 0u;
    const u32 startLmax = startLmaxOf(P, Lread);
    u32 Nsplit = 0, LgoodMin = 0, nGroups = 0, nUnits = 0, maxNstart = 0;
    { u32 iR = 0, iFrag = 0, pS = 0, pL = 0;
      while (Nsplit < P.maxNsplit && nextPiece(R, Lread, iR, iFrag, pS, pL)) {
          if (pL > LgoodMin) LgoodMin = pL;
          if (pL < P.seedSplitMin) continue;
          Nsplit++;
          const u32 Nstart = nStartOf(P, startLmax, pL);
          if (Nstart > maxNstart) maxNstart = Nstart;
          nGroups += 2u * Nstart; nUnits += 2u * Nstart - 1u;
      } }
    SeedPlan pl; pl.group0 = 0; pl.nGroups = (u16)min(nGroups, 0xFFFFu); pl.nSplit = (u16)Nsplit; pl.LgoodMin = (u16)min(LgoodMin, 0xFFFFu); pl.handOn = 0; pl.pad = 0;
    // groups and units of the wavefront's 64 reads in one piece each: two atomics per wavefront (an exclusive scan over the lanes gives every read its place)
    // a read of more groups than k_seed_merge has lanes for (one per group), or with more start points per piece than SeedUnit::istart / nstart hold (u8):
    // the general kernel -- handed on HERE, so that no unit of it is allocated, searched and thrown away
    const bool huge = nGroups > 64u || maxNstart > 255u;
    u32 packed = huge ? 0u : (nGroups | (nUnits << 16)), incl = packed;
    for (u32 d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl((int)incl, (int)(lane >= d ? lane - d : 0u), 64); if (lane >= d) incl += o; }
    const u32 tot = (u32)__shfl((int)incl, 63, 64);
    u32 gBase = 0, uBase = 0;
    if (lane == 0 && tot) { gBase = atomicAdd(&B.cursors[CUR_SEED_GROUPS], tot & 0xFFFFu); uBase = atomicAdd(&B.cursors[CUR_SEED_UNITS], tot >> 16); }
    gBase = first32(gBase); uBase = first32(uBase);
    if (!live) return;
    if (huge) pl.handOn = 1;
    else if (nGroups) {
        const u32 g0 = gBase + ((incl - packed) & 0xFFFFu), u0 = uBase + ((incl - packed) >> 16);
        const bool fits = g0 + nGroups <= W.groupCap && u0 + nUnits <= W.unitCap && Lread <= 0x7FFFu;
        if (!fits) {
            // the unit slots of this read that lie inside the pool are marked empty (the cursor has counted them); the read takes the general kernel
            for (u32 u = u0; u < u0 + nUnits && u < W.unitCap; u++) W.units[u].read = 0xFFFFFFFFu;
            pl.handOn = 1;
        } else {
            pl.group0 = g0;
            u32 iR = 0, iFrag = 0, pS = 0, pL = 0, g = g0, u = u0, k = 0;
            while (k < Nsplit && nextPiece(R, Lread, iR, iFrag, pS, pL)) {
                if (pL < P.seedSplitMin) continue;
                k++;
                const u32 Nstart = nStartOf(P, startLmax, pL);
                for (u32 istart = 0; istart < Nstart; istart++)
                    for (u32 iDir = 0; iDir < 2; iDir++) {
                        if (istart == 0 && iDir == 1) continue;                       // backward from start point 0: second half of the unit of (forward, 0)
                        SeedUnit un; un.read = ir; un.group = g + iDir * Nstart + istart; un.pS = (u16)pS; un.pL = (u16)pL; un.iFrag = (u8)iFrag; un.istart = (u8)istart;
                        un.nstart = (u8)Nstart; un.kind = (u8)(istart == 0 ? 0u : 1u + iDir);
                        W.units[u++] = un;
                    }
                g += 2u * Nstart;
            }
        }
    }
    W.plan[ir] = pl;
}

extern "C" __global__ void __launch_bounds__(256, SEED_WAVES) k_seed_units(const DevIndex *__restrict__ Xp, DevBatch B, SeedWork W) {
    const DevIndex &X = *Xp;
    SeedCnt cn = {0, 0, 0};
    const u32 nUnits = min(B.cursors[CUR_SEED_UNITS], W.unitCap);
    for (;;) {
        if (*(volatile u32 *)&B.cursors[CUR_TICKET_SEED_UNITS] >= nUnits) break;      // (a lane that comes for nothing does not queue for the counter)
        const u32 u = atomicAdd(&B.cursors[CUR_TICKET_SEED_UNITS], 1u);
        if (u >= nUnits) break;
        const SeedUnit un = W.units[u];
        if (un.read == 0xFFFFFFFFu) continue;
        const u8 *R = B.bases + B.readOffset[un.read];
        const u32 pS = un.pS, pL = un.pL, Nstart = un.nstart, Lstart = pL / Nstart, istart = un.istart, iFrag = un.iFrag;
        bool flagDirMap = true;
        const u32 dir0 = un.kind == 2u ? 1u : 0u, dir1 = un.kind == 1u ? 0u : 1u;      // kind 0: forward then backward; 1: forward; 2: backward
        for (u32 iDir = dir0; iDir <= dir1; iDir++) {
            const u32 g = un.group + (un.kind == 0u && iDir == 1u ? Nstart : 0u);
            SlotSink sink; sink.slot = W.slots + (u64)g * SEED_SLOTS; sink.n = 0; sink.limit = W.slotLimit; sink.over = false;
            searchFromStart(X, R, sink, pS, pL, iDir, istart, Lstart, iFrag, flagDirMap, cn);
            W.groupHead[g] = sink.n | (iDir << 8) | (iFrag << 16) | (sink.over ? 0x80000000u : 0u);
        }
    }
    addCounters(B, cn);
}

// storeAligns (ReadAlign_storeAligns.cpp:10-160) over ALL seeds of a read at once, wave = read, lane = seed (in schedule order).  What the sequential inserts produce is a
// function of the set: the table sorted by (rStart ascending, Length descending) (:31-42: a seed goes in front of the shorter ones with its rStart and behind everything that
// starts earlier), a seed dropped when an EARLIER one of the schedule has its (rStart, Length) (:38: the first one stays -- their `dir` may differ), seeds with more loci than
// seedMultimapNmax never stored (:14-21); multNmin / multNminL = the smallest number of loci among the seeds that were too many and the stored multimappers, the first of the
// schedule among equals (:16-19, :152-158: strictly smaller replaces).  So: a key per lane, two all-pairs passes over broadcast keys (duplicates, then ranks) -- no sorted
// inserts through memory, which took k_seed_merge 1.45 ms with a lane per read.
extern "C" __global__ void __launch_bounds__(256) k_seed_merge(const DevIndex *__restrict__ Xp, DevBatch B, SeedWork W) {
    const DevIndex &X = *Xp; const staramd_params &P = X.P;
    const u32 lane = laneId(), nWaves = gridDim.x * (blockDim.x >> 6), wave = blockIdx.x * (blockDim.x >> 6) + WAVE_INDEX(threadIdx.x >> 6);
    u32 poolNext = 0, poolEnd = 0;                   // the wavefront's piece of the seed pool: taken 256 rows at a time (one allocation per read would be 400 k atomics on one counter)
    u32 nSeedsTot = 0;
    for (u32 ir = wave; ir < B.nReads; ir += nWaves) {
        const SeedPlan pl = W.plan[ir];
        const u32 nG = pl.nGroups;
        u32 head = 0;
        if (lane < nG) head = W.groupHead[pl.group0 + lane];
        bool handOn = pl.handOn != 0 || nG > 64u || __ballot(lane < nG && (head & 0x80000000u)) != 0;          // a unit ran out of slots (or met an interval of > 2^32 entries)
        // seeds of the groups, one behind the other: inclusive sums of the counts over the lanes
        u32 incl = lane < nG ? (head & 0xFFu) : 0u;
        for (u32 d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl((int)incl, (int)(lane >= d ? lane - d : 0u), 64); if (lane >= d) incl += o; }
        const u32 T = (u32)__shfl((int)incl, 63, 64);
        if (T > 64u) handOn = true;
        if (handOn) { if (lane == 0) { const u32 k = atomicAdd(&B.cursors[CUR_OVF_SEED], 1u); W.handOn[k] = ir; } continue; }
        // lane i < T: the group its seed is in = the first group whose inclusive sum is above i
        u32 g = 0;
        for (u32 k = 0; k < nG; k++) { const u32 e = laneGet32(incl, k); if (lane >= e) g = k + 1u; }
        const bool have = lane < T;
        if (!have) g = 0;
        const u32 hg = (u32)__shfl((int)head, (int)g, 64), eg = (u32)__shfl((int)incl, (int)g, 64);
        SeedSlot c; c.i0 = 0; c.nrep = 0; c.shift = 0; c.L = 0;
        if (have) c = W.slots[(u64)(pl.group0 + g) * SEED_SLOTS + (lane - (eg - (hg & 0xFFu)))];
        const u32 iDir = (hg >> 8) & 1u, iFrag = (hg >> 16) & 0xFFu, L = c.L & 0x7FFFu;
        const bool tooMany = have && ((c.L & 0x8000u) != 0 || c.nrep > P.seedMultimapNmax), stor = have && !tooMany;
        const u32 rStart = iDir == 0 ? c.shift : c.shift + 1u - L;
        const u32 key = (rStart << 16) | (0xFFFFu - L);
        const u64 storMask = __ballot(stor);
        bool dup = false;
        for (u64 m = storMask; m; m &= m - 1) { const u32 j = firstLane(m); const u32 kj = laneGet32(key, j); dup |= j < lane && kj == key; }
        const bool kept = stor && !dup;
        const u64 keptMask = __ballot(kept);
        u32 rank = 0;
        for (u64 m = keptMask; m; m &= m - 1) { const u32 j = firstLane(m); const u32 kj = laneGet32(key, j); rank += kj < key ? 1u : 0u; }
        SeedState st; st.PC = nullptr; st.cap = 0;
        st.nP = (u32)__popcll(keptMask); st.nA = storMask ? 1u : 0u;
        st.fatal = st.nP > P.seedPerReadNmax;
        // multNmin / multNminL
        const bool counts = tooMany || (kept && c.nrep != 1u);
        const u32 v = counts ? c.nrep : 0xFFFFFFFFu;          // (a number of loci beyond 32 bits is kept as 0xFFFFFFFF, as storeAligns keeps it)
        const u64 cm = __ballot(counts);
        st.multNmin = 0; st.multNminL = 0;
        if (cm) { const u32 vmin = ~waveMaxU32(~v); const u64 at = __ballot(counts && v == vmin); const u32 f = firstLane(at); st.multNmin = vmin; st.multNminL = laneGet32(L, f); }
        nSeedsTot += st.nP;
        DRead rd;
        const bool wants = classifyRead(P, rd, (u32)(B.readOffset[ir + 1] - B.readOffset[ir]), st, pl.nSplit, pl.LgoodMin);
        if (wants) {
            if (poolNext + st.nP > poolEnd) { u32 b = 0; if (lane == 0) b = atomicAdd(&B.cursors[CUR_SEED], 256u); poolNext = first32(b); poolEnd = poolNext + 256u; }
            if (poolNext + st.nP > B.seedCap) { if (lane == 0) atomicOr(&B.cursors[CUR_FLAGS], (u32)OVF_SEEDPOOL); }
            else {
                rd.seedOffset = poolNext; rd.nSeeds = st.nP;
                if (kept) { DSeed sd; sd.saStart = c.i0; sd.nrep = c.nrep; sd.rStart = (u16)rStart; sd.L = (u16)L; sd.dir = (u8)iDir; sd.iFrag = (u8)iFrag; for (int k = 0; k < 6; k++) sd.pad[k] = 0; B.seedPool[poolNext + rank] = sd; }
            }
            poolNext += st.nP;
        }
        if (lane == 0) B.reads[ir] = rd;
    }
    if (lane == 0 && nSeedsTot) atomicAdd((unsigned long long *)&B.counters[DC_nSeeds], (unsigned long long)nSeedsTot);
}

// ---- the companion array: per suffix-array entry {entry | bases-before-the-first-non-ACGT << 58, 32 bases of the suffix behind the SAindex prefix, 2 bits each} ------------
// lane = entry.  The text of an entry of the reverse strand is the complement of the genome walked backwards -- what compareSeqToGenome's dirG = false branch compares against.
extern "C" __global__ void __launch_bounds__(256) k_sak_build(const DevIndex *__restrict__ Xp, u64 *__restrict__ out_, u64 n0, u64 n1) {
    SakRec *out = (SakRec *)out_;
    const DevIndex &X = *Xp;
    const u32 P0 = X.saiNbases;
    for (u64 i = n0 + (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += (u64)gridDim.x * blockDim.x) {
        const u64 v = packedGet(X.SA, i, X.saBits, X.saMask);
        const bool fwd = (v >> X.strandBit) == 0;
        const u64 p = v & X.strandMask;
        const u8 *g = fwd ? X.G + p + P0 : X.G + (X.nGenome - 1 - p) - P0;
        u64 key = 0; u32 klen = 32;
#pragma unroll
        for (u32 j = 0; j < 4; j++) {
            u64 raw = fwd ? load8(g + 8u * j) : load8rev(g - 8u * j);
            const u64 bad = raw & 0xFCFCFCFCFCFCFCFCull;
            if (bad && klen == 32) klen = 8u * j + ((u32)__builtin_ctzll(bad) >> 3);
            if (!fwd) raw ^= 0x0303030303030303ull;
            key |= pack8to16(raw) << (16u * j);
        }
        SakRec r; r.w0 = v | ((u64)klen << 58); r.key = key; out[i] = r;
    }
}
