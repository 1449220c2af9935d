// stitch_scalar.h -- the three per-base routines of the stitcher for ONE LANE: growing an end of a transcript (growOnLane), finding an annotated
// junction (junctionOnLane), joining the next seed to the working transcript (joinOnLane).  Same results as
//   extendAlign (source/extendAlign.cpp:6-93), binarySearch2 (source/binarySearch2.cpp:3-43), stitchAlignToTranscript (source/stitchAlignToTranscript.cpp:9-415),
// which is also what the wave-cooperative versions of k_stitch.hip compute with 64 lanes per call.
// Two users:
//   * k_stitch_lane.hip (product): one lane walks one read, 64 independent walks per wavefront;
//   * the shadow-validation build of the cooperative stitcher (-DSTARAMD_SHADOW, k_stitch.hip): every cooperative call is re-run through these on every lane
//     and disagreements are counted (DC_shadow*).
//
// Shape.  A lane has no neighbours to share a scan with, so the unit of work is a RUN: 8 consecutive positions of a scan, fetched in one trip -- 8 genome bytes
// out of two aligned 8-byte words (the genome is one byte per base), 8 read codes out of the 4-bit packed read in the lane's LDS slot -- and reduced at once to
// 8-bit masks (equal / both A,C,G,T / padding / mate spacer) by byte-parallel arithmetic on the two 64-bit words.  The rules of the three routines are then
// applied to the set bits of those masks: a scan that the reference walks base by base with two or three loads per base costs two or three loads per EIGHT
// bases here, and the sequential decisions (where a prefix is recorded, where the mismatch budget ends a scan, where a junction scores best) are taken once per
// mismatch / per candidate position on register-resident masks.
#pragma once
#include "stitch_common.h"

struct ExtRes { i32 maxScore; u32 extendL, nMatch, nMM; };

// ---- byte-parallel helpers: a run = 8 codes in the 8 bytes of a u64, byte k = k-th position of the scan ---------------------------------------------
#define B8(x) ((u64)(x) * 0x0101010101010101ull)
// bit k of the result = byte k of t is zero
__device__ __forceinline__ u32 zeroBytes(u64 t) {
    u64 y = (t & B8(0x7F)) + B8(0x7F);
    y = ~(y | t | B8(0x7F));                                   // 0x80 in every byte of t that is zero, nothing else
    return (u32)(((y >> 7) * 0x0102040810204080ull) >> 56);
}
__device__ __forceinline__ u32 sameBytes(u64 a, u64 b) { return zeroBytes(a ^ b); }
__device__ __forceinline__ u32 acgtBytes(u64 a) { return zeroBytes(a & B8(0xFC)); }            // codes 0..3
__device__ __forceinline__ u32 bytesEqual(u64 a, u32 code) { return zeroBytes(a ^ B8(code)); }
__device__ __forceinline__ u64 spreadNibbles(u32 n) {         // nibble k -> byte k
    u64 x = n;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull; x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    return x;
}
__device__ __forceinline__ u64 complementBytes(u64 c) {       // codes 0..3 -> 3 - code, everything else stays (complementSeqNumbers, SequenceFuns.cpp:4-14)
    u64 big = c & B8(0x0C); big = ((big >> 2) | (big >> 3)) & B8(0x01);
    return c ^ ((B8(0x01) ^ big) * 3u);
}
__device__ __forceinline__ u32 lowBits(u32 n) { return n >= 32u ? 0xFFFFFFFFu : ((1u << n) - 1u); }

// 8 genome bases pos .. pos+7 (up = true) or pos, pos-1, .. pos-7 (up = false) as a run; two aligned words per trip, the word the next trip of the same scan
// starts in is kept.  GPAD bytes of padding lie on both sides of the genome on the device (dev.h): a run may reach into them, its bytes there are code 5.
struct GenomeRuns { i64 base; u64 w0, w1; };
__device__ __forceinline__ void runsInit(GenomeRuns &s) { s.base = (i64)0x7fffffffffffff00ll; s.w0 = s.w1 = 0; }
__device__ __forceinline__ u64 genomeRun(StitchCtx &c, GenomeRuns &s, i64 pos, bool up) {
    const i64 first = up ? pos : pos - 7;
    const i64 b = first & ~7ll;
    const u8 *G = c.X->G;
    if (b != s.base) {
        if (b == s.base + 8) { s.w0 = s.w1; s.w1 = *GLOBAL(u64, G + b + 8); }
        else if (b == s.base - 8) { s.w1 = s.w0; s.w0 = *GLOBAL(u64, G + b); }
        else { s.w0 = *GLOBAL(u64, G + b); s.w1 = *GLOBAL(u64, G + b + 8); }
        s.base = b;
    }
    const u32 sh = (u32)(first & 7) * 8u;
    const u64 v = sh ? ((s.w0 >> sh) | (s.w1 << (64u - sh))) : s.w0;
    return up ? v : __builtin_bswap64(v);
}
// 8 read codes R[i], R[i+1] .. (up) or R[i], R[i-1] .. (down), R[] = the read as the window's strand sees it (ReadAlign_stitchPieces.cpp:321: Read1[0] for + windows,
// the reverse complement for - windows).  Positions outside 0 .. Lread come back as whatever lies there: the callers bound their scans by lengths.
__device__ __forceinline__ u32 nibbleWord(const StitchCtx &c, i32 w) {
    if (w < 0 || (u32)w > (c.Lread >> 3)) return 0u;
    return ((const __attribute__((address_space(3))) u32 *)ldsReads)[(c.ldsByte >> 2) + (u32)w];
}
__device__ __forceinline__ u64 nibbles8(const StitchCtx &c, i32 first) {       // packed-read codes first .. first+7 as a run
    const i32 w = first >> 3; const u32 sh = (u32)(first & 7) * 4u;
    const u64 two = (u64)nibbleWord(c, w) | ((u64)nibbleWord(c, w + 1) << 32);
    return spreadNibbles((u32)(two >> sh));
}
__device__ __forceinline__ u64 readRun(const StitchCtx &c, i32 i, bool up) {
    if (c.str == 0) { const u64 v = nibbles8(c, up ? i : i - 7); return up ? v : __builtin_bswap64(v); }
    // reverse strand: R[i] = complement of the packed read's base Lread-1-i
    const i32 j = (i32)c.Lread - 1 - i;
    const u64 v = complementBytes(nibbles8(c, up ? j - 7 : j));
    return up ? __builtin_bswap64(v) : v;
}

// ---- growing an end (extendAlign.cpp:6-93) ---------------------------------------------------------------------------------------------------------
// dR == dG == +1 (3' end) or -1 (5' end) at every call site; L bases at most.
__device__ static bool growOnLane(StitchCtx &c, u32 rStart, u64 gStart, int dR, int dG, u32 L, u32 Lprev, u32 nMMprev, u32 nMMmax, double pMMmax, bool toTheEnd, ExtRes &e) {
    (void)dG;
    DIAG(c.nExtendCalls++);
    e.maxScore = 0; e.extendL = 0; e.nMatch = 0; e.nMM = 0;
    // (extendAlign.cpp:59 runs `i < (int) L`: a length that went "negative" is no extension at all)
    if ((int)L <= 0) return false;
    const bool up = dR > 0;
    GenomeRuns gs; runsInit(gs);
    int score = 0; u32 nMatch = 0, nMM = 0;
    if (toTheEnd) {                        // --alignEndsType Extend*: everything up to the end of the mate counts, padding in the way voids the extension (:18-56)
        u32 done = 0;
        for (u32 at = 0; at < L; at += 8) {
            const u64 g = genomeRun(c, gs, (i64)gStart + (up ? (i64)at : -(i64)at), up), r = readRun(c, (i32)rStart + (up ? (i32)at : -(i32)at), up);
            const u32 span = lowBits(min(8u, L - at));
            const u32 pad = bytesEqual(g, 5) & span, spacer = bytesEqual(r, STARAMD_SPACER_BASE) & span;
            // the genome is looked at before the read at every position: padding at or before the spacer voids, a spacer before any padding ends the scan
            const u32 firstPad = pad ? (u32)__builtin_ctz(pad) : 8u, firstSp = spacer ? (u32)__builtin_ctz(spacer) : 8u;
            if (firstPad <= firstSp && firstPad < 8u) { c.nGstitch += firstPad + 1u; e.extendL = 0; e.maxScore = -999999999; e.nMatch = 0; e.nMM = nMMmax + 1; return true; }
            const u32 use = span & lowBits(firstSp);
            const u32 both = acgtBytes(g) & acgtBytes(r) & use, same = sameBytes(g, r) & both;
            nMatch += (u32)__builtin_popcount(same); nMM += (u32)__builtin_popcount(both & ~same);
            const u32 n = (u32)__builtin_popcount(use);
            done += n; c.nGstitch += n;
            if (firstSp < 8u || n < 8u) break;
        }
        if (done > 0) { e.extendL = done; e.maxScore = (int)nMatch - (int)nMM; e.nMatch = nMatch; e.nMM = nMM; return true; }
        return false;
    }
    const double budgetAll = fmin(pMMmax * (double)(u64)(Lprev + L), (double)nMMmax);
    for (u32 at = 0; at < L; at += 8) {
        const u64 g = genomeRun(c, gs, (i64)gStart + (up ? (i64)at : -(i64)at), up), r = readRun(c, (i32)rStart + (up ? (i32)at : -(i32)at), up);
        const u32 span = lowBits(min(8u, L - at));
        const u32 ends = (bytesEqual(g, 5) | bytesEqual(r, STARAMD_SPACER_BASE)) & span;                 // :61-63
        const u32 use = span & (ends ? lowBits((u32)__builtin_ctz(ends)) : 0xFFu);
        const u32 both = acgtBytes(g) & acgtBytes(r) & use;                                               // :65 an N on either side scores nothing
        const u32 same = sameBytes(g, r) & both;
        u32 miss = both & ~same, seen = 0;               // seen: positions of the run dealt with so far
        bool over = false;
        // Between two mismatches the score only rises, and the mismatch allowance of a prefix only rises with its length: of a stretch of matches only the LAST
        // one can be where a new best prefix is recorded (:69-75).  So the run is taken mismatch by mismatch.
        for (;;) {
            const u32 upto = miss ? lowBits((u32)__builtin_ctz(miss)) : 0xFFu;          // positions before the next mismatch
            const u32 stretch = same & upto & ~seen;
            if (stretch) {
                const u32 k = (u32)__builtin_popcount(stretch), last = 31u - (u32)__builtin_clz(stretch);
                nMatch += k; score += (int)k;
                if (score > e.maxScore && (double)(u32)(nMM + nMMprev) <= fmin(pMMmax * (double)(u64)(Lprev + at + last + 1u), (double)nMMmax)) {
                    e.extendL = at + last + 1u; e.maxScore = score; e.nMatch = nMatch; e.nMM = nMM;
                }
            }
            if (!miss) break;
            if ((double)(u32)(nMM + nMMprev) >= budgetAll) { over = true; c.nGstitch += (u32)__builtin_ctz(miss) + 1u; break; }   // :78 the mismatches before this one exhaust the budget
            nMM++; score--;
            seen = lowBits((u32)__builtin_ctz(miss) + 1u);
            miss &= miss - 1u;
        }
        if (over) break;
        const u32 n = (u32)__builtin_popcount(use);
        c.nGstitch += n;
        if (n < 8u) break;
    }
    return e.extendL > 0;
}

// ---- annotated junction (start, end) -> its index, -1 when there is none (binarySearch2.cpp:3-43: the starts are sorted, equal starts lie together) ----
__device__ static int junctionOnLane(u64 start, u64 end, const u64 *starts_, const u64 *ends_, int n) {
    const __attribute__((address_space(1))) u64 *starts = GLOBAL(u64, starts_), *ends = GLOBAL(u64, ends_);
    if (n == 0 || start > starts[n - 1] || start < starts[0]) return -1;
    int lo = 0, hi = n - 1;
    while (hi > lo + 1) { const int mid = (lo + hi) / 2; if (starts[mid] > start) hi = mid; else lo = mid; }
    int at;
    if (starts[lo] == start) at = lo; else if (starts[hi] == start) at = hi; else return -1;
    for (int k = at; k >= 0 && starts[k] == start; k--) if (ends[k] == end) return k;
    for (int k = at; k < n; k++) { if (starts[k] != start) return -1; if (ends[k] == end) return k; }
    return -2;
}

// the splice-site class of a junction whose donor side reads d1 d2 and whose acceptor side reads a1 a2 (stitchAlignToTranscript.cpp:127-155):
// 1 GT/AG  2 CT/AC  3 GC/AG  4 CT/GC  5 AT/AC  6 GT/AT  0 anything else; codes A 0, C 1, G 2, T 3
__device__ __forceinline__ int spliceClass(u32 d1, u32 d2, u32 a1, u32 a2) {
    switch (d1 | (d2 << 4) | (a1 << 8) | (a2 << 12)) {
        case 0x2032: return 1;      // G T . A G
        case 0x1031: return 2;      // C T . A C
        case 0x2012: return 3;      // G C . A G
        case 0x1231: return 4;      // C T . G C
        case 0x1030: return 5;      // A T . A C
        case 0x3032: return 6;      // G T . A T
        default: return 0;
    }
}

// ---- joining seed B (read rBstart.., genome gBstart.., L bases, mate iFragB, annotated junction sjAB or -1) to the transcript that ends at read rAend / genome gAend
// (stitchAlignToTranscript.cpp:9-415).  h / eA are working copies: the caller commits them (and eN when `added`) only when the returned score is > -1000000.
// ex0R / ex0G = start of the transcript's first exon (the mate-pair branch looks at it).
__device__ static int joinOnLane(StitchCtx &c, u32 rAend, u64 gAend, u32 rBstart, u64 gBstart, u32 L, u32 iFragB, i32 sjAB,
                                 Hdr &h, staramd_exon &eA, staramd_exon &eN, bool &added, u32 ex0R, u64 ex0G) {
    const DevIndex &X = *c.X; const staramd_params &P = X.P;
    DIAG(c.nStitchCalls++);
    added = false;
    if (h.nExons >= STARAMD_MAX_N_EXONS) return -1000010;
    int total = 0;
    if (sjAB != -1 && eA.sjA == sjAB && eA.iFrag == iFragB && rBstart == rAend + 1 && gAend + 1 < gBstart) {
        // the two seeds are the two halves of one inserted junction sequence: the junction is the annotated one (:18-34)
        const u32 sInfo = GLOBAL(u32, X.sjdbInfo)[sjAB], motif = SJ_INFO_MOTIF(sInfo), shL = SJ_INFO_SHL(sInfo), shR = SJ_INFO_SHR(sInfo);
        if (motif == 0 && (L <= shR || eA.L <= shL)) return -1000006;
        eN.L = (u16)L; eN.R = (u16)rBstart; eN.G = gBstart;
        eA.canonSJ = (i8)motif; eA.shiftSJ[0] = (u16)shL; eA.shiftSJ[1] = (u16)shR; eA.sjAnnot = 1; eA.sjStr = (u8)SJ_INFO_STRAND(sInfo);
        added = true; h.nMatch += L;
        total = (int)L + P.sjdbScore;
    } else if (eA.iFrag == iFragB) {
        eA.sjAnnot = 0; eA.sjStr = 0;
        const u64 gBend = gBstart + L - 1; const u32 rBend = rBstart + L - 1;
        if (rBend <= rAend) return -1000001;
        if (gBend <= gAend) return -1000002;
        if (rBstart <= rAend) { gBstart += rAend - rBstart + 1; rBstart = rAend + 1; L = rBend - rBstart + 1; }      // B starts inside A: only what lies behind A is new
        total = (int)(rBend - rBstart + 1);
        const int gapG = (int)(gBstart - gAend - 1), gapR = (int)(rBstart - rAend - 1);
        u32 nMatch = L, nMM = 0, nDel = 0, nIns = 0, insLen = 0; u64 delLen = 0;
        int cut = 0;                       // the junction / insertion sits behind read base rAend + cut
        int kind = 999;                    // canonSJ of the exon: splice class 0..6, -1 deletion, -2 insertion
        // the acceptor side of the genome, indexed like the donor side: base rAend + j of the read lies on gAend + j (donor) or acc + j (acceptor)
        const u64 acc = gBstart - (u64)(i64)gapR - 1;
        GenomeRuns ga, gb; runsInit(ga); runsInit(gb);
        if (gapG == 0 && gapR == 0) {
            // B continues A
        } else if (gapG > 0 && gapR > 0 && gapR == gapG) {
            // same distance in read and genome: the bases in between are compared (:66-80)
            for (int at = 1; at <= gapR; at += 8) {
                const u32 span = lowBits((u32)min(8, gapR - at + 1));
                const u64 g = genomeRun(c, ga, (i64)gAend + at, true), r = readRun(c, (i32)rAend + at, true);
                const u32 both = acgtBytes(g) & acgtBytes(r) & span, same = sameBytes(g, r) & both;
                const u32 k = (u32)__builtin_popcount(same), x = (u32)__builtin_popcount(both & ~same);
                total += (int)k - (int)x; nMatch += k; nMM += x;
                c.nGstitch += (u32)__builtin_popcount(span);
            }
        } else if (gapG > gapR) {
            // ---- the genome has more between the seeds than the read: a deletion or an intron; where does it go? (:82-243)
            nDel = 1; delLen = (u64)(i64)(gapG - gapR);
            if (delLen > P.alignIntronMax && P.alignIntronMax > 0) return -1000003;
            const bool intron = delLen >= P.alignIntronMin;
            // 1. how far to the left of A's end may the junction move: back until the read has disagreed scoreStitchSJshift + 1 times with the acceptor side where it
            //    agrees with the donor side, or A's exon is used up (:88-99); position j = read base rAend + j, j = 0, -1, -2 ..
            int left = 0;
            {
                int allowance = P.scoreStitchSJshift;
                for (int at = 0, going = 1; going; at -= 8) {
                    const u64 gA = genomeRun(c, ga, (i64)gAend + at, false), gB = genomeRun(c, gb, (i64)acc + at, false), r = readRun(c, (i32)rAend + at, false);
                    const u32 lose = ~sameBytes(r, gB) & acgtBytes(gB) & sameBytes(r, gA);          // bit k = position at - k
                    u32 k = 0;
                    for (; k < 8u; k++) {
                        left = at - (int)k;
                        allowance -= (int)((lose >> k) & 1u);
                        if (!(allowance >= 0 && (int)eA.L + left > 1)) { going = 0; break; }
                    }
                    c.nGstitch += 2u * min(k + 1u, 8u);
                }
            }
            // 2. every position from there to the end of B (exclusive) is a candidate; the score of a position = (bases right of A's end, up to it, that agree with the
            //    donor side only) - (those that agree with the acceptor side only) + the penalty of its splice class; the first best wins (:101-160)
            int best = -999999, bestPen = 0, run = 0;
            const int stopAt = (int)rBend - (int)rAend;               // exclusive
            for (int at = left; at < stopAt; at += 8) {
                const u64 gA = genomeRun(c, ga, (i64)gAend + at, true), gB = genomeRun(c, gb, (i64)acc + at, true), r = readRun(c, (i32)rAend + at, true);
                const u32 span = lowBits((u32)min(8, stopAt - at));
                const u32 onA = sameBytes(r, gA), onB = sameBytes(r, gB);
                const u32 gain = onA & ~onB & span, loss = ~onA & onB & span;
                u64 d1 = 0, d2 = 0, a1 = 0;
                if (intron) { d1 = genomeRun(c, ga, (i64)gAend + at + 1, true); d2 = genomeRun(c, ga, (i64)gAend + at + 2, true); a1 = genomeRun(c, gb, (i64)acc + at - 1, true); }
                const u32 n = (u32)__builtin_popcount(span);
                for (u32 k = 0; k < n; k++) {
                    run += (int)((gain >> k) & 1u) - (int)((loss >> k) & 1u);
                    int cls = -1, pen = 0;
                    if (intron) {
                        cls = spliceClass((u32)(d1 >> (8u * k)) & 255u, (u32)(d2 >> (8u * k)) & 255u, (u32)(a1 >> (8u * k)) & 255u, (u32)(gB >> (8u * k)) & 255u);
                        pen = cls == 0 ? P.scoreGapNoncan : cls <= 2 ? 0 : cls <= 4 ? P.scoreGapGCAG : P.scoreGapATAC;
                    }
                    if (run + pen > best) { best = run + pen; cut = at + (int)k; kind = cls; bestPen = pen; }
                }
                c.nGstitch += (intron ? 5u : 2u) * n;
            }
            // 3. the junction can slide over bases that are the same on both sides: how far to the left and to the right (:162-182)
            u32 slideL = 0, slideR = 0;
            for (;;) {
                if (!((u64)((i64)gAend + cut) >= slideL)) break;
                const u64 gA = genomeRun(c, ga, (i64)gAend + cut - (i64)slideL, false), gB = genomeRun(c, gb, (i64)acc + cut - (i64)slideL, false);
                u32 rep = sameBytes(gA, gB) & acgtBytes(gA);
                // position p of the run is reached only while slideL + p <= 255 and gAend + cut >= slideL + p
                u32 k = (u32)__builtin_ctz(~rep | 0x100u);
                const u64 room = (u64)((i64)gAend + cut) - slideL;                 // slideL + p may go up to gAend + cut
                if ((u64)k > room + 1u) k = (u32)(room + 1u);
                if (slideL + k > 256u) k = 256u - slideL;
                slideL += k; c.nGstitch += 2u * min(k + 1u, 8u);
                if (k < 8u) break;
            }
            for (;;) {
                if (!((u64)((i64)gAend + cut) + slideR + 1u < X.nGenome)) break;
                const u64 gA = genomeRun(c, ga, (i64)gAend + cut + 1 + (i64)slideR, true), gB = genomeRun(c, gb, (i64)acc + cut + 1 + (i64)slideR, true);
                u32 rep = sameBytes(gA, gB) & acgtBytes(gA);
                u32 k = (u32)__builtin_ctz(~rep | 0x100u);
                const u64 room = X.nGenome - ((u64)((i64)gAend + cut) + slideR + 1u);    // positions left inside the genome
                if ((u64)k > room) k = (u32)room;
                if (slideR + k > 256u) k = 256u - slideR;
                slideR += k; c.nGstitch += 2u * min(k + 1u, 8u);
                if (k < 8u) break;
            }
            if (kind <= 0) {               // no canonical site: the junction goes as far left as it can (:184-191)
                cut -= (int)slideL;
                if ((int)eA.L + cut < 1) return -1000005;
                slideR += slideL; slideL = 0;
            }
            // 4. what moving the junction does to the score: bases between the old and the new position change sides (:193-212)
            {
                const int from = min(1, cut + 1), to = max(gapR, cut);
                for (int at = from; at <= to; at += 8) {
                    const u32 span = lowBits((u32)min(8, to - at + 1));
                    const u64 gA = genomeRun(c, ga, (i64)gAend + at, true), gB = genomeRun(c, gb, (i64)acc + at, true), r = readRun(c, (i32)rAend + at, true);
                    const u32 donorSide = cut >= at ? lowBits((u32)min(8, cut - at + 1)) : 0u;            // positions <= cut read the donor side
                    const u32 okA = acgtBytes(gA) & acgtBytes(r), okB = acgtBytes(gB) & acgtBytes(r);
                    const u32 both = ((okA & donorSide) | (okB & ~donorSide)) & span;
                    const u32 same = ((sameBytes(gA, r) & donorSide) | (sameBytes(gB, r) & ~donorSide)) & both;
                    const u32 inGap = (at <= gapR ? lowBits((u32)min(8, gapR - at + 1)) : 0u) & ~(at < 1 ? lowBits((u32)min(8, 1 - at)) : 0u);   // positions 1 .. gapR
                    const u32 hit = (u32)__builtin_popcount(same & inGap), missIn = (u32)__builtin_popcount(both & ~same & inGap), missOut = (u32)__builtin_popcount(both & ~same & ~inGap);
                    total += (int)hit - (int)missIn - 2 * (int)missOut;
                    nMatch += hit; nMatch -= missOut; nMM += missIn + missOut;
                    c.nGstitch += (u32)__builtin_popcount(span);
                }
            }
            // 5. is it an annotated junction? (:214-243)
            int known = -1;
            if (X.sjdbN > 0) known = X.sjdbHash ? sjdbHashFind(X.sjdbHash, X.sjdbHashMask, (u64)((i64)gAend + cut + 1), (u64)((i64)acc + cut))
                                                : junctionOnLane((u64)((i64)gAend + cut + 1), (u64)((i64)acc + cut), X.sjdbStart, X.sjdbEnd, (int)X.sjdbN);
            if (known < 0) {
                if (intron) total += P.scoreGap + bestPen;
                else { total += (int)delLen * P.scoreDelBase + P.scoreDelOpen; kind = -1; eA.sjAnnot = 0; }
            } else {
                const u32 jInfo = GLOBAL(u32, X.sjdbInfo)[known], motif = SJ_INFO_MOTIF(jInfo), shL = SJ_INFO_SHL(jInfo);
                kind = (int)motif;
                if (motif == 0) {
                    if (L <= shL || eA.L <= shL) return -1000006;
                    cut += (int)shL;
                    if ((u64)rAend + (i64)cut >= rBend) return -1000006;
                    slideL = shL; slideR = SJ_INFO_SHR(jInfo);
                }
                eA.sjAnnot = 1; eA.sjStr = (u8)SJ_INFO_STRAND(jInfo);
                total += P.sjdbScore;
            }
            eA.shiftSJ[0] = (u16)slideL; eA.shiftSJ[1] = (u16)slideR; eA.canonSJ = (i8)kind;
            if (eA.sjAnnot == 0) eA.sjStr = (kind > 0) ? (u8)(2 - kind % 2) : 0;
        } else if (gapR > gapG) {
            // ---- the read has more between the seeds than the genome: an insertion (:245-300)
            insLen = (u32)(gapR - gapG); nIns = 1;
            if (gapG < 0) total -= -gapG;              // the seeds overlap on the genome
            if (gapG > 0) {
                // the inserted bases may sit anywhere in the gap: behind read base rAend + cut the read skips insLen bases.  Moving the insertion one to the right turns
                // a base of the gap from "compared behind the insertion" into "compared before it".
                int run = 0, best = 0;
                const int tie = P.alignInsertionFlushRight ? 0 : 1;          // flush right: an equal score moves the insertion on (:273)
                for (int at = 1; at <= gapG; at += 8) {
                    const u32 span = lowBits((u32)min(8, gapG - at + 1));
                    const u64 g = genomeRun(c, ga, (i64)gAend + at, true), r0 = readRun(c, (i32)rAend + at, true), r1 = readRun(c, (i32)(rAend + insLen) + at, true);
                    const u32 real = acgtBytes(g) & span, before = sameBytes(r0, g), behind = sameBytes(r1, g);
                    const u32 n = (u32)__builtin_popcount(span);
                    for (u32 k = 0; k < n; k++) {
                        if ((real >> k) & 1u) run += (((before >> k) & 1u) ? 1 : -1) + (((behind >> k) & 1u) ? -1 : 1);
                        if (run >= best + tie) { best = run; cut = at + (int)k; }
                    }
                    c.nGstitch += n;
                }
                for (int at = 1; at <= gapG; at += 8) {
                    const u32 span = lowBits((u32)min(8, gapG - at + 1));
                    const u64 g = genomeRun(c, ga, (i64)gAend + at, true), r0 = readRun(c, (i32)rAend + at, true), r1 = readRun(c, (i32)(rAend + insLen) + at, true);
                    const u32 early = cut >= at ? lowBits((u32)min(8, cut - at + 1)) : 0u;                  // positions <= cut are read before the insertion
                    const u32 both = ((acgtBytes(r0) & early) | (acgtBytes(r1) & ~early)) & acgtBytes(g) & span;
                    const u32 same = ((sameBytes(r0, g) & early) | (sameBytes(r1, g) & ~early)) & both;
                    const u32 k = (u32)__builtin_popcount(same), x = (u32)__builtin_popcount(both & ~same);
                    total += (int)k - (int)x; nMatch += k; nMM += x;
                    c.nGstitch += (u32)__builtin_popcount(span);
                }
            }
            if (P.alignInsertionFlushRight) {         // the insertion moves right over bases that agree with the genome (:281-291)
                const int lim = (int)rBend - (int)rAend - (int)insLen;
                for (; cut < lim; cut++) {
                    const u8 g = (u8)(genomeRun(c, ga, (i64)gAend + cut + 1, true) & 255u), r = (u8)(readRun(c, (i32)rAend + cut + 1, true) & 255u);
                    c.nGstitch++;
                    if (r != g || g == 4) break;
                }
                if (cut == lim) return -1000009;
            }
            total += (int)insLen * P.scoreInsBase + P.scoreInsOpen;
            kind = -2;
        }
        if ((h.nMM + nMM) <= c.mmMaxTotal && (kind < 0 || (kind < 7 && (u64)nMM <= (u64)(i64)P.alignSJstitchMismatchNmax[(kind + 1) / 2]))) {
            h.nMM += nMM; h.nMatch += nMatch;
            if (delLen >= P.alignIntronMin) { h.nGap += nDel; h.lGap += (u32)delLen; } else { h.nDel += nDel; h.lDel += (u32)delLen; }
            if (delLen == 0 && insLen == 0) eA.L = (u16)(eA.L + (rBend - rAend));
            else if (delLen > 0) {
                eA.L = (u16)((int)eA.L + cut);
                eN.L = (u16)((int)(rBend - rAend) - cut); eN.R = (u16)((int)rAend + cut + 1); eN.G = acc + (i64)cut + 1;
                added = true;
            } else {
                h.nIns += nIns; h.lIns += insLen;
                eA.L = (u16)((int)eA.L + cut);
                eN.L = (u16)((int)(rBend - rAend) - cut - (int)insLen); eN.R = (u16)((int)rAend + cut + (int)insLen + 1); eN.G = gAend + 1 + (i64)cut;
                eA.canonSJ = -2; eA.sjAnnot = 0;
                added = true;
            }
        } else return -1000007;
    } else {
        eA.sjAnnot = 0; eA.sjStr = 0;
        // ---- B is on the other mate (:382-411): allowed when it does not start before the transcript does (beyond what --alignEndsProtrude lets through)
        if (!(gBstart + ex0R + (i64)P.alignEndsProtrudeNbasesMax >= ex0G || ex0G < ex0R)) return -1000008;
        if (P.alignMatesGapMax > 0 && gBstart > eA.G + eA.L + P.alignMatesGapMax) return -1000004;
        total = (int)L;
        ExtRes e;
        // the end of mate A grows towards its 3' end, the start of mate B towards its 5' end
        if (growOnLane(c, rAend + 1, gAend + 1, 1, 1, STARAMD_READ_LEN_MAX, h.nMatch, h.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax, P.alignEndsTypeExt[eA.iFrag][1] != 0, e)) {
            h.nMatch += e.nMatch; h.nMM += e.nMM; total += e.maxScore; eA.L = (u16)(eA.L + e.extendL);
        }
        eN.R = (u16)rBstart; eN.G = gBstart; eN.L = (u16)L; h.nMatch += L;
        // (when the transcript has one exon it is the one that just grew: its start has not moved, only its length)
        const u32 room = P.alignEndsTypeExt[iFragB][1] ? STARAMD_READ_LEN_MAX : (u32)(gBstart - ex0G + ex0R);
        if (growOnLane(c, rBstart - 1, gBstart - 1, -1, -1, room, h.nMatch, h.nMM, c.mmMaxTotal, P.outFilterMismatchNoverLmax, P.alignEndsTypeExt[iFragB][1] != 0, e)) {
            h.nMatch += e.nMatch; h.nMM += e.nMM; total += e.maxScore;
            eN.R = (u16)(eN.R - e.extendL); eN.G -= e.extendL; eN.L = (u16)(eN.L + e.extendL);
        }
        eA.canonSJ = -3; eA.sjAnnot = 0;
        added = true;
    }
    // the last exon carries the mate and the junction index of the last seed (:413-414)
    if (added) { eN.iFrag = (u8)iFragB; eN.sjA = sjAB; eN.canonSJ = 0; eN.sjAnnot = 0; eN.sjStr = 0; eN.shiftSJ[0] = eN.shiftSJ[1] = 0; eN.pad0 = 0; eN.pad1 = 0; }
    else { eA.iFrag = (u8)iFragB; eA.sjA = sjAB; }
    return total;
}
