// sjdb_core.h -- splice-junction insertion into a suffix-array index, as data-parallel passes over arrays in HBM
// (SURVEY.md section 8(f) row 2).
//
// What it replaces: sjdbBuildIndex (source/sjdbBuildIndex.cpp:15-333), the stage of sjdbInsertJunctions that costs time:
//   search   :53-86    insertion point of every suffix of every new junction sequence in the OLD suffix array
//                      (suffixArraySearch1 / compareSeqToGenome1 / compareRefEnds, source/SuffixArrayFuns.cpp:209-351)
//   sort     :88-101   new suffixes ordered by (insertion point, suffix text up to its spacer, offset)
//                      (funCompareUintAndSuffixes, source/funCompareUintAndSuffixes.cpp:6-40)
//   merge    :141-207  old and new entries interleaved into the new packed array, old entries re-based
//   SAindex  :209-284  patched by the reference; here REBUILT from the merged array with the generation-time builder
//                      (index_core.h buildSAindex) -- the result is the same table (tests/test_sjdb_device.py)
// sjdbPrepare (which junctions, their flanking sequences Gsj, motifs, shifts) stays on the host: it is small.
//
// Same backend interface as index_core.h: HipBackend is the product, oracle/index_emul.cpp the plain-loop twin for CPU tests.
#pragma once
#include "index_core.h"

namespace staridx {

struct SjdbParams {
    u64 nGenomeOld;        // bytes of the old genome text (chromosomes + old junction block)
    u64 nGenomeReal;       // chrStart[nChrReal]: where the junction block starts
    u64 nSAold;
    u32 GstrandBit;
    u32 sjdbN, sjdbLength; // new junction table (old junctions included); sjdbLength = 2*overhang+1
    u32 oldSjdbN;
    u64 sjNew;             // junctions that are not in the old index
    u32 saIndexNbases;
};

enum { SJ_GPAD = 1024 };   // spacer bytes the caller guarantees either side of the old genome text (a comparison runs at most sjdbLength past an end)

// character k of old suffix-array entry v, in read direction (the 2N text of index_core.h, addressed through G)
IDX_HD u8 oldSuffixChar(const u8 *G, u64 nGenomeOld, u32 GstrandBit, u64 v, u64 k) {
    const u64 s = v & ~(1ull << GstrandBit);
    if ((v >> GstrandBit) == 0) return G[s + k];
    return compCode(G[(i64)(nGenomeOld - 1 - s) - (i64)k]);
}

// suffixArraySearch1 with gInsert = -1: first old index whose suffix is greater than the query q (query = junction text up to and
// including its spacer).  A query that ties with an old suffix up to the spacer goes AFTER it on the + strand, BEFORE it on the - strand
// (compareRefEnds, SuffixArrayFuns.cpp:209-219).  Returns nSAold when the query is greater than every old suffix (the reference's -2).
template <class SAget> IDX_HD u64 sjdbSearchOne(const u8 *G, u64 nGenomeOld, u32 GstrandBit, u64 nSAold, SAget sa, const u8 *q) {
    auto cmp = [&](u64 iSA, u64 L, int &res) -> u64 {          // compareSeqToGenome1 from offset L; returns the new common length
        const u64 v = sa(iSA);
        const bool fwd = (v >> GstrandBit) == 0;
        for (u64 ii = L;; ii++) {
            u8 a = q[ii], b = oldSuffixChar(G, nGenomeOld, GstrandBit, v, ii);
            if (a != b) { res = a > b ? 1 : -1; return ii; }
            if (a == 5) { res = fwd ? 1 : -1; return ii; }
        }
    };
    int r = 0;
    u64 i1 = 0, i2 = nSAold - 1;
    u64 L1 = cmp(i1, 0, r);
    if (r < 0) return 0;
    u64 L2 = cmp(i2, 0, r);
    if (r > 0) return nSAold;
    u64 L = L1 < L2 ? L1 : L2;
    while (i1 + 1 < i2) {
        u64 i3 = i1 / 2 + i2 / 2 + (i1 % 2 + i2 % 2) / 2;
        u64 L3 = cmp(i3, L, r);
        if (r > 0) { i1 = i3; L1 = L3; } else { i2 = i3; L2 = L3; }
        L = L1 < L2 ? L1 : L2;
    }
    return i2;
}

struct SjdbDeviceResult {
    u64 nInd, nSAnew, nGenomeNew;
    u64 *dSApacked; u64 saWords;      // new packed suffix array (device), caller frees with be.free
    u8 *dGnew;                        // new genome text with padOut spacer bytes either side (device): dGnew + padOut = base 0
    u64 *dSAiPacked; u64 saiWords;    // new packed SAindex (device)
    int badFirstSuffix;
};

// dGold: old genome text, base 0 at dGold (SJ_GPAD bytes of 5 readable either side).  dSAold: packed old suffix array (u64 words).
// hGsj: junction sequences of the NEW table, forward half only: sjdbN blocks of sjdbLength codes, the last code of a block = spacer.
// hIsOld[sjdbN]: 1 = the junction is already in the old index (no new suffixes).  hOldSJind[oldSjdbN]: new number of every old junction.
template <class BE> int sjdbInsertDevice(BE &be, const SjdbParams &P, const u8 *dGold, const u64 *dSAold, const u8 *hGsj, const u8 *hIsOld,
                                         const u32 *hOldSJind, const u64 *saiStart, SjdbDeviceResult &R, u64 padOut = SJ_GPAD) {
    const u64 nGsj = (u64)P.sjdbN * P.sjdbLength, nQ = 2 * nGsj + 1;
    const u32 Lsj = P.sjdbLength, GstrandBit = P.GstrandBit, saBits = P.GstrandBit + 1;
    const u64 nGenomeOld = P.nGenomeOld, nSAold = P.nSAold, nGenomeReal = P.nGenomeReal;
    be.stage(nullptr);
    // ---- query text: forward blocks, their reverse complement, one closing spacer (sjdbBuildIndex.cpp:30-39)
    u8 *dQ = be.template alloc<u8>(nQ + 64);
    be.copyToDevice(dQ, hGsj, nGsj);
    be.forEach(nGsj + 65, [=] IDX_L (u64 i) {
        if (i < nGsj) { u8 c = dQ[i]; dQ[2 * nGsj - 1 - i] = compCode(c); }
        else dQ[2 * nGsj + (i - nGsj)] = 5;
    });
    u8 *dIsOld = be.template alloc<u8>(P.sjdbN + 1);
    be.copyToDevice(dIsOld, hIsOld, P.sjdbN);
    // ---- candidates: every offset that starts with ACGT inside a junction that is new, in offset order (:62-83, :90-97)
    const u32 sjdbN = P.sjdbN;
    auto isCand = [=] IDX_L (u64 off) {
        u64 isj = off / Lsj; u64 isj1 = isj < sjdbN ? isj : 2 * (u64)sjdbN - 1 - isj;
        return !dIsOld[isj1] && dQ[off] < 4;
    };
    const u64 nInd = compactIf(be, 2 * nGsj, isCand, [=] IDX_L (u64, u64) {});
    R.nInd = nInd; R.nSAnew = nSAold + nInd; R.nGenomeNew = nGenomeReal + nGsj;
    u64 *off = be.template alloc<u64>(nInd + 1), *pos = be.template alloc<u64>(nInd + 1);
    compactIf(be, 2 * nGsj, isCand, [=] IDX_L (u64 o, u64 j) { off[j] = o; });
    be.stage("sjdb: candidates");
    // ---- search
    {
        const u64 saMask = saBits >= 64 ? ~0ull : ((1ull << saBits) - 1);
        be.forEach(nInd, [=] IDX_L (u64 j) {
            auto sa = [=](u64 i) { return packedGetW(dSAold, i, saBits) & saMask; };
            pos[j] = sjdbSearchOne(dGold, nGenomeOld, GstrandBit, nSAold, sa, dQ + off[j]);
        });
    }
    be.stage("sjdb: search");
    // ---- sort by (insertion point, text up to the spacer, offset): LSD radix passes over 21-code chunks of the text (stable, so the
    //      offset order of the input breaks the remaining ties), then one stable pass on the insertion point
    if (nInd > 1) {
        // end[j] = offset of the first spacer at or after off[j] (a suffix is compared up to and including it)
        u64 *endp = be.template alloc<u64>(nInd);
        be.forEach(nInd, [=] IDX_L (u64 j) { u64 e = off[j]; while (dQ[e] != 5) e++; endp[j] = e; });
        u64 *key = be.template alloc<u64>(nInd), *keyAlt = be.template alloc<u64>(nInd), *perm = be.template alloc<u64>(nInd), *permAlt = be.template alloc<u64>(nInd);
        be.forEach(nInd, [=] IDX_L (u64 j) { perm[j] = j; });
        const u32 nChunks = (Lsj + KEY_CODES - 1) / KEY_CODES;
        for (u32 c = nChunks; c-- > 0;) {
            { u64 *k = key; const u64 *pm = perm;
              be.forEach(nInd, [=] IDX_L (u64 j) {
                  u64 e = pm[j]; u64 o = off[e] + (u64)c * KEY_CODES, last = endp[e]; u64 v = 0;
                  for (u32 t = 0; t < KEY_CODES; t++) { u64 q = o + t; v = (v << 3) | (q <= last ? (u64)dQ[q] : 0ull); }
                  k[j] = v;
              }); }
            be.sortPairs(key, keyAlt, perm, permAlt, nInd, 0, 63);
        }
        { u64 *k = key; const u64 *pm = perm; be.forEach(nInd, [=] IDX_L (u64 j) { k[j] = pos[pm[j]]; }); }
        be.sortPairs(key, keyAlt, perm, permAlt, nInd, 0, (int)bitsFor(nSAold + 1));
        // gather into (pos, off) order
        { u64 *k = keyAlt; const u64 *pm = perm; be.forEach(nInd, [=] IDX_L (u64 j) { k[j] = off[pm[j]]; }); }
        { const u64 *k = key, *k2 = keyAlt; be.forEach(nInd, [=] IDX_L (u64 j) { pos[j] = k[j]; off[j] = k2[j]; }); }
        be.free(key); be.free(keyAlt); be.free(perm); be.free(permAlt); be.free(endp);
    }
    be.stage("sjdb: sort");
    // ---- merge (:141-207): new entry j lands at output index pos[j] + j; old entries fill the rest in order
    u32 *dOldSJind = be.template alloc<u32>((u64)P.oldSjdbN + 1);
    if (P.oldSjdbN) be.copyToDevice(dOldSJind, hOldSJind, P.oldSjdbN);
    const u64 nSAnew = R.nSAnew;
    R.saWords = packedWords(nSAnew + 1, saBits);
    R.dSApacked = be.template alloc<u64>(R.saWords);
    {
        const u64 N2bit = 1ull << GstrandBit, strandMask = ~N2bit, nGenomeNew = R.nGenomeNew, nGsjNew = P.sjNew * Lsj, saMask = saBits >= 64 ? ~0ull : ((1ull << saBits) - 1);
        u64 *out = R.dSApacked; const u64 *ps = pos, *of = off;
        const u64 nGroups = (nSAnew + 1 + 63) / 64;              // one more entry: the 0 the reference writes behind the array (sjdbInsertJunctions.cpp:66-68)
        // Two passes.  (1) per group of 64 output entries: how many NEW entries lie before it (a bisection in the sorted insert positions: the only search of the
        // merge).  (2) ONE THREAD PER OUTPUT WORD: the two or three entries whose bits fall into the word are produced (a new suffix, or the next old entry re-based)
        // and assembled -- neighbouring threads read neighbouring input words and write neighbouring output words, so the 52 GB the merge of a human index moves
        // stream through whole cache lines.  (The first form gave a thread 64 entries = 33 words of its own: every load and store of a wavefront touched 64
        // different lines, 0.39 TB/s.)
        u64 *jnG = be.template alloc<u64>(nGroups + 1);
        be.forEach(nGroups, [=] IDX_L (u64 g) {
            const u64 o0 = g * 64;
            u64 lo = 0, hi = nInd;                               // first j with pos[j] + j >= o0
            while (lo < hi) { u64 mid = lo + (hi - lo) / 2; if (ps[mid] + mid < o0) lo = mid + 1; else hi = mid; }
            jnG[g] = lo;
        });
        const u64 nWords = nGroups * saBits;                     // the words of whole groups (packedWords() has two more, left zero below)
        be.forEach(nWords, [=] IDX_L (u64 w) {
            const u64 bit0 = w * 64;
            u64 e = bit0 / saBits;                                // first entry with bits in this word
            u64 jn = jnG[e >> 6];
            { const u64 o0 = e & ~63ull; (void)o0; while (jn < nInd && ps[jn] + jn < e) jn++; }      // new entries before entry e (0.5 % of the entries are new: a step now and then)
            u64 word = 0;
            for (; e * saBits < bit0 + 64; e++) {
                u64 v = 0;
                if (e < nSAnew) {
                    if (jn < nInd && ps[jn] + jn == e) {
                        const u64 f = of[jn++];
                        v = f < nGsj ? f + nGenomeReal : ((f - nGsj) | N2bit);
                    } else {
                        u64 ind1 = packedGetW(dSAold, e - jn, saBits) & saMask;
                        if (ind1 & N2bit) {
                            u64 ind1s = nGenomeOld - (ind1 & strandMask);
                            if (ind1s >= nGenomeReal) {                   // an old junction suffix: its junction may have a new number
                                u64 sj1 = (ind1s - nGenomeReal) / Lsj;
                                ind1s += ((u64)dOldSJind[sj1] - sj1) * Lsj;
                                ind1 = (nGenomeNew - ind1s) | N2bit;
                            } else ind1 += nGsjNew;
                        } else if (ind1 >= nGenomeReal) {
                            u64 sj1 = (ind1 - nGenomeReal) / Lsj;
                            ind1 += ((u64)dOldSJind[sj1] - sj1) * Lsj;
                        }
                        v = ind1;
                    }
                }
                const u64 b = e * saBits;                         // first bit of the entry
                word |= b >= bit0 ? (v << (b - bit0)) : (v >> (bit0 - b));
            }
            out[w] = word;
        });
        be.forEach(packedWords(nSAnew + 1, saBits) - nWords, [=] IDX_L (u64 k) { out[nWords + k] = 0; });
        be.free(jnG);
    }
    be.free(pos); be.free(off); be.free(dOldSJind); be.free(dIsOld);
    be.stage("sjdb: merge");
    // ---- new genome text: chromosomes + forward junction block, spacers either side
    R.dGnew = be.template alloc<u8>(R.nGenomeNew + 2 * padOut);
    {
        u8 *gn = R.dGnew; const u64 nGn = R.nGenomeNew;
        be.forEach(nGn + 2 * padOut, [=] IDX_L (u64 i) {
            u8 c = 5;
            if (i >= padOut && i < padOut + nGn) { u64 p = i - padOut; c = p < nGenomeReal ? dGold[p] : dQ[p - nGenomeReal]; }
            gn[i] = c;
        });
    }
    be.free(dQ);
    be.stage("sjdb: new genome text");
    // ---- SAindex of the merged array
    R.saiWords = packedWords(saiStart[P.saIndexNbases], GstrandBit + 3);
    R.dSAiPacked = be.template alloc<u64>(R.saiWords);
    {
        const u64 N = R.nGenomeNew;
        u8 *dTraw = be.template alloc<u8>(2 * N + 2 * TPAD);
        u8 *T = buildText(be, R.dGnew + padOut, N, dTraw);
        u64 *dSApos = be.template alloc<u64>(nSAnew);
        { const u64 *sp = R.dSApacked; const u64 N2bit = 1ull << GstrandBit;
          be.forEach(nSAnew, [=] IDX_L (u64 i) { u64 v = packedGetW(sp, i, saBits); dSApos[i] = (v & N2bit) ? N + (v & ~N2bit) : v; }); }
        const u64 nSAi = saiStart[P.saIndexNbases];
        u64 *dSAiU = be.template alloc<u64>(nSAi);
        be.stage("sjdb: SAindex text + unpack");
        R.badFirstSuffix = buildSAindex(be, T, dSApos, nSAnew, P.saIndexNbases, GstrandBit, saiStart, dSAiU);
        const u64 *su = dSAiU;
        packArray(be, nSAi, GstrandBit + 3, R.dSAiPacked, [=] IDX_L (u64 i) { return su[i]; });
        be.free(dSAiU); be.free(dSApos); be.free(dTraw);
    }
    be.stage("sjdb: SAindex build + pack");
    return 0;
}

// host buffers in, host buffers out (include/star_amd_index.h staramd_sjdb_insert); returns 0, -2 capacity, -3 bad first suffix
struct SjdbHostArgs {
    const u8 *G; const u8 *SA; u64 nSAbyteOld; const u8 *Gsj; const u8 *isOld; const u32 *oldSJind;
    u8 *SAout; u64 saCap; u8 *SAiOut; u64 saiCap;
};
template <class BE> int sjdbInsertHost(BE &be, const SjdbParams &P, const SjdbHostArgs &A, u64 &nInd, u64 &nSAbyteNew, u64 &nSAibyte) {
    u64 saiStart[17]; saiStart[0] = 0;
    for (u32 i = 1; i <= P.saIndexNbases; i++) saiStart[i] = saiStart[i - 1] + (1ull << (2 * i));
    u8 *dGraw = be.template alloc<u8>(P.nGenomeOld + 2 * SJ_GPAD);
    { u8 *g = dGraw; const u64 n = P.nGenomeOld; be.forEach(2 * SJ_GPAD, [=] IDX_L (u64 i) { g[i < SJ_GPAD ? i : n + i] = 5; }); }
    be.copyToDevice(dGraw + SJ_GPAD, A.G, P.nGenomeOld);
    const u64 wOld = (A.nSAbyteOld + 7) / 8 + 2;
    u64 *dSAold = be.template alloc<u64>(wOld);
    { u64 *w = dSAold; be.forEach(3, [=] IDX_L (u64 i) { w[wOld - 1 - i] = 0; }); }
    be.copyToDevice((u8 *)dSAold, A.SA, A.nSAbyteOld);
    SjdbDeviceResult R;
    sjdbInsertDevice(be, P, dGraw + SJ_GPAD, dSAold, A.Gsj, A.isOld, A.oldSJind, saiStart, R);
    be.free(dSAold); be.free(dGraw);
    nInd = R.nInd;
    nSAbyteNew = packedBytes(R.nSAnew, P.GstrandBit + 1); nSAibyte = packedBytes(saiStart[P.saIndexNbases], P.GstrandBit + 3);
    int rc = R.badFirstSuffix ? -3 : 0;
    if (A.saCap < nSAbyteNew || A.saiCap < nSAibyte) rc = -2;
    if (!rc) { be.copyToHost(A.SAout, (const u8 *)R.dSApacked, nSAbyteNew); be.copyToHost(A.SAiOut, (const u8 *)R.dSAiPacked, nSAibyte); }
    be.free(R.dSApacked); be.free(R.dGnew); be.free(R.dSAiPacked);
    return rc;
}

} // namespace staridx
