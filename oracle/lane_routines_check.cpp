// lane_routines_check.cpp -- TEST INFRASTRUCTURE: the one-lane routines of the engine (star_amd/csrc/engine/stitch_scalar.h: 8 bases per trip, byte-parallel masks) against the
// base-by-base restatement of oracle/lane_routines_ref.h on random plausible inputs -- every kind of gap between two seeds, both strands, both mates, annotated junctions,
// every failure code.  Host build through the wavefront emulator headers (oracle/wave_emul).  usage: lane_routines_check [trials]
#include "stitch_common.h"
#define ExtRes ExtResOld
#include "lane_routines_ref.h"
#undef ExtRes
#include "stitch_scalar.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
extern thread_local uint32_t ldsReads[];

static std::mt19937_64 rng(12345);
static u32 rnd(u32 n) { return (u32)(rng() % n); }

int main(int argc, char **argv) {
    const long trials = argc > 1 ? atol(argv[1]) : 200000;
    const u64 NG = 300000;
    std::vector<u8> Gbuf(NG + 2 * GPAD, 5);
    u8 *G = Gbuf.data() + GPAD;
    for (u64 i = 0; i < NG; i++) { u32 x = rnd(1000); G[i] = x < 3 ? 4 : (u8)rnd(4); }
    // low-complexity stretches (repeats around junctions)
    for (int k = 0; k < 300; k++) { u64 p = rnd(NG - 400); u32 len = 5 + rnd(60), per = 1 + rnd(3); for (u32 i = per; i < len; i++) G[p + i] = G[p + i % per]; }
    DevIndex X; memset(&X, 0, sizeof(X));
    X.G = G; X.nGenome = NG;
    long bad = 0, nJoin = 0, nOK = 0, nGrow = 0;
    std::vector<u64> sjS, sjE; std::vector<u8> sjM, sjL, sjR, sjStr; std::vector<u32> sjInfo;
    for (long t = 0; t < trials; t++) {
        staramd_params &P = X.P;
        P.scoreStitchSJshift = (int)rnd(3); P.alignIntronMin = 21; P.alignIntronMax = rnd(3) == 0 ? 500 : 0;
        P.scoreGap = 0; P.scoreGapNoncan = -8; P.scoreGapGCAG = -4; P.scoreGapATAC = -8; P.scoreDelOpen = -2; P.scoreDelBase = -2; P.scoreInsOpen = -2; P.scoreInsBase = -2;
        P.alignInsertionFlushRight = rnd(3) == 0; P.sjdbScore = 2; P.alignMatesGapMax = rnd(4) == 0 ? 300 : 0; P.alignEndsProtrudeNbasesMax = rnd(5) == 0 ? 10 : 0;
        P.alignSJstitchMismatchNmax[0] = 0; P.alignSJstitchMismatchNmax[1] = -1; P.alignSJstitchMismatchNmax[2] = 0; P.alignSJstitchMismatchNmax[3] = 0;
        if (rnd(4) == 0) for (int k = 0; k < 4; k++) P.alignSJstitchMismatchNmax[k] = (int)rnd(4) - 1;
        P.outFilterMismatchNoverLmax = rnd(3) == 0 ? 0.05 : 0.3;
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) P.alignEndsTypeExt[a][b] = rnd(8) == 0;
        // ---- a read made of genome piece A, something in between, genome piece B
        const u32 lenA = 15 + rnd(80), lenB = 15 + rnd(80);
        const u64 gA0 = 1000 + rnd((u32)NG - 20000);
        int kindGap = (int)rnd(6);      // 0 continuous, 1 small deletion, 2 intron, 3 insertion, 4 equal gap with junk, 5 overlap on genome
        u32 between = 0; u64 gB0;
        if (kindGap == 0) gB0 = gA0 + lenA;
        else if (kindGap == 1) gB0 = gA0 + lenA + 1 + rnd(20);
        else if (kindGap == 2) { gB0 = gA0 + lenA + 21 + rnd(3000);
            if (rnd(2)) { u32 m = rnd(6); const u8 mot[6][4] = {{2,3,0,2},{1,3,0,1},{2,1,0,2},{1,3,2,1},{0,3,0,1},{2,3,0,3}}; G[gA0 + lenA] = mot[m][0]; G[gA0 + lenA + 1] = mot[m][1]; G[gB0 - 2] = mot[m][2]; G[gB0 - 1] = mot[m][3]; } }
        else if (kindGap == 3) { between = 1 + rnd(12); gB0 = gA0 + lenA + (rnd(2) ? 0 : rnd(10)); }
        else if (kindGap == 4) { between = 1 + rnd(15); gB0 = gA0 + lenA + between; }
        else gB0 = gA0 + lenA - std::min<u32>(lenA - 1, 1 + rnd(5));
        std::vector<u8> r;
        for (u32 i = 0; i < lenA; i++) r.push_back(G[gA0 + i]);
        for (u32 i = 0; i < between; i++) r.push_back(kindGap == 4 && rnd(4) ? G[gA0 + lenA + i] : (u8)rnd(4));
        const u32 rB0 = (u32)r.size();
        for (u32 i = 0; i < lenB; i++) r.push_back(G[gB0 + i]);
        const u32 tail = rnd(20); for (u32 i = 0; i < tail; i++) r.push_back((u8)rnd(4));
        for (auto &b : r) { u32 x = rnd(100); if (x < 3) b = (u8)rnd(4); else if (x == 3) b = 4; }
        const bool otherMate = rnd(5) == 0;
        if (otherMate) r[rB0 - 1] = STARAMD_SPACER_BASE;
        const u32 Lread = (u32)r.size();
        StitchCtx c; memset(&c, 0, sizeof(c)); c.X = &X; c.ldsByte = 64; c.Lread = Lread; c.str = rnd(2); c.readLength[0] = Lread; c.mmMaxTotal = rnd(4) == 0 ? 2 : 10;
        gcInit(c.ca); gcInit(c.cb);
        // pack: R[i] = r[i]; str 1: R[i] = comp(packed[Lread-1-i])
        memset(ldsReads, 0xCD, 4096);
        { u8 *pk = (u8 *)ldsReads + c.ldsByte; memset(pk, 0xFF, (Lread + 16) / 2 + 8);
          for (u32 j = 0; j <= Lread; j++) { u8 code = j < Lread ? (c.str == 0 ? r[j] : compBase(r[Lread - 1 - j])) : (u8)15; u8 &b = pk[j >> 1]; b = (j & 1) ? (u8)((b & 0x0F) | (code << 4)) : (u8)((b & 0xF0) | code); } }
        // ---- junction table: sometimes holds the true junction
        sjS.clear(); sjE.clear(); sjM.clear(); sjL.clear(); sjR.clear(); sjStr.clear();
        X.sjdbN = 0; X.sjdbHash = nullptr;
        if (rnd(3) == 0) {
            std::vector<std::pair<u64, u64>> js;
            for (int k = 0; k < 6; k++) js.push_back({rnd((u32)NG), 0});
            if (kindGap == 2 || kindGap == 1) { js.push_back({gA0 + lenA, gB0 - 1}); if (rnd(2)) js.push_back({gA0 + lenA, gB0 + 5}); if (rnd(2)) js.push_back({gA0 + lenA - rnd(3), gB0 - 1 - rnd(3)}); }
            for (auto &j : js) if (!j.second) j.second = j.first + 30 + rnd(500);
            std::sort(js.begin(), js.end());
            for (auto &j : js) { sjS.push_back(j.first); sjE.push_back(j.second); sjM.push_back((u8)rnd(7)); sjL.push_back((u8)rnd(4)); sjR.push_back((u8)rnd(4)); sjStr.push_back((u8)rnd(3)); }
            X.sjdbN = (u32)js.size(); X.sjdbStart = sjS.data(); X.sjdbEnd = sjE.data(); X.sjdbMotif = sjM.data(); X.sjdbShiftLeft = sjL.data(); X.sjdbShiftRight = sjR.data(); X.sjdbStrand = sjStr.data();
            sjInfo.clear(); for (size_t k = 0; k < js.size(); k++) sjInfo.push_back(SJ_INFO(sjM[k], sjStr[k], sjL[k], sjR[k]));      // the engine's packed form of the same four (dev.h); the restatement reads the arrays
            X.sjdbInfo = sjInfo.data();
        }
        // ---- the two seeds: A ends somewhere in piece A, B starts somewhere around the start of piece B
        const u32 cutA = 5 + rnd(lenA - 5);                       // A covers read [.., cutA-1]
        const u32 rAend = cutA - 1; const u64 gAend = gA0 + cutA - 1;
        int shiftB = (int)rnd(12) - 4;                             // B may start a little early (overlapping A / the gap) or late
        if ((int)rB0 + shiftB < 1) shiftB = 0;
        if (otherMate) { if (shiftB < 0) shiftB = 0; if (cutA + 1 > rB0) continue; }
        u32 rBstart = (u32)((int)rB0 + shiftB); u64 gBstart = (u64)((i64)gB0 + shiftB);
        u32 Lb = 5 + rnd(lenB - 4); if (rBstart + Lb > Lread) Lb = Lread - rBstart; if (Lb == 0) continue;
        Hdr h; memset(&h, 0, sizeof(h)); h.nExons = 1 + rnd(3); h.nMM = rnd(3); h.nMatch = cutA; h.rStart = 0; h.gStart = gA0; h.tR2 = rAend; h.tG2 = gAend;
        if (rnd(50) == 0) h.nExons = STARAMD_MAX_N_EXONS;
        staramd_exon eA; memset(&eA, 0, sizeof(eA)); eA.G = gA0 + (cutA > 20 && rnd(2) ? cutA - 10 : 0); eA.R = (u16)(eA.G - gA0); eA.L = (u16)(cutA - eA.R); eA.iFrag = 0; eA.sjA = -1;
        u32 iFragB = otherMate ? 1 : 0; i32 sjAB = -1;
        if (rnd(25) == 0 && X.sjdbN) { sjAB = (i32)rnd(X.sjdbN); if (rnd(2)) eA.sjA = sjAB; if (rnd(2) && !otherMate) { rBstart = rAend + 1; if (rBstart + Lb > Lread) Lb = Lread - rBstart; if (!Lb) continue; } }
        if (rnd(60) == 0 && !otherMate && rBstart > 6) { rBstart = rAend > 8 ? rAend - 3 - rnd(4) : rBstart; Lb = 1 + rnd(3); gBstart = gAend - (rAend - rBstart); }     // B inside A
        const u32 ex0R = 0; const u64 ex0G = (otherMate && rnd(10) == 0) ? gBstart + 20 + rnd(100) : gA0;
        // ---- old against new
        Hdr h1 = h, h2 = h; staramd_exon a1 = eA, a2 = eA, n1, n2; memset(&n1, 0x5A, sizeof(n1)); memset(&n2, 0x5A, sizeof(n2)); bool ad1 = false, ad2 = false;
        StitchCtx c1 = c, c2 = c;
        const int s1 = stitchAlignToTranscript(c1, rAend, gAend, rBstart, gBstart, Lb, iFragB, sjAB, h1, a1, n1, ad1, ex0R, ex0G);
        const int s2 = joinOnLane(c2, rAend, gAend, rBstart, gBstart, Lb, iFragB, sjAB, h2, a2, n2, ad2, ex0R, ex0G);
        nJoin++;
        { static long hist[64]; static long canon[16], annot = 0, shifted = 0; int k = s1 > -1000000 ? 0 : -(s1 + 1000000); hist[k < 63 ? k : 63]++; if (s1 > -1000000) { canon[(a1.canonSJ + 4) & 15]++; annot += a1.sjAnnot; shifted += (a1.shiftSJ[0] || a1.shiftSJ[1]); }
          if (t == trials - 1) { printf("return codes: ok %ld", hist[0]); for (int i = 1; i < 12; i++) printf("  -10000%02d: %ld", i, hist[i]); printf("\ncanonSJ -3..6:"); for (int i = 1; i <= 10; i++) printf(" %ld", canon[i]); printf("  annotated %ld  with repeat shifts %ld\n", annot, shifted); } }
        bool diff = s1 != s2;
        if (!diff && s1 > -1000000) {
            nOK++;
            diff = memcmp(&h1, &h2, sizeof(h)) != 0 || ad1 != ad2 || memcmp(&a1, &a2, sizeof(a1)) != 0 || (ad1 && memcmp(&n1, &n2, sizeof(n1)) != 0);
        }
        if (diff && bad++ < 12) {
            printf("JOIN DIFF trial %ld kind %d str %u other %d: score %d / %d added %d/%d  rAend %u rBstart %u Lb %u gap g %lld  eA.L %u/%u canon %d/%d shift %u,%u / %u,%u annot %u/%u sjStr %u/%u  nMM %u/%u nMatch %u/%u nGap %u/%u lGap %u/%u nDel %u/%u nIns %u/%u  eN R %u/%u L %u/%u G %llu/%llu\n",
                   t, kindGap, c.str, (int)otherMate, s1, s2, ad1, ad2, rAend, rBstart, Lb, (long long)((i64)gBstart - (i64)gAend - 1), a1.L, a2.L, a1.canonSJ, a2.canonSJ, a1.shiftSJ[0], a1.shiftSJ[1], a2.shiftSJ[0], a2.shiftSJ[1],
                   a1.sjAnnot, a2.sjAnnot, a1.sjStr, a2.sjStr, h1.nMM, h2.nMM, h1.nMatch, h2.nMatch, h1.nGap, h2.nGap, h1.lGap, h2.lGap, h1.nDel, h2.nDel, h1.nIns, h2.nIns, n1.R, n2.R, n1.L, n2.L, (unsigned long long)n1.G, (unsigned long long)n2.G);
        }
        // ---- growing an end, both directions
        for (int rep = 0; rep < 2; rep++) {
            const int dir = rep ? 1 : -1;
            u32 rs = rnd(Lread); u64 gsx = gA0 + rs + (rnd(4) == 0 ? rnd(5) : 0);
            if (rnd(40) == 0) gsx = dir < 0 ? rnd(12) : NG - 1 - rnd(12);                   // near the ends of the genome: padding
            u32 Lx = dir > 0 ? Lread - rs : rs + 1; if (rnd(6) == 0) Lx = rnd(Lx + 1); if (rnd(60) == 0) Lx = (u32)-3;
            if (dir > 0 && rnd(3) == 0) Lx = STARAMD_READ_LEN_MAX;                            // the mate-gap call: runs to the spacer
            const u32 Lprev = rnd(100), nMMprev = rnd(4), nMMmax = rnd(3) == 0 ? 2 : 10; const double p = rnd(2) ? 0.3 : 0.05; const bool toEnd = rnd(6) == 0;
            if (dir > 0 && Lx == STARAMD_READ_LEN_MAX) { bool sp = false; for (u32 i = rs; i < Lread; i++) if (r[i] == STARAMD_SPACER_BASE) sp = true; if (!sp) Lx = Lread - rs; }
            ExtResOld e1; ExtRes e2; StitchCtx d1 = c, d2 = c;
            const bool b1 = extendAlign(d1, rs, gsx, dir, dir, Lx, Lprev, nMMprev, nMMmax, p, toEnd, e1);
            const bool b2 = growOnLane(d2, rs, gsx, dir, dir, Lx, Lprev, nMMprev, nMMmax, p, toEnd, e2);
            nGrow++;
            if (b1 != b2 || e1.maxScore != e2.maxScore || e1.extendL != e2.extendL || e1.nMatch != e2.nMatch || e1.nMM != e2.nMM) {
                if (bad++ < 12) printf("GROW DIFF trial %ld dir %d str %u toEnd %d rs %u L %d Lread %u: %d/%d score %d/%d len %u/%u nMatch %u/%u nMM %u/%u\n", t, dir, c.str, (int)toEnd, rs, (int)Lx, Lread, b1, b2, e1.maxScore, e2.maxScore, e1.extendL, e2.extendL, e1.nMatch, e2.nMatch, e1.nMM, e2.nMM);
            }
        }
    }
    printf("%ld joins (%ld with a valid score), %ld extensions: %ld differences\n", nJoin, nOK, nGrow, bad);
    return bad ? 1 : 0;
}
