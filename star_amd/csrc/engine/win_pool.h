// win_pool.h -- k_windows with the seed lists of a read in LDS (-DWIN_POOL_ROWS=64 on k_window.hip AND engine.hip).
//
// STATUS: a candidate, NOT part of the product build (WIN_POOL_ROWS defaults to 0 and every line it guards compiles out; `make` builds the
// same code objects with or without this header's macros).  It was written after the last GPU minute of round 4: the wavefront emulator shows
// that its results are the product kernel's (tests/test_wave_emul.py, `win_pool*` cases), nothing shows what it costs on the MI355X.  It lands or
// is deleted by the first A/B of the next session (tools/build_variants.sh pool:k_window+engine:"-DWIN_POOL_ROWS=64").
//
// Why: k_windows keeps the seed list of every window (the reference's WA[iW][], ReadAlign_assignAlignToWindow.cpp:6-130) in global memory, one
// block of WA_MAX rows per window.  Every assignment of a seed to a window is then a load of the list (a round trip to L2), a sorted insert by
// shifted stores and a fence (a second round trip) -- 26 assignments per read pair, ~50 dependent round trips of the ~120 a read waits for; the
// emission loop reads every list once more, one window after the other (20 more).  A pair has 26 rows in all (nWA per pair, 3.1 Gb index).
// With WIN_POOL_ROWS = R the rows of a read live in ONE packed pool of R rows in LDS, the lists of its windows one behind the other in the order
// they were started: lane r holds pool row r (R <= 64), so opening a gap for a sorted insert is "every row from the gap on moves up by one" -- one
// LDS read and one LDS write per lane, no memory round trip -- and the list words of the windows behind the gap move their base by one.
// A list that reaches WIN_POOL_LIST rows, or needs the reference's eviction rule (seedPerWindowNmax rows), or the longest list when the pool is
// full, moves to a block of the arena in global memory and goes on there exactly as before.  Nothing overflows that did not overflow before.
//
// LDS per wavefront: rows of 6 words instead of 8 (block / lrec / count share one word) = 3 KB for 128 rows, 2 KB owner map, 64 x 24 B pool = 1.5 KB:
// 6.5 KB, 26 KB per block of four wavefronts, six blocks per CU as before.
#pragma once
#ifndef WIN_POOL_ROWS
#define WIN_POOL_ROWS 0
#endif
#ifndef WIN_POOL_LIST
#define WIN_POOL_LIST 24            // rows a list may have and still live in the pool
#endif
// the list word of a window: count | lrec << 7 | where << 18 | at << 20
//   where 0: no list yet   1: rows [at, at + count) of the pool   2: block `at` of the arena in global memory
#define LST_NWA(x) ((x) & 127u)
#define LST_LREC(x) (((x) >> 7) & 2047u)
#define LST_WHERE(x) (((x) >> 18) & 3u)
#define LST_AT(x) ((x) >> 20)
#define LST_WORD(where, at, lrec, nwa) (((at) << 20) | ((where) << 18) | ((lrec) << 7) | (nwa))
#define LST_MAX_AT 4095u
// LDS words of one wavefront of the first k_windows launch
__host__ __device__ inline unsigned winLdsWords(unsigned capW, unsigned hashBits, bool pool) { return pool ? capW * 6u + hashBits / 32u + WIN_POOL_ROWS * 6u : capW * 8u + hashBits / 32u; }
