#!/usr/bin/env python3
"""Load round trips per wavefront of the seed search under three shapes of control flow, from a trace of the emulated engine.

  STARAMD_SEED_FLAT=1 STARAMD_SEED_TRACE=trace.txt  <an emulated run: tests/emul_run.py ...>      (k_seed_flat.hip logs every search of every read)
  python tools/seed_divergence.py trace.txt

A read is a lane; 64 consecutive reads are a wavefront (the ticket order of the kernel).  Unit of cost: one dependent load round trip of a lane
(a SAindex look-up, a suffix-array probe, one 8-base compare step).  A wavefront pays for a loop as many trips as its slowest lane makes, level by level:

  call tree   (k_seed_search)       the reference's nest: per search  max(SAindex look-ups) + for every compare of L1, L2, the main bisection, findMultRange x 2,
                                    aligned by position in their loop:  1 + max(words);  searches aligned by (piece, direction, start, step, sparse offset)
  flat search (k_seed_search_flat)  per search  max(SAindex look-ups) + max over lanes of the SUM of (1 + words) over the lane's compares;  same alignment of searches
  flat read   (not written)         max over lanes of the sum of everything the lane loads
  ideal                             mean over lanes (what 64 lanes that never wait would need)
"""
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    reads = defaultdict(list)          # ir -> list of searches: (key, nSAi, [(phase, words)])
    cur = None
    for line in open(path):
        t = line.split()
        if t[0] == "S":
            ir = int(t[1]); key = tuple(int(x) for x in t[2:7])
            cur = [key, 0, []]
            reads[ir].append(cur)
        elif t[0] == "a":
            cur[1] += 1
        else:
            cur[2].append((int(t[1]), int(t[2])))
    irs = sorted(reads)
    n_waves = 0; tot = defaultdict(float)
    for w0 in range(0, len(irs), 64):
        lanes = irs[w0:w0 + 64]
        if len(lanes) < 64:
            break
        n_waves += 1
        groups = defaultdict(dict)     # search key -> lane -> search
        lane_sum = []
        for l in lanes:
            s = 0
            seen = defaultdict(int)
            for key, nsai, cmps in reads[l]:
                k = key + (seen[key],); seen[key] += 1          # (the same key twice: the two passes of a sparse suffix array)
                groups[k][l] = (nsai, cmps)
                s += nsai + sum(1 + wds for _, wds in cmps)
            lane_sum.append(s)
        tree = flat = 0
        for k, by_lane in groups.items():
            sai = max(v[0] for v in by_lane.values())
            # call tree: compares aligned by (phase, ordinal inside the phase's loop)
            slots = defaultdict(int)
            for nsai, cmps in by_lane.values():
                ordn = defaultdict(int)
                for ph, wds in cmps:
                    key2 = (ph, ordn[ph]); ordn[ph] += 1
                    slots[key2] = max(slots[key2], wds)
            tree += sai + sum(1 + v for v in slots.values())
            flat += sai + max(sum(1 + wds for _, wds in v[1]) for v in by_lane.values())
        tot["tree"] += tree; tot["flat"] += flat; tot["read"] += max(lane_sum); tot["ideal"] += sum(lane_sum) / 64.0
    print("%d wavefronts of 64 reads; load round trips per wavefront:" % n_waves)
    for k, what in (("tree", "call tree (k_seed_search)"), ("flat", "flat search (k_seed_search_flat)"), ("read", "flat over the whole read"), ("ideal", "mean lane")):
        print("  %-34s %9.0f   x%.2f of the mean lane" % (what, tot[k] / n_waves, tot[k] / tot["ideal"]))


if __name__ == "__main__":
    main()
