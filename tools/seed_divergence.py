#!/usr/bin/env python3
"""Load round trips per wavefront of the seed search under three shapes of control flow, from a trace of the emulated engine.

  STARAMD_SEED_FLAT=1 STARAMD_SEED_TRACE=trace.txt  <an emulated run: tests/emul_run.py ...>      (k_seed_flat.hip logs every search of every read)
  python tools/seed_divergence.py trace.txt
  python tools/seed_divergence.py trace.txt --trips      (trace of STARAMD_SEED_FLAT=4: how often a wavefront executes each block of the whole-read state machine)

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


def trips(path):
    """whole-read state machine (k_seed_search_read*): 'M <read> <letters>' lines = the blocks of the loop a lane went through for that read, trip by trip.  A lane
    is taken to map one read (a launch has about as many lanes as a batch has reads), 64 consecutive reads are a wavefront.  A wavefront executes a block in a
    trip if ANY of its lanes does; it makes as many trips as its longest lane."""
    lanes = {}
    for line in open(path):
        if line.startswith("M "):
            t = line.split()
            lanes[int(t[1])] = [ord(c) - 65 for c in (t[2] if len(t) > 2 else "")]
    ids = sorted(lanes)
    names = ((1, "TICKET"), (2, "SCHED"), (4, "load site"), (8, "POST"))
    tot = {b: 0 for b, _ in names}; lane_tot = {b: 0 for b, _ in names}; trips_w = 0; trips_l = 0; nw = 0; reads = 0
    for w0 in range(0, len(ids) - 63, 64):
        ws = [lanes[i] for i in ids[w0:w0 + 64]]
        nw += 1
        T = max(len(x) for x in ws); trips_w += T; trips_l += sum(len(x) for x in ws) / 64.0
        for t in range(T):
            u = 0
            for x in ws:
                if t < len(x):
                    u |= x[t]
                    for b, _ in names:
                        if x[t] & b:
                            lane_tot[b] += 1
            for b, _ in names:
                if u & b:
                    tot[b] += 1
    print("%d wavefronts; trips per wavefront %.0f (mean lane %.0f)" % (nw, trips_w / nw, trips_l / nw))
    for b, nm in names:
        print("  %-10s executed by the wavefront in %5.1f %% of its trips; a lane is in it in %5.1f %% of its own" % (nm, 100.0 * tot[b] / trips_w, 100.0 * lane_tot[b] / 64.0 / trips_l))


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "--trips":
        return trips(sys.argv[1])
    path = sys.argv[1]
    reads = defaultdict(list)          # ir -> list of searches: (key, nSAi, [(phase, words)])
    cur = None
    for line in open(path):
        t = line.split()
        if t[0] == "M":
            continue
        if t[0] == "S":
            ir = int(t[1]); key = tuple(int(x) for x in t[2:7])
            cur = [key, 0, []]
            reads[ir].append(cur)
        elif t[0] == "a":
            cur[1] += 1
        else:
            cur[2].append((int(t[1]), int(t[2])))
    irs = sorted(reads)
    if not irs:
        sys.exit("no search records in the trace (STARAMD_SEED_FLAT=1 logs them; a whole-read trace has trip lines only: --trips)")
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
