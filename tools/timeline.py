#!/usr/bin/env python3
"""GPU timeline of a rocprofv3 run (`--kernel-trace --memory-copy-trace --output-format csv`): how busy the device was in the steady state of the bench's
timed region and where the gaps are.   tools/timeline.py <dir with *_kernel_trace.csv [+ *_memory_copy_trace.csv]> [seconds of steady state to analyse, from the end]"""
import csv, glob, os, sys

d = sys.argv[1]; tail_s = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
mf = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
ev = []
for f in kf:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K:" + r["Kernel_Name"][:28], r.get("Stream_Id", r.get("Queue_Id", "?"))))
for f in mf:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M:" + r.get("Direction", "copy")[:24], "-"))
ev.sort()
if not ev:
    sys.exit("no events")
# steady state = the window that ends with the last k_gather of the main run's densest stretch: simply the last `tail_s` seconds before the last kernel
t_end = max(e[1] for e in ev if e[2].startswith("K:")); t0 = t_end - int(tail_s * 1e9)
win = [e for e in ev if e[1] > t0 and e[0] < t_end]
def union(evs):
    tot = 0; cur_s = cur_e = None
    for s, e, *_ in sorted(evs):
        s = max(s, t0); e = min(e, t_end)
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
k = [e for e in win if e[2].startswith("K:")]; m = [e for e in win if e[2].startswith("M:")]
span = t_end - t0
print("window %.3f s: kernels busy (union) %.1f %%, copies busy %.1f %%, kernels or copies %.1f %%" % (span / 1e9, 100 * union(k) / span, 100 * union(m) / span, 100 * union(win) / span))
by = {}
for s, e, n, q in k:
    by.setdefault(n, [0, 0]); by[n][0] += 1; by[n][1] += min(e, t_end) - max(s, t0)
for n, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:14]:
    print("  %-32s %5d launches %9.2f ms  %5.1f %% of the window" % (n, c, t / 1e6, 100 * t / span))
# overlap: time with >= 2 kernels in flight
pts = []
for s, e, *_ in k:
    pts.append((max(s, t0), 1)); pts.append((min(e, t_end), -1))
pts.sort(); depth = 0; last = t0; two = 0
for t, dlt in pts:
    if depth >= 2: two += t - last
    depth += dlt; last = t
print("  two or more kernels in flight: %.1f %% of the window" % (100 * two / span))
# largest idle gaps of the kernel timeline
gaps = []; cur = t0
for s, e, n, q in sorted(k):
    if s > cur: gaps.append((s - cur, cur, n))
    cur = max(cur, e)
gaps.sort(reverse=True)
print("  largest gaps with no kernel running (ms, before which kernel):", [(round(g / 1e6, 2), n) for g, _, n in gaps[:8]])
