#!/usr/bin/env python3
"""exec-mask branches (s_*_saveexec, s_cbranch_execz/nz) and lane moves of spilled scalars (v_readlane / v_writelane) of one kernel by source line
   tools/isa_by_line.py k_stitch k_stitch_win [extra flags]"""
import collections, os, re, subprocess, sys
f, kernel, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = ["-fno-unroll-loops", "-DSTITCH_WAVES=3"] if f == "k_stitch" else []
S = "/tmp/isl_%d.s" % os.getpid()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "-gline-tables-only", "--cuda-device-only", "-S"] + flags + extra +
                      [os.path.join(root, "star_amd/csrc/engine", f + ".hip"), "-o", S], stderr=subprocess.DEVNULL)
s = open(S).read(); os.unlink(S)
body = re.search(r'^%s:.*?^\.Lfunc_end\d+:' % kernel, s, re.S | re.M).group(0)
files = {m.group(1): (m.group(3) or m.group(2)).split('/')[-1] for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"\s*(?:"([^"]*)")?', s)}
cur = None; h = collections.defaultdict(collections.Counter)
for line in body.splitlines():
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', line)
    if m: cur = (files.get(m.group(1), '?'), int(m.group(2))); continue
    m = re.match(r'\s+([a-z_0-9]+)', line)
    if not m: continue
    i = m.group(1)
    k = 'saveexec' if 'saveexec' in i else 'execbr' if i.startswith('s_cbranch_exec') else 'readlane' if i.startswith('v_readlane') else 'writelane' if i.startswith('v_writelane') else 'scratch' if i.startswith('scratch_') else None
    if k: h[cur][k] += 1
    h[cur]['all'] += 1
tot = collections.Counter()
for c in h.values(): tot.update(c)
print(kernel, dict(tot))
for (fl, l), c in sorted(h.items(), key=lambda x: -(x[1]['saveexec'] + x[1]['execbr']))[:40]:
    if c['saveexec'] + c['execbr'] == 0: break
    print("  %-16s %4d  saveexec %3d  exec branches %3d  readlane %3d writelane %3d  (instructions %4d)" % (fl, l, c['saveexec'], c['execbr'], c['readlane'], c['writelane'], c['all']))
