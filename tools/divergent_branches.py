#!/usr/bin/env python3
"""Which branches of a kernel does the compiler take for divergent?  (LLVM's uniformity analysis on the optimised device IR of one engine source file)
A wave-cooperative kernel branches on wave-uniform conditions almost everywhere; a branch the compiler cannot prove uniform costs an exec-mask sequence
(s_and_saveexec / s_cbranch_execz / s_or) instead of an s_cbranch_scc, and every value merged behind it lives in a VGPR instead of an SGPR.
  tools/divergent_branches.py k_stitch k_stitch_win [extra hipcc flags]      prints function:line of every divergent branch, in source order"""
import collections, os, re, subprocess, sys
f, kernel, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = ["-fno-unroll-loops", "-DSTITCH_WAVES=4"] if f == "k_stitch" else []
ll = "/tmp/div_%d.ll" % os.getpid()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "-gline-tables-only", "--cuda-device-only",
                       "-emit-llvm", "-S"] + flags + extra + [os.path.join(root, "star_amd/csrc/engine", f + ".hip"), "-o", ll], stderr=subprocess.DEVNULL)
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/opt", "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-passes=print<uniformity>", "-disable-output", ll], stderr=subprocess.PIPE, text=True).stderr
src = open(ll).read(); os.unlink(ll)
loc = {m.group(1): (int(m.group(2)), m.group(3)) for m in re.finditer(r'^!(\d+) = !DILocation\(line: (\d+), column: \d+, scope: !(\d+)', src, re.M)}
sub = {m.group(1): m.group(2) for m in re.finditer(r'^!(\d+) = distinct !DISubprogram\(name: "([^"<]+)', src, re.M)}
lex = {m.group(1): m.group(2) for m in re.finditer(r'^!(\d+) = (?:distinct )?!DILexicalBlock(?:File)?\(scope: !(\d+)', src, re.M)}
def fn_of(s):
    n = 0
    while s in lex and n < 64: s = lex[s]; n += 1
    return sub.get(s, "?")
sec = txt.split("UniformityInfo for function '%s'" % kernel)[1].split("UniformityInfo for function")[0]
br = collections.Counter(); nval = 0
for line in sec.splitlines():
    if "DIVERGENT" not in line: continue
    if not re.search(r"DIVERGENT:\s+(br|switch) ", line): nval += 1; continue
    m = re.search(r"!dbg !(\d+)", line)
    br[(fn_of(loc[m.group(1)][1]), loc[m.group(1)][0]) if m and m.group(1) in loc else ("?", 0)] += 1
print("%s: %d divergent branches, %d other divergent values" % (kernel, sum(br.values()), nval))
for (fn, l), n in sorted(br.items()): print("  %-24s line %4d  x%d" % (fn, l, n))
