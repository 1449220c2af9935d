#!/bin/bash
# static figures of the kernels of one engine source file: registers and spills as the compiler reports them, and the instruction mix of the ISA
# (exec-mask bookkeeping, lane moves of spilled scalars, scratch traffic)       tools/isa_stats.sh k_stitch [extra hipcc flags]
f=$1; shift
extra=""; if [ "$f" = k_stitch ]; then extra="-fno-unroll-loops -DSTITCH_WAVES=${STITCH_WAVES:-4}"; fi
S=/tmp/isa_$$.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $extra "$@" -Rpass-analysis=kernel-resource-usage --cuda-device-only -S star_amd/csrc/engine/$f.hip -o $S 2> /tmp/isa_$$.err || { grep error /tmp/isa_$$.err | head; exit 1; }
python3 - $S /tmp/isa_$$.err <<'PY'
import re, sys
s = open(sys.argv[1]).read(); err = open(sys.argv[2]).read()
res = {}
for b in err.split('Function Name: ')[1:]:
    name = b.split()[0]
    g = lambda k: int(re.search(k + r': (\d+)', b).group(1))
    res[name] = (g('VGPRs'), g('SGPRs Spill'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'))
for m in re.finditer(r'^(\w+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if name not in res: continue
    ins = re.findall(r'^\s+([a-z_0-9]+)', body, re.M)
    c = lambda p: sum(1 for i in ins if re.match(p, i))
    v, ss, vs, sc, occ = res[name]
    print("%-18s VGPRs %3d  waves %d  SGPR spills %3d  VGPR spills %3d (scratch %3d B)   instr %5d: s_ %4d v_ %4d | saveexec %3d  exec-mask s_*_b64 %4d  cbranch_execz/nz %3d  cbranch_scc %3d | readlane %3d writelane %3d readfirstlane %3d | scratch ld/st %3d" % (
        name, v, occ, ss, vs, sc, len(ins), c(r's_'), c(r'v_'), c(r's_(and|or|xor|andn2|orn2)_saveexec'), c(r's_(and|or|andn2|xor|mov|cselect)_b64$'), c(r's_cbranch_exec'), c(r's_cbranch_scc'), c(r'v_readlane'), c(r'v_writelane'), c(r'v_readfirstlane'), c(r'scratch_')))
PY
rm -f $S /tmp/isa_$$.err
