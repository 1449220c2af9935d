#!/bin/bash
# Where do the lanes of a wavefront hand data to one another through memory without a LOCKSTEP() / fence?  (wavefront emulator, oracle/wave_emul/emu.h)
#   tests/tools/emul_hunt.sh <dataset> <nReads> [emul_run flags]
# Runs the emulated engine with the lanes scheduled in ascending and in descending order, an LDS hash logged at every rendezvous (STARAMD_EMUL_HASHLOG),
# and prints, per kernel and wavefront, the first rendezvous at which the two runs differ, with the source lines of that operation and of the one before:
# the code between the two is where the order matters.  (Work lists filled through atomics make later kernels legitimately differ: look at the FIRST kernel.)
cd "$(dirname "$0")/../.."
make -s oracle/_build/libstaramd_emul.so || exit 1
D=$1; N=$2; shift; shift
T=$(mktemp -d /tmp/emuhunt.XXXXXX)
STARAMD_EMUL_HASHLOG=$T/asc.log timeout 1800 python tests/emul_run.py $D $T/a $N "$@" 2>&1 | tail -1
STARAMD_EMUL_ORDER=desc STARAMD_EMUL_HASHLOG=$T/desc.log timeout 1800 python tests/emul_run.py $D $T/d $N "$@" 2>&1 | tail -1
python - $T/asc.log $T/desc.log > $T/diff.txt <<'PY'
import sys
from collections import OrderedDict
def streams(path):
    d = OrderedDict()
    for l in open(path):
        p = l.split()
        if len(p) >= 10:
            d.setdefault((p[0], p[2], p[4]), []).append((p[7], p[9], l.rstrip()))
    return d
A, B = streams(sys.argv[1]), streams(sys.argv[2])
shown = 0
for k in A:
    x, y = A[k], B.get(k, [])
    for i, (u, v) in enumerate(zip(x, y)):
        if u[:2] != v[:2]:
            print(k, "first difference at rendezvous", i); print("  before:", x[i - 1][2] if i else None); print("  asc   :", u[2]); print("  desc  :", v[2]); shown += 1; break
    if shown >= 3: break
if not shown: print("the two lane orders agree at every rendezvous")
PY
cat $T/diff.txt
for a in $(grep -oE "lib\+0x[0-9a-f]+" $T/diff.txt | head -3 | sort -u | sed 's/lib+//'); do echo "== $a"; /opt/rocm/lib/llvm/bin/llvm-symbolizer --obj=oracle/_build/libstaramd_emul.so $a 2>/dev/null | grep -E "csrc/engine" | head -4; done
rm -rf $T
