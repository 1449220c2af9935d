#!/bin/bash
# round 4, GPU session 3: host diagnostics (is the container CPU-throttled? are writes into page-locked memory slow?), k_windows owner map A/B, pass-A section profile
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s3; mkdir -p $O
V=star_amd/lib/variants
{ echo "nproc $(nproc)"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null; lscpu | grep -E "Model name|Socket|NUMA|Thread|Core|^CPU\(s\)|MHz"; numactl -H 2>/dev/null | head -12; cat /sys/fs/cgroup/memory.max; cat /sys/kernel/mm/transparent_hugepage/enabled; python -c "import os;print('affinity', len(os.sched_getaffinity(0)))"; } > $O/host.txt 2>&1
cat $O/host.txt
timeout 900 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "map_w4c256|-|" "nomap_w4c256|-|STARAMD_WIN_OWNER_MAP=0" "map_w6c128|$V/libstaramd_w6.so|STARAMD_CAP_WINDOWS=128" "nomap_w6c128|$V/libstaramd_w6.so|STARAMD_CAP_WINDOWS=128 STARAMD_WIN_OWNER_MAP=0" \
  "map_w6c128_h8k|$V/libstaramd_w6.so|STARAMD_CAP_WINDOWS=128 STARAMD_WIN_HASH_BITS=8192" "map_w6c192|$V/libstaramd_w6.so|STARAMD_CAP_WINDOWS=192" > $O/ab.txt 2> $O/ab.err
grep -v "counts per pair" $O/ab.txt | tail -14
timeout 600 python tools/ab_kernels.py --genome-mb 3100 --batches 2 --repeat 2 --rounds 1 --out $O/prof.json "profile|star_amd/lib/libstaramd_profile.so|" > $O/prof.txt 2> $O/prof.err
cat $O/prof.txt
run() { tag=$1; shift; cat /sys/fs/cgroup/cpu.stat > $O/cpustat_$tag.before 2>/dev/null
  env "$@" STARAMD_HOST_TIMING=1 timeout 400 python bench.py --steps 16 --warmup 4 --reads 400000 --no-cpu-baseline --no-extra-legs --no-exclusive ${EXTRA:-} > $O/b_$tag.json 2> $O/b_$tag.err
  cat /sys/fs/cgroup/cpu.stat > $O/cpustat_$tag.after 2>/dev/null
  python - <<PY
import json, re
try:
    d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); e = open("$O/b_$tag.err").read()
    al = [float(x) for x in re.findall(r"parse alloc\s+([0-9.]+) ms", e)]; fi = [float(x) for x in re.findall(r"parse fill\s+([0-9.]+) ms", e)]
    print("%-14s value %.3f  ms/step %.1f  parse alloc mean %.1f ms  fill mean %.1f ms" % ("$tag", d["value"], d["ms_per_step"], sum(al) / max(1, len(al)), sum(fi) / max(1, len(fi))))
except Exception as ex:
    print("$tag FAILED", ex); print(open("$O/b_$tag.err").read()[-600:])
PY
}
run default X=1
run pageable STARAMD_PAGEABLE_BATCHES=1
EXTRA="--host-threads 32" run threads32 X=1
EXTRA="--host-threads 16" run threads16 X=1
run default2 X=1
