#!/bin/bash
# round 4, GPU session 5: stitch tail (STARAMD_LIGHT_EST), k_windows preload variants, GPU timeline of the pipelined run, thread count of the CPU baseline
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s5; mkdir -p $O
V=star_amd/lib/variants
timeout 900 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "base|-|" "nopre|$V/libstaramd_nopre.so|" "pre_w5|$V/libstaramd_pre_w5.so|" "est4096|-|STARAMD_LIGHT_EST=4096" "est512|-|STARAMD_LIGHT_EST=512" "est64|-|STARAMD_LIGHT_EST=64" "est16|-|STARAMD_LIGHT_EST=16" "c3|-|STARAMD_LANE_CLASS=3" "c5|-|STARAMD_LANE_CLASS=5" > $O/ab.txt 2> $O/ab.err
grep -v "counts per pair" $O/ab.txt | tail -18
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --no-exclusive > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/bench_traced.err
cd $GRAFT_REPO_ROOT
python tools/timeline.py $O/trace 0.6 > $O/timeline.txt 2>&1; cat $O/timeline.txt
find $O/trace -name "*.csv" -size +20M -delete
python - <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import bench
from oracle import refstar
g = bench.genome_dir(type("A", (), {"read_len": 101, "workdir": "/dev/shm/star_amd_bench"})(), 3100); idx = os.path.join(g, "idx")
import glob
runs = sorted(glob.glob(os.path.join(g, "run_w1_n*")), key=os.path.getmtime); rd = runs[-1]
fq = [os.path.join(rd, "reads_r0_%d.fq" % m) for m in (1, 2)]
small = ["--limitIObufferSize", "2000000", "50000000"]
out = {}
t = time.perf_counter(); refstar.align(idx, fq, os.path.join(rd, "tc_"), threads=16, extra=["--readMapNumber", "1"] + small, timeout=600); tl = time.perf_counter() - t
t = time.perf_counter(); refstar.align(idx, fq, os.path.join(rd, "tc_"), threads=16, extra=["--readMapNumber", "1"] + small, timeout=600); tl = time.perf_counter() - t
for th in (16, 24, 32, 48):
    t = time.perf_counter(); refstar.align(idx, fq, os.path.join(rd, "tc_"), threads=th, extra=["--readMapNumber", "3000000"] + small, timeout=600); tf = time.perf_counter() - t
    out[th] = 3.0 / max(tf - tl, 1e-3); print("reference STAR %d threads: %.3f Mreads/s (load %.1f s, full %.1f s)" % (th, out[th], tl, tf), flush=True)
json.dump(out, open("gpurun_out/s5/cpu_threads.json", "w"))
PY
