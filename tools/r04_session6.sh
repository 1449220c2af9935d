#!/bin/bash
# round 4, GPU session 6: lane class down to off; contexts per GPU and host threads end to end; 2-pass with host timing
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s6; mkdir -p $O
timeout 900 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "c4|-|" "c3|-|STARAMD_LANE_CLASS=3" "c2|-|STARAMD_LANE_CLASS=2" "nolane|-|STARAMD_LANE=0" "nolane_lean0|-|STARAMD_LANE=0 STARAMD_LEAN_DEPTH=0" "noskip|-|STARAMD_PRUNE=3" > $O/ab.txt 2> $O/ab.err
grep -v "counts per pair" $O/ab.txt | tail -12; grep "counts per pair" $O/ab.txt | head -1 | cut -c1-900
run() { tag=$1; shift
  env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive ${EXTRA:-} > $O/b_$tag.json 2> $O/b_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"]))
    print("%-14s value %.3f  ms/step %.1f  throttled %s  parse %.1f convert %.1f emit %.1f M/s" % ("$tag", d["value"], d["ms_per_step"], e.get("cpu_throttled_in_timed_region"), e["pipeline"]["parse_Mreads_s"], e["pipeline"]["convert_Mreads_s"], e["pipeline"]["postmap_write_Mreads_s"]))
except Exception as ex:
    print("$tag FAILED", ex); print(open("$O/b_$tag.err").read()[-600:])
PY
}
run ctx2_t16 X=1
run ctx1_t16 STARAMD_CONTEXTS_PER_GPU=1
run ctx3_t16 STARAMD_CONTEXTS_PER_GPU=3
EXTRA="--host-threads 12" run ctx2_t12 X=1
EXTRA="--host-threads 8" run ctx2_t8 X=1
EXTRA="--host-threads 24" run ctx2_t24 X=1
run ctx2_t16b X=1
# 2-pass with the host stages timed
G=$(ls -d /dev/shm/star_amd_bench/genome_3100mb_*); R=$(ls -d $G/run_w1_n10000000)
STARAMD_HOST_TIMING=1 STARAMD_VERBOSE=1 timeout 600 star_amd/bin/star_amd --runMode alignReads --genomeDir $G/idx --readFilesIn $R/reads_r0_1.fq $R/reads_r0_2.fq --outFileNamePrefix $R/tp_ --runThreadN 16 --gpuBatchReads 400000 --twopassMode Basic --readMapNumber 4000000 > $O/twopass.out 2> $O/twopass.err
grep -E "end of pass 1|sjdb insert|staramd index stage|1st pass|star_amd:" $O/twopass.err | head -40
