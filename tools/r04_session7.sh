#!/bin/bash
# round 4, GPU session 7: window rows per chunk in k_stitch_win, kernel turns of two contexts, 2-pass after the host cuts
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s7; mkdir -p $O
V=star_amd/lib/variants
timeout 900 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "new|-|" "oldloop|$V/libstaramd_oldloop.so|" "new_c4|-|STARAMD_LANE_CLASS=4" > $O/ab.txt 2> $O/ab.err
grep -v "counts per pair" $O/ab.txt | tail -6
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
for rep in a b c; do
run turns_c2_$rep X=1
run noturns_c2_$rep STARAMD_KERNEL_TURNS=0
run c1_$rep STARAMD_CONTEXTS_PER_GPU=1
done
run turns_c3 STARAMD_CONTEXTS_PER_GPU=3
G=$(ls -d /dev/shm/star_amd_bench/genome_3100mb_*); R=$(ls -d $G/run_w1_n10000000)
STARAMD_HOST_TIMING=1 STARAMD_VERBOSE=1 timeout 600 star_amd/bin/star_amd --runMode alignReads --genomeDir $G/idx --readFilesIn $R/reads_r0_1.fq $R/reads_r0_2.fq --outFileNamePrefix $R/tp_ --runThreadN 16 --gpuBatchReads 400000 --twopassMode Basic --readMapNumber 4000000 > $O/twopass.out 2> $O/twopass.err
grep -E "end of pass 1|sjdb insert|1st pass|star_amd:" $O/twopass.err | head -12
