#!/bin/bash
# round 4, GPU session 8: mapper threads that sleep while their batch is on the device (vs spinning), contexts per GPU, batch size; 2-pass stages
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s8; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive ${EXTRA:-} > $O/b_$tag.json 2> $O/b_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"]))
    print("%-16s value %.3f  ms/step %.1f  parse %.1f convert %.1f emit %.1f M/s" % ("$tag", d["value"], d["ms_per_step"], e["pipeline"]["parse_Mreads_s"], e["pipeline"]["convert_Mreads_s"], e["pipeline"]["postmap_write_Mreads_s"]))
except Exception as ex:
    print("$tag FAILED", ex); print(open("$O/b_$tag.err").read()[-600:])
PY
}
for rep in a b; do
run c2_sleep_$rep X=1
run c2_spin_$rep STARAMD_SPIN_WAIT=1
run c1_sleep_$rep STARAMD_CONTEXTS_PER_GPU=1
run c1_spin_$rep STARAMD_CONTEXTS_PER_GPU=1 STARAMD_SPIN_WAIT=1
run c2_sleep_noturns_$rep STARAMD_KERNEL_TURNS=0
done
EXTRA="--reads 800000" run c2_sleep_800k X=1
EXTRA="--reads 800000" run c1_sleep_800k STARAMD_CONTEXTS_PER_GPU=1
G=$(ls -d /dev/shm/star_amd_bench/genome_3100mb_*); R=$(ls -d $G/run_w1_n10000000)
STARAMD_HOST_TIMING=1 STARAMD_VERBOSE=1 timeout 600 star_amd/bin/star_amd --runMode alignReads --genomeDir $G/idx --readFilesIn $R/reads_r0_1.fq $R/reads_r0_2.fq --outFileNamePrefix $R/tp_ --runThreadN 16 --gpuBatchReads 400000 --twopassMode Basic --readMapNumber 4000000 > $O/twopass.out 2> $O/twopass.err
grep -E "end of pass 1|sjdb insert|1st pass|star_amd:" $O/twopass.err | head -12
grep -E "emit: " $O/twopass.err | tail -4
