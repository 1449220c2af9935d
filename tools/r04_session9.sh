#!/bin/bash
# round 4, GPU session 9: the whole GPU suite (hardware fuzzer included) on the round's kernels; 2-pass stage times with the new merge
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s9; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -n 4 > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --no-exclusive > $O/b.json 2> $O/b.err; tail -c 400 $O/b.json
G=$(ls -d /dev/shm/star_amd_bench/genome_3100mb_*); R=$(ls -d $G/run_w1_n2000000)
STARAMD_HOST_TIMING=1 STARAMD_VERBOSE=1 timeout 600 star_amd/bin/star_amd --runMode alignReads --genomeDir $G/idx --readFilesIn $R/reads_r0_1.fq $R/reads_r0_2.fq --outFileNamePrefix $R/tp_ --runThreadN 16 --gpuBatchReads 400000 --twopassMode Basic --readMapNumber 2000000 > $O/twopass.out 2> $O/twopass.err
grep -E "end of pass 1|sjdb insert|sjdbInsertJunctions|staramd index stage|1st pass|star_amd:" $O/twopass.err | head -30
