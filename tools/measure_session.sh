#!/bin/bash
# measurement session (run from the repo root on the GPU box): bench line + rocprofv3 kernel stats + PMC passes on the SAME workload
#   tools/measure_session.sh <tag> <genome_mb> [steps] [passes: "stats sq1 sq2 sq3 fetch write"]
R=$PWD; TAG=${1:-r02}; MB=${2:-3100}; STEPS=${3:-3}; PASSES=${4:-"stats sq1 sq2 fetch write"}
O=$R/gpurun_out/$TAG; mkdir -p $O
export STARAMD_BENCH_GENOME_MB=$MB
B="python $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-sweep --no-two-pass --no-extra-legs"
# data + an unprofiled line first (the profiled runs reuse the cached genome / reads in /dev/shm)
STARAMD_VERBOSE=1 timeout 1200 $B > $O/bench_plain.json 2> $O/bench_plain.err || { tail -5 $O/bench_plain.err; exit 1; }
python -c "import json;d=json.load(open('$O/bench_plain.json'));print('plain', d['value'], d['roofline']['per_kernel_ms'])"
cd /tmp && export TMPDIR=/tmp
for p in $PASSES; do
  case $p in
    stats) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o st -- $B > $O/bench_prof.json 2> $O/prof_stats.err ;;
    sq1) timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_sq1 -o c -- $B > /dev/null 2> $O/pmc_sq1.err ;;
    sq2) timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq2 -o c -- $B > /dev/null 2> $O/pmc_sq2.err ;;
    sq3) timeout 900 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_sq3 -o c -- $B > /dev/null 2> $O/pmc_sq3.err ;;
    tcp) timeout 900 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcp -o c -- $B > /dev/null 2> $O/pmc_tcp.err ;;
    fetch) timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o c -- $B > /dev/null 2> $O/pmc_fetch.err ;;
    write) timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o c -- $B > /dev/null 2> $O/pmc_write.err ;;
  esac
  echo "pass $p done: $(find $O -name '*counter_collection.csv' -o -name '*kernel_stats.csv' | wc -l) csv so far"
done
cd $R
# the dependent-gather ceiling of this box, this session (roofline.per_kernel.*.of_gather_ceiling)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/gather_ceiling.hip -o /tmp/gather_ceiling 2> /dev/null && timeout 120 /tmp/gather_ceiling 16384 256 > $O/gather_ceiling.txt 2>&1
# calibration of FETCH_SIZE on this engine's own access pattern (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern"): the dependent random
# 8-byte gathers of tools/gather_ceiling.hip under the same counter -- a known number of gathers, each of which moves one 64-byte sector
cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_calib -o c -- /tmp/gather_ceiling 16384 64 > $O/gather_ceiling_under_pmc.txt 2> $O/pmc_calib.err; cd $R
for d in pmc_sq1 pmc_sq2 pmc_sq3 pmc_tcp pmc_fetch pmc_write pmc_calib; do
  f=$(find $O/$d -name "*counter_collection.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && python profiles/pmc_summary.py $f > $O/$d.summary.json
done
f=$(find $O/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
f=$(find $O/prof_stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python profiles/trace_summary.py $f 40 > $O/kernel_trace_last.txt
# keep the merge small
find $O -type f -name "*.csv" -size +8M -delete
find $O -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
rm -rf $O/prof_stats
du -sh $O; ls $O
