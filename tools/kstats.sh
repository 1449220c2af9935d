#!/bin/bash
# per-kernel times of one bench configuration under rocprofv3 --kernel-trace --stats:  tools/kstats.sh <outdir> <genome_mb> [ENV=VAL ...]
R=$PWD; O=$R/$1; MB=$2; shift 2; mkdir -p $O
export STARAMD_BENCH_GENOME_MB=$MB
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-two-pass --no-extra-legs"
env "$@" timeout 600 $B > $O/plain.json 2> $O/plain.err || tail -3 $O/plain.err
cd /tmp; export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o st -- $B > $O/prof.json 2> $O/prof.err
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -14 $O/kernel_stats.csv | cut -c1-110
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python profiles/trace_summary.py $f 30 > $O/kernel_trace_last.txt
rm -rf $O/prof
