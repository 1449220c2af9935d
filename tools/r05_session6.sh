#!/bin/bash
# round 5, GPU session 6: per-kernel durations (rocprofv3 --kernel-trace --stats) of the seed stage in rounds and with the units' searches back to back
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r05s6; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for m in 2 1; do
STARAMD_SEED_UNITS=$m timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_m$m -o st -- python $R/tools/ab_kernels.py --genome-mb 3100 --batches 2 --repeat 2 --rounds 1 "only|-|" > $O/ab_m$m.txt 2> $O/ab_m$m.err
f=$(find $O/prof_m$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_m$m.csv
f=$(find $O/prof_m$m -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/profiles/trace_summary.py $f 60 > $O/kernel_trace_last_m$m.txt
rm -rf $O/prof_m$m
echo "mode $m:"; head -30 $O/kernel_stats_m$m.csv | cut -c1-160
done
