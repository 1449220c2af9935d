#!/bin/bash
# Memory check of the KERNELS on a machine without a GPU: the wavefront-emulated engine built with AddressSanitizer (oracle/wave_emul/build.sh asan), so that
# every access of the kernels to the "device" buffers (host allocations of the same sizes, padding included) is bounds-checked -- an out-of-range global access
# is a memory fault on the device.  Runs the data sets and forced paths of tests/test_wave_emul.py.   usage: tests/tools/emul_asan.sh
cd "$(dirname "$0")/../.."
bash oracle/wave_emul/build.sh asan || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
bad=0
run() {   # run <env assignments or -> <dataset> <n> [flags]
  local envs=$1; shift
  local W; W=$(mktemp -d /tmp/emuasan.XXXXXX)
  [ "$envs" = "-" ] && envs=""
  out=$(env $envs LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 STARAMD_EMUL_LIB=$PWD/oracle/_build/libstaramd_emul_asan.so timeout 1500 python tests/emul_run.py "$1" $W "${@:2}" 2>&1 | tail -40)
  rm -rf $W
  last=$(echo "$out" | tail -1)
  if [[ "$last" == OK* ]]; then echo "ok    [$envs] $* : $last"; else echo "FAIL  [$envs] $*"; echo "$out" | grep -v longer_pathname | head -30; bad=$((bad+1)); fi
}
run - se50 150
run - pe101 40
run - pe101 25 --gpuResultSelect All
run - pe150_indel 12
run - pe76_overlap 40 --peOverlapNbasesMin 10 --peOverlapMMp 0.1
run - pe150_chim 20 --chimSegmentMin 15 --chimJunctionOverhangMin 15
run "STARAMD_POOL_SLACK=64 STARAMD_SEEDS_PER_READ=1 STARAMD_WINDOWS_PER_READ=1 STARAMD_WA_PER_READ=1 STARAMD_TR_PER_READ=1" pe101 25 --gpuResultSelect All
run "STARAMD_CAP_WINDOWS=2 STARAMD_CAP_WA_BLOCKS=2" pe101 25 --gpuResultSelect All
run "STARAMD_CAP_WINDOWS=1 STARAMD_CAP_WA_BLOCKS=1 STARAMD_CAP_WINDOWS_MID=3 STARAMD_CAP_WA_BLOCKS_MID=3" pe101 25 --gpuResultSelect All
run "STARAMD_CAP_WINDOWS=2 STARAMD_CAP_WA_BLOCKS=2 STARAMD_CAP_WINDOWS_MID=0" pe101 30
run "STARAMD_STITCH_ARENA=256" pe101 25 --gpuResultSelect All
run "STARAMD_CAND_KB_PER_WAVE=1" pe101 30
run "STARAMD_LIGHT_EST=0" pe101 30
run "STARAMD_LIGHT_EST=4000000000" pe101 25 --gpuResultSelect All
run "STARAMD_PRUNE=0" pe101 30
run "STARAMD_LANE_CLASS=31" pe101 60 --gpuResultSelect All
run "STARAMD_LANE_CLASS=31 STARAMD_LANE_ARENA=256" pe125_protrude 160
# the experimental seed kernel (one load site, both words of a funnel load always read: the padding of the index arrays and of the read buffer is what makes that legal)
# and the post-pruning cost class; stitching made cheap so that thousands of reads go through the seed search
echo "$bad case(s) with a report"
exit $bad
