#!/bin/bash
# One GPU session (run from the repo root on the GPU box: gpurun -- 'bash tools/session.sh <mode> ...'); output under gpurun_out/<tag>/.
# The eleven sessions of round 5 (profiles/r05_*session*.txt name them) were instances of these five modes; their one-off scripts are in the history.
#
#   tools/session.sh ab <tag> "<variant>" ["<variant>" ...]    kernel A/B on ONE box: tools/ab_kernels.py at 3.1 Gb, 3 batches x 2 passes x 2 rounds (ABAB...), result buffers
#                                                              of every variant compared with the first.  variant = "name|lib or -|ENV=V ENV2=V"
#                                                              (libraries: tools/build_variants.sh <tag>:<kernel file>:"<flags>" BEFORE the call -- they travel with the snapshot)
#   tools/session.sh e2e <tag> "<name> ENV=V ..." [...]        end to end, alternating: bench.py --steps 20 --warmup 5 without CPU baseline / optional legs, with the pipeline
#                                                              event log; one line per run: value, ms per step, device ms, CPU per pair by stage, fast-path counts
#   tools/session.sh tests <tag> [pytest args]                 the GPU suite (default: tests -m gpu) + the hardware fuzzer (80 combinations)
#   tools/session.sh measure <tag>                             tools/measure_session.sh: plain bench + rocprofv3 --stats + PMC passes (sq1, fetch, write) + gather ceiling;
#                                                              afterwards, here: tools/make_traffic_json.py gpurun_out/<tag> 3100 400000 profiles/rNN_pmc_hbm_traffic.json
#   tools/session.sh final <tag>                               the default bench run (CPU baseline, full-size parity, optional legs) + the whole GPU suite
cd ${GRAFT_REPO_ROOT:-.}
mode=$1; tag=${2:-s}; shift 2
O=gpurun_out/$tag; mkdir -p $O
case $mode in
  ab)
    STARAMD_VERBOSE=1 timeout 900 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json "$@" > $O/ab.txt 2> $O/ab.err; echo "ab rc $?"
    grep -v "counts per pair" $O/ab.txt | tail -40; grep "staramd: k_\|seed units" $O/ab.err | sort | uniq -c | cut -c1-220 | head -12; tail -2 $O/ab.err ;;
  e2e)
    for spec in "$@"; do
      name=${spec%% *}; envs=${spec#"$name"}
      env $envs STARAMD_PIPELINE_LOG=$PWD/$O/plog_$name.txt timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive > $O/b_$name.json 2> $O/b_$name.err
      python - <<PY
import json
d = json.loads(open("$O/b_$name.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"])); p = e["pipeline"]
print("%-18s value %.3f ms/step %.1f device %s (device only %.2f M/s) map calls ms %.1f | cpu us/pair %s | fast %s" % ("$name", d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"],
      d.get("device_only_value") or 0, p["map_batch_call_s"] / 20 * 1e3, p["cpu_us_per_pair_by_stage"], p["fast_path_batches"]))
PY
    done ;;
  tests)
    timeout 1200 python -m pytest ${@:-tests -m gpu} -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
    timeout 400 python tests/tools/fuzz_engine.py 80 $RANDOM > $O/fuzz_engine_hardware.log 2>&1; echo "fuzz rc $?"; tail -1 $O/fuzz_engine_hardware.log ;;
  measure)
    bash tools/measure_session.sh $tag 3100 3 "stats sq1 fetch write" > $O.log 2>&1; tail -12 $O.log ;;
  final)
    timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-800 $O/bench_line.json
    cp /dev/shm/star_amd_bench/bench_extra.json $O/bench_extra.json 2>/dev/null
    timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log ;;
  *) echo "modes: ab e2e tests measure final"; exit 2 ;;
esac
