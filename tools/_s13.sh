cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s13; mkdir -p $O
run() { tag=$1; cpus=$2; th=$3; shift 3
env "$@" taskset -c $cpus timeout 400 python bench.py --steps 10 --warmup 3 --host-threads $th --no-cpu-baseline --no-extra-legs --no-exclusive > $O/b_$tag.json 2> $O/b_$tag.err
python - <<PY
import json
d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"])); p = e["pipeline"]
print("%-22s value %.3f ms/step %.1f device %.1f | cpu us/pair %s" % ("$tag", d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"]["device_total"], p["cpu_us_per_pair_by_stage"]))
PY
}
run two_cpus_nap150 0-1 2 X=1
run two_cpus_no_nap 0-1 2 STARAMD_WAIT_NAP_US=0
run four_cpus_nap150 0-3 4 X=1
run four_cpus_no_nap 0-3 4 STARAMD_WAIT_NAP_US=0
run sixteen_nap150 0-15 16 STARAMD_WAIT_NAP_US=150
