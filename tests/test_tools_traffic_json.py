"""tools/make_traffic_json.py (HBM bytes / issue figures per kernel from the rocprofv3 PMC summaries of a measurement session) re-run on the summaries committed under
profiles/: it reproduces the figures `bench.py` reports as roofline.traffic / roofline.issue.  Protects the last step of a GPU measurement session from a tool bug."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def test_traffic_json_from_the_committed_pmc_summaries(tmp_path):
    d = str(tmp_path)
    for src, dst in (("r06_3100mb_pmc_fetch.summary.json", "pmc_fetch.summary.json"), ("r06_3100mb_pmc_write.summary.json", "pmc_write.summary.json"),
                     ("r06_3100mb_pmc_sq1.summary.json", "pmc_sq1.summary.json"), ("r06_3100mb_kernel_stats.csv", "kernel_stats.csv"),
                     ("r06_3100mb_engine_launch_geometry.txt", "bench_plain.err"), ("r06_gather_ceiling_16g.txt", "gather_ceiling.txt")):
        shutil.copy(os.path.join(P, src), os.path.join(d, dst))
    out = os.path.join(d, "traffic.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_traffic_json.py"), d, "3100", "400000", out], cwd=ROOT)
    new, old = json.load(open(out)), json.load(open(os.path.join(P, "r06_pmc_hbm_traffic.json")))
    assert new["genome_mb"] == old["genome_mb"] == 3100 and new["reads_per_launch"] == old["reads_per_launch"]
    for k in ("k_seed_search", "k_windows", "k_stitch_win"):
        for f in ("hbm_bytes_per_launch", "valu_insts_per_launch", "salu_insts_per_launch", "valu_busy_frac"):
            assert abs(new[k][f] - old[k][f]) <= 1e-9 * abs(old[k][f]), (k, f, new[k][f], old[k][f])
        assert new[k]["waves_per_simd_resident"] == old[k]["waves_per_simd_resident"] and new[k]["scratch_bytes_per_lane"] == old[k]["scratch_bytes_per_lane"], k
    assert new["gather_ceiling_Gsectors_s"] == old["gather_ceiling_Gsectors_s"] > 40
