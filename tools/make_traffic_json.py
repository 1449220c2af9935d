#!/usr/bin/env python3
"""profiles/r04_pmc_hbm_traffic.json from the PMC passes of tools/measure_session.sh (FETCH_SIZE, WRITE_SIZE, SQ counters: separate runs):
   tools/make_traffic_json.py <session dir> <genome_mb> <reads per launch> <out.json>
Per kernel: (sum of the counter over every dispatch of the kernel) / (number of batches) * 1024 -- rocprofv3 reports both in KB; a batch
is one launch of the hot path (k_seed_search runs once per batch, k_windows three times, the stitch stage = k_stitch_lane + k_stitch_win).
FETCH_SIZE is NOT doubled: MI355X_MICROARCH.md's gfx950 x2 correction is calibrated for wide coalesced 16 B/lane streams only; these kernels
issue 1- to 8-byte gathers, for which the guide calls the counter uncalibrated, so the figure is a lower bound of the bytes read.
valu_busy_frac (the `roofline.issue` of the bench line): SQ_INSTS_VALU x 4 cycles (a wave64 VALU instruction occupies its SIMD16 for four
cycles) / (kernel time from the --stats pass x 2.4 GHz x 1024 SIMDs).  kernel_src_sha: sha256 of the files the kernel is written in (bench.KERNEL_SOURCES) at measurement time; bench.py reports a kernel's
figures only while those files hash to the same value (engine_src_sha: all of star_amd/csrc/engine, kept for reference)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    d, mb, reads, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import bench
    fe = json.load(open(d + "/pmc_fetch.summary.json"))
    wr = json.load(open(d + "/pmc_write.summary.json"))
    sq = json.load(open(d + "/pmc_sq1.summary.json")) if os.path.isfile(d + "/pmc_sq1.summary.json") else {}
    dur = {}
    if os.path.isfile(d + "/kernel_stats.csv"):
        for r in csv.DictReader(open(d + "/kernel_stats.csv")):
            dur[r["Name"]] = float(r["TotalDurationNs"])
    nb = fe["k_seed_search"]["dispatches"]
    assert nb == wr["k_seed_search"]["dispatches"]
    res = {"genome_mb": mb, "reads_per_launch": reads, "batches_in_each_pass": nb, "engine_src_sha": bench.engine_src_sha(),
           "_how": __doc__.split("\n", 2)[2].strip()}
    names = {"k_windows": ["k_windows", "k_windows_big"], "k_stitch_win": ["k_stitch_win", "k_stitch_lane"], "k_seed_search": ["k_seed_search"],
             "k_stitch_replay": ["k_stitch_replay"], "k_stitch_finish": ["k_stitch_finish"], "k_gather": ["k_gather"]}
    for k, parts in names.items():
        f = sum(fe[p]["counters"]["FETCH_SIZE"] for p in parts if p in fe)
        w = sum(wr[p]["counters"]["WRITE_SIZE"] for p in parts if p in wr)
        res[k] = {"kernel_src_sha": bench.kernel_src_sha(k), "FETCH_SIZE_KB_per_launch": f / nb, "WRITE_SIZE_KB_per_launch": w / nb, "hbm_bytes_per_launch": (f + w) / nb * 1024.0}
        valu = sum(sq[p]["counters"].get("SQ_INSTS_VALU", 0) for p in parts if p in sq)
        salu = sum(sq[p]["counters"].get("SQ_INSTS_SALU", 0) for p in parts if p in sq)
        t_ns = sum(dur.get(p, 0.0) for p in parts)
        if valu and t_ns:
            res[k]["valu_insts_per_launch"] = valu / nb; res[k]["salu_insts_per_launch"] = salu / nb
            res[k]["kernel_ms_per_launch_stats_pass"] = t_ns / nb / 1e6
            res[k]["valu_busy_frac"] = valu * 4.0 / (t_ns * 1e-9 * 2.4e9 * 1024)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: (round(v["hbm_bytes_per_launch"] / 1e9, 2), round(v.get("valu_busy_frac", 0), 3)) for k, v in res.items() if isinstance(v, dict)}))


if __name__ == "__main__":
    main()
