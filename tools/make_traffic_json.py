#!/usr/bin/env python3
"""profiles/r02_pmc_hbm_traffic.json from the two PMC passes of tools/measure_session.sh (FETCH_SIZE and WRITE_SIZE, separate runs):
   tools/make_traffic_json.py <session dir> <genome_mb> <reads per launch> <out.json>
Per kernel: (sum of the counter over every dispatch of the kernel) / (number of batches) * 1024 -- rocprofv3 reports both in KB; a batch
is one launch of the hot path (k_seed_search runs once per batch, k_windows three times, k_stitch_win up to three times).  FETCH_SIZE is
NOT doubled: MI355X_MICROARCH.md's gfx950 x2 correction is calibrated for wide coalesced 16 B/lane streams only; these kernels issue
1- to 8-byte gathers, for which the guide calls the counter uncalibrated, so the figure is a lower bound of the bytes read."""
import json
import sys


def main():
    d, mb, reads, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    fe = json.load(open(d + "/pmc_fetch.summary.json"))
    wr = json.load(open(d + "/pmc_write.summary.json"))
    nb = fe["k_seed_search"]["dispatches"]
    assert nb == wr["k_seed_search"]["dispatches"]
    res = {"genome_mb": mb, "reads_per_launch": reads, "batches_in_each_pass": nb,
           "_how": __doc__.split("\n", 2)[2].strip()}
    names = {"k_windows": ["k_windows", "k_windows_big"], "k_stitch_win": ["k_stitch_win"], "k_seed_search": ["k_seed_search"],
             "k_stitch_replay": ["k_stitch_replay"], "k_stitch_finish": ["k_stitch_finish"], "k_gather": ["k_gather"]}
    for k, parts in names.items():
        f = sum(fe[p]["counters"]["FETCH_SIZE"] for p in parts if p in fe)
        w = sum(wr[p]["counters"]["WRITE_SIZE"] for p in parts if p in wr)
        res[k] = {"FETCH_SIZE_KB_per_launch": f / nb, "WRITE_SIZE_KB_per_launch": w / nb, "hbm_bytes_per_launch": (f + w) / nb * 1024.0}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_launch"] / 1e9 for k, v in res.items() if isinstance(v, dict)}))


if __name__ == "__main__":
    main()
