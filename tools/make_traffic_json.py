#!/usr/bin/env python3
"""profiles/r05_pmc_hbm_traffic.json from the PMC passes of tools/measure_session.sh (FETCH_SIZE, WRITE_SIZE, SQ counters: separate runs):
   tools/make_traffic_json.py <session dir> <genome_mb> <reads per launch> <out.json>
The seed stage is the sum of its kernels (k_seed_plan + k_seed_units + k_seed_merge + k_seed_search).  Beside the counters: the dependent-gather ceiling of the same session
(<session dir>/gather_ceiling.txt, tools/gather_ceiling at 16 GiB, 524 288 lanes in flight: G sectors/s) and the static resources of the dominant kernel of each stage as the
compiler reports them (registers, scratch bytes per lane, wavefronts per SIMD: -Rpass-analysis=kernel-resource-usage on the committed sources).
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
    nb = fe["k_windows_big"]["dispatches"]             # (launched once per batch whatever the data)
    assert nb == wr["k_windows_big"]["dispatches"]
    res = {"genome_mb": mb, "reads_per_launch": reads, "batches_in_each_pass": nb, "engine_src_sha": bench.engine_src_sha(),
           "_how": __doc__.split("\n", 2)[2].strip()}
    names = {"k_windows": ["k_windows", "k_windows_big"], "k_stitch_win": ["k_stitch_win", "k_stitch_lane"], "k_seed_search": ["k_seed_search", "k_seed_plan", "k_seed_units", "k_seed_merge"],
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
            res[k]["valu_busy_frac"] = valu * 4.0 / (t_ns * 1e-9 * 2.4e9 * 1024)          # (4 cycles per wave64 instruction as in rounds 2-5, for the series; MI355X_MICROARCH.md gives 2 on the SIMD-32 of CDNA4: valu_busy_frac_simd32)
            res[k]["valu_busy_frac_simd32"] = valu * 2.0 / (t_ns * 1e-9 * 2.4e9 * 1024)
            res[k]["wave_instructions_per_read"] = (valu + salu) / nb / reads
    # static resources of the kernel that dominates each stage
    import re, subprocess
    main_kernel = {"k_seed_search": ("k_seed", "k_seed_units", []), "k_windows": ("k_window", "k_windows", []), "k_stitch_win": ("k_stitch", "k_stitch_win", ["-fno-unroll-loops", "-DSTITCH_WAVES=4"])}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k, (src, kern, extra) in main_kernel.items():
        try:
            p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"] + extra +
                               ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-S", os.path.join(root, "star_amd", "csrc", "engine", src + ".hip"), "-o", "/dev/null"],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            b = [x for x in p.stderr.split("Function Name: ")[1:] if x.split()[0] == kern][0]
            g = lambda key: int(re.search(key + r": (\d+)", b).group(1))
            res[k].update({"vgprs": g("VGPRs"), "scratch_bytes_per_lane": g(r"ScratchSize \[bytes/lane\]"), "waves_per_simd": g(r"Occupancy \[waves/SIMD\]"), "static_resources_of": kern})
        except Exception as e:
            res[k]["static_resources_error"] = repr(e)[:200]
    # resident blocks of 4 wavefronts per CU = wavefronts per SIMD, as the engine sized its launches in the session (STARAMD_VERBOSE line of the plain bench run)
    try:
        m = re.search(r"staramd: k_stitch_win (\d+) blocks/CU \(LDS \d+ B/block\), k_windows (\d+) blocks/CU, k_seed_search (\d+) blocks/CU", open(os.path.join(d, "bench_plain.err")).read())
        if m:
            res["k_stitch_win"]["waves_per_simd_resident"] = int(m.group(1)); res["k_windows"]["waves_per_simd_resident"] = int(m.group(2)); res["k_seed_search"]["waves_per_simd_resident"] = int(m.group(3))
        m = re.search(r"staramd: k_stitch_win main launch: depth (\d+), (\d+) blocks/CU", open(os.path.join(d, "bench_plain.err")).read())
        if m:           # (the launch that walks all but a handful of the windows; the full-depth launch behind it keeps the figure above)
            res["k_stitch_win"]["waves_per_simd_resident_full_depth_launch"] = res["k_stitch_win"].get("waves_per_simd_resident"); res["k_stitch_win"]["waves_per_simd_resident"] = int(m.group(2))
    except Exception:
        pass
    cal = os.path.join(d, "pmc_calib.summary.json")
    if os.path.isfile(cal):
        try:
            cj = json.load(open(cal))
            ck = [k for k in cj if k.startswith("chase")][0]
            steps = 64                                                        # tools/measure_session.sh: gather_ceiling 16384 64; per lane count 1, 2, 4, 8 blocks per CU a warm-up launch of 16 steps + the timed one
            gathers = 256 * 256 * (1 + 2 + 4 + 8) * (16 + steps)
            fetch_bytes = cj[ck]["counters"]["FETCH_SIZE"] * 1024.0
            res["fetch_size_calibration"] = {"pattern": "dependent random 8-byte gathers over a 16 GiB table (tools/gather_ceiling.hip), rocprofv3 --pmc FETCH_SIZE", "gathers": gathers,
                                             "FETCH_SIZE_bytes": fetch_bytes, "FETCH_SIZE_bytes_per_gather": fetch_bytes / gathers,
                                             "reading": "a random 8-byte gather moves one 64-byte sector from HBM; FETCH_SIZE_bytes_per_gather / 64 is what the counter reports of it -- divide the traffic figures of the gather-bound kernels by that ratio for absolute bytes"}
        except Exception as e:
            res["fetch_size_calibration"] = {"error": repr(e)[:200]}
    gc = os.path.join(d, "gather_ceiling.txt")
    if os.path.isfile(gc):
        for line in open(gc):
            m = re.search(r"524288 lanes in flight: ([0-9.]+) G gathers/s", line)
            if m:
                res["gather_ceiling_Gsectors_s"] = float(m.group(1)); res["gather_ceiling_how"] = "tools/gather_ceiling 16384 MiB table, 524288 lanes in flight, same session"
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: (round(v["hbm_bytes_per_launch"] / 1e9, 2), round(v.get("valu_busy_frac", 0), 3)) for k, v in res.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}))


if __name__ == "__main__":
    main()
