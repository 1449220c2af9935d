"""The kernel SOURCES of the engine (star_amd/csrc/engine/*.hip) executed on the CPU by the wavefront emulator (oracle/wave_emul/emu.h: one fiber per
work-item, cross-lane operations as rendezvous of the 64 lanes) and compared with the oracle, result buffer by result buffer, byte for byte.
This is the `-m "not gpu"` twin of tests/test_gpu_parity.py: seed search, window building, the wave-cooperative stitcher, verify / replay / finish and the
result gather run exactly as written, including the rarely taken paths forced through the STARAMD_* knobs.  What it cannot show is what only hardware
shows (memory ordering, address spaces, occupancy, speed) -- that is the `-m gpu` suite.

Every case runs with the lanes of a wavefront scheduled in ascending and in descending order between two rendezvous: the results must not depend on
it.  A dependence means that lanes hand data to one another through memory and rely on running in lock step; the place gets a LOCKSTEP() (dev.h),
which is nothing on the device and a rendezvous here."""
import os
import subprocess
import sys

import pytest

from util import refstar

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(ROOT, "oracle", "_build", "libstaramd_emul.so")
SELFTEST = os.path.join(ROOT, "oracle", "_build", "wave_emul_selftest")
CLANG = os.environ.get("EMUL_CXX", "/opt/rocm/lib/llvm/bin/clang++")
pytestmark = pytest.mark.skipif(not refstar.have_ref() or not os.path.exists(CLANG), reason="oracle/_ref/STAR (index generation) or the host clang++ of ROCm missing")


@pytest.fixture(scope="module", autouse=True)
def emul_lib(built):
    """the emulated engine is (re)built from the kernel sources as they are now"""
    import fcntl
    os.makedirs(os.path.join(ROOT, "oracle", "_build"), exist_ok=True)
    with open(os.path.join(ROOT, "oracle", "_build", ".emul.lock"), "w") as lock:          # (pytest-xdist workers would build side by side)
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "oracle/_build/libstaramd_emul.so"], cwd=ROOT)
    assert os.path.exists(LIB) and os.path.exists(SELFTEST)


def _run(dataset, n, more, tmp_path, env=None, order="asc"):
    import pickle
    from util import prepare
    e = dict(os.environ, **(env or {}))
    e["STARAMD_EMUL_ORDER"] = order
    os.makedirs(str(tmp_path), exist_ok=True)
    info = prepare(dataset, str(tmp_path), need_ref=False)          # (generated once per test session)
    pkl = os.path.join(str(tmp_path), "info.pkl")
    pickle.dump(info, open(pkl, "wb"))
    p = subprocess.run([sys.executable, os.path.join(HERE, "emul_run.py"), pkl, str(tmp_path), str(n)] + more, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    last = (p.stdout.strip().splitlines() or [""])[-1]
    assert p.returncode == 0 and last.startswith("OK"), (last, p.stderr[-1500:])


def test_primitives(built):
    p = subprocess.run([SELFTEST], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0 and "primitives OK" in p.stdout, p.stdout + p.stderr


# (data set, reads, flags): the read counts stop short of the few reads of each set whose walks take minutes in the emulator
CASES = [("se50", 150, []), ("pe101", 40, []), ("pe101", 25, ["--gpuResultSelect", "All"]), ("pe150_indel", 12, []), ("pe76_overlap", 40, []),
         ("pe150_chim", 20, ["--chimSegmentMin", "15", "--chimJunctionOverhangMin", "15"]),
         # --seedSearchLmax: the backward search is given Shift + 1 bases and runs over the start of its piece into an N / the mate spacer; the reference adds
         # those codes into the L-mer prefix as they are (carries and borrows included) -- found by the emulated fuzzer, k_seed.hip searchOneDist
         ("pe101", 30, ["--gpuResultSelect", "All", "--seedSearchLmax", "30", "--seedSearchStartLmax", "12"]),
         # --alignEndsProtrude + 5' clipping: the second mate starts before the first exon, the extension length of the mate-gap stitch is "negative"
         # (no extension in the reference: `(int) L` loop bound, extendAlign.cpp:59) -- found by the hardware fuzzer, k_stitch.hip coopExtendBody
         ("pe125_protrude", 160, ["--gpuResultSelect", "All", "--alignEndsProtrude", "15", "ConcordantPair", "--clip5pNbases", "20", "20"]),
         # 2x300: the packed reads of a k_stitch_lane block outgrow its dynamic LDS; the batch takes the cooperative launches alone (engine.hip launchAll)
         ("pe300", 12, [])]


# every data set with one lane order, the paired-end set of the forced cases with both (tests/tools/fuzz_engine.py alternates the order over hundreds of
# combinations; the suite stays short)
@pytest.mark.parametrize("dataset,n,more,order", [c + ("asc" if i % 2 == 0 else "desc",) for i, c in enumerate(CASES)] + [CASES[1] + ("asc",), CASES[2] + ("desc",)])
def test_kernels_match_oracle(dataset, n, more, order, tmp_path, built):
    _run(dataset, n, more, tmp_path, order=order)


# the rarely taken paths of tests/test_gpu_parity.py::test_forced_rare_paths (same knobs)
FORCED = {
    "tiny_pools": {"STARAMD_POOL_SLACK": "64", "STARAMD_SEEDS_PER_READ": "1", "STARAMD_WINDOWS_PER_READ": "1", "STARAMD_WA_PER_READ": "1", "STARAMD_TR_PER_READ": "1"},
    "window_overflow": {"STARAMD_CAP_WINDOWS": "2", "STARAMD_CAP_WA_BLOCKS": "2"},
    "window_overflow_twice": {"STARAMD_CAP_WINDOWS": "1", "STARAMD_CAP_WA_BLOCKS": "1", "STARAMD_CAP_WINDOWS_MID": "3", "STARAMD_CAP_WA_BLOCKS_MID": "3"},
    "window_overflow_no_middle": {"STARAMD_CAP_WINDOWS": "2", "STARAMD_CAP_WA_BLOCKS": "2", "STARAMD_CAP_WINDOWS_MID": "0"},
    "block_overflow": {"STARAMD_CAP_WA_BLOCKS": "2", "STARAMD_CAP_WA_BLOCKS_MID": "3"},      # seed-list blocks run out before table rows do (first and middle launch)
    "arena_overflow": {"STARAMD_STITCH_ARENA": "256"},
    "log_overflow": {"STARAMD_CAND_KB_PER_WAVE": "1"},
    "all_heavy": {"STARAMD_LIGHT_EST": "0"},
    "all_light": {"STARAMD_LIGHT_EST": "4000000000"},
    "no_pruning": {"STARAMD_PRUNE": "0"},
    "lane_all_classes": {"STARAMD_LANE_CLASS": "31"},         # every light read of few seeds per window through the lane-per-read stitcher (k_stitch_lane.hip), not only the cheapest classes
    "lane_off": {"STARAMD_LANE": "0"},                        # ... and none of them: the cooperative walk alone
    "lean_tiny_lane_off": {"STARAMD_LEAN_DEPTH": "3", "STARAMD_LANE": "0"},
    "main_depth_tiny": {"STARAMD_MAIN_DEPTH": "3"},           # the main cooperative launch behind the lane kernel holds windows of 2 seeds: almost everything goes on to the full-depth launch
    "main_depth_off": {"STARAMD_MAIN_DEPTH": "0"},            # one cooperative launch at full depth (three blocks per CU)
    "lane_tiny_arena": {"STARAMD_LANE_CLASS": "31", "STARAMD_LANE_ARENA": "256"},   # records outgrow the lane's arena: the read goes on to the cooperative kernel
    "no_sjdb_hash": {"STARAMD_NO_SJDB_HASH": "1"},            # annotated junctions looked up by bisection (what an index does whose coordinates / junction count do not fit the hash)
    "no_leaf_skipping": {"STARAMD_PRUNE": "3"},               # window pruning as in round 3, every leaf of every walked window finalised
    "win_owner_map_off": {"STARAMD_WIN_OWNER_MAP": "0"},      # k_windows: the covered bins in a Bloom filter + serial owner look-ups (what a read does whose windows cover more bins than the owner map holds)
    "win_owner_map_tiny": {"STARAMD_WIN_HASH_BITS": "1024", "STARAMD_WIN_HASH_BITS_MID": "4096"},     # 32-slot owner map: reads switch between the map and the filter
    "seed_lane_per_read": {"STARAMD_SEED_UNITS": "0"},         # the seed stage as one nest per lane (k_seed_search over every read) instead of lane = unit
    "seed_unit_pool_tiny": {"STARAMD_SEED_GROUPS_PER_READ": "1"},   # the group / unit pools hold less than half of the batch: the rest is handed on to k_seed_search
    "seed_one_slot_per_unit": {"STARAMD_SEED_SLOT_LIMIT": "1"},     # a unit that finds a second seed hands its read on
    "no_sa_keys": {"STARAMD_SA_KEYS": "0"},                         # the seed stage probes the packed suffix array and the genome (what it does with a sparse suffix array, or when 16 bytes per suffix do not fit)
}


@pytest.mark.parametrize("case", sorted(FORCED))
def test_forced_rare_paths(case, tmp_path, built):
    i = sorted(FORCED).index(case)                 # mode and lane order alternate over the cases
    if i % 2 == 0:
        _run("pe101", 25, ["--gpuResultSelect", "All"], tmp_path, env=FORCED[case], order="asc" if i % 4 == 0 else "desc")
    else:
        _run("pe101", 30, [], tmp_path, env=FORCED[case], order="desc" if i % 4 == 1 else "asc")


# ---- the shipped front end (main.cpp + cli_run.cpp) linked against the emulated engine: oracle/_build/star_amd_emul_cli -------------------------
EMUL_CLI = os.path.join(ROOT, "oracle", "_build", "star_amd_emul_cli")
SMALL = {"STARAMD_WIN_BLOCKS_BIG": "2"}       # (the 64 blocks of the last k_windows launch own 3.9 GB of work space, which the emulated hipMalloc fills with its pattern)


@pytest.fixture(scope="module")
def emul_cli(emul_lib):
    subprocess.check_call(["make", "-s", "oracle/_build/star_amd_emul_cli"], cwd=ROOT)
    return EMUL_CLI


@pytest.mark.parametrize("name", ["annot"])
def test_front_end_generates_the_index_on_the_emulated_device(name, tmp_path, emul_cli):
    """`--runMode genomeGenerate` through the product's own device path -- HipBackend (hip_backend.h: sliced k_forEach launches, the sort / scan calls),
    index_gpu.hip, junction insertion of the annotation on the device -- with the kernels emulated: every genomeDir file equals the reference's"""
    import test_index_build
    test_index_build._run_case(emul_cli, str(tmp_path), name)


@pytest.mark.parametrize("more,n", [(["--twopassMode", "Basic"], 60), (["--twopassMode", "Basic", "--outFilterType", "BySJout", "--sjdbGTFfile", "GTF"], 40)])
def test_front_end_two_pass_with_resident_junction_insertion(more, n, tmp_path, emul_cli):
    """the whole binary on the emulated engine: 1st pass, junction insertion ON THE ARRAYS RESIDENT IN THE ENGINE CONTEXT (staramd_insert_junctions +
    staramd_update_tables behind cli_run.cpp's hook), 2nd pass -- every output file against one reference run with the same flags"""
    from test_cli_pipeline import run_cli_case
    run_cli_case(emul_cli, "pe101", more + ["--readMapNumber", str(n)], 40, tmp_path, env=SMALL)


def test_front_end_two_pass_when_the_device_has_no_room_for_resident_insertion(tmp_path, emul_cli):
    """staramd_insert_junctions_fits says no (STARAMD_SJDB_FITS_FREE_GB: a pretended amount of free device memory): the front end keeps its host copy of the suffix
    array, the junctions of the 1st pass are inserted through host buffers and the index is uploaded again -- same outputs as the reference's 2-pass run"""
    from test_cli_pipeline import run_cli_case
    run_cli_case(emul_cli, "pe101", ["--twopassMode", "Basic", "--readMapNumber", "60"], 40, tmp_path, env=dict(SMALL, STARAMD_SJDB_FITS_FREE_GB="0.001"))


@pytest.mark.parametrize("more", [[], ["--sjdbInsertSave", "All"]])
def test_front_end_two_pass_without_any_junction(more, tmp_path, emul_cli):
    """2-pass on an index without annotation whose 1st pass yields NO junction (filters nobody passes): nothing is inserted, the engine contexts keep the
    index they hold (the host copy of the suffix array is gone by then: a re-upload read a null pointer) and only get the new tables; with
    --sjdbInsertSave All the unchanged suffix array is written into _STARgenome as the reference does"""
    from test_cli_pipeline import run_cli_case
    none = ["--outSJfilterCountUniqueMin", "1000000", "1000000", "1000000", "1000000", "--outSJfilterCountTotalMin", "1000000", "1000000", "1000000", "1000000"]
    run_cli_case(emul_cli, "se50", ["--twopassMode", "Basic"] + none + more + ["--readMapNumber", "60"], 40, tmp_path, env=SMALL)


CHIM_FE = [("pe150_chim", ["--chimSegmentMin", "15", "--chimJunctionOverhangMin", "15"], 30),
           ("pe150_chim", ["--chimSegmentMin", "20", "--chimOutJunctionFormat", "1", "--chimScoreDropMax", "30", "--chimScoreSeparation", "5", "--chimSegmentReadGapMax", "3"], 30),
           ("se50", ["--chimSegmentMin", "12", "--chimJunctionOverhangMin", "12", "--chimFilter", "None"], 120),
           ("pe76_overlap", ["--chimSegmentMin", "10", "--chimJunctionOverhangMin", "10", "--chimMainSegmentMultNmax", "1"], 40),
           ("pe150_chim", ["--chimSegmentMin", "15", "--chimJunctionOverhangMin", "15", "--twopassMode", "Basic"], 30)]      # (the 1st pass runs without chimeric detection, twoPassRunPass1.cpp:24)


@pytest.mark.parametrize("name,more,n", CHIM_FE)
@pytest.mark.parametrize("on_device", ["1", "0", "noprune"])
def test_front_end_chimeric_detection_partner_chosen_on_the_device(name, more, n, on_device, tmp_path, emul_cli):
    """--chimSegmentMin > 0 through the shipped front end on the emulated engine: the partner loop of chimericDetectionOld runs in k_stitch_finish
    (staramd_params::resultSelect 2; STARAMD_CHIM_ON_DEVICE=0: on the host over every transcript of every window, as before) -- Chimeric.out.junction, SAM,
    SJ.out.tab against one reference run"""
    from test_cli_pipeline import run_cli_case
    # ("noprune": on the device, every window of every read stitched -- STARAMD_PRUNE without bit 3; default: only the reads whose best alignment can be the main segment of a chimera)
    env = dict(SMALL, STARAMD_CHIM_ON_DEVICE="1", STARAMD_PRUNE="7") if on_device == "noprune" else dict(SMALL, STARAMD_CHIM_ON_DEVICE=on_device)
    run_cli_case(emul_cli, name, more + ["--readMapNumber", str(n)], 25, tmp_path, env=env)


def test_front_end_two_contexts_on_one_device(tmp_path, emul_cli):
    """`--gpuDevices 0,0`: two engine contexts, two mapper threads (each OS thread runs its own emulated launches), batches emitted in input order;
    with the 1st-pass junctions inserted into BOTH contexts' resident arrays"""
    from test_cli_pipeline import run_cli_case
    run_cli_case(emul_cli, "pe101", ["--twopassMode", "Basic", "--readMapNumber", "60", "--gpuDevices", "0,0"], 12, tmp_path, env=SMALL)
