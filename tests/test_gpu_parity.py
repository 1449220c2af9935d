"""GPU parity tests (the parity tests proper): the HIP engine is called through the C ABI
(include/star_amd.h) and must reproduce
  (a) the oracle's result buffers byte for byte (read results, transcripts, exons), and
  (b) the reference's Aligned.out.sam / SJ.out.tab / Log.final.out counters byte for byte.
Bit-exact bar: integer / index work only, no tolerance."""
import ctypes as C

import pytest

from util import DATASETS, PARAM_SWEEP, capi, compare_outputs, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.gpu


def _engine(g, p):
    return capi.Engine(g, p, device=0, max_reads=4096)


@pytest.mark.parametrize("name", sorted(DATASETS))
def test_engine_matches_reference_outputs(name, tmp_path, built):
    if not refstar.have_ref():
        pytest.skip("oracle/_ref/STAR missing")
    info = prepare(name, str(tmp_path))
    new = run_with_engine(info, str(tmp_path / name / "gpu_"), _engine)
    problems = compare_outputs(info["ref_prefix"], new)
    assert not problems, problems


@pytest.mark.parametrize("name,select", [(n, "Selected") for n in sorted(DATASETS)] + [("pe101", "All"), ("pe150_indel", "All")])
def test_engine_matches_oracle_buffers(name, select, tmp_path, built):
    """select = Selected: only the transcripts multMapSelect can pick are returned (default); All: the whole trAll[][]."""
    if not refstar.have_ref():
        pytest.skip("oracle/_ref/STAR missing (needed to build the index)")
    info = prepare(name, str(tmp_path), need_ref=False)
    _compare_buffers(info, ["--gpuResultSelect", select], str(tmp_path / "x_"))


@pytest.mark.parametrize("name", ["pe101", "pe101_sparse3"])
def test_seed_search_lmax_prefix_codes(name, tmp_path, built):
    """--seedSearchLmax: the backward search is given Shift + 1 bases (ReadAlign_mapOneRead.cpp:81-86) and runs over the start of its piece into an N, the
    mate spacer or the other mate; the reference adds those codes into the L-mer prefix as they are, carries and borrows included
    (ReadAlign_maxMappableLength2strands.cpp:23-37).  Found by the emulated fuzzer (tests/tools/fuzz_engine.py, FUZZ_EMUL=1)."""
    if not refstar.have_ref():
        pytest.skip("oracle/_ref/STAR missing (needed to build the index)")
    info = prepare(name, str(tmp_path), need_ref=False)
    _compare_buffers(info, ["--gpuResultSelect", "All", "--seedSearchLmax", "30", "--seedSearchStartLmax", "12", "--readMapNumber", "300"], str(tmp_path / "x_"))    # (the reads the emulator was run on)


@pytest.mark.parametrize("mode", ["All", "Selected"])
def test_negative_mate_extension_length(mode, tmp_path, built):
    """--alignEndsProtrude + --clip5pNbases: the second mate starts before the first exon of the transcript, the extension length of the mate-gap stitch
    (stitchAlignToTranscript.cpp:390) wraps around and the reference's `(int) L` loop bound (extendAlign.cpp:59) makes it no extension at all.
    Found by tests/tools/fuzz_engine.py on hardware in round 3 (1 of 150 combinations)."""
    if not refstar.have_ref():
        pytest.skip("oracle/_ref/STAR missing (needed to build the index)")
    info = prepare("pe125_protrude", str(tmp_path), need_ref=False)
    _compare_buffers(info, ["--gpuResultSelect", mode, "--sjdbScore", "0", "--outFilterMismatchNmax", "3", "--alignEndsProtrude", "15", "ConcordantPair", "--clip5pNbases", "20", "20"], str(tmp_path / "x_"))


def _compare_buffers(info, more, prefix):
    argv = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", prefix] + info["extra"] + more
    run = capi.HostRun(argv)
    eng = _engine(run.genome, run.params)
    orc = oracle_lib.Oracle(run.genome, run.params)
    try:
        nb = 0
        while True:
            b = run.next_batch(1500)
            if b is None:
                break
            n = b.nReads
            bg = capi.ResultBuffers(n, tr_cap=n * 400)     # All mode + EndToEnd / chimeric mode record ~80+ transcripts per read
            bo = capi.ResultBuffers(n, tr_cap=n * 400)
            eng.map_batch(b, bg)
            orc.map_batch(b, bo)
            rg, tg, eg = bg.as_bytes(n)
            ro, to, eo = bo.as_bytes(n)
            assert bg.res.trCount == bo.res.trCount and bg.res.exCount == bo.res.exCount
            if rg != ro:
                for i in range(n):
                    a, o = bg.reads[i], bo.reads[i]
                    fa = (a.status, a.nW, a.nTr, a.trOffset, a.trBest, a.maxScoreMate[0], a.maxScoreMate[1], a.unmappedLength)
                    fo = (o.status, o.nW, o.nTr, o.trOffset, o.trBest, o.maxScoreMate[0], o.maxScoreMate[1], o.unmappedLength)
                    # (resultSelect 1 returns maxScoreMate[] as 0 -- engine and oracle alike, include/star_amd.h; resultSelect 0 the reference's values)
                    assert fa == fo, "read %d of batch %d: gpu %r oracle %r" % (i, nb, fa, fo)
            assert tg == to, "transcript records differ in batch %d" % nb
            assert eg == eo, "exon records differ in batch %d" % nb
            nb += 1
        assert nb > 0
    finally:
        eng.close(); orc.close(); run.close()


def test_result_arrays_too_small_then_copied_without_a_second_run(tmp_path, built):
    """STARAMD_ERR_RESULT_OVERFLOW (include/star_amd.h): the call says what the batch needs, its results stay resident, and the same batch handed in again with
    larger arrays gets them copied -- the same bytes as a call that had room at once, and no kernel runs a second time (the engine's launch counter does not move).
    A DIFFERENT batch in the same host arrays is mapped, not answered from what is resident."""
    import ctypes as C
    if not refstar.have_ref():
        pytest.skip("oracle/_ref/STAR missing (needed to build the index)")
    info = prepare("pe101", str(tmp_path), need_ref=False)
    argv = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "ro_"), "--gpuResultSelect", "All"] + info["extra"]
    run = capi.HostRun(argv)
    eng = _engine(run.genome, run.params)
    try:
        b = run.next_batch(1200)
        n = b.nReads
        full = capi.ResultBuffers(n, tr_cap=n * 400)
        eng.map_batch(b, full)
        want = full.as_bytes(n)
        assert full.res.trCount > 64
        launches0 = eng.L.staramd_launch_count(eng.ctx)
        small = capi.ResultBuffers(n, tr_cap=64, ex_cap=64)
        rc = eng.L.staramd_map_batch(eng.ctx, C.byref(b), C.byref(small.res))
        assert rc == -3 and small.res.trCount == full.res.trCount and small.res.exCount == full.res.exCount
        launches1 = eng.L.staramd_launch_count(eng.ctx)
        assert launches1 == launches0 + 1
        big = capi.ResultBuffers(n, tr_cap=n * 400)
        rc = eng.L.staramd_map_batch(eng.ctx, C.byref(b), C.byref(big.res))
        assert rc == 0 and big.as_bytes(n) == want
        assert eng.L.staramd_launch_count(eng.ctx) == launches1, "the batch was mapped a second time"
        # the overflow again, then another batch: mapped (the resident results are not handed out for it)
        rc = eng.L.staramd_map_batch(eng.ctx, C.byref(b), C.byref(small.res))
        assert rc == -3
        b2 = run.next_batch(1200)
        if b2 is not None:
            n2 = b2.nReads
            r2 = capi.ResultBuffers(n2, tr_cap=n2 * 400)
            before = eng.L.staramd_launch_count(eng.ctx)
            eng.map_batch(b2, r2)
            assert eng.L.staramd_launch_count(eng.ctx) == before + 1
    finally:
        eng.close(); run.close()


@pytest.fixture(scope="module")
def sweep_data(tmp_path_factory, built):
    if not refstar.have_ref():
        pytest.skip("oracle/_ref/STAR missing (needed to build the index)")
    return prepare("pe150_indel", str(tmp_path_factory.mktemp("sweep")), need_ref=False)


@pytest.mark.parametrize("combo", sorted(PARAM_SWEEP))
def test_engine_matches_oracle_with_flags(combo, sweep_data, tmp_path, built):
    """non-default flags (util.PARAM_SWEEP; the oracle is pinned against the reference with the same flags in
    test_oracle_vs_reference.py): result buffers byte for byte, all recorded transcripts"""
    _compare_buffers(sweep_data, PARAM_SWEEP[combo] + ["--gpuResultSelect", "All"], str(tmp_path / "x_"))


@pytest.mark.parametrize("name", ["pe101", "pe150_indel"])
def test_shadow_validation(name, tmp_path, built):
    """Shadow-validation build: every wave-cooperative stitch / extend call is re-run on the GPU through the scalar
    restatement and compared; zero disagreements over all calls of the data set."""
    import json, os, subprocess, sys
    if not refstar.have_ref():
        pytest.skip("oracle/_ref/STAR missing (needed to build the index)")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, STARAMD_ENGINE_LIB="shadow")
    out = subprocess.check_output([sys.executable, os.path.join(here, "shadow_run.py"), name, str(tmp_path)], env=env)
    c = json.loads(out.decode().strip().splitlines()[-1])
    assert c["stitchN"] > 1000 and c["extendN"] > 1000, c
    assert c["stitchBad"] == 0 and c["extendBad"] == 0, c


# Rarely taken paths of the engine, forced by shrinking its work space through the STARAMD_* knobs:
#   tiny pools            -> bump allocators overflow, the host doubles the pool and re-runs the batch
#   2 windows / 2 blocks  -> every read with more windows goes through the middle k_windows launch (larger table in LDS);
#                            with a 3-window middle table, on to the last launch (table in global memory); with the
#                            middle launch off, straight to the last launch
#   256-byte record arena -> windows re-walked with the record arena in HBM
#   tiny candidate logs   -> maxScoreMate-sensitive windows re-walked (k_stitch_win mode 1) instead of replayed
#   every read heavy / every read light -> both kinds of stitch work items
FORCED = {
    "tiny_pools": {"STARAMD_POOL_SLACK": "64", "STARAMD_SEEDS_PER_READ": "1", "STARAMD_WINDOWS_PER_READ": "1", "STARAMD_WA_PER_READ": "1", "STARAMD_TR_PER_READ": "1"},
    "window_overflow": {"STARAMD_CAP_WINDOWS": "2", "STARAMD_CAP_WA_BLOCKS": "2"},
    "window_overflow_twice": {"STARAMD_CAP_WINDOWS": "1", "STARAMD_CAP_WA_BLOCKS": "1", "STARAMD_CAP_WINDOWS_MID": "3", "STARAMD_CAP_WA_BLOCKS_MID": "3"},
    "window_overflow_no_middle": {"STARAMD_CAP_WINDOWS": "2", "STARAMD_CAP_WA_BLOCKS": "2", "STARAMD_CAP_WINDOWS_MID": "0"},
    "arena_overflow": {"STARAMD_STITCH_ARENA": "256"},
    "log_overflow": {"STARAMD_CAND_KB_PER_WAVE": "1"},
    "all_heavy": {"STARAMD_LIGHT_EST": "0"},
    "all_light": {"STARAMD_LIGHT_EST": "4000000000"},
    "block_overflow": {"STARAMD_CAP_WA_BLOCKS": "2", "STARAMD_CAP_WA_BLOCKS_MID": "3"},      # seed-list blocks run out before table rows do (first and middle launch)
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
    import os, subprocess, sys
    if not refstar.have_ref():
        pytest.skip("oracle/_ref/STAR missing")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ); env.update(FORCED[case])
    p = subprocess.run([sys.executable, os.path.join(here, "engine_run.py"), "pe150_indel", str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    ok = p.returncode == 0 and (p.stdout.decode().strip().splitlines() or [""])[-1] == "OK"
    msg = (p.stdout.decode() + p.stderr.decode())[-2000:]
    assert ok, msg
