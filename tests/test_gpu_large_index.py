"""Large-coordinate parity inside the GPU suite: a 400 Mb synthetic genome with annotation (nSA = 8 * 10^8, 13-base SAindex,
strand bit 32, ~45 k junctions; index generated on the device by the product, byte-identical to the reference's genomeGenerate:
tests/test_index_build.py), 200 k pairs 2x101 through the shipped pipeline and through oracle/_ref/STAR with the same flags:
SAM bodies as a multiset, SJ.out.tab and the Log.final.out counters.  Until now this size class was only covered by bench.py:full_size_parity."""
import argparse
import os
import sys

import pytest

from util import refstar

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built")]


def test_400mb_index_200k_pairs_against_the_reference(tmp_path, built):
    sys.path.insert(0, ROOT)
    import shutil
    import tempfile
    work = tempfile.mkdtemp(prefix="staramd_t400_", dir="/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path))      # ~9 GB of genome, index, reads and SAM
    try:
        _run(work)
    finally:
        shutil.rmtree(work, ignore_errors=True)


def _run(work):
    import bench
    from star_amd.capi import run_cli
    args = argparse.Namespace(read_len=101, reads=100000, workdir=work)
    notes = []
    g, ginfo = bench.build_genome(args, 400, notes.append)
    assert ginfo["index_bytes"] > 4e9 and ginfo["junctions_in_index"] > 10000
    n = 200000
    fq = bench.make_reads(args, g, os.path.join(g, "t"), "reads", n, 4242)
    idx = os.path.join(g, "idx")
    new, ref = os.path.join(g, "t", "gpu_"), os.path.join(g, "t", "ref_")
    rc, rep = run_cli(["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", new, "--runThreadN", "16", "--gpuBatchReads", "100000"])
    assert rc == 0 and int(rep.reads) == n
    refstar.align(idx, fq, ref, threads=min(32, os.cpu_count() or 8), timeout=900)
    par = bench.full_size_parity(ref, new)
    assert par and par["sam_multiset_identical"] and par["sj_out_tab_identical"] and par["log_final_counters_identical"], par
    assert par["sam_records_star_amd"] > 1.5 * n          # (mapped pairs write two records)


def test_100mb_index_two_pass_against_the_reference(tmp_path, built):
    """config 4 pinned in the GPU suite at a size where coordinates and the junction count are not toy-sized: --twopassMode Basic on a 100 Mb index with
    annotation, 100 k pairs -- the 1st pass's junctions inserted into the index resident in HBM -- against the reference's own 2-pass run: SAM multiset, SJ.out.tab
    (column 6), the inserted junction list, Log.final.out counters (bench.py:two_pass_parity is the same at 400 Mb in every default bench run)"""
    sys.path.insert(0, ROOT)
    import shutil
    import tempfile
    import bench
    work = tempfile.mkdtemp(prefix="staramd_t100_", dir="/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path))
    try:
        args = argparse.Namespace(read_len=101, reads=50000, workdir=work)
        notes = []
        orig = bench.build_genome
        par = None
        # the leg itself, on a 100 Mb genome
        bench.build_genome = lambda a, mb, log: orig(a, 100, log)
        try:
            par = bench.two_pass_parity(args, notes.append)
        finally:
            bench.build_genome = orig
        assert par and "error" not in par, par
        assert par["sam_multiset_identical"] and par["sj_out_tab_identical"] and par["log_final_counters_identical"] and par["inserted_junction_list_identical"], par
        assert par["junctions_inserted_by_pass1"] > 100, par
    finally:
        shutil.rmtree(work, ignore_errors=True)
