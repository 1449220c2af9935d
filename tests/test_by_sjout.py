"""--outFilterType BySJout (ENCODE's standard option): stage 1 maps everything, holds the reads that have an unannotated junction and
collects the junctions of ALL reads; the unannotated junctions that pass the SJ filters become a whitelist; stage 2 maps the held
reads again and the stitcher drops transcripts with unannotated junctions that are not on the list (stitchWindowAligns.cpp:169-177;
on the device: staramd_set_novel_junctions).  The reference run with the same flags is the truth."""
import os

import pytest

from util import capi, compare_outputs, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

CASES = {
    "plain": ["--outFilterType", "BySJout"],
    "two_pass": ["--outFilterType", "BySJout", "--twopassMode", "Basic"],
    "encode": ["--outFilterType", "BySJout", "--outFilterMultimapNmax", "20", "--alignSJoverhangMin", "8", "--alignSJDBoverhangMin", "1",
               "--outFilterMismatchNmax", "999", "--outFilterMismatchNoverReadLmax", "0.04", "--alignIntronMin", "20", "--alignIntronMax", "1000000",
               "--alignMatesGapMax", "1000000", "--sjdbScore", "1", "--outSAMattributes", "NH", "HI", "AS", "NM", "MD"],
    "threads": ["--outFilterType", "BySJout", "--runThreadN", "3"],
}


def _case(name, flags, tmp_path, factory):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    info["extra"] = flags if name == "pe76_overlap" and "--alignSJoverhangMin" in flags else list(info["extra"]) + flags
    d = os.path.dirname(info["fastq"][0])
    ref_flags = [f for f in info["extra"]]
    if "--runThreadN" in ref_flags:                   # refstar.align passes its own --runThreadN
        k = ref_flags.index("--runThreadN"); del ref_flags[k:k + 2]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "ref_"), threads=1, extra=ref_flags)
    new = run_with_engine(info, os.path.join(d, "new_"), factory, batch_reads=1000)
    assert not compare_outputs(ref, new)


@pytest.mark.parametrize("name,case", [("pe101", "plain"), ("pe101", "two_pass"), ("pe101", "encode"), ("pe101", "threads"), ("se50", "plain"),
                                       ("pe150_indel", "plain"), ("pe150_indel", "encode"), ("pe76_overlap", "plain")])
def test_by_sjout_oracle(name, case, tmp_path, built):
    _case(name, CASES[case], tmp_path, lambda g, p: oracle_lib.Oracle(g, p))


@pytest.mark.gpu
@pytest.mark.parametrize("name,case", [("pe101", "plain"), ("pe101", "two_pass"), ("pe150_indel", "encode"), ("se50", "plain")])
def test_by_sjout_engine(name, case, tmp_path, built):
    _case(name, CASES[case], tmp_path, lambda g, p: capi.Engine(g, p, device=0, max_reads=4096))


@pytest.mark.gpu
def test_by_sjout_cli(tmp_path, built):
    """the command line itself: stage 1, whitelist upload, stage 2 over the held reads kept in memory"""
    import subprocess
    from util import ROOT
    subprocess.check_call(["make", "-s", "cli"], cwd=ROOT)
    info = prepare("pe101", str(tmp_path), need_ref=False)
    d = os.path.dirname(info["fastq"][0])
    flags = ["--outFilterType", "BySJout", "--twopassMode", "Basic"]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "ref_"), threads=1, extra=flags)
    prefix = os.path.join(d, "cli_")
    subprocess.check_call([os.path.join(ROOT, "star_amd", "bin", "star_amd"), "--runMode", "alignReads", "--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] +
                          ["--outFileNamePrefix", prefix, "--runThreadN", "4", "--gpuBatchReads", "700"] + flags)
    assert not compare_outputs(ref, prefix)
