"""Chimeric detection (--chimSegmentMin > 0, default algorithm --chimMultimapNmax 0, --chimOutType Junctions): Chimeric.out.junction, the
"Number of chimeric reads" counter and every other output against the reference.  Host post-map code over the transcripts of ALL windows,
which the hot path returns for it (resultSelect 0 + the chimSegmentMin clause of stitchWindowAligns.cpp:247)."""
import os

import pytest

from util import capi, compare_outputs, oracle_lib, prepare, refstar, run_with_engine
from star_amd import synth

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

CASES = [("pe150_chim", ["--chimSegmentMin", "15", "--chimJunctionOverhangMin", "15"]),
         ("pe150_chim", ["--chimSegmentMin", "20", "--chimOutJunctionFormat", "1", "--chimScoreDropMax", "30", "--chimScoreSeparation", "5", "--chimSegmentReadGapMax", "3",
                         "--twopassMode", "Basic"]),
         ("pe101", ["--chimSegmentMin", "12", "--chimJunctionOverhangMin", "12", "--outFilterType", "BySJout"]),
         ("se50", ["--chimSegmentMin", "12", "--chimJunctionOverhangMin", "12", "--chimFilter", "None"]),
         ("pe150_indel", ["--chimSegmentMin", "15", "--runThreadN", "3"]),
         ("pe76_overlap", ["--chimSegmentMin", "10", "--chimJunctionOverhangMin", "10", "--chimMainSegmentMultNmax", "1"])]

# data sets made of chimeras: mates from different loci (junction type -1) and reads whose halves come from different loci (types 0 / 1 / 2)
STRESS = {"pe": (dict(seed=9, chr_lengths=(300000, 250000, 200000), n_tr=100, n_reads=3000, read_len=125, paired=True, sub_rate=0.005, chim_rate=0.6),
                 ["--chimSegmentMin", "12", "--chimJunctionOverhangMin", "10", "--chimScoreDropMax", "80", "--chimScoreSeparation", "1", "--chimSegmentReadGapMax", "5"]),
          "se": (dict(seed=10, chr_lengths=(300000, 250000), n_tr=80, n_reads=3000, read_len=100, paired=False, sub_rate=0.005, chim_rate=0.6),
                 ["--chimSegmentMin", "15", "--chimJunctionOverhangMin", "12", "--chimScoreDropMax", "60", "--chimScoreSeparation", "3", "--chimOutJunctionFormat", "1"])}


def _compare(info, d, factory, min_lines=0):
    rf = list(info["extra"])
    if "--runThreadN" in rf:
        k = rf.index("--runThreadN"); del rf[k:k + 2]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refC_"), threads=1, extra=rf)
    new = run_with_engine(info, os.path.join(d, "newC_"), factory, batch_reads=800)
    problems = compare_outputs(ref, new)
    lines = lambda p: [l for l in open(p + "Chimeric.out.junction") if not l.startswith("# 2.7.11b")]       # (that comment line holds the command line)
    a, b = lines(ref), lines(new)
    if a != b:
        problems.append("Chimeric.out.junction differs: %d vs %d lines" % (len(a), len(b)))
    assert not problems, problems
    assert len(a) >= min_lines


@pytest.mark.parametrize("name,more", CASES)
def test_chimeric_oracle(name, more, tmp_path, built):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    info["extra"] = list(info["extra"]) + more
    _compare(info, os.path.dirname(info["fastq"][0]), lambda g, p: oracle_lib.Oracle(g, p))


def _stress(tag, tmp_path):
    kw, more = STRESS[tag]
    d = os.path.join(str(tmp_path), tag)
    info = synth.make_dataset(d, **kw)
    refstar.genome_generate(info["fasta"], d + "/idx", gtf=info["gtf"], sa_index_nbases=8, sjdb_overhang=kw["read_len"] - 1)
    info["idx"] = d + "/idx"; info["extra"] = more
    return info, d


@pytest.mark.parametrize("tag", sorted(STRESS))
def test_chimeric_stress_oracle(tag, tmp_path, built):
    info, d = _stress(tag, tmp_path)
    _compare(info, d, lambda g, p: oracle_lib.Oracle(g, p), min_lines=400)


@pytest.mark.gpu
def test_chimeric_stress_engine(tmp_path, built):
    """the device side of chimeric mode: every transcript of every window returned, chimSegmentMin clause on -- buffers against the oracle,
    then the whole run against the reference"""
    from test_gpu_parity import _compare_buffers
    info, d = _stress("pe", tmp_path)
    _compare_buffers(info, [], os.path.join(d, "x_"))
    _compare(info, d, lambda g, p: capi.Engine(g, p, device=0, max_reads=4096), min_lines=400)
