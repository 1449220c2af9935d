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
         ("pe101_sparse3", ["--chimSegmentMin", "15", "--runThreadN", "3"]),
         ("pe76_overlap", ["--chimSegmentMin", "10", "--chimJunctionOverhangMin", "10", "--chimMainSegmentMultNmax", "1"])]

MULT = [("pe150_chim", ["--chimSegmentMin", "15", "--chimJunctionOverhangMin", "15", "--chimMultimapNmax", "10"]),
        ("pe150_chim", ["--chimSegmentMin", "12", "--chimJunctionOverhangMin", "8", "--chimOutJunctionFormat", "1", "--chimMultimapNmax", "20", "--chimMultimapScoreRange", "3",
                        "--chimScoreJunctionNonGTAG", "-4", "--chimNonchimScoreDropMin", "10", "--alignSJstitchMismatchNmax", "5", "-1", "5", "5", "--outSAMattrRGline", "ID:GRPundef",
                        "--alignInsertionFlush", "Right", "--alignSplicedMateMapLminOverLmate", "0", "--alignSplicedMateMapLmin", "30"]),      # STAR-Fusion's options (without mate merging)
        ("se50", ["--chimSegmentMin", "12", "--chimJunctionOverhangMin", "12", "--chimMultimapNmax", "5", "--chimFilter", "None", "--chimScoreDropMax", "30"]),
        ("pe101", ["--chimSegmentMin", "12", "--chimMultimapNmax", "2", "--chimMultimapScoreRange", "0", "--outFilterType", "BySJout", "--twopassMode", "Basic"]),
        ("pe76_overlap", ["--chimSegmentMin", "10", "--chimJunctionOverhangMin", "10", "--chimMultimapNmax", "50", "--chimScoreMin", "1", "--chimScoreSeparation", "1",
                          "--chimScoreJunctionNonGTAG", "0", "--chimSegmentReadGapMax", "3", "--runThreadN", "3"])]                     # Arriba's options (without mate merging)


@pytest.mark.parametrize("name,more", MULT)
def test_chimeric_multimapping_oracle(name, more, tmp_path, built):
    """--chimMultimapNmax > 0: every pair of recorded alignments is a candidate, the junction is re-located and both sides re-scored; all chimeras
    within --chimMultimapScoreRange of the best are reported (table with a header line and six more columns)"""
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    info["extra"] = list(info["extra"]) + more
    _compare(info, os.path.dirname(info["fastq"][0]), lambda g, p: oracle_lib.Oracle(g, p))


@pytest.mark.parametrize("tag", ["pe", "se"])
def test_chimeric_multimapping_stress_oracle(tag, tmp_path, built):
    kw, flags = STRESS[tag]
    d = str(tmp_path / tag)
    info = synth.make_dataset(d, **kw)
    info["idx"] = os.path.join(d, "idx")
    refstar.genome_generate(info["fasta"], info["idx"], gtf=info["gtf"], sa_index_nbases=8, sjdb_overhang=100)
    info["extra"] = flags + ["--chimMultimapNmax", "10", "--chimMultimapScoreRange", "2", "--chimNonchimScoreDropMin", "15"]
    _compare(info, d, lambda g, p: oracle_lib.Oracle(g, p), min_lines=500)


WITHIN = [("pe150_chim", ["--chimSegmentMin", "15", "--chimJunctionOverhangMin", "15", "--chimOutType", "WithinBAM"], "Unsorted"),
          ("pe150_chim", ["--chimSegmentMin", "15", "--chimJunctionOverhangMin", "15", "--chimOutType", "Junctions", "WithinBAM", "SoftClip", "--outSAMattributes", "NH", "HI", "AS", "nM", "ch", "MC"], "SortedByCoordinate"),
          ("pe150_chim", ["--chimSegmentMin", "12", "--chimJunctionOverhangMin", "8", "--chimMultimapNmax", "20", "--chimMultimapScoreRange", "3", "--chimNonchimScoreDropMin", "10",
                          "--chimOutType", "WithinBAM", "Junctions", "--outSAMunmapped", "Within"], "Unsorted"),
          ("se50", ["--chimSegmentMin", "12", "--chimJunctionOverhangMin", "12", "--chimOutType", "WithinBAM", "HardClip", "--outSAMattributes", "All"], "Unsorted"),
          ("se50", ["--chimSegmentMin", "12", "--chimJunctionOverhangMin", "12", "--chimMultimapNmax", "5", "--chimOutType", "WithinBAM", "SoftClip", "--chimScoreDropMax", "30"], "SortedByCoordinate"),
          ("pe76_overlap", ["--chimSegmentMin", "10", "--chimJunctionOverhangMin", "10", "--chimOutType", "WithinBAM", "--outFilterType", "BySJout", "--quantMode", "GeneCounts"], "Unsorted")]


def _within(info, d, kind, min_suppl):
    from util import bam_parts
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refW_"), threads=1, extra=info["extra"])
    new = run_with_engine(info, os.path.join(d, "newW_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=800)
    f = "Aligned.out.bam" if kind == "Unsorted" else "Aligned.sortedByCoord.out.bam"
    (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + f), bam_parts(new + f)
    assert ra == rb
    assert len(rr) == len(nr)
    for x, y in zip(rr, nr):
        assert x == y
    flag = lambda rec: int.from_bytes(rec[18:20], "little")
    n_chim = [int(l.split("|")[1]) for l in open(ref + "Log.final.out") if "Number of chimeric reads" in l][0]
    print("chimeric reads %d, supplementary records %d, records with SA %d" % (n_chim, sum(1 for x in rr if flag(x) & 0x800), sum(1 for x in rr if b"SAZ" in x)))
    assert n_chim >= min_suppl
    assert open(ref + "SJ.out.tab", "rb").read() == open(new + "SJ.out.tab", "rb").read()
    assert refstar.final_log_counters(ref + "Log.final.out") == refstar.final_log_counters(new + "Log.final.out")
    if "Junctions" in info["extra"]:
        lines = lambda p: [l for l in open(p + "Chimeric.out.junction") if not l.startswith("# 2.7.11b")]
        assert lines(ref) == lines(new)
    else:
        assert not os.path.exists(new + "Chimeric.out.junction")
    if "GeneCounts" in info["extra"]:
        assert open(ref + "ReadsPerGene.out.tab", "rb").read() == open(new + "ReadsPerGene.out.tab", "rb").read()


@pytest.mark.parametrize("name,more,kind", WITHIN)
def test_chimeric_within_bam(name, more, kind, tmp_path, built):
    """--chimOutType WithinBAM: the chimera replaces the read's linear alignments in the BAM: representative + supplementary records (hard / soft clips,
    0x800, mate fields of the other segment, SA tags both ways); nothing else is output or counted for such a read"""
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    info["extra"] = [x for x in info["extra"]] + more + ["--outSAMtype", "BAM", kind]
    if "--outSAMattributes" in more:          # the data set's own attribute list goes
        k = info["extra"].index("--outSAMattributes")
        if k < len(info["extra"]) - len(more) - 3:
            j = k + 1
            while j < len(info["extra"]) and not info["extra"][j].startswith("--"):
                j += 1
            del info["extra"][k:j]
    _within(info, os.path.dirname(info["fastq"][0]), kind, 1)


@pytest.mark.parametrize("tag", ["pe", "se"])
@pytest.mark.parametrize("mult", [False, True])
def test_chimeric_within_bam_stress(tag, mult, tmp_path, built):
    info, d = _stress(tag, tmp_path)
    info["extra"] = list(info["extra"]) + ["--chimOutType", "WithinBAM", "Junctions", "--outSAMtype", "BAM", "Unsorted", "SortedByCoordinate"] + \
        (["--chimMultimapNmax", "10", "--chimMultimapScoreRange", "2", "--chimNonchimScoreDropMin", "15"] if mult else [])
    _within(info, d, "Unsorted", 300)
    from util import bam_parts
    assert bam_parts(d + "/refW_Aligned.sortedByCoord.out.bam")[1:] == bam_parts(d + "/newW_Aligned.sortedByCoord.out.bam")[1:]


@pytest.mark.parametrize("tag", ["pe", "se"])
def test_chimeric_separate_sam_old(tag, tmp_path, built):
    """--chimOutType SeparateSAMold: Chimeric.out.sam, the two segments as SAM records that point at each other; the linear alignments stay in Aligned.out.sam"""
    info, d = _stress(tag, tmp_path)
    info["extra"] = list(info["extra"]) + ["--chimOutType", "SeparateSAMold", "Junctions", "--outSAMattributes", "NH", "HI", "AS", "nM", "NM", "MD"]
    _compare(info, d, lambda g, p: oracle_lib.Oracle(g, p), min_lines=400)
    body = lambda p: [l for l in open(p, "rb") if not l.startswith(b"@")]
    a, b = body(d + "/refC_Chimeric.out.sam"), body(d + "/newC_Chimeric.out.sam")
    assert a == b and len(a) > 800


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


@pytest.mark.parametrize("more,want", [(["--chimSegmentMin", "15"], 1), (["--chimSegmentMin", "15", "--twopassMode", "Basic"], 1),
                                        (["--chimSegmentMin", "15", "--chimMultimapNmax", "10"], 0), ([], 0),
                                        (["--chimSegmentMin", "15", "--peOverlapNbasesMin", "10", "--chimOutType", "WithinBAM", "--outSAMtype", "BAM", "Unsorted"], 0)])
def test_which_runs_ask_the_engine_for_the_partner(more, want, tmp_path, built):
    """sah_chim_select_on_device (include/star_amd_host.h): staramd_params::resultSelect 2 for --chimSegmentMin > 0 with the default algorithm and no mate merging --
    also when a 1st pass comes first (it runs without chimeric detection, the 2nd pass needs the choice made) -- and for nothing else"""
    import ctypes as C
    info = prepare("pe150_chim", str(tmp_path), need_ref=False)
    run = capi.HostRun(["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "sel_")] + list(info["extra"]) + more)
    try:
        L = run.L
        L.sah_chim_select_on_device.restype = C.c_int; L.sah_chim_select_on_device.argtypes = [C.c_void_p]
        before = run.params.contents.resultSelect
        assert L.sah_chim_select_on_device(run.h) == want
        assert run.params.contents.resultSelect == (2 if want else before)
        assert run.params.contents.chimSegmentMin == (15 if "--chimSegmentMin" in more else 0)
    finally:
        run.close()
