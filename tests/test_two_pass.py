"""2-pass mapping (--twopassMode Basic) and junction insertion at the mapping stage (--sjdbFileChrStartEnd): SURVEY.md 8d config 4.

The host restatement of sjdbPrepare / sjdbBuildIndex (star_amd/csrc/host/sjdb_insert.cpp) must leave the SAME index as the
reference: with --sjdbInsertSave All both write _STARgenome/{Genome,SA,SAindex,sjdbInfo.txt,sjdbList.out.tab}, compared byte
for byte; then the final Aligned.out.sam / SJ.out.tab / Log.final.out and the 1st-pass SJ.out.tab / Log.final.out must match too.
CPU tests drive the passes with the oracle, GPU tests with the HIP engine (index replaced through staramd_update_index)."""
import os

import pytest

from util import capi, compare_outputs, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

INDEX_FILES = ["sjdbInfo.txt", "sjdbList.out.tab", "Genome", "SA", "SAindex"]


def _junction_list(sj_out_tab, path):
    """A --sjdbFileChrStartEnd file from columns 1-4 of an SJ.out.tab, plus the adversarial cases of sjdbPrepare's
    collapsing: the same intron on the other strand, with undefined strand, shifted copies inside the repeat."""
    with open(path, "w") as o:
        for k, l in enumerate(open(sj_out_tab)):
            c = l.split("\t")
            st = c[3] if k % 2 else {"0": ".", "1": "+", "2": "-"}[c[3]]
            o.write("\t".join([c[0], c[1], c[2], st]) + "\n")
            if k % 5 == 0:
                o.write("\t".join([c[0], c[1], c[2], "-" if st in ("+", "1") else "+"]) + "\n")
            if k % 7 == 0:
                o.write("\t".join([c[0], str(int(c[1]) + 1), str(int(c[2]) + 1), "."]) + "\n")
    return path


def _check(info, tag, factory, tmp, two_pass):
    d = os.path.dirname(info["fastq"][0])
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "ref_%s_" % tag), threads=1, extra=info["extra"])
    new = run_with_engine(info, os.path.join(d, "new_%s_" % tag), factory)
    problems = compare_outputs(ref, new)
    for f in INDEX_FILES:
        if open(ref + "_STARgenome/" + f, "rb").read() != open(new + "_STARgenome/" + f, "rb").read():
            problems.append("_STARgenome/%s differs" % f)
    if two_pass:
        if open(ref + "_STARpass1/SJ.out.tab", "rb").read() != open(new + "_STARpass1/SJ.out.tab", "rb").read():
            problems.append("_STARpass1/SJ.out.tab differs")
        if refstar.final_log_counters(ref + "_STARpass1/Log.final.out") != refstar.final_log_counters(new + "_STARpass1/Log.final.out"):
            problems.append("_STARpass1/Log.final.out counters differ")
    assert not problems, problems


def _oracle(g, p):
    return oracle_lib.Oracle(g, p)


def _engine(g, p):
    return capi.Engine(g, p, device=0, max_reads=4096)


def _two_pass_case(name, tmp_path, factory):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    info["extra"] = list(info["extra"]) + ["--twopassMode", "Basic", "--sjdbInsertSave", "All"]
    _check(info, "2p", factory, tmp_path, True)


def _file_insert_case(name, tmp_path, factory, more):
    info = dict(prepare(name, str(tmp_path), need_ref=True))
    sjf = _junction_list(info["ref_prefix"] + "SJ.out.tab", os.path.join(os.path.dirname(info["fastq"][0]), "sjlist.tab"))
    info["extra"] = list(info["extra"]) + ["--sjdbFileChrStartEnd", sjf, "--sjdbInsertSave", "All"] + more
    _check(info, "sjf", factory, tmp_path, "--twopassMode" in more)


@pytest.mark.parametrize("name", ["pe101", "se50", "pe150_indel", "pe76_overlap"])
def test_two_pass(name, tmp_path, built):
    """genomes generated WITH a GTF (old junctions keep their place, novel ones of the 1st pass are inserted) and without"""
    _two_pass_case(name, tmp_path, _oracle)


def test_file_insertion_then_two_pass(tmp_path, built):
    """both insertions in one run: junction file before the 1st pass (limited to 1000 reads), 1st-pass junctions before the 2nd"""
    _file_insert_case("pe101", tmp_path, _oracle, ["--twopassMode", "Basic", "--twopass1readsN", "1000"])


def test_file_insertion_into_plain_genome(tmp_path, built):
    """genome generated WITHOUT junctions (sjdbOverhang defaults to 100): everything is new"""
    _file_insert_case("se50", tmp_path, _oracle, [])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pe101", "se50", "pe150_indel"])
def test_two_pass_on_the_engine(name, tmp_path, built):
    _two_pass_case(name, tmp_path, _engine)


@pytest.mark.gpu
def test_file_insertion_then_two_pass_on_the_engine(tmp_path, built):
    """both insertions in one run: junction file before the 1st pass (limited to 1000 reads), 1st-pass junctions before the 2nd"""
    _file_insert_case("pe101", tmp_path, _engine, ["--twopassMode", "Basic", "--twopass1readsN", "1000"])


def test_two_pass_parameter_errors(tmp_path, built):
    info = prepare("se50", str(tmp_path), need_ref=False)
    base = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "e_")]
    for extra, text in [(["--twopass1readsN", "10"], "--twopass1readsN is defined, but --twoPassMode is not defined"),
                        (["--twopassMode", "Basic", "--twopass1readsN", "0"], "--twopass1readsN = 0 in the 2-pass mode"),
                        (["--twopassMode", "Fancy"], "unrecognized value of --twopassMode=Fancy"),
                        (["--twopassMode", "Basic", "--twopassMode", "Basic"], "duplicate parameter \"twopassMode\""),
                        (["--sjdbFileChrStartEnd", str(tmp_path / "missing.tab")], "could not open input file pGe.sjdbFileChrStartEnd")]:
        with pytest.raises(RuntimeError) as e:
            capi.HostRun(base + extra)
        assert text in str(e.value)


def test_compressed_input_through_read_command(tmp_path, built):
    """--readFilesCommand zcat / gunzip -c: the text comes from a pipe; the 2nd pass runs the command again"""
    import subprocess
    info = dict(prepare("pe101", str(tmp_path), need_ref=False))
    gz = []
    for f in info["fastq"]:
        subprocess.check_call("gzip -c '%s' > '%s.gz'" % (f, f), shell=True)
        gz.append(f + ".gz")
    info["fastq"] = gz
    d = os.path.dirname(gz[0])
    for k, cmd in enumerate((["zcat"], ["gunzip", "-c"])):
        info["extra"] = ["--readFilesCommand"] + cmd + ["--twopassMode", "Basic"]
        ref = refstar.align(info["idx"], gz, os.path.join(d, "refgz%d_" % k), threads=1, extra=info["extra"])
        new = run_with_engine(info, os.path.join(d, "newgz%d_" % k), _oracle)
        assert not compare_outputs(ref, new)


GTF_FILES = INDEX_FILES + ["sjdbList.fromGTF.out.tab", "exonInfo.tab", "transcriptInfo.tab", "geneInfo.tab", "exonGeTrInfo.tab"]


@pytest.mark.parametrize("name,more", [("se50", []), ("pe101", ["--twopassMode", "Basic"])])
def test_gtf_at_the_mapping_stage(name, more, tmp_path, built):
    """--sjdbGTFfile with alignReads (star_amd/csrc/host/gtf.cpp): junctions of the annotation inserted before mapping; the tables
    the reference derives from the GTF are written too and must be identical"""
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    info["extra"] = list(info["extra"]) + ["--sjdbGTFfile", info["gtf"], "--sjdbInsertSave", "All"] + more
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refG_"), threads=1, extra=info["extra"])
    new = run_with_engine(info, os.path.join(d, "newG_"), _oracle)
    problems = compare_outputs(ref, new)
    for f in GTF_FILES:
        if open(ref + "_STARgenome/" + f, "rb").read() != open(new + "_STARgenome/" + f, "rb").read():
            problems.append("_STARgenome/%s differs" % f)
    assert not problems, problems


def test_threaded_junction_insertion(tmp_path, built, monkeypatch):
    """--runThreadN 7 with tiny slices: the suffix searches and the SA merge of the insertion run on threads, slice boundaries fall
    inside packed words; the index must still be byte-identical to the reference's"""
    monkeypatch.setenv("STARAMD_SJDB_SLICE", "5000")
    info = dict(prepare("pe150_indel", str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    flags = ["--twopassMode", "Basic", "--sjdbInsertSave", "All"]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refT_"), threads=1, extra=flags)
    info["extra"] = flags + ["--runThreadN", "7"]
    new = run_with_engine(info, os.path.join(d, "newT_"), _oracle)
    problems = compare_outputs(ref, new)
    for f in INDEX_FILES:
        if open(ref + "_STARgenome/" + f, "rb").read() != open(new + "_STARgenome/" + f, "rb").read():
            problems.append("_STARgenome/%s differs" % f)
    assert not problems, problems
