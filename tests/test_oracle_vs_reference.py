"""Pin the CPU restatement (oracle/) against the reference itself (oracle/_ref/STAR):
whole-run outputs must be byte-identical on synthetic data sets that exercise the adversarial
branches listed in SURVEY.md section 4.  Needs the reference binary (built from /root/reference by
oracle/Makefile.ref; it travels to the GPU box inside oracle/_ref/)."""
import pytest

from util import DATASETS, PARAM_SWEEP, compare_outputs, prepare, run_with_engine, oracle_lib, refstar

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")


@pytest.mark.parametrize("name", sorted(DATASETS))
def test_oracle_matches_reference(name, tmp_path, built):
    info = prepare(name, str(tmp_path))
    new = run_with_engine(info, str(tmp_path / name / "orc_"), lambda g, p: oracle_lib.Oracle(g, p))
    problems = compare_outputs(info["ref_prefix"], new)
    assert not problems, problems


def test_threaded_postmap_matches_reference(tmp_path, built):
    """--runThreadN > 1: the host post-map (multMapSelect ... SAM / SJ / Stats) runs on several threads over read ranges
    of a batch; SAM text is concatenated in read order, so the OUTPUT FILE is identical to the single-thread one."""
    info = dict(prepare("pe101", str(tmp_path)))
    one = run_with_engine(info, str(tmp_path / "pe101" / "t1_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=3000)
    info["extra"] = list(info["extra"]) + ["--runThreadN", "5"]
    many = run_with_engine(info, str(tmp_path / "pe101" / "t5_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=3000)
    assert not compare_outputs(info["ref_prefix"], many)
    a = [l for l in open(one + "Aligned.out.sam", "rb") if not l.startswith(b"@")]
    b = [l for l in open(many + "Aligned.out.sam", "rb") if not l.startswith(b"@")]
    assert a == b                     # same records in the same (input) order


@pytest.fixture(scope="module")
def sweep_data(tmp_path_factory, built):
    return prepare("pe101", str(tmp_path_factory.mktemp("sweep")), need_ref=False)


@pytest.mark.parametrize("combo", sorted(PARAM_SWEEP))
def test_oracle_matches_reference_with_flags(combo, sweep_data, built):
    """non-default flags (util.PARAM_SWEEP): the reference run with the same flags is the truth"""
    import os
    info = dict(sweep_data); info["extra"] = list(info["extra"]) + PARAM_SWEEP[combo]
    d = os.path.dirname(info["fastq"][0])
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "ref_%s_" % combo), threads=1, extra=info["extra"])
    new = run_with_engine(info, os.path.join(d, "orc_%s_" % combo), lambda g, p: oracle_lib.Oracle(g, p))
    assert not compare_outputs(ref, new)
