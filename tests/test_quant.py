"""--quantMode GeneCounts (ReadsPerGene.out.tab): host post-map code over the unique alignment of each read, three strandedness
columns.  Byte-identical to the reference, also combined with a GTF given at the mapping stage, 2-pass, BySJout and threads."""
import os

import pytest

from util import capi, compare_outputs, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

CASES = [("pe101", []), ("pe101", ["--twopassMode", "Basic", "--outFilterType", "BySJout"]), ("se50", ["GTF"]), ("pe150_chim", ["--runThreadN", "3"]),
         ("pe76_overlap", [])]


def _case(name, more, tmp_path, factory):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    m = ["--sjdbGTFfile", info["gtf"]] if more == ["GTF"] else more
    info["extra"] = list(info["extra"]) + ["--quantMode", "GeneCounts"] + m
    rf = list(info["extra"])
    if "--runThreadN" in rf:
        k = rf.index("--runThreadN"); del rf[k:k + 2]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refQ_"), threads=1, extra=rf)
    new = run_with_engine(info, os.path.join(d, "newQ_"), factory, batch_reads=900)
    problems = compare_outputs(ref, new)
    if open(ref + "ReadsPerGene.out.tab", "rb").read() != open(new + "ReadsPerGene.out.tab", "rb").read():
        problems.append("ReadsPerGene.out.tab differs")
    assert not problems, problems


@pytest.mark.parametrize("name,more", CASES)
def test_gene_counts_oracle(name, more, tmp_path, built):
    _case(name, more, tmp_path, lambda g, p: oracle_lib.Oracle(g, p))


@pytest.mark.gpu
@pytest.mark.parametrize("name,more", CASES[:2])
def test_gene_counts_engine(name, more, tmp_path, built):
    _case(name, more, tmp_path, lambda g, p: capi.Engine(g, p, device=0, max_reads=4096))
