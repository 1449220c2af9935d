"""--outSAMtype BAM Unsorted: BAM records (ReadAlign::alignBAM) in BGZF blocks.  The decompressed record stream must be byte-identical to the
reference's (single-thread order = input order on both sides); header text may differ only in the @PG / @CO command-line lines."""
import os

import pytest

from util import bam_parts, capi, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

CASES = [("pe101", []),
         ("se50", ["--outSAMunmapped", "Within"]),
         ("pe150_indel", ["--outSAMattributes", "NH", "HI", "AS", "nM", "NM", "MD", "jM", "jI", "MC", "XS", "--outSAMunmapped", "Within", "--outSAMstrandField", "intronMotif"]),
         ("pe150_chim", ["--outSAMattributes", "All", "--outSAMunmapped", "Within", "--runThreadN", "3", "--outBAMcompression", "6"]),
         ("pe76_overlap", ["--twopassMode", "Basic"])]


def _case(name, more, tmp_path, factory):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    info["extra"] = list(info["extra"]) + ["--outSAMtype", "BAM", "Unsorted"] + more
    rf = list(info["extra"])
    if "--runThreadN" in rf:
        k = rf.index("--runThreadN"); del rf[k:k + 2]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refB_"), threads=1, extra=rf)
    new = run_with_engine(info, os.path.join(d, "newB_"), factory, batch_reads=700)
    (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + "Aligned.out.bam"), bam_parts(new + "Aligned.out.bam")
    assert ra == rb
    assert len(rr) == len(nr)
    assert rr == nr
    keep = lambda t: [l for l in t.split(b"\n") if not l.startswith((b"@PG", b"@CO"))]
    assert keep(ta) == keep(tb)
    assert open(new + "Aligned.out.bam", "rb").read()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")     # BGZF EOF marker
    assert open(ref + "SJ.out.tab", "rb").read() == open(new + "SJ.out.tab", "rb").read()
    assert not os.path.exists(new + "Aligned.out.sam")


@pytest.mark.parametrize("name,more", CASES)
def test_bam_unsorted_oracle(name, more, tmp_path, built):
    _case(name, more, tmp_path, lambda g, p: oracle_lib.Oracle(g, p))


@pytest.mark.gpu
@pytest.mark.parametrize("name,more", CASES[2:4])
def test_bam_unsorted_engine(name, more, tmp_path, built):
    _case(name, more, tmp_path, lambda g, p: capi.Engine(g, p, device=0, max_reads=4096))


SORTED = [("pe101", ["SortedByCoordinate"], []),
          ("se50", ["Unsorted", "SortedByCoordinate"], ["--outSAMunmapped", "Within"]),
          ("pe150_indel", ["SortedByCoordinate"], ["--outSAMunmapped", "Within", "--outFilterType", "BySJout", "--outSAMattributes", "NH", "HI", "AS", "nM", "NM", "MD", "MC"]),
          ("pe150_chim", ["SortedByCoordinate"], ["--outSAMunmapped", "Within", "--runThreadN", "3", "--twopassMode", "Basic"])]


def _sorted_case(name, types, more, tmp_path, factory):
    """Aligned.sortedByCoord.out.bam: records ordered by (reference, position, read order, order of production), unmapped reads last in
    read order -- the reference's bin sort (BAMbinSortByCoordinate.cpp / BAMbinSortUnmapped.cpp) leaves exactly this order"""
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    info["extra"] = list(info["extra"]) + ["--outSAMtype", "BAM"] + types + more
    rf = list(info["extra"])
    if "--runThreadN" in rf:
        k = rf.index("--runThreadN"); del rf[k:k + 2]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refS_"), threads=1, extra=rf)
    new = run_with_engine(info, os.path.join(d, "newS_"), factory, batch_reads=700)
    for f in ["Aligned.sortedByCoord.out.bam"] + (["Aligned.out.bam"] if "Unsorted" in types else []):
        (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + f), bam_parts(new + f)
        assert ra == rb and rr == nr, f
        assert ta.split(b"\n")[0] == tb.split(b"\n")[0]
    if "Unsorted" not in types:
        assert not os.path.exists(new + "Aligned.out.bam")


@pytest.mark.parametrize("name,types,more", SORTED)
def test_bam_sorted_oracle(name, types, more, tmp_path, built):
    _sorted_case(name, types, more, tmp_path, lambda g, p: oracle_lib.Oracle(g, p))


@pytest.mark.gpu
@pytest.mark.parametrize("name,types,more", SORTED[2:3])
def test_bam_sorted_engine(name, types, more, tmp_path, built):
    _sorted_case(name, types, more, tmp_path, lambda g, p: capi.Engine(g, p, device=0, max_reads=4096))


def test_out_sam_type_none_and_errors(tmp_path, built):
    info = prepare("se50", str(tmp_path), need_ref=False)
    base = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"]
    run = capi.HostRun(base + ["--outFileNamePrefix", str(tmp_path / "n_"), "--outSAMtype", "None"])
    run.close()
    assert not os.path.exists(str(tmp_path / "n_Aligned.out.sam")) and not os.path.exists(str(tmp_path / "n_Aligned.out.bam"))
    for flags, text in [(["--outSAMtype", "BAM"], "missing BAM option"), (["--outSAMtype", "BAM", "Sorted"], "unknown value for the word 2 of outSAMtype"),
                        (["--outSAMtype", "SAM", "Unsorted"], "can cannot be combined"), (["--outSAMattributes", "NH", "ch"], "requires BAM output")]:
        with pytest.raises(RuntimeError) as e:
            capi.HostRun(base + ["--outFileNamePrefix", str(tmp_path / "e_")] + flags)
        assert text in str(e.value)
