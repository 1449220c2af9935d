"""Corner-case reads (util.make_edge_reads): 1-base and 650-base reads, all-N, homopolymers, N-riddled reads, chromosome ends,
fully overlapping / chimeric / junk mates, lower case and IUPAC codes, far too many mismatches, a 6-base middle exon, very
different mate lengths.  The reference run on the same FASTQ is the truth for the oracle (CPU) and the engine (GPU)."""
import os

import pytest

from util import capi, compare_outputs, make_edge_reads, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

CASES = [("pe150_indel", True), ("se50", False), ("pe101", True)]


def _case(name, paired, tmp_path, factory):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.join(str(tmp_path), name)
    info["fastq"] = make_edge_reads(info, d, paired)
    info["extra"] = list(info["extra"]) + ["--outSAMunmapped", "Within"]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refE_"), threads=1, extra=info["extra"])
    new = run_with_engine(info, os.path.join(d, "newE_"), factory, batch_reads=100)
    assert not compare_outputs(ref, new)
    return info


@pytest.mark.parametrize("name,paired", CASES)
def test_oracle_on_edge_reads(name, paired, tmp_path, built):
    _case(name, paired, tmp_path, lambda g, p: oracle_lib.Oracle(g, p))


@pytest.mark.gpu
@pytest.mark.parametrize("name,paired", CASES)
def test_engine_on_edge_reads(name, paired, tmp_path, built):
    from test_gpu_parity import _compare_buffers
    info = _case(name, paired, tmp_path, lambda g, p: capi.Engine(g, p, device=0, max_reads=4096))
    _compare_buffers(info, ["--gpuResultSelect", "All"], str(tmp_path / "x_"))       # and every recorded transcript, byte for byte
