"""BASELINE config 1 / SURVEY.md 8d "plumbing" case at its stated size: yeast-like genome (16 chromosomes, 12.1 Mb, uniform random,
--genomeSAindexNbases 10, no junction database), 100 000 single-end 50-bp reads from both strands with 1 % substitutions; reference
with --runThreadN 1.  Pass = byte-identical sorted SAM body + SJ.out.tab + Log.final.out counters."""
import os

import pytest

from util import capi, compare_outputs, oracle_lib, refstar, run_with_engine
from star_amd import synth

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")


def _prepare(tmp):
    d = os.path.join(str(tmp), "yeast")
    info = synth.make_dataset(d, seed=1, chr_lengths=(756250,) * 16, n_tr=0, n_reads=100000, read_len=50, paired=False, sub_rate=0.01,
                              n_rate=0.0, repeat_families=(), n_runs=0, frac_spliced=0.0)
    refstar.genome_generate(info["fasta"], d + "/idx", gtf=None, sa_index_nbases=10)
    info["idx"] = d + "/idx"; info["extra"] = []
    info["ref_prefix"] = refstar.align(info["idx"], info["fastq"], d + "/ref_", threads=1)
    return info


def test_config1_oracle(tmp_path, built):
    info = _prepare(tmp_path)
    new = run_with_engine(info, os.path.join(str(tmp_path), "orc_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=20000)
    assert not compare_outputs(info["ref_prefix"], new)


@pytest.mark.gpu
def test_config1_engine(tmp_path, built):
    info = _prepare(tmp_path)
    new = run_with_engine(info, os.path.join(str(tmp_path), "gpu_"), lambda g, p: capi.Engine(g, p, device=0, max_reads=32768), batch_reads=32768)
    assert not compare_outputs(info["ref_prefix"], new)
