"""Junction insertion ON THE DEVICE (star_amd/csrc/index/sjdb_core.h): the search of every new junction suffix
in the old suffix array, the sort of the new suffixes, the merge into the new packed array and the SAindex of the result.

Same cases and the same bar as tests/test_two_pass.py (the index left in _STARgenome/ with --sjdbInsertSave All is compared with the
reference's byte for byte: Genome, SA, SAindex, sjdbInfo.txt, sjdbList.out.tab), with the insertion switched from the host restatement to
  * CPU tests: the device algorithm on the plain-loop backend (oracle/_build/libindex_emul.so, test infrastructure)
  * -m gpu:    staramd_sjdb_insert of the HIP library, the passes themselves on the HIP engine
"""
import os

import pytest

import test_two_pass as t2
from util import capi, compare_outputs, prepare, refstar, run_with_engine, ROOT

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")
EMUL = os.path.join(ROOT, "oracle", "_build", "libindex_emul.so")


def _emul():
    return capi.device_sjdb_insertion(EMUL, "sjdb_emul_insert")


@pytest.mark.parametrize("name", ["pe101", "se50", "pe150_indel", "pe76_overlap"])
def test_two_pass_device_algorithm(name, tmp_path, built):
    """old junctions from the generated genome keep their place and get new numbers, novel ones of the 1st pass are inserted"""
    with _emul():
        t2._two_pass_case(name, tmp_path, t2._oracle)


def test_file_insertion_then_two_pass_device_algorithm(tmp_path, built):
    with _emul():
        t2._file_insert_case("pe101", tmp_path, t2._oracle, ["--twopassMode", "Basic", "--twopass1readsN", "1000"])


def test_file_insertion_into_plain_genome_device_algorithm(tmp_path, built):
    with _emul():
        t2._file_insert_case("se50", tmp_path, t2._oracle, [])


@pytest.mark.parametrize("name,more", [("se50", []), ("pe101", ["--twopassMode", "Basic"])])
def test_gtf_at_the_mapping_stage_device_algorithm(name, more, tmp_path, built):
    with _emul():
        t2.test_gtf_at_the_mapping_stage(name, more, tmp_path, built)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pe101", "se50", "pe150_indel", "pe76_overlap"])
def test_two_pass_device(name, tmp_path, built):
    with capi.device_sjdb_insertion():
        t2._two_pass_case(name, tmp_path, t2._engine)


@pytest.mark.gpu
def test_file_insertions_device(tmp_path, built):
    with capi.device_sjdb_insertion():
        t2._file_insert_case("pe101", tmp_path / "a", t2._engine, ["--twopassMode", "Basic", "--twopass1readsN", "1000"])
        t2._file_insert_case("se50", tmp_path / "b", t2._engine, [])
