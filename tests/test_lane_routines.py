"""The one-lane routines of the engine (stitch_scalar.h: growOnLane / junctionOnLane / joinOnLane, 8 bases per trip on byte-parallel masks) against the base-by-base
restatement kept under oracle/ (lane_routines_ref.h) on random plausible inputs: every kind of gap between two seeds, both strands, both mates, annotated junctions,
all failure codes; every output field compared.  Host build through the wavefront emulator's headers."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = os.environ.get("EMUL_CXX", "/opt/rocm/lib/llvm/bin/clang++")
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="the host clang++ of ROCm is missing")


def test_lane_routines_against_the_base_by_base_restatement(tmp_path):
    exe = str(tmp_path / "lane_routines_check")
    subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O2", "-Wno-unknown-attributes", "-Wno-unused-result", "-D_GNU_SOURCE", "-I", "oracle/wave_emul", "-I", "star_amd/csrc/engine", "-I", "include", "-I", "oracle",
                           "oracle/lane_routines_check.cpp", "oracle/wave_emul/emu.cpp", "oracle/wave_emul/emu_lds.cpp", "-o", exe, "-ldl"], cwd=ROOT)
    p = subprocess.run([exe, "400000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    last = p.stdout.strip().splitlines()[-1]
    assert p.returncode == 0 and last.endswith(": 0 differences"), p.stdout[-3000:]
