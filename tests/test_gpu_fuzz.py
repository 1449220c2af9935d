"""The hardware fuzzer inside the GPU suite (tests/tools/fuzz_engine.py: random data sets x random hot-path flags, the HIP engine's result
buffers against the oracle's byte for byte, both result-selection modes, merged-mate and clipped batches).  Fixed seeds: the same ~150
combinations on every run; a new seed for a bug hunt is `python tests/tools/fuzz_engine.py <n> <seed>` on the GPU box.  This is the tool that
found the only hot-path difference of round 3 (`(int) L` in extendAlign.cpp:59) after thousands of 40-read emulator combinations had not."""
import os
import subprocess
import sys

import pytest

from util import ROOT, refstar

FUZZ = os.path.join(ROOT, "tests", "tools", "fuzz_engine.py")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", [(73, 50), (977, 50), (20260923, 50)])
def test_random_datasets_and_flags_match_oracle(seed, n, built):
    if not refstar.have_ref():
        pytest.skip("oracle/_ref/STAR missing (the random data sets are indexed by the reference)")
    env = {k: v for k, v in os.environ.items() if k not in ("FUZZ_EMUL", "STARAMD_ENGINE_LIB")}
    p = subprocess.run([sys.executable, FUZZ, str(n), str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    lines = p.stdout.strip().splitlines()
    fails = [l for l in lines if l.startswith(("FAIL", "      "))]
    assert p.returncode == 0 and lines and lines[-1].startswith("0 of %d" % n), "\n".join(fails[-12:] or lines[-12:])
