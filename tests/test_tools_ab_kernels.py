"""tools/ab_kernels.py (the same-box kernel A/B of the GPU sessions) over the wavefront-emulated engine: the tool parses its variant specs, loads the index and the batches once,
gives every variant a fresh context, applies the variant's environment, compares the result buffers of the variants and prints one line per variant and round.  A bug in the tool
costs a GPU session (it did once); this runs it end to end on a machine without a GPU."""
import os
import shutil
import subprocess
import sys

import pytest

from util import prepare, refstar

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "libstaramd_emul.so")
pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR missing (index generation)")


def test_ab_kernels_runs_variants_and_compares_their_results(tmp_path, built):
    subprocess.check_call(["make", "-s", "oracle/_build/libstaramd_emul.so"], cwd=ROOT)
    info = prepare("pe101", str(tmp_path), need_ref=False)
    d = str(tmp_path / "keep"); os.makedirs(d)
    idx = os.path.join(d, "idx"); shutil.copytree(os.path.realpath(info["idx"]), idx)
    fq = []
    for f in info["fastq"]:
        shutil.copy(os.path.realpath(f), os.path.join(d, os.path.basename(f))); fq.append(os.path.join(d, os.path.basename(f)))
    out = str(tmp_path / "ab.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_kernels.py"), "--genome-dir", idx, "--fastq"] + fq + ["--reads", "30", "--batches", "2", "--repeat", "1", "--rounds", "2", "--out", out,
                        "base|%s|" % LIB, "no_lane_kernel|%s|STARAMD_LANE=0 STARAMD_PRUNE=3" % LIB],
                       env=dict(os.environ, STARAMD_WIN_BLOCKS_BIG="2"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith(("base", "no_lane_kernel"))]
    assert len(lines) == 4 and all("seed" in l and "windows" in l and "stitch" in l for l in lines), p.stdout[-2000:]
    assert not any("RESULTS DIFFER" in l for l in lines), p.stdout[-2000:]
    assert os.path.getsize(out) > 100
