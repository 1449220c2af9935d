"""Static figures of the two wave-cooperative kernels as the compiler reports them (tools/isa_stats.sh; hipcc cross-compiles without a GPU).
The register allocator has two regimes for k_stitch_win -- wave-uniform state in scalar registers (106 VGPRs, no scratch) or in vector registers (168 + spills) -- and
small edits of the kernel flip it back into the second (tools/isa_stats.sh k_stitch shows which one a build is in).  This test is what notices."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc missing")


def _figures(src, kernel, *flags):
    out = subprocess.run([os.path.join(ROOT, "tools", "isa_stats.sh"), src] + list(flags), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900).stdout
    line = [l for l in out.splitlines() if l.split() and l.split()[0] == kernel]
    assert line, out[-2000:]
    g = lambda k: int(re.search(k + r"\s+(\d+)", line[0]).group(1))
    return {"vgprs": g("VGPRs"), "vgpr_spills": g("VGPR spills"), "scratch": int(re.search(r"scratch\s+(\d+) B", line[0]).group(1)), "saveexec": g("saveexec")}


def test_stitch_kernel_keeps_its_uniform_state_in_scalar_registers():
    f = _figures("k_stitch", "k_stitch_win")
    assert f["vgprs"] <= 112 and f["vgpr_spills"] == 0 and f["scratch"] == 0 and f["saveexec"] <= 160, f


def test_lane_kernel_spills_no_vector_registers():
    """k_stitch_lane with spilled vector registers (beside its spilled scalar registers and per-lane arrays in scratch) corrupted results on hardware in round 6
    (k_stitch_lane.hip LANE_WAVES has the story): the kernel is held to an occupancy at which nothing of its vector state spills."""
    f = _figures("k_stitch_lane", "k_stitch_lane")
    assert f["vgpr_spills"] == 0, f


def test_window_kernel_does_not_spill_vector_registers():
    f = _figures("k_window", "k_windows")
    assert f["vgprs"] <= 80 and f["vgpr_spills"] == 0 and f["saveexec"] <= 200, f
