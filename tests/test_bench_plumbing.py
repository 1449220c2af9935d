"""bench.py's own plumbing on a box without a GPU (`--cpu-selftest`: gloo, the oracle behind the front end, a tiny genome):
the line is ONE JSON object under 4 KB that parses from the tail of stdout, `--gpus 2` really runs two ranks, and the variables that
redirect the product pipeline to test stand-ins are refused outside the self-test."""
import json
import os
import subprocess
import sys

import pytest

from util import ROOT

BENCH = os.path.join(ROOT, "bench.py")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(ROOT, "oracle", "_build", "libstaramd_cli_oracle.so")), reason="oracle front end not built")


def _line(stdout):
    tail = stdout[-4096:]                       # what a driver that keeps the last 4 KB of stdout sees
    last = tail.strip().splitlines()[-1]
    return json.loads(last), len(last)


@pytest.mark.parametrize("gpus", [1, 2])
def test_line_is_compact_and_counts_ranks(gpus, tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, BENCH, "--cpu-selftest", "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--workdir", str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d, n = _line(p.stdout)
    assert n < 4000
    assert d["n_gpus"] == gpus and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "roofline", "loaded_libs", "extra"):
        assert k in d
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "kernel", "achieved", "peak", "frac", "traffic", "kernel_ms", "algorithmic_bytes_per_launch"):
        assert k in d["roofline"]
    assert "selftest" in d                      # a CPU run can never pass for a measurement
    assert d["value"] > 0
    extra = json.load(open(d["extra"]))
    assert "counters_per_pair" in extra and "pipeline" in extra
    # reads processed by all ranks: every rank maps (steps) x 2000 pairs in the timed region
    assert abs(d["value"] * d["ms_per_step"] * d["steps"] * 1e3 - gpus * 2 * 2000) < 0.05 * gpus * 4000      # (value is rounded to 4 digits: ~1 % at this tiny rate)


def test_world_size_must_match_gpus(tmp_path):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--cpu-selftest", "--gpus", "2", "--workdir", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


@pytest.mark.parametrize("var", ["STARAMD_CLI_LIB", "STARAMD_ENGINE_LIB"])
def test_stand_in_libraries_are_refused(var, tmp_path):
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    env[var] = os.path.join(ROOT, "oracle", "_build", "libstaramd_cli_replay.so")
    p = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0", "--workdir", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, env=env)
    assert p.returncode != 0 and var in p.stderr and p.stdout.strip() == ""
