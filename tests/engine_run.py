"""Run one data set through the HIP engine in a fresh process (so that STARAMD_* environment knobs of the engine take
effect) and compare with the reference outputs.  Prints OK or the list of problems.
Usage: python tests/engine_run.py <dataset> <workdir>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import capi, compare_outputs, prepare, run_with_engine  # noqa: E402


def main():
    name, workdir = sys.argv[1], sys.argv[2]
    info = prepare(name, workdir)
    new = run_with_engine(info, os.path.join(workdir, name, "gpu_"), lambda g, p: capi.Engine(g, p, device=0, max_reads=2048), batch_reads=1999)
    problems = compare_outputs(info["ref_prefix"], new)
    print("OK" if not problems else "PROBLEMS %r" % problems)


if __name__ == "__main__":
    main()
