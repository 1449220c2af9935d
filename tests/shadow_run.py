"""Run one data set through the SHADOW-VALIDATION build of the engine (libstaramd_shadow.so) and print the
disagreement counters as JSON: in that build every wave-cooperative stitch / extend call is re-run on the GPU through
the scalar restatement (star_amd/csrc/engine/stitch_scalar.h) and compared field by field.
Usage: STARAMD_ENGINE_LIB=shadow python tests/shadow_run.py <dataset> <workdir>"""
import json
import os
import sys

os.environ["STARAMD_ENGINE_LIB"] = "shadow"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import capi, prepare  # noqa: E402

IDX = {"stitchBad": 17, "stitchN": 18, "extendBad": 19, "extendN": 20}


def main():
    name, workdir = sys.argv[1], sys.argv[2]
    info = prepare(name, workdir, need_ref=False)
    argv = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", os.path.join(workdir, "sh_")] + info["extra"]
    run = capi.HostRun(argv)
    eng = capi.Engine(run.genome, run.params, device=0, max_reads=4096)
    tot = dict((k, 0) for k in IDX)
    try:
        while True:
            b = run.next_batch(1500)
            if b is None:
                break
            bufs = capi.ResultBuffers(b.nReads, tr_cap=b.nReads * 64)
            eng.map_batch(b, bufs)
            c = eng.counters(24)
            for k, i in IDX.items():
                tot[k] += c[i]
    finally:
        eng.close(); run.close()
    print(json.dumps(tot))


if __name__ == "__main__":
    main()
