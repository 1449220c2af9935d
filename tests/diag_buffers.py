"""Diagnostic (not a test): first read whose exon / transcript records differ between the engine and the oracle.
usage: python tests/diag_buffers.py <dataset> <workdir> [flags...]"""
import sys
import ctypes as C
from util import capi, oracle_lib, prepare


def rec(x):
    return {f[0]: (list(getattr(x, f[0])) if hasattr(getattr(x, f[0]), "__len__") else getattr(x, f[0])) for f in x._fields_}


def main():
    name, wd, more = sys.argv[1], sys.argv[2], sys.argv[3:]
    info = prepare(name, wd, need_ref=False)
    argv = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", wd + "/d_"] + info["extra"] + more
    run = capi.HostRun(argv)
    eng = capi.Engine(run.genome, run.params, device=0, max_reads=4096)
    orc = oracle_lib.Oracle(run.genome, run.params)
    shown = 0
    while shown < 3:
        b = run.next_batch(1500)
        if b is None:
            break
        n = b.nReads
        bg = capi.ResultBuffers(n, tr_cap=n * 200); bo = capi.ResultBuffers(n, tr_cap=n * 200)
        eng.map_batch(b, bg); orc.map_batch(b, bo)
        for i in range(n):
            a, o = bg.reads[i], bo.reads[i]
            ta = [rec(bg.tr[a.trOffset + k]) for k in range(a.nTr)]; to = [rec(bo.tr[o.trOffset + k]) for k in range(o.nTr)]
            ea = [[rec(bg.ex[t["exonOffset"] + k]) for k in range(t["nExons"])] for t in ta]
            eo = [[rec(bo.ex[t["exonOffset"] + k]) for k in range(t["nExons"])] for t in to]
            for t in ta + to:
                t.pop("exonOffset")
            if ta != to or ea != eo:
                L = b.readOffset[i + 1] - b.readOffset[i]
                print("READ", i, "Lread", L, "mate1", b.mate1Length[i], "bases", "".join("ACGTN......#"[b.bases[b.readOffset[i] + k]] for k in range(L)))
                for k in range(max(len(ta), len(to))):
                    if k >= len(ta) or k >= len(to) or ta[k] != to[k] or ea[k] != eo[k]:
                        print(" tr", k, "GPU", ta[k] if k < len(ta) else None); print("      ORC", to[k] if k < len(to) else None)
                        for x in (ea[k] if k < len(ea) else []): print("   gpu ex", {q: x[q] for q in ("G", "R", "L", "iFrag", "canonSJ", "sjAnnot", "shiftSJ", "sjA")})
                        for x in (eo[k] if k < len(eo) else []): print("   orc ex", {q: x[q] for q in ("G", "R", "L", "iFrag", "canonSJ", "sjAnnot", "shiftSJ", "sjA")})
                shown += 1
                if shown >= 3:
                    break
    print("done, shown", shown)


main()
