"""Run the first N reads of a data set through the WAVE-EMULATED engine (oracle/_build/libstaramd_emul.so: the kernel sources of
star_amd/csrc/engine compiled for the host, oracle/wave_emul/emu.h) and through the oracle, and compare the result buffers byte for
byte.  Test infrastructure; a fresh process so that STARAMD_ENGINE_LIB and the STARAMD_* knobs take effect.
Usage: python tests/emul_run.py <dataset | prepared info.pkl> <workdir> <nReads> [--gpuResultSelect All|Selected] [host flags...]      prints OK or the first difference"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["STARAMD_ENGINE_LIB"] = os.environ.get("STARAMD_EMUL_LIB") or os.path.join(os.path.dirname(HERE), "oracle", "_build", "libstaramd_emul.so")     # STARAMD_EMUL_LIB: the AddressSanitizer build
os.environ.setdefault("STARAMD_WIN_BLOCKS_BIG", "2")      # 64 blocks of the last k_windows launch = 3.9 GB of work space that the emulated hipMalloc would fill with its pattern
sys.path.insert(0, HERE)
from util import capi, oracle_lib, prepare  # noqa: E402


def main():
    name, wd, n_reads, more = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4:]
    if name.endswith(".pkl"):                      # a data set the caller has prepared already (tests/test_wave_emul.py: once per session, not once per process)
        import pickle
        info = pickle.load(open(name, "rb"))
    else:
        info = prepare(name, wd, need_ref=False)
    argv = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", os.path.join(wd, "e_"), "--readMapNumber", str(n_reads)] + list(info["extra"]) + more
    run = capi.HostRun(argv)
    eng = capi.Engine(run.genome, run.params, device=0, max_reads=max(64, n_reads))
    orc = oracle_lib.Oracle(run.genome, run.params)
    selected = "All" not in more
    t0 = time.time()
    try:
        b = run.next_batch(n_reads)
        n = b.nReads
        bg = capi.ResultBuffers(n, tr_cap=n * 400); bo = capi.ResultBuffers(n, tr_cap=n * 400)
        eng.map_batch(b, bg); orc.map_batch(b, bo)
        rg, tg, eg = bg.as_bytes(n); ro, to, eo = bo.as_bytes(n)
        if bg.res.trCount != bo.res.trCount or bg.res.exCount != bo.res.exCount:
            bad = [(i, bg.reads[i].nTr, bo.reads[i].nTr, bg.reads[i].nW, bo.reads[i].nW, bg.reads[i].status, bo.reads[i].status) for i in range(n) if bg.reads[i].nTr != bo.reads[i].nTr][:5]
            print("DIFF transcript / exon counts: engine %d %d, oracle %d %d; (read, nTr engine, oracle, nW engine, oracle, status engine, oracle): %r" % (bg.res.trCount, bg.res.exCount, bo.res.trCount, bo.res.exCount, bad)); return
        for i in range(n):
            a, o = bg.reads[i], bo.reads[i]
            fa = (a.status, a.nW, a.nTr, a.trOffset, a.trBest, a.unmappedLength, a.maxScoreMate[0], a.maxScoreMate[1])
            fo = (o.status, o.nW, o.nTr, o.trOffset, o.trBest, o.unmappedLength, o.maxScoreMate[0], o.maxScoreMate[1])
            same = fa == fo                                      # (Selected: maxScoreMate[] is 0 on both sides, include/star_amd.h)
            if not same:
                print("DIFF read %d: engine %r oracle %r" % (i, fa, fo)); return
        if tg != to:
            print("DIFF transcript records"); return
        if eg != eo:
            print("DIFF exon records"); return
        print("OK %d reads, %d transcripts, %.1f s" % (n, bg.res.trCount, time.time() - t0))
    finally:
        eng.close(); orc.close(); run.close()


main()
