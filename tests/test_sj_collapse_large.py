"""OutSJ::collapse on a table large enough for its parallel path (star_amd/csrc/host/sjout_stats.cpp: key ranges cut by sampled
splitters, one thread per range) against a numpy group-by of the same records: counts add up, overhangs take the maximum
(Junction::collapseOneSJ, source/OutSJ.cpp:74-98), records come out sorted by (start, gap) (OutSJ::collapseSJ, OutSJ.cpp:42-72)."""
import ctypes as C

import numpy as np

from util import capi, prepare
from star_amd import multi_gpu

REC = np.dtype([("start", "<u8"), ("gap", "<u4"), ("cu", "<u4"), ("cm", "<u4"), ("ol", "<u2"), ("orr", "<u2"),
                ("strand", "i1"), ("motif", "i1"), ("annot", "i1"), ("pad", "u1", (5,))])


def test_parallel_collapse_matches_group_by(tmp_path, built):
    assert REC.itemsize == multi_gpu.SJ_RECORD_BYTES
    info = prepare("se50", str(tmp_path), need_ref=False)
    run = capi.HostRun(["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "c_")])
    L = run.L
    multi_gpu._bind(L)
    rng = np.random.default_rng(7)
    for n, nLoci in ((1000, 300), (1_500_000, 200_000), (700_000, 3)):       # serial path, parallel path, parallel path with almost one key
        loci_start = rng.integers(1, 1 << 33, size=nLoci, dtype=np.uint64)
        loci_gap = rng.integers(21, 500000, size=nLoci, dtype=np.uint32)
        loci_gap[: nLoci // 2] = loci_gap[0]                                  # many loci share the start OR the gap
        loci_start[nLoci // 3:] = np.sort(loci_start[nLoci // 3:])
        loci_start[-(nLoci // 4):] = loci_start[-1]
        which = rng.integers(0, nLoci, size=n)
        a = np.zeros(n, dtype=REC)
        a["start"] = loci_start[which]; a["gap"] = loci_gap[which]
        a["cu"] = rng.integers(0, 3, size=n); a["cm"] = rng.integers(0, 3, size=n)
        a["ol"] = rng.integers(1, 100, size=n); a["orr"] = rng.integers(1, 100, size=n)
        key = a["start"] + a["gap"]                                            # strand / motif / annotation are functions of the locus
        a["strand"] = (key % 3).astype(np.int8); a["motif"] = (key % 7).astype(np.int8); a["annot"] = (key % 2).astype(np.int8)
        L.sah_sj_clear(run.h)
        assert L.sah_sj_import(run.h, a.ctypes.data_as(C.c_void_p), n) == 0
        m = int(L.sah_sj_export(run.h, None, 0))
        out = np.zeros(m, dtype=REC)
        assert int(L.sah_sj_export(run.h, out.ctypes.data_as(C.c_void_p), m)) == m
        # expected: group by (start, gap)
        order = np.lexsort((a["gap"], a["start"]))
        s = a[order]
        first = np.ones(n, dtype=bool); first[1:] = (s["start"][1:] != s["start"][:-1]) | (s["gap"][1:] != s["gap"][:-1])
        idx = np.flatnonzero(first)
        assert m == len(idx)
        assert np.array_equal(out["start"], s["start"][idx]) and np.array_equal(out["gap"], s["gap"][idx])
        assert np.array_equal(out["cu"], np.add.reduceat(s["cu"].astype(np.uint64), idx).astype(np.uint32))
        assert np.array_equal(out["cm"], np.add.reduceat(s["cm"].astype(np.uint64), idx).astype(np.uint32))
        assert np.array_equal(out["ol"], np.maximum.reduceat(s["ol"], idx)) and np.array_equal(out["orr"], np.maximum.reduceat(s["orr"], idx))
        for f in ("strand", "motif", "annot"):
            assert np.array_equal(out[f], s[f][idx])
    L.sah_sj_clear(run.h)
    run.close()
