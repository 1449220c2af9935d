"""End-of-run exchange between ranks (one process per GPU).

The hot path shards over read batches with NO data-path collective: every rank holds a full index replica in its
GPU's HBM and maps its own shard.  The only cross-rank step is the reduction the reference does across its threads at
the end of a run:
  * junction tables   -- outputSJ k-way merge of per-thread tables       source/outputSJ.cpp:39-83
  * Stats counters    -- Stats::addStats under mutexStats                source/ReadAlignChunk_mapChunk.cpp:124-127
Here both travel in ONE all_gather over RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests): fixed 32-byte junction
records (already collapsed per rank) padded to the largest rank + 32 u64 counters.  Rank 0 then imports everything and
runs the unchanged collapse / filter / SJ.out.tab / Log.final.out code of the host library.
"""
import ctypes as C

import numpy as np
import torch

SJ_RECORD_BYTES = 32
NSTAT = 32


def _bind(L):
    if getattr(L, "_mg_bound", False):
        return
    L.sah_sj_export.restype = C.c_uint64; L.sah_sj_export.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.sah_sj_import.restype = C.c_int; L.sah_sj_import.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.sah_sj_clear.restype = None; L.sah_sj_clear.argtypes = [C.c_void_p]
    L.sah_stats_export.restype = C.c_int; L.sah_stats_export.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.sah_stats_import_add.restype = C.c_int; L.sah_stats_import_add.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L._mg_bound = True


def export_rank_tables(run):
    """(junction records as uint8[n*32], stats as int64[32]) of this rank's HostRun."""
    L = run.L
    _bind(L)
    n = int(L.sah_sj_export(run.h, None, 0))
    buf = np.zeros(max(n, 1) * SJ_RECORD_BYTES, dtype=np.uint8)
    got = int(L.sah_sj_export(run.h, buf.ctypes.data_as(C.c_void_p), n))
    assert got == n
    st = (C.c_uint64 * NSTAT)()
    L.sah_stats_export(run.h, st)
    return buf[:n * SJ_RECORD_BYTES], np.frombuffer(bytes(st), dtype=np.int64).copy()


def import_rank_tables(run, sj_bytes, stats):
    L = run.L
    _bind(L)
    n = len(sj_bytes) // SJ_RECORD_BYTES
    if n:
        a = np.ascontiguousarray(sj_bytes, dtype=np.uint8)
        L.sah_sj_import(run.h, a.ctypes.data_as(C.c_void_p), n)
    st = (C.c_uint64 * NSTAT)(*[int(x) & 0xFFFFFFFFFFFFFFFF for x in np.asarray(stats, dtype=np.int64).astype(np.uint64)])
    L.sah_stats_import_add(run.h, st)


def merge_run_outputs(run, dist, dev, rank, world):
    """all_gather of (padded junction table, counters); rank 0 ends up holding the union.  Returns bytes moved per rank."""
    sj, st = export_rank_tables(run)
    n = torch.tensor([len(sj)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), SJ_RECORD_BYTES)
    payload = torch.zeros(cap + NSTAT * 8, dtype=torch.uint8, device=dev)
    if len(sj):
        payload[:len(sj)] = torch.from_numpy(sj).to(dev)
    payload[cap:] = torch.from_numpy(st.view(np.uint8)).to(dev)
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload)
    if rank == 0:
        for r in range(1, world):
            g = gathered[r].cpu().numpy()
            import_rank_tables(run, g[:sizes[r]], g[cap:].view(np.int64))
    return int(payload.numel())
