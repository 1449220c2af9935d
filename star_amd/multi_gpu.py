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
    L.sah_sj_select.restype = None; L.sah_sj_select.argtypes = [C.c_void_p, C.c_int]
    L.sah_in_stage1.restype = C.c_int; L.sah_in_stage1.argtypes = [C.c_void_p]
    L.sah_quant_export.restype = C.c_uint64; L.sah_quant_export.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64]
    L.sah_quant_import_add.restype = C.c_int; L.sah_quant_import_add.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64]
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


def _gather_tables(run, dist, dev, rank, world, everyone, with_stats=True):
    """all_gather of (padded junction table, counters) of the table selected with sah_sj_select; imported by rank 0 only, or by
    every rank (everyone=True: all ranks continue with the same union)."""
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
    if rank == 0 or everyone:
        for r in range(world):
            if r == rank:
                continue
            g = gathered[r].cpu().numpy()
            import_rank_tables(run, g[:sizes[r]], g[cap:].view(np.int64) if (rank == 0 and with_stats) else np.zeros(NSTAT, dtype=np.int64))
    return int(payload.numel())


def next_phase(run, dist, dev, rank, world):
    """Multi-rank form of HostRun.next_phase() for --twopassMode Basic and --outFilterType BySJout: what the reference merges across its
    threads between the phases has to be merged across ranks first -- the junctions of the 1st pass before
    they are inserted into every rank's index replica, the junctions of all reads of BySJout stage 1 before the whitelist is built.
    Every rank imports every other rank's table, so all ranks go on with the same index / whitelist.  Returns HostRun.next_phase()."""
    L = run.L
    _bind(L)
    if run.in_pass1():
        L.sah_sj_select(run.h, 0)
        _gather_tables(run, dist, dev, rank, world, everyone=True)
    elif L.sah_in_stage1(run.h):
        L.sah_sj_select(run.h, 1)
        _gather_tables(run, dist, dev, rank, world, everyone=True, with_stats=False)     # Stats travel once, at the end of the run
        L.sah_sj_select(run.h, 0)
    return run.next_phase()


def _merge_gene_counts(run, dist, dev, rank, world):
    L = run.L
    n = int(L.sah_quant_export(run.h, None, 0))
    if n == 0:
        return
    buf = (C.c_uint64 * n)()
    L.sah_quant_export(run.h, buf, n)
    t = torch.from_numpy(np.frombuffer(bytes(buf), dtype=np.int64).copy()).to(dev)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    if rank == 0:
        for r in range(1, world):
            a = gathered[r].cpu().numpy().astype(np.uint64)
            L.sah_quant_import_add(run.h, (C.c_uint64 * n)(*[int(x) for x in a]), n)


def merge_run_outputs(run, dist, dev, rank, world):
    """all_gather of (padded junction table, counters); rank 0 ends up holding the union.  Returns bytes moved per rank."""
    _bind(run.L)
    run.L.sah_sj_select(run.h, 0)
    _merge_gene_counts(run, dist, dev, rank, world)
    return _gather_tables(run, dist, dev, rank, world, everyone=False)


# ---- the same exchanges for the in-process front end (include/star_amd_cli.h: staramd_cli_hooks.exchange hands out the sah_* handle) ----

class HandleRun:
    """Adapter: a raw sah_* handle + the host library, in the shape the functions above expect."""

    def __init__(self, L, h):
        self.L = L
        self.h = C.c_void_p(h) if not isinstance(h, C.c_void_p) else h

    def in_pass1(self):
        self.L.sah_in_pass1.restype = C.c_int; self.L.sah_in_pass1.argtypes = [C.c_void_p]
        return bool(self.L.sah_in_pass1(self.h))


def exchange_before_phase(L, h, dist, dev, rank, world):
    """Hook body for a phase that is NOT the last one (1st pass of --twopassMode Basic, stage 1 of --outFilterType BySJout): every rank
    imports every other rank's junction table, so that all ranks insert the same junctions / build the same whitelist."""
    run = HandleRun(L, h)
    _bind(L)
    if run.in_pass1():
        L.sah_sj_select(run.h, 0)
        _gather_tables(run, dist, dev, rank, world, everyone=True)
    elif L.sah_in_stage1(run.h):
        L.sah_sj_select(run.h, 1)
        _gather_tables(run, dist, dev, rank, world, everyone=True, with_stats=False)
        L.sah_sj_select(run.h, 0)


def merge_handle_outputs(L, h, dist, dev, rank, world):
    """Hook body for the end of the run: rank 0 ends up with the union of the junction tables, the summed counters and gene counts."""
    return merge_run_outputs(HandleRun(L, h), dist, dev, rank, world)
