"""ctypes mirror of include/star_amd.h and of the host library's `sah_*` interface.

Python is only plumbing here (tests, bench.py, smoke): the product is the C-ABI library
star_amd/lib/libstaramd.so (hand-written HIP for gfx950) plus the C++ host library
star_amd/lib/libstaramd_host.so.  Loading fails loudly if a library is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
# STARAMD_ENGINE_LIB=shadow selects the shadow-validation build (tests only; see star_amd/csrc/engine/stitch_scalar.h)
_VARIANT = os.environ.get("STARAMD_ENGINE_LIB", "")
ENGINE_PATH = os.path.join(LIB_DIR, {"shadow": "libstaramd_shadow.so", "profile": "libstaramd_profile.so"}.get(_VARIANT, "libstaramd.so"))
if "/" in _VARIANT:                      # an explicit library path (build experiments)
    ENGINE_PATH = _VARIANT
HOST_PATH = os.path.join(LIB_DIR, "libstaramd_host.so")

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class Genome(C.Structure):
    _fields_ = [
        ("G", u8p), ("nGenome", C.c_uint64),
        ("SA", u8p), ("nSA", C.c_uint64), ("nSAbyte", C.c_uint64),
        ("SAi", u8p), ("nSAi", C.c_uint64), ("nSAibyte", C.c_uint64),
        ("GstrandBit", C.c_uint32), ("gSAindexNbases", C.c_uint32),
        ("genomeSAindexStart", C.c_uint64 * 17),
        ("gSAsparseD", C.c_uint32), ("gChrBinNbits", C.c_uint32),
        ("chrStart", u64p), ("chrLength", u64p), ("nChrReal", C.c_uint32),
        ("chrBin", u32p), ("chrBinN", C.c_uint64),
        ("sjGstart", C.c_uint64), ("sjdbOverhang", C.c_uint32), ("sjdbLength", C.c_uint32), ("sjdbN", C.c_uint32),
        ("sjDstart", u64p), ("sjAstart", u64p), ("sjdbStart", u64p), ("sjdbEnd", u64p),
        ("sjdbMotif", u8p), ("sjdbShiftLeft", u8p), ("sjdbShiftRight", u8p), ("sjdbStrand", u8p),
    ]


class Params(C.Structure):
    _fields_ = [
        ("readNmates", C.c_uint32),
        ("seedSearchStartLmax", C.c_uint32), ("seedSearchStartLmaxOverLread", C.c_double),
        ("seedSearchLmax", C.c_uint32), ("seedMultimapNmax", C.c_uint32), ("seedPerReadNmax", C.c_uint32),
        ("seedPerWindowNmax", C.c_uint32), ("seedSplitMin", C.c_uint32), ("seedMapMin", C.c_uint32), ("maxNsplit", C.c_uint32),
        ("winAnchorMultimapNmax", C.c_uint32), ("winBinNbits", C.c_uint32), ("winAnchorDistNbins", C.c_uint32),
        ("winFlankNbins", C.c_uint32), ("winBinChrNbits", C.c_uint32), ("winBinN", C.c_uint64),
        ("alignWindowsPerReadNmax", C.c_uint32), ("alignTranscriptsPerWindowNmax", C.c_uint32), ("alignTranscriptsPerReadNmax", C.c_uint32),
        ("alignIntronMin", C.c_uint64), ("alignIntronMax", C.c_uint64), ("alignMatesGapMax", C.c_uint64),
        ("alignSJoverhangMin", C.c_uint32), ("alignSJDBoverhangMin", C.c_uint32),
        ("alignSJstitchMismatchNmax", C.c_int32 * 4),
        ("alignSplicedMateMapLmin", C.c_uint32), ("alignSplicedMateMapLminOverLmate", C.c_double),
        ("alignEndsTypeExt", (C.c_uint8 * 2) * 2),
        ("alignEndsProtrudeNbasesMax", C.c_int32),
        ("alignEndsProtrudeConcordantPair", C.c_uint8), ("alignSoftClipAtReferenceEnds", C.c_uint8),
        ("alignInsertionFlushRight", C.c_uint8), ("outFilterIntronStrandsRemoveInconsistent", C.c_uint8),
        ("outFilterIntronMotifs", C.c_uint8), ("outSAMstrandFieldIntronMotif", C.c_uint8),
        ("chimSegmentMinPositive", C.c_uint8), ("outFilterBySJoutStage", C.c_uint8),
        ("scoreGap", C.c_int32), ("scoreGapNoncan", C.c_int32), ("scoreGapGCAG", C.c_int32), ("scoreGapATAC", C.c_int32),
        ("scoreDelOpen", C.c_int32), ("scoreDelBase", C.c_int32), ("scoreInsOpen", C.c_int32), ("scoreInsBase", C.c_int32),
        ("scoreStitchSJshift", C.c_int32), ("sjdbScore", C.c_int32),
        ("scoreGenomicLengthLog2scale", C.c_double),
        ("outFilterMultimapScoreRange", C.c_int32),
        ("outFilterMismatchNoverLmax", C.c_double),
        ("outFilterMatchNmin", C.c_uint32),
        ("resultSelect", C.c_uint32),
        ("chimSegmentMin", C.c_uint32), ("chimSegmentReadGapMax", C.c_uint32),
    ]


class Batch(C.Structure):
    _fields_ = [("nReads", C.c_uint32), ("bases", u8p), ("readOffset", u64p), ("mate1Length", u16p), ("mmMaxTotal", u16p)]


class ReadResult(C.Structure):
    _fields_ = [("status", C.c_uint32), ("nW", C.c_uint32), ("nTr", C.c_uint32), ("trOffset", C.c_uint32),
                ("trBest", C.c_int32), ("maxScoreMate", C.c_int32 * 2), ("unmappedLength", C.c_uint32)]


class Transcript(C.Structure):
    _fields_ = [("iW", C.c_uint32), ("exonOffset", C.c_uint32), ("nExons", C.c_uint16),
                ("rStart", C.c_uint16), ("rLength", C.c_uint16), ("roStart", C.c_uint16),
                ("Str", C.c_uint8), ("roStr", C.c_uint8), ("iFrag", C.c_int8), ("sjMotifStrand", C.c_uint8),
                ("Chr", C.c_uint32), ("gStart", C.c_uint64), ("gLength", C.c_uint64), ("maxScore", C.c_int32),
                ("nMatch", C.c_uint32), ("nMM", C.c_uint32), ("mappedLength", C.c_uint32),
                ("nGap", C.c_uint32), ("lGap", C.c_uint32), ("nDel", C.c_uint32), ("lDel", C.c_uint32), ("nIns", C.c_uint32), ("lIns", C.c_uint32),
                ("nUnique", C.c_uint16), ("nAnchor", C.c_uint16), ("intronMotifs", C.c_uint16 * 3), ("pad0", C.c_uint16), ("pad1", C.c_uint32)]


class Exon(C.Structure):
    _fields_ = [("G", C.c_uint64), ("R", C.c_uint16), ("L", C.c_uint16), ("sjA", C.c_int32), ("iFrag", C.c_uint8),
                ("canonSJ", C.c_int8), ("sjAnnot", C.c_uint8), ("sjStr", C.c_uint8), ("shiftSJ", C.c_uint16 * 2), ("pad0", C.c_uint32), ("pad1", C.c_uint32)]


class Results(C.Structure):
    _fields_ = [("reads", C.POINTER(ReadResult)),
                ("tr", C.POINTER(Transcript)), ("trCapacity", C.c_uint64), ("trCount", C.c_uint64),
                ("ex", C.POINTER(Exon)), ("exCapacity", C.c_uint64), ("exCount", C.c_uint64),
                ("msSeed", C.c_float), ("msWindows", C.c_float), ("msStitch", C.c_float), ("msTotalDevice", C.c_float)]


class ResultBuffers:
    """Caller-owned result arrays for one batch."""

    def __init__(self, n_reads, tr_cap=None, ex_cap=None):
        tr_cap = tr_cap or max(1024, n_reads * 24)
        ex_cap = ex_cap or tr_cap * 4
        self.reads = (ReadResult * n_reads)()
        self.tr = (Transcript * tr_cap)()
        self.ex = (Exon * ex_cap)()
        self.res = Results()
        self.res.reads = C.cast(self.reads, C.POINTER(ReadResult))
        self.res.tr = C.cast(self.tr, C.POINTER(Transcript)); self.res.trCapacity = tr_cap
        self.res.ex = C.cast(self.ex, C.POINTER(Exon)); self.res.exCapacity = ex_cap

    def as_bytes(self, n_reads):
        """(reads, transcripts, exons) raw bytes of the filled part -- for bit-exact comparisons."""
        r = self.res
        return (bytes(memoryview(self.reads))[:n_reads * C.sizeof(ReadResult)],
                bytes(memoryview(self.tr))[:r.trCount * C.sizeof(Transcript)],
                bytes(memoryview(self.ex))[:r.exCount * C.sizeof(Exon)])


def map_in_pieces(map_one, batch, bufs, max_reads):
    """A batch larger than an engine context takes (e.g. the re-mapping batch of --waspOutputMode, where a read over a dense SNV cluster has hundreds of
    copies): mapped in pieces of at most max_reads reads through map_one(piece, piece_bufs); the results are appended into bufs as if it had been one call."""
    n, done, tr_n, ex_n = batch.nReads, 0, 0, 0
    ro = C.cast(batch.readOffset, C.c_void_p).value; m1 = C.cast(batch.mate1Length, C.c_void_p).value; mm = C.cast(batch.mmMaxTotal, C.c_void_p).value
    while done < n:
        k = min(max_reads, n - done)
        piece = Batch()
        piece.nReads = k; piece.bases = batch.bases
        piece.readOffset = C.cast(ro + 8 * done, type(batch.readOffset)); piece.mate1Length = C.cast(m1 + 2 * done, type(batch.mate1Length)); piece.mmMaxTotal = C.cast(mm + 2 * done, type(batch.mmMaxTotal))
        pb = ResultBuffers(k, tr_cap=max(1024, bufs.res.trCapacity // 2))
        map_one(piece, pb)
        if tr_n + pb.res.trCount > bufs.res.trCapacity or ex_n + pb.res.exCount > bufs.res.exCapacity:
            raise RuntimeError("map_in_pieces: result buffers too small (-3)")
        for i in range(k):
            bufs.reads[done + i] = pb.reads[i]
            bufs.reads[done + i].trOffset += tr_n
        C.memmove(C.byref(bufs.tr, tr_n * C.sizeof(Transcript)), pb.tr, pb.res.trCount * C.sizeof(Transcript))
        for i in range(pb.res.trCount):
            bufs.tr[tr_n + i].exonOffset += ex_n
        C.memmove(C.byref(bufs.ex, ex_n * C.sizeof(Exon)), pb.ex, pb.res.exCount * C.sizeof(Exon))
        tr_n += pb.res.trCount; ex_n += pb.res.exCount; done += k
    bufs.res.trCount = tr_n; bufs.res.exCount = ex_n


def _need(path):
    if not os.path.isfile(path):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (or `make`) first; "
                           "there is no Python/CPU fallback for the hot path" % path)
    return path


_host = None
_engine = None


def host_lib():
    global _host
    if _host is None:
        L = C.CDLL(_need(HOST_PATH))
        L.sah_create.restype = C.c_void_p
        L.sah_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_int]
        L.sah_genome.restype = C.POINTER(Genome); L.sah_genome.argtypes = [C.c_void_p]
        L.sah_params.restype = C.POINTER(Params); L.sah_params.argtypes = [C.c_void_p]
        L.sah_batch_reads.restype = C.c_uint64; L.sah_batch_reads.argtypes = [C.c_void_p]
        L.sah_genome_load_seconds.restype = C.c_double; L.sah_genome_load_seconds.argtypes = [C.c_void_p]
        L.sah_next_batch.restype = C.c_int; L.sah_next_batch.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Batch)]
        L.sah_emit.restype = C.c_int; L.sah_emit.argtypes = [C.c_void_p, C.POINTER(Results)]
        L.sah_wasp_batch.restype = C.c_int; L.sah_wasp_batch.argtypes = [C.c_void_p, C.POINTER(Results), C.POINTER(Batch)]
        L.sah_wasp_results.restype = C.c_int; L.sah_wasp_results.argtypes = [C.c_void_p, C.POINTER(Results), C.POINTER(Results)]
        L.sah_merged_batch.restype = C.c_int; L.sah_merged_batch.argtypes = [C.c_void_p, C.POINTER(Batch)]
        L.sah_emit_merged.restype = C.c_int; L.sah_emit_merged.argtypes = [C.c_void_p, C.POINTER(Results), C.POINTER(Results)]
        L.sah_finish.restype = C.c_int; L.sah_finish.argtypes = [C.c_void_p]
        L.sah_error.restype = C.c_char_p; L.sah_error.argtypes = [C.c_void_p]
        L.sah_destroy.restype = None; L.sah_destroy.argtypes = [C.c_void_p]
        L.sah_in_pass1.restype = C.c_int; L.sah_in_pass1.argtypes = [C.c_void_p]
        L.sah_pass1_end.restype = C.c_int; L.sah_pass1_end.argtypes = [C.c_void_p]
        L.sah_insert_log.restype = C.c_char_p; L.sah_insert_log.argtypes = [C.c_void_p]
        L.sah_next_phase.restype = C.c_int; L.sah_next_phase.argtypes = [C.c_void_p]
        L.sah_novel_junctions.restype = C.c_uint64; L.sah_novel_junctions.argtypes = [C.c_void_p, C.POINTER(u64p), C.POINTER(u64p)]
        _host = L
    return _host


def engine_lib():
    """The HIP engine.  Raises if the extension has not been built: no fallback."""
    global _engine
    if _engine is None:
        L = C.CDLL(_need(ENGINE_PATH))
        L.staramd_create.restype = C.c_int
        L.staramd_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Genome), C.POINTER(Params), C.c_uint32, C.c_uint64]
        L.staramd_update_index.restype = C.c_int
        L.staramd_update_index.argtypes = [C.c_void_p, C.POINTER(Genome), C.POINTER(Params)]
        L.staramd_set_novel_junctions.restype = C.c_int
        L.staramd_set_novel_junctions.argtypes = [C.c_void_p, u64p, u64p, C.c_uint64, C.c_uint32]
        L.staramd_map_batch.restype = C.c_int
        L.staramd_map_batch.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Results)]
        L.staramd_map_resident.restype = C.c_int
        L.staramd_map_resident.argtypes = [C.c_void_p, C.POINTER(Results)]
        L.staramd_destroy.restype = None; L.staramd_destroy.argtypes = [C.c_void_p]
        L.staramd_last_error.restype = C.c_char_p
        L.staramd_get_timings.restype = C.c_int
        L.staramd_get_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        L.staramd_get_counters.restype = C.c_int
        L.staramd_get_counters.argtypes = [C.c_void_p, u64p, C.c_int]
        if hasattr(L, "staramd_launch_count"):
            L.staramd_launch_count.restype = C.c_uint64; L.staramd_launch_count.argtypes = [C.c_void_p]
        _engine = L
    return _engine


class HostRun:
    """One alignReads run on the host side: STAR-style argv in, SAM/SJ/Log files out."""

    def __init__(self, argv):
        L = host_lib()
        args = [b"star_amd"] + [a.encode() if isinstance(a, str) else a for a in argv]
        arr = (C.c_char_p * len(args))(*args)
        err = C.create_string_buffer(4096)
        self.h = L.sah_create(len(args), arr, err, 4096)
        if not self.h:
            raise RuntimeError(err.value.decode())
        self.L = L
        self.genome = L.sah_genome(self.h)
        self.params = L.sah_params(self.h)

    def next_batch(self, max_reads):
        b = Batch()
        n = self.L.sah_next_batch(self.h, max_reads, C.byref(b))
        if n < 0:
            raise RuntimeError(self.L.sah_error(self.h).decode())
        return b if n > 0 else None

    def merged_batch(self):
        """--peOverlapNbasesMin > 0: the pairs of the current batch whose mates overlap, merged into single reads (None when there are none).
        Map it with the same engine and pass its results to emit() as `merged`."""
        b = Batch()
        n = self.L.sah_merged_batch(self.h, C.byref(b))
        return b if n > 0 else None

    def wasp_batch(self, results):
        """--waspOutputMode SAMtag: the allele-swapped reads built from the results of the current batch (None when there are none); map them with the same
        engine and hand their results to wasp_results() before emit()."""
        b = Batch()
        n = self.L.sah_wasp_batch(self.h, C.byref(results), C.byref(b))
        return b if n > 0 else None

    def wasp_results(self, results, wasp_results):
        if self.L.sah_wasp_results(self.h, C.byref(results), C.byref(wasp_results) if wasp_results is not None else None) != 0:
            raise RuntimeError(self.L.sah_error(self.h).decode())

    def emit(self, results, merged=None):
        rc = self.L.sah_emit(self.h, C.byref(results)) if merged is None else self.L.sah_emit_merged(self.h, C.byref(results), C.byref(merged))
        if rc != 0:
            raise RuntimeError(self.L.sah_error(self.h).decode())

    def in_pass1(self):
        """True while the 1st pass of --twopassMode Basic is running (map every batch, then call pass1_end())."""
        return bool(self.L.sah_in_pass1(self.h))

    def pass1_end(self):
        """Write _STARpass1/, insert the junctions into the host index, rewind the reads; the engine must then be
        given the new index (Engine.update_index(run.genome, run.params))."""
        if self.L.sah_pass1_end(self.h) != 0:
            raise RuntimeError(self.L.sah_error(self.h).decode())

    def next_phase(self):
        """After the last batch: 0 = done (finish()); 1 = the index was rewritten by junction insertion (engine.update_index(run.genome,
        run.params), then map every batch again); 2 = the whitelist of the 2nd BySJout stage was built
        (engine.set_novel_junctions(*run.novel_junctions()), then map every batch again)."""
        r = self.L.sah_next_phase(self.h)
        if r < 0:
            raise RuntimeError(self.L.sah_error(self.h).decode())
        return r

    def novel_junctions(self):
        a, b = u64p(), u64p()
        n = self.L.sah_novel_junctions(self.h, C.byref(a), C.byref(b))
        return a, b, n

    def finish(self):
        if self.L.sah_finish(self.h) != 0:
            raise RuntimeError(self.L.sah_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.sah_destroy(self.h)
            self.h = None


class Engine:
    """The MI355X engine context (one per GPU)."""

    def __init__(self, genome_p, params_p, device=0, max_reads=65536, max_bases=None):
        L = engine_lib()
        self.L = L
        self.ctx = C.c_void_p()
        self.max_reads = max_reads
        max_bases = max_bases or max_reads * 660
        rc = L.staramd_create(C.byref(self.ctx), device, genome_p, params_p, max_reads, max_bases)
        if rc != 0:
            raise RuntimeError("staramd_create failed (%d): %s" % (rc, L.staramd_last_error().decode()))

    def update_index(self, genome_p, params_p):
        rc = self.L.staramd_update_index(self.ctx, genome_p, params_p)
        if rc != 0:
            raise RuntimeError("staramd_update_index failed (%d): %s" % (rc, self.L.staramd_last_error().decode()))

    def set_novel_junctions(self, start, end, n, stage=2):
        rc = self.L.staramd_set_novel_junctions(self.ctx, start, end, n, stage)
        if rc != 0:
            raise RuntimeError("staramd_set_novel_junctions failed (%d): %s" % (rc, self.L.staramd_last_error().decode()))

    def map_batch(self, batch, bufs):
        if batch.nReads > self.max_reads:        # more reads than the context was created for: in pieces
            return map_in_pieces(self.map_batch, batch, bufs, self.max_reads)
        rc = self.L.staramd_map_batch(self.ctx, C.byref(batch), C.byref(bufs.res))
        if rc != 0:
            raise RuntimeError("staramd_map_batch failed (%d): %s" % (rc, self.L.staramd_last_error().decode()))

    def map_resident(self, bufs):
        rc = self.L.staramd_map_resident(self.ctx, C.byref(bufs.res))
        if rc != 0:
            raise RuntimeError("staramd_map_resident failed (%d): %s" % (rc, self.L.staramd_last_error().decode()))

    def counters(self, n=40):
        out = (C.c_uint64 * n)()
        k = self.L.staramd_get_counters(self.ctx, out, n)
        return list(out)[:k]

    def timings(self):
        out = (C.c_float * 7)()
        k = self.L.staramd_get_timings(self.ctx, out, 7)
        return dict(zip(["seed", "windows", "order", "stitch_walk", "stitch_redecide", "gather", "total"], list(out)[:k]))

    def close(self):
        if self.ctx:
            self.L.staramd_destroy(self.ctx)
            self.ctx = C.c_void_p()


class device_sjdb_insertion:
    """Context manager: junction insertion (2-pass, --sjdbFileChrStartEnd / --sjdbGTFfile at the mapping stage) of every HostRun created inside
    runs through `fn_lib.fn_name` -- staramd_sjdb_insert of the engine library on a GPU box, or its plain-loop twin in the CPU tests
    (oracle/_build/libindex_emul.so: sjdb_emul_insert) -- instead of the host restatement.  Process-wide switch of the host library."""

    def __init__(self, lib_path=None, fn_name="staramd_sjdb_insert", device=0):
        self.lib = C.CDLL(lib_path or _need(ENGINE_PATH))
        self.fn = C.cast(getattr(self.lib, fn_name), C.c_void_p)
        self.device = device

    def __enter__(self):
        L = host_lib()
        L.sah_set_sjdb_device_fn.restype = None; L.sah_set_sjdb_device_fn.argtypes = [C.c_void_p, C.c_int]
        L.sah_set_sjdb_device_fn(self.fn, self.device)
        return self

    def __exit__(self, *a):
        host_lib().sah_set_sjdb_device_fn(None, 0)
        return False


# ---- the command-line front end as a function (include/star_amd_cli.h) ----

class CliHooks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("warmup_done", C.CFUNCTYPE(None, C.c_void_p)), ("exchange", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int))]


class CliReport(C.Structure):
    _fields_ = [("reads", C.c_uint64), ("wallMapping", C.c_double), ("timedReads", C.c_uint64), ("timedWall", C.c_double),
                ("genomeLoadSeconds", C.c_double), ("indexUploadSeconds", C.c_double), ("nDevices", C.c_int),
                ("deviceBusy", C.c_double * 16), ("deviceMs", C.c_double * 16), ("stageMs", C.c_double * 8), ("counters", C.c_uint64 * 64),
                ("parseBusy", C.c_double), ("emitBusy", C.c_double), ("batches", C.c_uint64), ("pass1Seconds", C.c_double), ("finishSeconds", C.c_double), ("nContexts", C.c_int), ("convertBusy", C.c_double), ("emitParts", C.c_double * 4), ("fastPaths", C.c_uint64 * 4), ("cpuSeconds", C.c_double * 8)]


def run_cli(argv, warmup_done=None, exchange=None, lib_path=None):
    """The whole front end (star_amd/csrc/host/cli_run.cpp) in this process: argv as on the command line; warmup_done() is called when the
    --benchWarmupReads reads are written and the pipeline is empty; exchange(handle, last) at the end of every mapping phase (cross-rank tables).
    Returns (exit code, CliReport)."""
    lib = C.CDLL(lib_path or os.environ.get("STARAMD_CLI_LIB") or _need(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libstaramd_cli.so")))      # (STARAMD_CLI_LIB: tests of bench.py's plumbing on a box without a GPU)
    lib.staramd_cli_main.restype = C.c_int
    lib.staramd_cli_main.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(CliHooks), C.POINTER(CliReport)]
    args = [b"star_amd"] + [a.encode() for a in argv]
    arr = (C.c_char_p * len(args))(*args)
    hooks = CliHooks()
    WD, EX = CliHooks._fields_[1][1], CliHooks._fields_[2][1]
    wd = WD(lambda u: warmup_done() if warmup_done else None)
    ex = EX(lambda u, h, last: int(exchange(h, last) or 0) if exchange else 0)
    hooks.user = None; hooks.warmup_done = wd; hooks.exchange = ex
    rep = CliReport()
    rc = lib.staramd_cli_main(len(args), arr, C.byref(hooks), C.byref(rep))
    return rc, rep


