"""ctypes loader for the CPU restatement (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by star_amd/."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle.so")


def build():
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-Wno-sign-compare", "-shared",
                           os.path.join(HERE, "star_oracle.cpp"), "-o", LIB])


class Oracle:
    N_COUNTERS = 16
    COUNTER_NAMES = ["nSAi", "nSAprobe", "nGcmp", "nSAenum", "nGstitchReread", "nGstitchSpan", "nSeeds", "nWindows",
                     "nWA", "nNodes", "nLeaves", "nStitchCalls", "nExtendCalls", "nTrOut", "maxNodesWin", "maxWAWin"]

    def __init__(self, genome_p, params_p):
        from star_amd import capi
        if not os.path.isfile(LIB):
            build()
        L = C.CDLL(LIB)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(capi.Genome), C.POINTER(capi.Params)]
        L.oracle_map_batch.restype = C.c_int
        L.oracle_map_batch.argtypes = [C.c_void_p, C.POINTER(capi.Batch), C.POINTER(capi.Results)]
        L.oracle_seed_table.restype = C.c_int
        L.oracle_seed_table.argtypes = [C.c_void_p, capi.u8p, C.c_uint64, C.c_uint64, capi.u64p, C.c_int]
        L.oracle_get_counters.argtypes = [C.c_void_p, capi.u64p, C.c_int]
        L.oracle_reset_counters.argtypes = [C.c_void_p]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_set_novel_junctions.restype = C.c_int
        L.oracle_set_novel_junctions.argtypes = [C.c_void_p, capi.u64p, capi.u64p, C.c_uint64, C.c_uint32]
        self.L = L
        self.h = L.oracle_create(genome_p, params_p)

    def update_index(self, genome_p, params_p):
        """the index was rewritten on the host (junction insertion): start over on the new arrays"""
        self.L.oracle_destroy(self.h)
        self.h = self.L.oracle_create(genome_p, params_p)

    def set_novel_junctions(self, start, end, n, stage=2):
        self.L.oracle_set_novel_junctions(self.h, start, end, n, stage)

    def map_batch(self, batch, bufs):
        rc = self.L.oracle_map_batch(self.h, C.byref(batch), C.byref(bufs.res))
        if rc != 0:
            raise RuntimeError("oracle_map_batch failed: %d" % rc)

    def counters(self):
        out = (C.c_uint64 * self.N_COUNTERS)()
        self.L.oracle_get_counters(self.h, out, self.N_COUNTERS)
        return dict(zip(self.COUNTER_NAMES, list(out)))

    def reset_counters(self):
        self.L.oracle_reset_counters(self.h)

    def close(self):
        if self.h:
            self.L.oracle_destroy(self.h)
            self.h = None
