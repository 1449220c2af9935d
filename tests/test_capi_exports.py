"""The C-ABI library loads without a GPU, exports every entry point include/star_amd.h declares, its struct layouts
agree with the ctypes mirror, and it FAILS LOUDLY (no CPU fallback) when no GPU is visible."""
import ctypes as C
import os
import re
import subprocess

import pytest

from util import ROOT, capi

HEADER = os.path.join(ROOT, "include", "star_amd.h")


def _declared():
    """every entry point of the engine's headers: the boundary itself (star_amd.h) and its optional companion (star_amd_async.h: upload ahead, the two halves of a batch)"""
    txt = open(HEADER).read() + open(os.path.join(ROOT, "include", "star_amd_async.h")).read()
    return sorted(set(re.findall(r"\b(staramd_[a-z_]+)\s*\(", txt)))


def test_engine_exports_every_declared_symbol():
    subprocess.check_call(["make", "-s", "engine"], cwd=ROOT)
    lib = C.CDLL(capi.ENGINE_PATH)
    names = _declared()
    assert {"staramd_create", "staramd_map_batch", "staramd_map_resident", "staramd_update_index", "staramd_destroy",
            "staramd_last_error", "staramd_get_counters", "staramd_get_timings", "staramd_prefetch_batch", "staramd_prefetch_cancel", "staramd_map_begin", "staramd_map_wait",
            "staramd_map_end"} <= set(names)
    for n in names:
        assert hasattr(lib, n), "libstaramd.so does not export " + n


def test_struct_layouts_match_ctypes_mirror(built):
    L = capi.host_lib()
    L.sah_sizeof.restype = C.c_uint64; L.sah_sizeof.argtypes = [C.c_int]
    mirrors = [capi.Genome, capi.Params, capi.Batch, capi.ReadResult, capi.Transcript, capi.Exon, capi.Results]
    for i, m in enumerate(mirrors):
        assert L.sah_sizeof(i) == C.sizeof(m), (i, m.__name__)
    assert C.sizeof(capi.Transcript) == 96 and C.sizeof(capi.Exon) == 32


def test_no_cpu_fallback_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: this test is about the GPU-less error path")
    subprocess.check_call(["make", "-s", "engine"], cwd=ROOT)
    L = capi.engine_lib()
    ctx = C.c_void_p()
    g = capi.Genome(); p = capi.Params()
    rc = L.staramd_create(C.byref(ctx), 0, C.byref(g), C.byref(p), 16, 0)
    assert rc == -2, rc                      # STARAMD_ERR_DEVICE
    assert b"no HIP device" in L.staramd_last_error() or b"hip" in L.staramd_last_error().lower()


def test_host_library_exports_every_declared_symbol(built):
    """include/star_amd_host.h (the job-level interface around the engine) against libstaramd_host.so"""
    txt = open(os.path.join(ROOT, "include", "star_amd_host.h")).read()
    names = sorted(set(re.findall(r"^\s*[a-z_A-Z0-9 \*]*?\b(sah_[a-z_0-9]+)\s*\(", txt, re.M)))
    assert len(names) >= 35, names
    L = capi.host_lib()
    for n in names:
        assert hasattr(L, n), "libstaramd_host.so does not export " + n
