"""The C-ABI shared library loads without a GPU and exports every symbol include/coltt_gpu.h declares; compute entry
points fail loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os

import pytest

import coltt_amd


def test_library_is_built_in_tree_and_exports_header():
    p = coltt_amd.lib_path()
    assert os.path.exists(p), "run __graft_entry__.build() first"
    L = coltt_amd.lib()
    syms = coltt_amd.declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert b"gfx950" in L.coltt_version()


def test_no_cpu_fallback_without_device():
    L = coltt_amd.lib()
    if L.coltt_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_uint64(0)
    rc = L.coltt_flat_create(C.c_uint32(8), 0, 0, C.byref(h))
    assert rc == -5 and b"no CPU fallback" in L.coltt_last_error()
    with pytest.raises(coltt_amd.ColttError):
        coltt_amd.Hnsw(8)


def test_argument_validation_needs_no_device():
    L = coltt_amd.lib()
    assert L.coltt_flat_create(C.c_uint32(0), 0, 0, C.byref(C.c_uint64(0))) == -1
    assert L.coltt_flat_create(C.c_uint32(8), 0, 9, C.byref(C.c_uint64(0))) == -4
    assert b"not support quantization type" in L.coltt_last_error()          # edge/vectorstore.go:79
    assert L.coltt_flat_destroy(C.c_uint64(12345)) == -3
    assert L.coltt_hnsw_search(C.c_uint64(1), None, C.c_size_t(0), 1, 0, None, None, None, None) == -3


def test_product_never_touches_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, fs in os.walk(os.path.join(root, "coltt_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "libcoltt_oracle" not in src \
                    and "orc_" not in src, f
