"""CPU-side checks of the C ABI: the library loads and exports every symbol
include/tsdrgpu.h declares; no compute calls are made (no GPU here)."""
import ctypes as C
import os
import re

from tempestsdr_amd import build, gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tsdrgpu_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    build.build(verbose=False)
    lib = C.CDLL(gpu.LIB_PATH)
    names = declared("tsdrgpu.h")
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), n
    # and the Python mirror binds exactly the declared set
    assert set(gpu.exported_symbols()) == set(names)


def test_no_cpu_fallback():
    """Without a GPU the context constructor must fail loudly."""
    import pytest
    lib = gpu.load_library()
    h = C.c_void_p()
    rc = lib.tsdrgpu_create(C.byref(h), 0)
    if rc == 0:  # running on a GPU box
        lib.tsdrgpu_destroy(h)
        pytest.skip("a GPU is present")
    assert rc < 0 and not h.value
    with pytest.raises(gpu.TsdrGpuError):
        gpu.TsdrGpu(0)
