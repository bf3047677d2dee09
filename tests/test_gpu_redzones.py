"""The red zones tests/conftest.py puts around every device buffer of the GPU suite: they must catch a write of one
element beyond either end (otherwise a green suite says nothing about where the kernels write), and the exact-fit
calls below — every output buffer sized to the element — must leave them intact."""
import numpy as np
import pytest

import conftest
from gpu_util import ctx

pytestmark = pytest.mark.gpu


def _redzoned(a):
    return hasattr(a, "redzones_intact")


def test_redzones_catch_one_element_either_side():
    g = ctx()
    a = g.empty(1000)
    if not _redzoned(a):
        pytest.skip("TSDR_TEST_REDZONES=0")
    src = g.to_device(np.ones(1001, np.float32))
    assert a.redzones_intact("self-test")
    g._ck(g.lib.tsdrgpu_copy(g.h, a.ptr, src.ptr, 1001 * 4))  # one float too many
    g.sync()
    assert not a.redzones_intact("self-test")
    assert any("above" in v for v in conftest._redzone_violations)
    conftest._redzone_violations.clear()
    assert a.redzones_intact("self-test")  # (re-armed)
    g._ck(g.lib.tsdrgpu_copy(g.h, a.ptr - 4, src.ptr, 4))  # one float in front
    g.sync()
    assert not a.redzones_intact("self-test")
    assert any("below" in v for v in conftest._redzone_violations)
    conftest._redzone_violations.clear()
    # what an out-of-bounds READ hands a kernel: NaN
    b = g.empty(4)
    g._ck(g.lib.tsdrgpu_copy(g.h, b.ptr, a.ptr + 1000 * 4, 16))  # four floats of the guard above `a`
    g.sync()
    assert np.all(np.isnan(b.download()))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4097, 262_145])
def test_exact_fit_outputs_of_the_elementwise_entry_points(n):
    """am_demod, the sample decode, frame -> RGB and the copies, every buffer sized to the element, at ragged sizes"""
    g = ctx()
    rng = np.random.default_rng(n)
    iq = rng.standard_normal(2 * n).astype(np.float32)
    d_iq, d_m = g.to_device(iq), g.empty(n)
    g.am_demod(d_iq, d_m, n)
    assert np.allclose(d_m.download(), np.hypot(iq[0::2], iq[1::2]), rtol=1e-6)  # (bit-exactness: test_gpu_demod_resample.py)
    for fmt, dt in (("int8", np.int8), ("uint8", np.uint8), ("int16", np.int16), ("uint16", np.uint16)):
        raw = rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, 2 * n, dtype=dt, endpoint=True)
        d_raw, d_out = g.to_device(raw, dt), g.empty(2 * n)
        g.decode_samples(d_raw, fmt, d_out, 2 * n)
        assert np.all(np.isfinite(d_out.download()))
    fr = rng.random(n).astype(np.float32)
    d_fr, d_rgb = g.to_device(fr), g.empty(n, np.int32)
    d_rgb.zero()
    g.frame_to_rgb(d_fr, d_rgb, n)
    g.sync()  # (values: test_gpu_extras.py; here only where they land — the fixture checks the guards after the test)


def test_library_redzone_mode_reports_an_overrun(tmp_path):
    """TSDRGPU_REDZONES (tsdrgpu_internal.h): the same for every allocation the LIBRARY makes.  A fresh process with the mode
    on allocates 100 bytes through the C ABI, copies 104 into them and frees: one report, naming the size and the edge."""
    import os
    import subprocess
    import sys
    log = tmp_path / "reports.txt"
    code = (
        "import ctypes as C\n"
        "from tempestsdr_amd import gpu\n"
        "g = gpu.TsdrGpu(0)\n"
        "a, b = C.c_void_p(), C.c_void_p()\n"
        "g._ck(g.lib.tsdrgpu_alloc(g.h, C.byref(a), 100)); g._ck(g.lib.tsdrgpu_alloc(g.h, C.byref(b), 4096))\n"
        "g._ck(g.lib.tsdrgpu_zero(g.h, b, 4096))\n"
        "g._ck(g.lib.tsdrgpu_copy(g.h, a, b, 100)); g.sync()\n"
        "g._ck(g.lib.tsdrgpu_free(g.h, a))\n"          # clean: no report
        "g._ck(g.lib.tsdrgpu_alloc(g.h, C.byref(a), 100))\n"
        "g._ck(g.lib.tsdrgpu_copy(g.h, a, b, 104)); g.sync()\n"
        "g._ck(g.lib.tsdrgpu_free(g.h, a)); g._ck(g.lib.tsdrgpu_free(g.h, b))\n"
        "g.close()\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TSDRGPU_REDZONES="2", TSDRGPU_REDZONE_LOG=str(log))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = log.read_text().splitlines() if log.exists() else []
    assert len(lines) == 1 and "4 bytes written above a device allocation of 100 bytes (first at byte +0" in lines[0], (lines, out.stderr[-500:])
    assert "RED ZONE VIOLATION" in out.stderr
    # =1 aborts the process at the violation
    env["TSDRGPU_REDZONES"] = "1"
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "RED ZONE VIOLATION" in out.stderr
