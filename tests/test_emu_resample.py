"""The closed-form per-pixel resampler math the HIP kernel runs
(tempestsdr_amd/csrc/resample_math.h), compiled for the host (tests/emu) and
checked bit-for-bit against the oracle's sequential loop — CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "emu", "emu.cpp")
    so = os.path.join(HERE, "emu", "libemu.so")
    hdr = os.path.join(HERE, "..", "tempestsdr_amd", "csrc", "resample_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src], check=True)
    lib = C.CDLL(so)
    lib.emu_resample_chunk.restype = C.c_uint
    lib.emu_resample_chunk.argtypes = [f32p, C.c_uint, C.c_double, C.c_double, C.c_double, C.c_double, f32p,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.emu_resample_chunk_up.restype = C.c_uint
    lib.emu_resample_chunk_up.argtypes = [f32p, C.c_uint, C.c_double, C.c_double, C.c_double, C.c_double, f32p]
    return lib


CASES = [(1.9963125e6, 1e6, [13333] * 6), (2.0, 1.0, [1000] * 5), (0.37, 1.0, [1000, 1013, 999, 5, 3, 1, 1, 2000]),
         (1.0, 1.0, [100] * 4), (0.999, 1, [777] * 5), (1.5, 1, [100, 101, 99] * 3), (3.25, 1, [500] * 4),
         (7.0, 1.0, [64] * 5), (0.05, 1.0, [7, 3, 50, 11, 200]), (2962 * 1125 * 60.0, 100e6, [166666] * 3),
         (0.5, 1.0, [10, 11, 12, 13]), (2.0, 1.0, [1, 1, 2, 3]), (507 * 525 * 60.0, 8e6, [13333] * 5),
         (800 * 525 * 60.0, 12.6e6, [21000] * 4), (1e-3, 1.0, [100, 5000, 77])]


@pytest.mark.parametrize("up,down,sizes", CASES)
def test_closed_form_equals_sequential_loop(orc, emu, up, down, sizes):
    rng = np.random.default_rng(5)
    rs = orc.Resampler()
    con, off = 0.0, 0.0
    for s in sizes:
        x = rng.random(s).astype(np.float32)
        want = rs.process(x, up, down)
        out = np.zeros(want.size + 4, np.float32)
        co, oo = C.c_double(), C.c_double()
        n = emu.emu_resample_chunk(x, s, up, down, off, con, out, C.byref(co), C.byref(oo))
        k = min(rs.last_emitted, want.size)
        assert n == want.size
        assert np.array_equal(out[:k], want[:k])
        assert np.all(out[k:n] == 0.0)  # pixels the reference's loop never stores
        assert (co.value, oo.value) == (rs.st.contrib, rs.st.offset)
        con, off = co.value, oo.value


@pytest.mark.parametrize("up,down,sizes", CASES + [(1.0000001, 1.0, [5000] * 3), (8.0, 1.0, [300, 301]), (2.5, 1.0, [1, 2, 1, 700]),
                                                   (1481 * 2 * 1125 * 60.0, 100e6, [166666] * 2), (4.0, 3.0, [999] * 4)])
def test_sample_parallel_form_equals_sequential_loop(orc, emu, up, down, sizes):
    """k_rs_area_up's lane scheme (one lane per input sample, neighbour values through a wave shift).  It is only
    dispatched for r >= 1, but it is exact for every ratio (the chain replay covers neighbours that did not fire)."""
    rng = np.random.default_rng(6)
    rs = orc.Resampler()
    con, off = 0.0, 0.0
    slow_total = 0
    for s in sizes:
        x = (rng.random(s).astype(np.float32) - np.float32(0.25))  # negative samples too (real-valued input)
        want = rs.process(x, up, down)
        out = np.full(want.size + 4, -7.0, np.float32)
        slow_total += emu.emu_resample_chunk_up(x, s, up, down, off, con, out)
        k = min(rs.last_emitted, want.size)
        assert np.array_equal(out[:k], want[:k])
        assert np.all(out[k:want.size] == 0.0)
        assert np.all(out[want.size:] == -7.0)
        con, off = rs.st.contrib, rs.st.offset
    if up / down >= 1.0:
        assert slow_total <= len(sizes)  # the chain replay is the exception when upsampling
