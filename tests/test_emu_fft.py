"""The three-trip autocorrelation kernels (tempestsdr_amd/csrc/fft4step.h) — the same source the GPU
runs — compiled for the host (tests/emu: a workgroup = OS threads, __shared__ = static, __syncthreads =
barrier) and checked against numpy's f64 transform of fft_autocorrelation's identity
answer == ifft(|fft(x[:N])|) (TempestSDR/src/fft.c:49-64).  CPU only.

Tolerance: float32 transforms of N points carry ~1e-7 of the largest term, R[0]; asserted 5e-7 * R[0]."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "emu", "emu_fft.cpp")
    so = os.path.join(HERE, "emu", "libemu_fft.so")
    deps = [src, os.path.join(HERE, "emu", "hipemu.h"), os.path.join(HERE, "..", "tempestsdr_amd", "csrc", "fft4step.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas", "-o", so, src], check=True)
    lib = C.CDLL(so)
    lib.emu_autocorr4.restype = C.c_int
    lib.emu_autocorr4.argtypes = [f32p, C.c_int, C.c_longlong, C.c_int, C.c_uint, f32p, f32p, C.c_int, C.c_int,
                                  C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    return lib


def _want(x, n, iq):
    if iq:
        seg = x[:2 * n].astype(np.float64)
        mag = np.sqrt(seg[0::2] ** 2 + seg[1::2] ** 2)
    else:
        mag = x[:n].astype(np.float64)
    return np.fft.ifft(np.abs(np.fft.fft(mag))).real


# log2(N1) = 4..7 covers every pass structure of the column kernel: one pass (16), a radix-2/4/8 first pass
# followed by a radix-16 pass; 2^8..2^10 add a second radix-16 pass (checked on the GPU: minutes here)
@pytest.mark.parametrize("logn1,iq,cnt", [(4, 0, 2), (4, 1, 1), (5, 1, 2), (6, 0, 1), (7, 1, 1)])
def test_three_trip_autocorrelation_matches_numpy(emu, logn1, iq, cnt):
    nh = 4096 << logn1
    n = 2 * nh
    rng = np.random.default_rng(100 + logn1)
    stride = n + 37  # odd: every second window starts on an 8-byte boundary only
    per = 2 if iq else 1
    x = rng.standard_normal((cnt * stride + 8) * per).astype(np.float32) if iq else rng.random(cnt * stride + 8).astype(np.float32)
    # a periodic component so that the correlation has structure besides R[0]
    period = n // 5 + 3
    comb = (np.arange(cnt * stride + 8) % period < period // 9).astype(np.float32)
    if iq:
        x[0::2] += comb
    else:
        x += comb
    work = np.zeros(cnt * nh * 2, np.float32)
    out = np.zeros(cnt * nh * 2, np.float32)
    assert emu.emu_autocorr4(x, iq, stride, cnt, nh, work, out, 0, -1, 0, 0, 0, 0) == 0
    for b in range(cnt):
        want = _want(x[per * b * stride:], n, iq)
        got = out[b * n:(b + 1) * n]
        assert np.max(np.abs(got - want)) <= 5e-7 * want[0]
        # the symmetric structure the detector relies on: R[j] == R[N-j]
        assert np.max(np.abs(got[1:] - got[:0:-1])) <= 1e-6 * want[0]


def test_three_trip_lag_window_filter(emu):
    """Trip 3 stores only the two lag windows (complex point m holds lags 2m, 2m+1) and point 0 (lag 0, the scale
    of the argmax certificate), except for the window that is kept whole for tsdrgpu_autocorr_last_corr."""
    logn1, cnt = 4, 2
    nh = 4096 << logn1
    n = 2 * nh
    rng = np.random.default_rng(7)
    x = rng.random(cnt * n).astype(np.float32)
    work = np.zeros(cnt * nh * 2, np.float32)
    full = np.zeros(cnt * nh * 2, np.float32)
    assert emu.emu_autocorr4(x, 0, n, cnt, nh, work, full, 0, -1, 0, 0, 0, 0) == 0
    lo0, hi0, lo1, hi1 = 20000, 31000, 17, 300
    part = np.full(cnt * nh * 2, -7.0, np.float32)
    assert emu.emu_autocorr4(x, 0, n, cnt, nh, work, part, 1, 1, lo0, hi0, lo1, hi1) == 0
    p0, f0 = part[:n].reshape(nh, 2), full[:n].reshape(nh, 2)
    keep = np.zeros(nh, bool)
    keep[lo0:hi0] = True
    keep[lo1:hi1] = True
    keep[0] = True
    assert np.array_equal(p0[keep], f0[keep]) and np.all(p0[~keep] == -7.0)
    assert np.array_equal(part[n:], full[n:])  # window 1 == full_b: stored whole
