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


# ---------------------------------------------------------------------------
# round 5: trip 1's side store into the retention ring, and the super-bandwidth stitch on the three trips
# ---------------------------------------------------------------------------
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def emu5(emu):
    emu.emu_autocorr4_retain.restype = C.c_int
    emu.emu_autocorr4_retain.argtypes = [f32p, C.c_longlong, C.c_int, C.c_uint, f32p, f32p, f32p]
    emu.emu_sb_xcorr.restype = C.c_int
    emu.emu_sb_xcorr.argtypes = [f32p, f32p, f32p, f32p, C.c_uint, f32p, f32p, f32p, i32p]
    emu.emu_sb_stitch.restype = C.c_int
    emu.emu_sb_stitch.argtypes = [f32p, f32p, f32p, f32p, C.c_uint, i32p, f32p, f32p, f32p]
    return emu


def test_trip1_retains_the_reference_sum_of_squares(emu5):
    """k_ac_cols_retain: the ring holds am_demod's re*re + im*im with the reference's roundings (TSDRLibrary.c:244-262: two
    rounded products, their rounded sum — numpy's float32 arithmetic is exactly that) for every sample of every window: the
    correctly rounded root of it, which a replay takes on its loads, is the reference's demodulated sample.  The transform of
    the (hardware) roots is the autocorrelation."""
    logn1, cnt = 4, 2
    nh = 4096 << logn1
    n = 2 * nh
    rng = np.random.default_rng(5)
    stride = n + 11
    x = rng.standard_normal((cnt * stride + 8) * 2).astype(np.float32)
    x[:64] = 0.0                      # zeros (an int8 recording's)
    x[200:264] *= np.float32(1e-30)   # squares underflow
    work = np.zeros(cnt * nh * 2, np.float32)
    out = np.zeros(cnt * nh * 2, np.float32)
    ring = np.full(cnt * n, -1.0, np.float32)
    assert emu5.emu_autocorr4_retain(x, stride, cnt, nh, work, out, ring) == 0
    for b in range(cnt):
        seg = x[2 * b * stride:2 * (b * stride + n)]
        re, im = seg[0::2], seg[1::2]
        want = re * re + im * im  # float32 throughout
        assert want.dtype == np.float32
        assert np.array_equal(ring[b * n:(b + 1) * n], want)
        assert np.array_equal(np.sqrt(ring[b * n:(b + 1) * n]), np.sqrt(re * re + im * im))  # (what the replay demodulates to)
        corr = _want(x[2 * b * stride:], n, 1)
        assert np.max(np.abs(out[b * n:(b + 1) * n] - corr)) <= 5e-7 * corr[0]


def _absdiff(h):
    z = h[0::2].astype(np.float64) + 1j * h[1::2].astype(np.float64)
    m = np.abs(z)
    d = np.empty_like(m)
    d[0] = m[0] - m[0] * m[0]  # superbandwidth.c:70: prev seeded with |z0|^2
    d[1:] = m[1:] - m[:-1]
    return d


@pytest.mark.parametrize("logn1", [4, 5])
def test_stitch_alignment_on_three_trips_matches_numpy(emu5, logn1):
    """superb_bestfit for the pairs (0, i), i = 1..3 (superbandwidth.c:83-119, fft.c:69-93) as k_sb_cols<5> -> k_sb_rows<XCORR>
    -> k_sb_cols_argmax: the first maxima of |c_i|, c_i = ifft(conj(D0) Di), against numpy in f64."""
    bn = 4096 << logn1
    rng = np.random.default_rng(40 + logn1)
    base = (rng.standard_normal(bn + 5000) + 1j * rng.standard_normal(bn + 5000)) * (1.0 + (np.arange(bn + 5000) % 977 < 40) * 3.0)
    shifts = [0, 1234, 77, 4999]
    hops = []
    for sft in shifts:
        z = base[sft:sft + bn] + 0.05 * (rng.standard_normal(bn) + 1j * rng.standard_normal(bn))
        h = np.empty(2 * bn, np.float32)
        h[0::2], h[1::2] = z.real, z.imag
        hops.append(h)
    work = np.zeros(4 * bn * 2, np.float32)
    v = np.zeros(2 * bn * 2, np.float32)
    pval = np.full(4 * 4096, -2.0, np.float32)
    pidx = np.full(4 * 4096, -2, np.int32)
    T = emu5.emu_sb_xcorr(hops[0], hops[1], hops[2], hops[3], bn, work, v, pval, pidx)
    assert T > 0
    D = [np.fft.fft(_absdiff(h)) / bn for h in hops]
    slot_of = {1: 2, 2: 0, 3: 3}  # c_1 = |re| of array 1, c_2 = |re| of array 0, c_3 = |im| of array 1
    for i in (1, 2, 3):
        c = np.abs(np.fft.ifft(np.conj(D[0]) * D[i]) * bn)
        vals, idxs = pval[slot_of[i] * T:(slot_of[i] + 1) * T], pidx[slot_of[i] * T:(slot_of[i] + 1) * T]
        k = int(np.argmax(vals))
        best = np.flatnonzero(vals == vals[k])
        at = int(np.min(idxs[best]))
        assert at == int(np.argmax(c))
        assert abs(vals[k] - c.max()) <= 2e-5 * c.max()
    # the slot nobody reads (|im| of the single correlation) is rounding noise: the packing premise (c_i real)
    assert pval[1 * T:2 * T].max() <= 1e-4 * pval[0:T].max()


@pytest.mark.parametrize("logn1,offs", [(4, (0, 0, 0, 0)), (4, (0, 2468, 154, 9998)), (5, (0, 2 * 70000, 2, 2 * 131071))])
def test_stitch_transform_on_three_trips_matches_numpy(emu5, logn1, offs):
    """superb_ondataready's transforms (superbandwidth.c:135-146): every hop rotated left by its offset, transformed (1/M), the
    spectra concatenated in hop order, one unscaled inverse transform of 4M points — as k_sb_cols<6> -> k_sb_rows<STITCH> ->
    k_ac_cols<.., 0, true>(nh = 4M), in natural order."""
    per = 4096 << logn1
    rng = np.random.default_rng(60 + logn1)
    hops = [rng.standard_normal(2 * per).astype(np.float32) for _ in range(4)]
    off = np.array(offs, np.int32)
    work = np.zeros(4 * per * 2, np.float32)
    v = np.zeros(4 * per * 2, np.float32)
    out = np.zeros(4 * per * 2, np.float32)
    assert emu5.emu_sb_stitch(hops[0], hops[1], hops[2], hops[3], per, off, work, v, out) == 0
    spec = []
    for h, o in zip(hops, offs):
        z = h[0::2].astype(np.float64) + 1j * h[1::2].astype(np.float64)
        spec.append(np.fft.fft(np.roll(z, -(o // 2))) / per)
    want = np.fft.ifft(np.concatenate(spec)) * (4 * per)
    got = out[0::2] + 1j * out[1::2]
    assert np.max(np.abs(got - want)) <= 2e-6 * np.max(np.abs(want))

