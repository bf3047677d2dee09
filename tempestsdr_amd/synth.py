"""Synthetic raster IQ generator.

A monitor with total raster W_t x H_t at f_v Hz emits pixel k = floor(t*f_p),
f_p = W_t*H_t*f_v.  The sample at t = i/fs carries the amplitude of the pixel
it lands on: 8 vertical bars alternating 0.3/0.8 over the active area, a
16x16-pixel +-0.1 checkerboard in the lower half, 0.05 in the blanking
intervals, plus seeded uniform noise (sigma 0.02).  IQ = a*(cos phi, sin phi),
phi = 0.37*i rad, interleaved float32 (the RawFile plugin's "float" format,
TSDRPlugin_RawFile/src/TSDRPlugin_RawFile.c:174-176,242-244).

Noise is a counter-based hash of the absolute sample index, so any slice of
the stream can be generated independently (per block, per rank, on any device).
"""
import numpy as np

# (total_w, total_h, active_w, active_h) — VESA/CEA totals, JavaGUI VideoMode.java:29,47,97
MODES = {
    "640x480": (800, 525, 640, 480),
    "1024x768": (1344, 806, 1024, 768),
    "1920x1080": (2576, 1125, 1920, 1080),
    "3840x2160": (4400, 2250, 3840, 2160),
}

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _hash01(idx, seed):
    """splitmix64 finaliser of (seed + idx*golden) -> uniform [0,1) float64."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) * _GOLD + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def amplitude(x, y, mode):
    tw, th, aw, ah = mode
    bar = (x * 8) // aw
    a = np.where(bar % 2 == 0, 0.3, 0.8)
    check = (((x // 16) + (y // 16)) % 2) * 0.2 - 0.1
    a = a + np.where(y >= ah // 2, check, 0.0)
    active = (x < aw) & (y < ah)
    return np.where(active, a, 0.05)


def synth_iq(fs, mode, fv, nsamples, start=0, seed=0x5EED0000, noise=0.02, dtype=np.float32):
    """Interleaved I,Q float32 array of 2*nsamples values for samples
    [start, start+nsamples) of the stream."""
    if isinstance(mode, str):
        mode = MODES[mode]
    tw, th, _, _ = mode
    i = np.arange(start, start + nsamples, dtype=np.int64)
    f_p = tw * th * fv
    k = np.floor(i.astype(np.float64) * (f_p / fs)).astype(np.int64)
    x = k % tw
    y = (k // tw) % th
    a = amplitude(x, y, mode)
    if noise:
        a = a + (_hash01(i, seed) - 0.5) * (noise * np.sqrt(12.0))
    phi = 0.37 * i.astype(np.float64)
    out = np.empty(2 * nsamples, dtype)
    out[0::2] = a * np.cos(phi)
    out[1::2] = a * np.sin(phi)
    return out
