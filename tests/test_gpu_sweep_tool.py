"""tempestsdr_amd/tsdr_sweep — the detector's lag sweep over a recording as a C program on the tsdrgpu_* ABI (the C host of
SURVEY 8(e) row 1: one thread, one detector and one RCCL rank per device in ONE process; csrc/host/tsdr_sweep.c).  The test
box has one GPU, so what runs here is its one-device form (the reference's running mean: the engine's detector) and, with
--force-comm, the sums + ncclAllReduce + finalize path on a one-rank communicator; both against the ORACLE
(frameratedetector.c:34-62,87-126; fft.c:49-64)."""
import json
import os
import subprocess

import numpy as np
import pytest

from tempestsdr_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tempestsdr_amd", "tsdr_sweep")


def _run(args):
    out = subprocess.run([TOOL] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def _oracle(orc, fs, x_iq, nwin):
    o = orc.Autocorr(fs)
    cap = orc.capture_size(fs)
    for k in range(nwin):
        o.run(orc.am_demod(x_iq[2 * k * cap:2 * (k + 1) * cap]))
    return o


@pytest.mark.parametrize("fs,mode,nwin", [(8_000_000, "640x480", 5), (25_000_000, "1024x768", 3)])
def test_sweep_tool_one_device_equals_the_oracle(orc, tmp_path, fs, mode, nwin):
    cap = orc.capture_size(fs)
    iq = synth.synth_iq(fs, mode, 60.0, nwin * cap + 1000, seed=0x5EED0040)
    path = tmp_path / "rec.f32"
    iq.tofile(path)
    o = _oracle(orc, fs, iq, nwin)
    want = (int(np.argmax(o.frame)), int(np.argmax(o.line)))
    # the exact detector: plots bit for bit
    plots = tmp_path / "plots.f64"
    d = _run([path, fs, "float", "--detector", "exact", "--plots", plots])
    assert d["windows"] == nwin and d["devices"] == [0] and d["windows_per_device"] == [nwin]
    got = np.fromfile(plots, np.float64)
    assert np.array_equal(got[:o.flen], o.frame) and np.array_equal(got[o.flen:], o.line)
    assert (d["frame_idx"], d["line_idx"]) == want
    assert d["frame_lag"] == o.flo + want[0] and d["line_lag"] == o.llo + want[1]
    assert d["framerate"] == pytest.approx(fs / (o.flo + want[0]), rel=1e-9)
    assert d["height"] == int(round((o.flo + want[0]) / (o.llo + want[1])))
    # the certified detector (default): the identical argmax, by certificate or by replay
    c = _run([path, fs])
    assert (c["frame_idx"], c["line_idx"]) == want and c["certified"] == 1
    if fs == 8_000_000:
        assert c["epoch_replayed_exact"] == 1  # the structural tie R[j] == R[N - j] inside this rate's frame-lag window
    # sums + ncclAllReduce (one rank) + finalize, the path N devices take
    f = _run([path, fs, "--force-comm"])
    assert "ncclAllReduce" in f["exchange"] and (f["frame_idx"], f["line_idx"]) == want and f["certified"] == 1
    assert f["epoch_replayed_exact"] == c["epoch_replayed_exact"]


def test_sweep_tool_decodes_narrow_recordings_like_rawfile(orc, tmp_path):
    """an int16 recording: decoded on the device exactly like TSDRPlugin_RawFile.c:249-252 (value / 32767.0)"""
    fs, nwin = 8_000_000, 3
    cap = orc.capture_size(fs)
    iq = synth.synth_iq(fs, "640x480", 60.0, nwin * cap, seed=0x5EED0041)
    raw = np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16)
    path = tmp_path / "rec.s16"
    raw.tofile(path)
    decoded = (raw.astype(np.float64) / 32767.0).astype(np.float32)
    o = _oracle(orc, fs, decoded, nwin)
    plots = tmp_path / "plots.f64"
    d = _run([path, fs, "int16", "--detector", "exact", "--plots", plots])
    got = np.fromfile(plots, np.float64)
    assert np.array_equal(got[:o.flen], o.frame) and np.array_equal(got[o.flen:], o.line)
    assert (d["frame_idx"], d["line_idx"]) == (int(np.argmax(o.frame)), int(np.argmax(o.line)))


def test_sweep_tool_reports_errors(tmp_path):
    path = tmp_path / "short.f32"
    np.zeros(1000, np.float32).tofile(path)
    out = subprocess.run([TOOL, str(path), "8000000"], capture_output=True, text=True)
    assert out.returncode == 1 and "less than one capture window" in out.stderr
    out = subprocess.run([TOOL, str(path), "8000000", "--devices", "99"], capture_output=True, text=True)
    assert out.returncode == 1
