"""The REAL caller of the upper boundary on top of the drop-in: the reference's JNI shim (JavaGUI/jni/TSDRLibraryNDK.c:168-429),
compiled from its source and linked the way its makefile links it (JavaGUI/jni/makefile:122: shim + static libTSDRLibrary.a) —
with OUR libTSDRLibrary.a + libtsdrgpu.so (oracle/Makefile: _ref/libTSDRLibraryNDK_ours.so; the image has no JDK, so the shim
sees oracle/jni_stub/jni.h).  tests/jni/fake_jvm.c plays the JVM: it looks the natives up by their JNI names and replays
martin.tempest.core.TSDRLibrary's call sequence init -> loadPlugin -> setResolution -> ... -> nativeStart -> stop -> unloadPlugin
-> free.  What arrives in the Java object's int[] through SetIntArrayRegion + notifyCallbacks must be oracle frame -> oracle RGB
(the RGB loop that runs here IS the reference's: TSDRLibraryNDK.c:222-276), what arrives in its double[] the oracle's plots."""
import os
import struct
import subprocess

import numpy as np
import pytest

import host_util as hu
from tempestsdr_amd import synth

ROOT = hu.ROOT
SHIM = os.path.join(ROOT, "oracle", "_ref", "libTSDRLibraryNDK_ours.so")
JVM = os.path.join(ROOT, "tests", "jni", "fake_jvm")
RAWFILE = os.path.join(ROOT, "oracle", "_ref", "libTSDRPlugin_RawFile.so")
FS, H, FV = 8_000_000, 525, 60.0
BLOCK = 524288

NATIVES = ["init", "setBaseFreq", "loadPlugin", "nativeStart", "stop", "unloadPlugin", "free", "setGain", "setMotionBlur", "setResolution",
           "isRunning", "setInvertedColors", "sync", "setParam", "setParamDouble"]


def _need_shim():
    if not (os.path.exists(SHIM) and os.path.exists(JVM)):
        pytest.skip("oracle/_ref/libTSDRLibraryNDK_ours.so or tests/jni/fake_jvm not in the tree (built by build() where /root/reference exists)")


def parse_dump(path):
    b = open(path, "rb").read()
    pos, frames, plots, values, exceptions, tail = 0, [], [], [], [], None
    while pos < len(b):
        tag = struct.unpack_from("<i", b, pos)[0]
        pos += 4
        if tag == ord("F"):
            w, h = struct.unpack_from("<ii", b, pos)
            pos += 8
            frames.append((w, h, np.frombuffer(b, np.int32, w * h, pos).copy()))
            pos += 4 * w * h
        elif tag == ord("P"):
            pid, off, size, rate = struct.unpack_from("<iiiq", b, pos)
            pos += 20
            plots.append((pid, off, np.frombuffer(b, np.float64, size, pos).copy(), rate))
            pos += 8 * size
        elif tag == ord("V"):
            vid, a0, a1 = struct.unpack_from("<idd", b, pos)
            pos += 20
            values.append((vid, a0, a1))
        elif tag == ord("E"):
            n = struct.unpack_from("<i", b, pos)[0]
            cls = b[pos + 4:pos + 4 + n].decode()
            pos += 4 + n
            n = struct.unpack_from("<i", b, pos)[0]
            exceptions.append((cls, b[pos + 4:pos + 4 + n].decode(errors="replace")))
            pos += 4 + n
        elif tag == ord("Z"):
            tail = struct.unpack_from("<iiii", b, pos)
            pos += 16
        else:
            raise AssertionError(f"bad tag {tag} at {pos - 4}")
    return frames, plots, values, exceptions, tail


def run_jvm(plugin, params, nframes, dump, *extra, height=H, refresh=FV, timeout=60):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="2")
    out = subprocess.run([JVM, SHIM, plugin, params, str(height), repr(float(refresh)), str(nframes), str(dump)] + list(extra),
                         capture_output=True, text=True, timeout=timeout, env=env)
    return out


def test_shim_exports_the_natives_of_the_java_class():
    """libTSDRLibraryNDK (the reference's shim on top of our archive) resolves at load time — every tsdr_* it calls is in our
    libTSDRLibrary.a, every tsdrgpu_* in libtsdrgpu.so — and exports the 15 natives martin.tempest.core.TSDRLibrary declares."""
    _need_shim()
    import ctypes as C
    lib = C.CDLL(SHIM, mode=os.RTLD_NOW)
    for n in NATIVES:
        assert hasattr(lib, "Java_martin_tempest_core_TSDRLibrary_" + n), n


def test_shim_throws_the_java_exception_without_a_device(tmp_path):
    """No GPU: nativeStart must end in TSDRCannotOpenDeviceException through the shim's own error mapping (TSDRLibraryNDK.c:47-100),
    with the library's text — the drop-in has no CPU path to fall back to.  (With a GPU present the session tests below apply.)"""
    _need_shim()
    from tempestsdr_amd import gpu
    try:
        gpu.TsdrGpu(0).close()
        pytest.skip("a GPU is present")
    except gpu.TsdrGpuError:
        pass
    hu.build_test_plugin()
    p = tmp_path / "iq.f32"
    synth.synth_iq(FS, "640x480", FV, 4 * (BLOCK // 2), seed=1).tofile(p)
    out = run_jvm(hu.PLUGIN, f"{p} {FS} {BLOCK} 0", 3, tmp_path / "dump.bin", "timeout=5", timeout=30)
    frames, plots, values, exceptions, tail = parse_dump(tmp_path / "dump.bin")
    assert out.returncode == 4, out.stdout + out.stderr
    assert not frames and len(exceptions) == 1
    assert exceptions[0][0] == "martin/tempest/core/exceptions/TSDRCannotOpenDeviceException"
    assert "no CPU path" in exceptions[0][1]


def _oracle_rgb(orc, frames, inverted):
    """oracle frames -> the oracle's restatement of the shim's pixel loop (pinned to that loop: tests/test_oracle_vs_ref.py)"""
    out = []
    prev = np.zeros(frames[0].size, np.int32)
    for fr in frames:
        rgb = prev.copy()  # PIXEL_SPECIAL_VALUE_TRANSPARENT keeps what the previous frame left
        orc.lib.orc_frame_to_rgb(np.ascontiguousarray(fr, np.float32), rgb, fr.size, int(inverted))
        out.append(rgb)
        prev = rgb
    return out


def _oracle_frames(orc, iq, cfg):
    geo = orc.geometry(FS, H, FV)
    pix, _ = orc.demod_resample_stream(iq, geo)
    P = geo.width * geo.height
    pp = orc.PostProcess(geo)
    mb, lbs, aap, ash = cfg
    return geo, [pp.run(pix[k * P:(k + 1) * P].copy(), mb, 0.1, lbs, aap, ash, 0, 0) for k in range(pix.size // P)]


def _match_in_order(got, want):
    k, hits = 0, []
    for (_, _, a) in got:
        while k < len(want) and not np.array_equal(a, want[k]):
            k += 1
        assert k < len(want), f"delivered frame {len(hits)} (after oracle frame {hits[-1] if hits else -1}) is no oracle frame"
        hits.append(k)
        k += 1
    return hits


@pytest.fixture(scope="module")
def recording(tmp_path_factory):
    n = 16 * (BLOCK // 2)  # 0.52 s
    iq = synth.synth_iq(FS, "640x480", FV, n, seed=0x5EED0001)
    p = tmp_path_factory.mktemp("jni") / "cfg0.f32"
    iq.tofile(p)
    return str(p), iq


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,inverted", [((0.0, 0, 0, 0), 0), ((0.5, 1, 0, 1), 1)])
def test_java_session_through_the_shim_equals_the_oracle(orc, recording, tmp_path, cfg, inverted):
    """The GUI's session — RawFile plugin (the reference's binary), 8 MS/s, 640x480@60 -> 507 x 525 frames — through the natives: every
    int[] the Java object is handed equals oracle frame -> oracle RGB, in order from the first frame, none missing; the first plot
    pair is the oracle's (bit for bit: at this rate the detector replays its epoch in the reference's arithmetic)."""
    _need_shim()
    if not os.path.exists(RAWFILE):
        pytest.skip("oracle/_ref/libTSDRPlugin_RawFile.so not shipped")
    path, iq = recording
    mb, lbs, aap, ash = cfg
    geo, want_f = _oracle_frames(orc, iq, cfg)
    want = _oracle_rgb(orc, want_f, inverted)
    nframes = len(want) - 4  # the first pass over the file (the plugin loops at EOF; the seam is a partial block)
    extra = [f"blur={mb}", f"inverted={inverted}", f"param6={lbs}", f"param7={aap}", f"param0={ash}", "timeout=40"]
    out = run_jvm(RAWFILE, f"{path} {FS} float", nframes, tmp_path / "dump.bin", *extra)
    assert out.returncode == 0, out.stdout + out.stderr
    frames, plots, values, exceptions, tail = parse_dump(tmp_path / "dump.bin")
    assert not exceptions and tail is not None and tail[3] == 0
    assert len(frames) == nframes and all((w, h) == (geo.width, H) for (w, h, _) in frames)
    hits = _match_in_order(frames, want)
    assert hits == list(range(nframes)), hits
    # the frames are pictures, not constants (a wrong-but-equal pair of all-black arrays would pass otherwise)
    assert len(np.unique(frames[-1][2])) > 16
    fp = [p for p in plots if p[0] == 0]
    lp = [p for p in plots if p[0] == 1]
    if fp and lp:  # 0.2 s of stream holds the detector's first window (3.1/55 s); a plot pair needs it correlated before the stop
        calls = [v for v in values if v[0] == 2]  # VALUE_ID_AUTOCORRECT_FRAMES_COUNT with the first plot pair
        assert calls and calls[0][2] >= 1.0
        n = int(calls[0][2])
        ac = orc.Autocorr(FS)
        cap = orc.capture_size(FS)
        for k in range(n):
            ac.run(orc.am_demod(iq[2 * k * cap:2 * (k + 1) * cap]))
        assert (fp[0][1], fp[0][2].size, fp[0][3]) == (ac.flo, ac.flen, FS)
        assert np.array_equal(fp[0][2], ac.frame) and np.array_equal(lp[0][2], ac.line)
    assert "running_seen 1 running_after_stop 0" in out.stdout


@pytest.mark.gpu
def test_java_session_with_a_manual_sync_and_a_wrong_plugin(orc, recording, tmp_path):
    """sync(3, LEFT) half way (TSDRLibraryNDK.c:381-403: the enum's name() travels as a String): the frames before it are the
    oracle's, frames keep arriving after it; a plugin path that does not exist ends in TSDRIncompatiblePluginException (what the reference's loader
    answers when dlopen fails, TSDRPluginLoader.c:49-55, TSDRLibrary.c:446-447), not in a crash."""
    _need_shim()
    hu.build_test_plugin()
    path, iq = recording
    geo, want_f = _oracle_frames(orc, iq, (0.0, 0, 0, 0))
    want = _oracle_rgb(orc, want_f, 0)
    n = 12
    out = run_jvm(hu.PLUGIN, f"{path} {FS} {BLOCK} 6000", n, tmp_path / "dump.bin", "sync=3:LEFT", "timeout=40")
    assert out.returncode == 0, out.stdout + out.stderr
    frames, plots, values, exceptions, tail = parse_dump(tmp_path / "dump.bin")
    assert not exceptions and len(frames) == n
    first = _match_in_order(frames[:n // 2 - 1], want)
    assert first == list(range(len(first)))
    out = run_jvm(os.path.join(ROOT, "no_such_plugin.so"), "x", 1, tmp_path / "dump2.bin", "timeout=5")
    frames, plots, values, exceptions, tail = parse_dump(tmp_path / "dump2.bin")
    assert out.returncode == 4 and exceptions and exceptions[0][0] == "martin/tempest/core/exceptions/TSDRIncompatiblePluginException"
