"""CPU-side checks of the tsdr_* drop-in (tempestsdr_amd/libTSDRLibrary.so):
symbol set, error behaviour (codes + texts as in TSDRLibrary.c) and that it
fails loudly without a GPU.  No compute is done here."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import host_util as hu
from tempestsdr_amd import build


@pytest.fixture(scope="module")
def libs():
    build.build(verbose=False)
    return hu.load(), hu.build_test_plugin()


def test_exports_exactly_the_reference_tsdr_symbols(libs):
    out = subprocess.run(["nm", "-D", "--defined-only", hu.LIB], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    ours = [s for s in exported if s.startswith("tsdr_")]
    assert ours == sorted(hu.TSDR_SYMBOLS)  # == `nm -D Release/dlls/LINUX/X64/libTSDRLibrary.so | grep tsdr_`
    # everything else the library exports is the extension API of include/TSDRLibraryExt.h, under its own prefix
    assert [s for s in exported if not s.startswith("tsdr_")] == ["tsdrx_get_stats", "tsdrx_readasync_rgb"]
    ref = "/root/reference/Release/dlls/LINUX/X64/libTSDRLibrary.so"
    if os.path.exists(ref):
        o = subprocess.run(["nm", "-D", "--defined-only", ref], capture_output=True, text=True, check=True).stdout
        theirs = sorted(l.split()[-1] for l in o.splitlines() if " T tsdr_" in l)
        assert ours == theirs


def test_static_archive_has_the_api(libs):
    a = os.path.join(os.path.dirname(hu.LIB), "libTSDRLibrary.a")
    out = subprocess.run(["nm", a], capture_output=True, text=True, check=True).stdout
    for s in hu.TSDR_SYMBOLS:
        assert f" T {s}" in out
    # ... and nothing else: the internals (engine_run, plugin_host_load, tsdr_set_error, drop_shift_with ...) are local to
    # the archive's one object, so a host that links it statically (JavaGUI/jni/makefile:122) cannot collide with them
    glob = sorted(l.split()[-1] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "TDBRW")
    assert glob == sorted(list(hu.TSDR_SYMBOLS) + ["tsdrx_get_stats", "tsdrx_readasync_rgb"])


def test_static_archive_links_into_a_host(libs, tmp_path):
    """the JNI shim's way of using the library: link libTSDRLibrary.a (+ -ltsdrgpu) into the host's own object"""
    src = tmp_path / "host.c"
    src.write_text('#include "TSDRLibrary.h"\n'
                   'int engine_run(void) { return 7; }  /* a host symbol with the name of one of our internals */\n'
                   'int main(void) { tsdr_lib_t *t = 0; tsdr_init(&t, 0, 0, 0); if (!t) return 1; tsdr_free(&t); return engine_run() - 7; }\n')
    d = os.path.dirname(hu.LIB)
    exe = tmp_path / "host"
    subprocess.run(["gcc", "-I" + os.path.join(hu.ROOT, "include"), str(src), os.path.join(d, "libTSDRLibrary.a"), "-L" + d, "-ltsdrgpu",
                    "-Wl,-rpath," + d, "-lpthread", "-ldl", "-lm", "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_error_paths_match_reference(libs):
    s = hu.Session()
    lib = s.lib
    assert lib.tsdr_isrunning(s.h) == 0
    assert s.err() is None
    # TSDRLibrary.c:473-474
    assert lib.tsdr_readasync(s.h, s._cbs[0], None) == 1
    assert s.err() == "Please load a working plugin first!"
    # TSDRLibrary.c:426
    assert lib.tsdr_unloadplugin(s.h) == 1
    assert s.err() == "No plugin has been loaded so it can't be unloaded"
    # TSDRLibrary.c:446-447
    assert lib.tsdr_loadplugin(s.h, b"/nonexistent/plugin.so", b"") == 7
    assert "cannot be loaded" in s.err()
    # a shared object that is not a plugin: TSDRLibrary.c:450
    assert lib.tsdr_loadplugin(s.h, hu.LIB.encode(), b"") == 1
    assert s.err() == "The selected library is not a valid TSDR plugin!"
    # TSDRLibrary.c:553-554
    assert lib.tsdr_setresolution(s.h, 0, 60.0) == 2
    assert lib.tsdr_setresolution(s.h, 525, -1.0) == 2
    assert lib.tsdr_setresolution(s.h, 525, 60.0) == 0 and s.err() is None
    # TSDRLibrary.c:569
    assert lib.tsdr_motionblur(s.h, 1.5) == 2 and lib.tsdr_motionblur(s.h, 0.5) == 0
    # TSDRLibrary.c:605-606,614-615
    assert lib.tsdr_setparameter_int(s.h, 9, 1) == 8 and lib.tsdr_setparameter_int(s.h, -1, 1) == 8
    assert lib.tsdr_setparameter_int(s.h, 8, 1) == 0
    assert lib.tsdr_setparameter_double(s.h, 2, 1.0) == 8 and lib.tsdr_setparameter_double(s.h, 1, 1.0) == 0
    # TSDRLibrary.c:182: no plugin
    assert lib.tsdr_getsamplerate(s.h) == 1
    # tsdr_sync bounds, TSDRLibrary.c:583-596
    assert lib.tsdr_sync(s.h, 0, 1) == 0
    assert lib.tsdr_sync(s.h, 10_000, hu_dir("UP")) == 2
    assert lib.tsdr_sync(s.h, -1, hu_dir("LEFT")) == 2
    assert lib.tsdr_stop(s.h) == 0  # not running: OK (TSDRLibrary.c:214)
    s.close()
    assert not s.h.value


def hu_dir(name):
    return {"CUSTOM": 0, "UP": 1, "DOWN": 2, "LEFT": 3, "RIGHT": 4}[name]


def test_plugin_parameter_errors_and_loud_gpu_failure(libs, tmp_path):
    lib, plugin = libs
    s = hu.Session()
    # the plugin's own error text is surfaced (TSDRLibrary.c:456-460)
    assert s.lib.tsdr_loadplugin(s.h, plugin.encode(), b"nothing") == 4
    assert "usage:" in s.err()
    f = tmp_path / "iq.bin"
    np.zeros(4096, np.float32).tofile(f)
    assert s.lib.tsdr_loadplugin(s.h, plugin.encode(), f"{f} 8000000 1024 0".encode()) == 0
    assert s.lib.tsdr_getsamplerate(s.h) == 0
    # resolution never set: width/height invalid -> TSDR_WRONG_VIDEOPARAMS (TSDRLibrary.c:489-491)
    assert s.lib.tsdr_readasync(s.h, s._cbs[0], None) == 2
    assert s.lib.tsdr_setresolution(s.h, 525, 60.0) == 0
    import torch  # only to learn whether this box has a GPU
    if not torch.cuda.is_available():
        # no GPU here: the library must refuse to run rather than fall back to a CPU path
        assert s.lib.tsdr_readasync(s.h, s._cbs[0], None) == 6
        assert "no CPU path" in s.err()
        assert s.lib.tsdr_isrunning(s.h) == 0
    assert s.lib.tsdr_unloadplugin(s.h) == 0
    s.close()


def test_reference_rawfile_plugin_loads(libs):
    raw = os.path.join(os.path.dirname(hu.ROOT + "/x"), "oracle", "_ref", "libTSDRPlugin_RawFile.so")
    if not os.path.exists(raw):
        pytest.skip("oracle/_ref not built")
    s = hu.Session()
    assert s.lib.tsdr_loadplugin(s.h, raw.encode(), b"") == 4  # RawFile: "File name was not specified..."
    assert "File name" in s.err() or "Sample rate" in s.err()
    assert s.lib.tsdr_loadplugin(s.h, raw.encode(), b"/tmp/x.bin 8000000 float") == 0
    assert s.lib.tsdr_getsamplerate(s.h) == 0
    s.close()


def test_sweep_tool_is_built_and_has_no_cpu_path(libs, tmp_path):
    """tempestsdr_amd/tsdr_sweep (csrc/host/tsdr_sweep.c): the multi-GPU lag sweep's C host; without a HIP device it says so"""
    tool = os.path.join(os.path.dirname(hu.LIB), "tsdr_sweep")
    assert os.access(tool, os.X_OK)
    out = subprocess.run([tool], capture_output=True, text=True)
    assert out.returncode == 2 and "usage:" in out.stderr
    import torch
    if not torch.cuda.is_available():
        f = tmp_path / "rec.f32"
        np.zeros(2 * 450909 + 16, np.float32).tofile(f)
        out = subprocess.run([tool, str(f), "8000000"], capture_output=True, text=True)
        assert out.returncode == 1 and "no usable HIP device" in out.stderr


def test_bench_help_renders():
    """argparse formats every help string with %: a bare percent sign in one of them breaks `bench.py --help` (it did, twice)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "--seconds" in out.stdout, out.stderr[-500:]
