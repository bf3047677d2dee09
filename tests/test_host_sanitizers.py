"""The host library's threads under ThreadSanitizer and AddressSanitizer + UBSan, on the CPU.

tests/sanitize/host_stress.c drives the tsdr_* API the way the reference's Java GUI does (one thread blocked in
tsdr_readasync, another calling every setter, two racing tsdr_stop calls) with the library's three C files compiled
under the sanitizer.  Here they are linked against tests/sanitize/stub_tsdrgpu.c — host memory, no signal processing,
TEST INFRASTRUCTURE, see its header — so the engine's plugin / device / download / video / plot / copy threads and their
queues run without a GPU; tests/test_gpu_host_pipeline.py::test_host_stress_on_the_device runs the same driver against
the real libtsdrgpu.so.  A data race, a heap error or undefined behaviour in the host code fails the run."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "sanitize")
MEM = os.path.join(ROOT, "tempestsdr_amd", "libTSDRPlugin_Mem.so")
TESTPLUGIN = os.path.join(ROOT, "tests", "plugins", "libtsdr_test_plugin.so")


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    from tempestsdr_amd import build as b
    b.build()  # the Mem plugin and libtsdrgpu.so (the unsanitized variants link it)
    if not os.path.exists(TESTPLUGIN):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-o", TESTPLUGIN,
                        os.path.join(ROOT, "tests", "plugins", "tsdr_test_plugin.c")], check=True)
    subprocess.run(["bash", os.path.join(ROOT, "scripts", "build_sanitized.sh")], check=True, capture_output=True)
    d = tmp_path_factory.mktemp("san")
    rng = np.random.default_rng(5)
    f32 = d / "iq.f32"
    rng.standard_normal(4_000_000).astype(np.float32).tofile(f32)
    u8 = d / "iq.u8"
    rng.integers(0, 255, 2_000_000, dtype=np.uint8).tofile(u8)
    probe = subprocess.run([os.path.join(SAN, "host_stress_tsan_stub")], capture_output=True, text=True)
    if "unexpected memory mapping" in probe.stderr:  # (kernels whose ASLR entropy this gcc's libtsan cannot map around)
        pytest.skip("ThreadSanitizer cannot start on this kernel")
    return {"f32": str(f32), "u8": str(u8)}


def _run(binary, plugin, params, height=100, sessions=3, secs=1.5):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66", ASAN_OPTIONS="detect_leaks=1", TSDR_GPU_STATS="1")
    out = subprocess.run([os.path.join(SAN, binary), plugin, params, str(height), "60", str(sessions), str(secs)],
                         capture_output=True, text=True, timeout=300, env=env)
    text = out.stdout + out.stderr
    assert "ThreadSanitizer" not in text and "AddressSanitizer" not in text and "runtime error" not in text, text[-4000:]
    assert out.returncode == 0, text[-4000:]
    assert "host_stress: ok" in out.stdout
    return text


CASES = [
    ("mem-float-zero-copy", "mem", "{f32} 1000000 65536 0 4000"),
    ("mem-uint8-raw-decode", "mem", "{u8} 1000000 65536 0 4000 uint8"),
    ("file-source-bounce-buffers-and-a-drop", "test", "{f32} 1000000 524288 20000 3 1000"),
    ("file-source-drop-as-empty-block", "test", "{f32} 1000000 524288 20000 3 1000 1"),
]


@pytest.mark.parametrize("name,plugin,params", CASES, ids=[c[0] for c in CASES])
def test_host_threads_under_threadsanitizer(built, name, plugin, params):
    text = _run("host_stress_tsan_stub", MEM if plugin == "mem" else TESTPLUGIN, params.format(**built))
    assert "frames made" in text  # the engine ran (TSDR_GPU_STATS)


def test_host_code_under_addresssanitizer_and_ubsan(built):
    _run("host_stress_asan_stub", MEM, "{f32} 1000000 65536 0 2000".format(**built))
    _run("host_stress_asan_stub", TESTPLUGIN, "{f32} 1000000 524288 20000 3 1000".format(**built), sessions=2)
