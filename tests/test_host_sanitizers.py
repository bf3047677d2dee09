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
    subprocess.run(["bash", os.path.join(ROOT, "scripts", "build_sanitized.sh")], capture_output=True)
    if not all(os.path.exists(os.path.join(SAN, b)) for b in ("host_stress_tsan_stub", "host_stress_asan_stub", "host_stress_sweep_tsan_stub")):
        pytest.skip("this host's gcc has no libtsan / libasan")
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


def _run(binary, plugin, params, height=100, sessions=3, secs=1.5, seed=None):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66", ASAN_OPTIONS="detect_leaks=1", TSDR_GPU_STATS="1")
    out = subprocess.run([os.path.join(SAN, binary), plugin, params, str(height), "60", str(sessions), str(secs)] + ([str(seed)] if seed is not None else []),
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


@pytest.mark.parametrize("seed", [3, 17, 29, 40])
def test_random_geometry_changes_under_addresssanitizer(built, seed):
    """resolution changes mid-stream to random geometries — one-line frames, 20 000-line frames whose width truncates to 0, refresh
    rates from 5 to 240 Hz — : the engine's sizing arithmetic (chunks, pixel counts, stream and batch buffers, the fused run's
    bookkeeping) with every write of the stand-in's "kernels" checked by AddressSanitizer"""
    _run("host_stress_asan_stub", MEM, "{f32} 1000000 65536 0 2000".format(**built), sessions=2, secs=1.0, seed=seed)


@pytest.mark.parametrize("binary", ["host_stress_tsan_stub", "host_stress_asan_stub"])
def test_a_device_call_that_fails_mid_session_ends_it_loudly(built, binary):
    """the stand-in's 8th post-processing call fails (early in the first session, whatever the host's load): tsdr_readasync comes back on its own with TSDR_CANNOT_OPEN_DEVICE and the
    failing stage's text, every thread joined and everything freed (engine.c gpu_ok: nothing is retried, nothing falls back); the
    next session on the same handle runs clean"""
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66", ASAN_OPTIONS="detect_leaks=1", STRESS_EXPECT_FAILURE="1", STUB_FAIL_POSTPROC_AFTER="8")
    out = subprocess.run([os.path.join(SAN, binary), MEM, "{f32} 1000000 65536 0 4000".format(**built), "100", "60", "2", "1.5"],
                         capture_output=True, text=True, timeout=120, env=env)
    text = out.stdout + out.stderr
    assert "ThreadSanitizer" not in text and "AddressSanitizer" not in text and "runtime error" not in text, text[-3000:]
    assert out.returncode == 0 and "host_stress: ok" in out.stdout, text[-2000:]
    assert "GPU stage 'postproc' failed" in out.stderr


def test_host_code_under_addresssanitizer_and_ubsan(built):
    _run("host_stress_asan_stub", MEM, "{f32} 1000000 65536 0 2000".format(**built))
    _run("host_stress_asan_stub", TESTPLUGIN, "{f32} 1000000 524288 20000 3 1000".format(**built), sessions=2)


# ---------------------------------------------------------------------------
# tsdr_sweep (the multi-GPU sweep's C host: one thread and one RCCL rank per device) with MORE THAN ONE rank.  One MI355X box
# cannot run that (RCCL refuses two ranks on one device), so its meeting points — the ranks agree before every step that leads
# into a collective, tsdr_sweep.c agree() / MEET — are exercised here, under ThreadSanitizer, on the stand-in, whose communicator
# creation and all-reduce BLOCK until every rank has arrived, like ncclCommInitRank / ncclAllReduce: a rank that skipped or
# entered a collective alone would hang the run (the timeout), not fail it.
# ---------------------------------------------------------------------------
SWEEP_CASES = [
    # name, environment, devices, exit code, what the output must hold
    ("healthy-4-ranks", {}, "0,1,2,3", 0, '"epoch_replayed_exact": 0'),
    ("healthy-8-ranks", {}, "0,1,2,3,4,5,6,7", 0, '"windows_per_device": [2, 1, 1, 1, 1, 1, 1, 1]'),
    ("one-rank-alone-is-refused-its-certificate", {"STUB_UNCERTIFIED_DEVICE": "2"}, "0,1,2,3", 0, '"epoch_replayed_exact": 1'),
    ("every-rank-is-refused", {"STUB_UNCERTIFIED_DEVICE": "all"}, "0,1,2,3", 0, '"epoch_replayed_exact": 1'),
    ("a-rank-fails-before-the-communicator", {"STUB_FAIL_CREATE_DEVICE": "1"}, "0,1,2,3", 1, "rank 1 (device 1): tsdrgpu_create"),
    ("a-rank-fails-before-the-exchange", {"STUB_FAIL_RUN_DEVICE": "3"}, "0,1,2,3", 1, "rank 3 (device 3): tsdrgpu_autocorr_run"),
    ("two-ranks-fail-at-different-points", {"STUB_FAIL_CREATE_DEVICE": "0", "STUB_FAIL_RUN_DEVICE": "2"}, "0,1,2,3", 1, "rank 0 (device 0)"),
    ("one-rank-with-a-communicator", {}, "0 --force-comm", 0, '"windows_per_device": [9]'),
]


@pytest.mark.parametrize("name,env,devices,rc,needle", SWEEP_CASES, ids=[c[0] for c in SWEEP_CASES])
def test_sweep_tool_ranks_meet_before_every_collective(built, tmp_path, name, env, devices, rc, needle):
    rec = tmp_path / "sweep.f32"
    np.random.default_rng(3).standard_normal(2 * 56363 * 9).astype(np.float32).tofile(rec)  # 9 capture windows at 1 MS/s
    e = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66", **env)
    out = subprocess.run([os.path.join(SAN, "host_stress_sweep_tsan_stub"), str(rec), "1000000", "--devices"] + devices.split() + ["--windows", "9"],
                         capture_output=True, text=True, timeout=60, env=e)  # a rank left alone in a collective = this timeout
    text = out.stdout + out.stderr
    assert "ThreadSanitizer" not in text, text[-3000:]
    assert out.returncode == rc, text[-2000:]
    assert needle in text, text[-2000:]
