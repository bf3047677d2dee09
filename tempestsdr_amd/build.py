"""Builds the gfx950 shared libraries IN-TREE (they travel to the GPU box with
the gpurun snapshot; a JIT cache would not):

  tempestsdr_amd/libtsdrgpu.so     HIP kernels + the tsdrgpu_* C ABI (include/tsdrgpu.h)
  tempestsdr_amd/libTSDRLibrary.so the drop-in tsdr_* host library (C) on top of it
  tempestsdr_amd/libTSDRPlugin_Mem.so an in-memory replay source plugin (tsdrplugin_* ABI)

hipcc cross-compiles for gfx950 without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
          "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(ROOT, "include")]
# bit-exact stages: no FMA contraction (the reference is built without it)
STRICT = ["-ffp-contract=off"]

# The FFT kernels are complex float32 arithmetic on register arrays.  hipcc's SLP vectoriser turns it into v_pk_*_f32
# pairs — which on gfx950 issue at half the rate of the plain instructions, so nothing is gained per flop — and pays for
# the pairing with register moves: k_ac_rows had 725 v_mov_b32 among 2481 VALU instructions (29 %), without the pass 59
# among 2321, all of them full-rate.  TSDRGPU_SLP=1 builds the old way (A/B).
NOSLP = [] if os.environ.get("TSDRGPU_SLP") == "1" else ["-fno-slp-vectorize"]

HIP_SOURCES = [("tsdrgpu_core.hip", STRICT), ("tsdrgpu_frame.hip", STRICT), ("tsdrgpu_fft.hip", NOSLP), ("tsdrgpu_fftx.hip", STRICT),
               ("tsdrgpu_extras.hip", STRICT), ("tsdrgpu_rccl.hip", [])]
LIB = os.path.join(HERE, "libtsdrgpu.so")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    return hs


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    procs = []
    for src, extra in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + _headers()):
            cmd = [HIPCC] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    if force or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    host = os.path.join(CSRC, "host")
    if os.path.isdir(host) and os.path.exists(os.path.join(host, "Makefile")):
        subprocess.run(["make", "-C", host, "-s"], check=True)
    # source plugins shipped with the library (tsdrplugin_* ABI, include/TSDRPlugin.h)
    mem_src = os.path.join(CSRC, "plugins", "TSDRPlugin_Mem.c")
    mem_so = os.path.join(HERE, "libTSDRPlugin_Mem.so")
    if force or _newer(mem_so, [mem_src] + _headers()):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", mem_so, mem_src], check=True)
    build_checks(force=force, verbose=verbose)
    build_host_stress(force=force)
    build_fake_jvm(force=force)
    return LIB


# Device self-checks (scripts/micro/*.hip) that the GPU suite RUNS: the GPU boxes have no hipcc, so the binaries are built here,
# in-tree, and travel with the snapshot like the libraries (git-ignored, not gpurun-ignored).
# (arith_check with the flags of the bit-exact stages it vouches for)
CHECKS = [("arith_check", ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"]), ("wave_reduce_check", [])]


def build_checks(force=False, verbose=True):
    micro = os.path.join(ROOT, "scripts", "micro")
    procs = []
    for name, extra in CHECKS:
        src = os.path.join(micro, name + ".hip")
        exe = os.path.join(micro, name)
        if not os.path.exists(src):
            continue
        if force or _newer(exe, [src] + _headers()):
            cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-w"] + extra + ["-o", exe, src]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))


def build_fake_jvm(force=False):
    """tests/jni/fake_jvm: the stand-in JVM that hosts the reference's JNI shim linked on top of our libTSDRLibrary.a
    (oracle/Makefile builds that shim where /root/reference exists).  Test infrastructure; built here because the GPU box has the
    compiler but the test should find it like it finds the libraries."""
    src = os.path.join(ROOT, "tests", "jni", "fake_jvm.c")
    exe = os.path.join(ROOT, "tests", "jni", "fake_jvm")
    stub = os.path.join(ROOT, "oracle", "jni_stub")
    if os.path.exists(src) and (force or _newer(exe, [src, os.path.join(stub, "jni.h")])):
        subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-I" + stub, "-o", exe, src, "-ldl", "-lpthread"], check=True)


def build_host_stress(force=False):
    """tests/sanitize/host_stress_*: the host library's C files compiled WITH their stress driver, plain and under
    ThreadSanitizer / AddressSanitizer (scripts/build_sanitized.sh).  Test infrastructure; built here so that the GPU box finds
    the plain variant (tests/test_gpu_host_pipeline.py::test_host_stress_on_the_device) like it finds the libraries."""
    san = os.path.join(ROOT, "tests", "sanitize")
    script = os.path.join(ROOT, "scripts", "build_sanitized.sh")
    if not os.path.exists(script):
        return
    host = os.path.join(CSRC, "host")
    srcs = [os.path.join(san, "host_stress.c"), os.path.join(san, "stub_tsdrgpu.c"), script] + [os.path.join(host, f) for f in os.listdir(host)] + _headers()
    if force or any(_newer(os.path.join(san, exe), srcs) for exe in ("host_stress_plain", "host_stress_tsan_stub", "host_stress_asan_stub")):
        r = subprocess.run(["bash", script], capture_output=True, text=True)
        if r.returncode != 0:  # test infrastructure: a host without libtsan / libasan still gets the product
            print("build_host_stress: scripts/build_sanitized.sh failed (the sanitizer tests will rebuild or skip):\n" + r.stderr[-1500:], flush=True)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
