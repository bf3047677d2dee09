import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the compiled reference)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="session")
def ref(orc):
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    return orc.ref()


# ---------------------------------------------------------------------------
# Red zones around every device buffer the GPU tests allocate.
# The HIP kernels are checked for WHAT they compute by the parity tests; this checks WHERE they write.  Every
# tempestsdr_amd.gpu.DeviceArray a test creates (g.empty / g.to_device) sits between two guard regions filled with a pattern
# that reads as NaN in float32: a kernel that writes one element outside the buffer it was given destroys the pattern (checked
# after every test and when the buffer is freed), one that READS outside poisons its output and fails the test's own
# comparison with the oracle.  TSDR_TEST_REDZONES=0 switches it off.
# ---------------------------------------------------------------------------
REDZONE_BYTES = 4096
_redzone_violations = []
_redzone_live = None


def _install_redzones():
    global _redzone_live
    import weakref

    import numpy as np

    from tempestsdr_amd import gpu
    if getattr(gpu.DeviceArray, "_redzoned", False):
        return
    _redzone_live = weakref.WeakSet()
    pattern = np.frombuffer(bytes([0xAD, 0xDE, 0xC0, 0xFF]) * (REDZONE_BYTES // 4), np.uint8).copy()
    Plain = gpu.DeviceArray

    class RedzonedArray(Plain):
        _redzoned = True

        def __init__(self, ctx, count, dtype):
            dt = np.dtype(dtype)
            inner = int(count)
            # the guards keep the buffer's start on the 4 KiB boundary hipMalloc gave it
            Plain.__init__(self, ctx, (max(1, inner) * dt.itemsize + 2 * REDZONE_BYTES + dt.itemsize - 1) // dt.itemsize, dt)
            self.base = self.ptr
            self.ptr = self.base + REDZONE_BYTES
            self.count = inner
            self._hi = self.ptr + max(1, inner) * dt.itemsize
            for at in (self.base, self._hi):
                ctx._ck(ctx.lib.tsdrgpu_upload(ctx.h, at, pattern.ctypes.data, REDZONE_BYTES))
            ctx.sync()
            _redzone_live.add(self)

        def redzones_intact(self, where):
            if not self.ptr or not self.ctx.h:
                return True
            got = np.empty(REDZONE_BYTES, np.uint8)
            ok = True
            for name, at in (("below", self.base), ("above", self._hi)):
                self.ctx._ck(self.ctx.lib.tsdrgpu_download(self.ctx.h, got.ctypes.data, at, REDZONE_BYTES))
                self.ctx.sync()
                bad = np.nonzero(got != pattern)[0]
                if bad.size:
                    ok = False
                    _redzone_violations.append(f"{where}: {bad.size} bytes written {name} a {self.dtype} buffer of {self.count} elements "
                                               f"(first at byte {int(bad[0]) - (REDZONE_BYTES if name == 'below' else 0)} relative to that edge)")
                    self.ctx._ck(self.ctx.lib.tsdrgpu_upload(self.ctx.h, at, pattern.ctypes.data, REDZONE_BYTES))  # report once
                    self.ctx.sync()
            return ok

        def free(self):
            if self.ptr and self.ctx.h:
                try:
                    self.redzones_intact("at free")
                finally:
                    self.ptr = self.base
            Plain.free(self)

    gpu.DeviceArray = RedzonedArray


def _redzones_wanted(session):
    return os.environ.get("TSDR_TEST_REDZONES", "1") != "0" and any(i.get_closest_marker("gpu") is not None for i in session.items)


@pytest.fixture(scope="session", autouse=True)
def _redzones_session(request):
    # before any module-scoped fixture of a GPU test allocates
    if _redzones_wanted(request.session):
        _install_redzones()
    yield


@pytest.fixture(autouse=True)
def _redzones(request):
    if request.node.get_closest_marker("gpu") is None or _redzone_live is None:
        yield
        return
    yield
    import gc
    gc.collect()
    for a in list(_redzone_live):
        a.redzones_intact(request.node.nodeid)
    if _redzone_violations:
        msg = "\n".join(_redzone_violations)
        _redzone_violations.clear()
        pytest.fail("a kernel wrote outside the buffer it was given:\n" + msg)
