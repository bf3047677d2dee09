"""SURVEY 8(e), the sharded autocorrelation sweep with the HIP kernels in every rank.

Two ranks on the one GPU of the test box (two processes, two contexts): each transforms its share of the capture
windows (k mod world == rank) and keeps per-lag sums; merged and finalised they equal the single-rank running mean
(accummulate, frameratedetector.c:51-60) to 1e-12.  RCCL refuses two ranks on one device, so the two-rank test
exchanges through gloo on the host; the RCCL-from-C path (tsdrgpu_comm_*, tsdrgpu_autocorr_allreduce) is exercised
with a one-rank communicator, which runs the same ncclAllReduce launch on the library's lane."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tempestsdr_amd import gpu
from gpu_util import ctx

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# (100 MS/s, 17 windows: BASELINE configs[3], one second of the stream in windows of 2^22 samples)
@pytest.mark.parametrize("fs,nwin", [(8_000_000, 7), (25_000_000, 5), (100_000_000, 17)])
def test_two_ranks_on_one_device_match_the_running_mean(fs, nwin):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), "2", str(port), str(fs), str(nwin)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "merged plots equal the single-rank running mean: True" in outs[0]


@pytest.mark.parametrize("world", [2, 3])
def test_row_bands_in_two_processes_equal_the_oracle(world):
    """SURVEY 8(e) row 2 end to end, one process per rank: band resampler -> band statistics -> exchanges -> relayed literal
    collapse -> chain -> pass; the reassembled frames are the ORACLE's (tests/band_worker.py)."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "band_worker.py"), str(r), str(world), str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "bands equal the oracle: True" in outs[0]


def test_rccl_from_c_one_rank(orc):
    """tsdrgpu_rccl_unique_id / tsdrgpu_comm_create / tsdrgpu_autocorr_allreduce with a communicator of one rank:
    the ncclAllReduce is queued by the library on the autocorrelation's lane; sums + all-reduce + finalise equal
    the running mean, on the main lane and on the side lane."""
    g = ctx()
    fs, nwin = 8_000_000, 4
    comm = gpu.Comm(g, 1, 0, gpu.Comm.unique_id(g))
    rng = np.random.default_rng(5)
    ref = gpu.Autocorr(g, fs)
    x = rng.random(nwin * ref.capture).astype(np.float32)
    d_in = g.to_device(x)
    ref.run(d_in, 0, ref.capture, nwin, mode=0)
    rf, rl, _ = ref.plots()
    for side in (False, True):
        ac = gpu.Autocorr(g, fs)
        ac.set_async(side)
        ac.run(d_in, 0, ac.capture, nwin, mode=1)
        ac.allreduce(comm, nwin)
        f, l, calls = ac.plots()
        assert calls == nwin
        assert np.allclose(f, rf, rtol=1e-12, atol=0) and np.allclose(l, rl, rtol=1e-12, atol=0)
    comm.destroy()
