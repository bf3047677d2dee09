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


@pytest.mark.parametrize("kind", ["certified", "flat"])
def test_eight_rank_sweep_config3_certified_and_replayed(kind):
    """BASELINE configs[3] in its named shape: the 100 MS/s lag sweep (17 windows of 2^22 samples) sharded over EIGHT ranks
    (eight processes on the one device; rank 0 owns windows 0, 8, 16, the others two each), the detector in its certified
    mode.  `certified`: every rank finds the certificate on the global plots, nothing is replayed, the argmax is the
    single-rank run's.  `flat`: the certificate fails on every rank alike -> every rank replays its own windows in the
    reference's arithmetic -> SECOND exchange -> the merged plots equal the single-rank exact running mean to 1e-12
    (tests/dist_worker.py; frameratedetector.c:34-62)."""
    world, fs, nwin = 8, 100_000_000, 17
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), str(port), str(fs), str(nwin), kind],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "window shares [3, 2, 2, 2, 2, 2, 2, 2]" in outs[0]
    assert f"kind {kind}: promoted {1 if kind == 'flat' else 0}" in outs[0]
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


def test_row_bands_config4_in_eight_processes_equal_the_oracle():
    """BASELINE configs[4] in its named shape — 200 MS/s, 2962x2250 frames, motion blur 15/16, EIGHT ranks (here eight
    processes on the one device, exchanges through gloo) — from IQ to frames: every rank resamples only its rows, the
    strip partials and extrema are all-reduced, the literal collapse of the blank frames' strips is relayed through all
    eight bands, the replicated chain decides, every rank passes its rows; the reassembled frames and the final sync
    state are the ORACLE's bit for bit (dsp.c:41-110, syncdetector.c:26-153,171-225)."""
    world = 8
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "band_worker.py"), str(r), str(world), str(port), "config4"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "frames 9 of 2962x2250 in 8 bands" in outs[0]
    assert "bands equal the oracle: True" in outs[0]


@pytest.mark.parametrize("world,config", [(2, "small"), (3, "small"), (8, "config4")])
def test_fused_row_bands_in_processes_equal_the_oracle(world, config):
    """the FUSED band run end to end from IQ, one process per rank (tests/band_worker.py ... fused): the band resampler tracks its
    band's share of every frame's range, the ranks exchange the range BEFORE the band is read, one trip writes the rows and gathers the
    strip partials, the replicated contract-exact chain (relays through all bands for the blank frames) decides; reassembled frames and
    sync state are the ORACLE's bit for bit — configs[4] in its named shape on eight ranks among them."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "band_worker.py"), str(r), str(world), str(port), config, "fused"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "bands equal the oracle: True" in outs[0]


@pytest.mark.parametrize("world,config", [(3, "small"), (8, "config4")])
def test_row_bands_gui_order_with_autoshift_in_processes_equal_the_oracle(world, config):
    """the GENERAL band run end to end from IQ, one process per rank: the GUI's stage order (low-pass before sync) with
    autoshift on — two statistics rounds, the relayed literal collapse, and the all-gather that carries the rolled rows
    across the ranks (tests/band_worker.py ... gui); reassembled frames and sync state are the ORACLE's."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "band_worker.py"), str(r), str(world), str(port), config, "gui"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
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


# ---------------------------------------------------------------------------
# the super-bandwidth stitch with one hop per GPU (tsdrgpu_superb_shard_*)
# ---------------------------------------------------------------------------
def _copy_dev(g, dst, src, nfloats):
    g._ck(g.lib.tsdrgpu_copy(g.h, dst, src, nfloats * 4))


def test_sharded_stitch_equals_the_single_gpu_stitch_and_the_oracle(orc):
    """Four hop objects in one process (one per 'rank'), the two exchanges done by device copies: hop offsets identical to
    the oracle's, the stitched signal bit-identical to tsdrgpu_superb_stitch and within 1e-4*max of the oracle's."""
    from gpu_util import golden
    import cases
    g = ctx()
    gold = golden()
    fs, fv = cases.SUPERB["fs"], cases.SUPERB["fv"]
    sif = int(fs / fv)
    hops = [h.copy() for h in gold["superb_hops"]]
    nh, gathered = len(hops), hops[0].size // 2
    want, offs = orc.superb_stitch([h.copy() for h in hops], sif)
    d_single = [g.to_device(h) for h in hops]
    d_ref_out = g.empty(want.size)
    single_offs, total = g.superb_stitch(d_single, gathered, sif, d_ref_out)
    shards = [gpu.SuperbShard(g, nh, k, gathered, sif) for k in range(nh)]
    d_hops = [g.to_device(h) for h in hops]
    refs = [sh.reference(d) for sh, d in zip(shards, d_hops)]
    for k in range(1, nh):  # broadcast from the rank of hop 0
        _copy_dev(g, refs[k][0], refs[0][0], refs[0][1])
    specs = [sh.spectrum(d) for sh, d in zip(shards, d_hops)]
    per2 = specs[0][1]
    for k in range(nh):  # all-gather: everybody's slot to everybody
        for j in range(nh):
            if j != k:
                _copy_dev(g, specs[j][0] + 4 * k * per2, specs[k][0] + 4 * k * per2, per2)
    assert [s[2] for s in specs] == [int(o) for o in offs] == [int(o) for o in single_offs]
    for k, sh in enumerate(shards):
        d_out = g.empty(want.size)
        assert sh.finish(d_out) == total
        got = d_out.download()
        assert np.array_equal(got, d_ref_out.download()), k
        assert np.max(np.abs(got - want)) <= 1e-4 * np.max(np.abs(want))
        sh.destroy()
    # every hop buffer is left holding its spectrum, like the single-GPU call's (and the reference's)
    for a, b in zip(d_hops, d_single):
        assert np.array_equal(a.download(), b.download())


def test_sharded_stitch_collectives_over_rccl_one_rank(orc):
    """ncclBroadcast / ncclAllGather from C (tsdrgpu_comm_broadcast_f32 / _allgather_f32) on a one-rank communicator
    between the phases of a one-hop stitch."""
    g = ctx()
    rng = np.random.default_rng(4)
    gathered, sif = 70_000, 5_000
    hop = rng.standard_normal(2 * gathered).astype(np.float32)
    d_a, d_b = g.to_device(hop), g.to_device(hop)
    per = 1 << (gathered.bit_length() - 1)
    d_want, d_got = g.empty(2 * per), g.empty(2 * per)
    g.superb_stitch([d_a], gathered, sif, d_want)
    comm = gpu.Comm(g, 1, 0, gpu.Comm.unique_id(g))
    sh = gpu.SuperbShard(g, 1, 0, gathered, sif)
    p, n = sh.reference(d_b)
    comm.broadcast_f32(p, n, 0)
    p, n, off = sh.spectrum(d_b)
    comm.allgather_f32(p, n)
    assert off == 0 and sh.finish(d_got) == per
    assert np.array_equal(d_got.download(), d_want.download())
    sh.destroy()
    comm.destroy()


def test_sharded_stitch_in_four_processes_equals_the_oracle():
    """one process per hop, exchanges through gloo (tests/stitch_worker.py)"""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "stitch_worker.py"), str(r), "4", str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(4)]
    outs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "sharded stitch equals the oracle: True" in outs[0]
