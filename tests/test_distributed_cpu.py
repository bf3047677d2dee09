"""The N>1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.
Each rank correlates its round-robin share of the capture windows (the oracle
stands in for the GPU kernels here — no GPU in this container), the per-lag sums
are all-reduced exactly like bench.py does over RCCL, and the result must equal
the single-process running mean of ALL windows (frameratedetector.c:34-62)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS = 300_000
NWIN = 5


def windows_for_rank(total_windows, rank, world):
    """the partition bench.py / tests/dist_worker.py use: window k belongs to rank k mod world"""
    return list(range(rank, total_windows, world))


def make_windows(orc):
    rng = np.random.default_rng(77)
    cap = orc.capture_size(FS)
    per = FS // 60
    xs = []
    for k in range(NWIN):
        x = rng.random(cap).astype(np.float32) * np.float32(0.5)
        x += (np.arange(cap) % per < per // 12).astype(np.float32)
        xs.append(x)
    return xs


def worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    xs = make_windows(orc)
    flo, flen, llo, llen = orc.lag_windows(FS)
    sums = np.zeros(flen + llen)
    mine = windows_for_rank(NWIN, rank, world)
    for w in mine:  # mode 1 of tsdrgpu_autocorr_run: plain sums of |R| per lag
        corr = orc.fft_autocorrelation(xs[w]).astype(np.float64)
        mag = np.sqrt(corr[0::2] ** 2 + corr[1::2] ** 2)
        sums[:flen] += mag[flo:flo + flen]
        sums[flen:] += mag[llo:llo + llen]
    # tsdrgpu_autocorr_allreduce: one sum all-reduce of the per-lag sums, then the division by the global window count
    t = torch.from_numpy(sums)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    cnt = torch.tensor([float(len(mine))], dtype=torch.float64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    total = int(round(float(cnt.item())))
    if rank == 0:
        np.save(out_path, np.concatenate([[total], (t / total).numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_autocorrelation_equals_running_mean(orc, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "plots.npy")
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert int(got[0]) == NWIN
    ac = orc.Autocorr(FS)
    for x in make_windows(orc):
        ac.run(x)
    want = np.concatenate([ac.frame, ac.line])
    assert np.allclose(got[1:], want, rtol=1e-12, atol=0)
    # lag -> frame rate as the GUI derives it (PlotVisualizer.java:233-236 first maximum, Main.java:1301-1303)
    frame_lag = ac.flo + int(np.argmax(got[1:1 + ac.flen]))
    assert frame_lag == ac.flo + int(np.argmax(ac.frame))
    assert abs(FS / frame_lag - 60.0) < 0.5


def test_window_partition_covers_everything():
    for world in (1, 2, 3, 8):
        for total in (0, 1, 7, 17, 64):
            seen = sorted(w for r in range(world) for w in windows_for_rank(total, r, world))
            assert seen == list(range(total))
            sizes = [len(windows_for_rank(total, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
