"""Worker of tests/test_gpu_distributed.py: one rank of the sharded autocorrelation sweep running the HIP kernels.
usage: dist_worker.py <rank> <world> <port> <fs> <nwindows>
Every rank builds the same seeded stream, transforms windows rank, rank+world, ... on the GPU (tsdrgpu_autocorr_run
mode 1: per-lag sums), the sums meet in a gloo all-reduce on the host (two ranks share ONE device on the test box,
which RCCL refuses; the production path is tsdrgpu_autocorr_allreduce), go back to the device and are finalised.
Rank 0 also runs the whole sweep alone (mode 0, the reference's running mean) and compares."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port, fs, nwin = (int(a) for a in sys.argv[1:6])
    import torch
    import torch.distributed as dist
    from tempestsdr_amd import gpu
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = gpu.TsdrGpu(0)
    ac = gpu.Autocorr(g, fs)
    rng = np.random.default_rng(77)
    period = int(fs / 60.0)
    tot = nwin * ac.capture
    x = rng.random(tot).astype(np.float32) * np.float32(0.3) + (np.arange(tot) % period < period // 10).astype(np.float32)
    d_in = g.to_device(x)
    mine = list(range(rank, nwin, world))
    ac.run(d_in, 0, ac.capture * world, len(mine), mode=1, in_offset=rank * ac.capture)
    ptr, count = ac.device_sums()  # the lags and the certificate's lag-0 scale
    sums = np.empty(count, np.float64)
    g._ck(g.lib.tsdrgpu_download(g.h, sums.ctypes.data, ptr, sums.nbytes))
    g.sync()
    t = torch.from_numpy(sums)
    dist.all_reduce(t)  # the exchange step (RCCL over xGMI in production)
    g._ck(g.lib.tsdrgpu_upload(g.h, ptr, sums.ctypes.data, sums.nbytes))
    ac.finalize_sums(nwin)
    f, l, calls = ac.plots()
    assert calls == nwin
    ok = True
    if rank == 0:
        ref = gpu.Autocorr(g, fs)
        ref.run(d_in, 0, ac.capture, nwin, mode=0)
        rf, rl, _ = ref.plots()
        ok = bool(np.allclose(f, rf, rtol=1e-12, atol=0) and np.allclose(l, rl, rtol=1e-12, atol=0))
        ok = ok and ac.argmax() == ref.argmax()
        print("merged plots equal the single-rank running mean:", ok, flush=True)
    # every rank ends up with the same plots
    both = [None] * world
    dist.all_gather_object(both, (f[:64].tobytes(), l[:64].tobytes()))
    ok = ok and all(b == both[0] for b in both)
    g.close()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
