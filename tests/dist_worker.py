"""Worker of tests/test_gpu_distributed.py: one rank of the sharded autocorrelation sweep running the HIP kernels.
usage: dist_worker.py <rank> <world> <port> <fs> <nwindows> [kind]
Every rank builds the same seeded stream, transforms windows rank, rank+world, ... on the GPU (tsdrgpu_autocorr_run
mode 1: per-lag sums), the sums meet in a gloo all-reduce on the host (the ranks share ONE device on the test box,
which RCCL refuses; the production path is tsdrgpu_autocorr_allreduce), go back to the device and are finalised.
Rank 0 also runs the whole sweep alone (mode 0, the reference's running mean) and compares.

kind (default "plain"):
  plain      the float32 form, no certificate (rounds 1-3)
  certified  the detector's CERTIFIED mode as bench.py / the engine run it: after the exchange every rank takes the argmax
             and its certificate on the (identical) global plots; the raster-like stream must be certified on every rank
             and no epoch replayed
  flat       a constant stream (every lag of every plot ties): the certificate MUST fail on every rank alike, every rank replays ITS windows in the
             reference's arithmetic (tsdrgpu_autocorr_promote), the exact sums are exchanged a SECOND time and the merged
             plots equal the single-rank exact running mean (frameratedetector.c:34-62, fft.c:49-64,96-176)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port, fs, nwin = (int(a) for a in sys.argv[1:6])
    kind = sys.argv[6] if len(sys.argv) > 6 else "plain"
    import torch
    import torch.distributed as dist
    from tempestsdr_amd import gpu
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = gpu.TsdrGpu(0)
    ac = gpu.Autocorr(g, fs)
    if kind != "plain":
        ac.set_certify(2)  # the windows stay where they are (the HBM-resident stream): no copy
    rng = np.random.default_rng(77)
    period = int(fs / 60.0)
    tot = nwin * ac.capture
    if kind == "flat":
        x = np.full(tot, 0.25, np.float32)  # every lag ties (best == runner-up): the certificate cannot hold
    else:
        x = rng.random(tot).astype(np.float32) * np.float32(0.3) + (np.arange(tot) % period < period // 10).astype(np.float32)
    d_in = g.to_device(x)
    mine = list(range(rank, nwin, world))

    def exchange():
        ptr, count = ac.device_sums()  # the lags and the certificate's lag-0 scale
        sums = np.empty(count, np.float64)
        g._ck(g.lib.tsdrgpu_download(g.h, sums.ctypes.data, ptr, sums.nbytes))
        g.sync()
        t = torch.from_numpy(sums)
        dist.all_reduce(t)  # the exchange step (RCCL over xGMI in production)
        g._ck(g.lib.tsdrgpu_upload(g.h, ptr, sums.ctypes.data, sums.nbytes))
        ac.finalize_sums(nwin)

    ac.run(d_in, 0, ac.capture * world, len(mine), mode=1, in_offset=rank * ac.capture)
    exchange()
    promoted = 0
    arg = None
    if kind != "plain":
        arg = ac.argmax()
        c = ac.certificate()
        # the ranks AGREE before any of them enters the replay's collective: the certificate also folds in the premise check of
        # the rank's own newest window, which one rank alone may fail (bench.py's settle() does the same)
        flag = torch.tensor([0 if (c.frame_certified and c.line_certified) else 1], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            ac.promote()   # this rank's windows once more, in the reference's arithmetic (sums)
            exchange()     # the second exchange
            arg = ac.argmax()
            promoted = 1
            c = ac.certificate()
            assert c.exact_epoch == 1
    f, l, calls = ac.plots()
    assert calls == nwin
    ok = True
    everyone = [None] * world
    dist.all_gather_object(everyone, (promoted, arg, len(mine), f[:64].tobytes(), l[:64].tobytes()))
    ok = ok and all(e[0] == everyone[0][0] and e[1] == everyone[0][1] and e[3:] == everyone[0][3:] for e in everyone)  # all ranks alike
    if rank == 0:
        shares = [e[2] for e in everyone]
        print("window shares", shares, flush=True)
        ok = ok and sum(shares) == nwin and max(shares) - min(shares) <= 1
        ref = gpu.Autocorr(g, fs)
        if kind == "flat":
            ref.set_exact(True)  # what the replay computes
        ref.run(d_in, 0, ac.capture, nwin, mode=0)
        rf, rl, _ = ref.plots()
        same = bool(np.allclose(f, rf, rtol=1e-12, atol=0) and np.allclose(l, rl, rtol=1e-12, atol=0))
        if kind == "plain":
            same = same and ac.argmax() == ref.argmax()
        elif kind == "certified":
            same = same and promoted == 0 and arg == ref.argmax()
        else:
            # a flat plot's maximum is decided in the last bits, and sums over ranks differ from the running mean there
            # (documented: 1e-15 relative): the argmax must be the first maximum of the merged plots themselves
            same = same and promoted == 1 and arg == (int(np.argmax(f)), int(np.argmax(l)))
        ok = ok and same
        print(f"kind {kind}: promoted {promoted}, argmax {arg}", flush=True)
        print("merged plots equal the single-rank running mean:", ok, flush=True)
    g.close()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
