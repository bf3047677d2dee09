"""Worker of tests/test_gpu_distributed.py::test_sharded_stitch_in_four_processes_equals_the_oracle: rank r holds hop r of
the super-bandwidth stitch (superbandwidth.c:121-152) and runs the HIP kernels on it; the two
exchanges — hop 0's reference spectrum to everybody, everybody's hop spectrum to everybody — go through gloo on the
host (the test box's ranks share one device; production: tsdrgpu_comm_broadcast_f32 / _allgather_f32 over xGMI).
usage: stitch_worker.py <rank> <world> <port>"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))


def main():
    rank, world, port = (int(a) for a in sys.argv[1:4])
    import torch
    import torch.distributed as dist
    import cases
    from tempestsdr_amd import gpu
    from oracle import oracle as orc
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = gpu.TsdrGpu(0)
    gold = np.load(os.path.join(HERE, "golden", "golden.npz"))
    fs, fv = cases.SUPERB["fs"], cases.SUPERB["fv"]
    sif = int(fs / fv)
    hops = [h.copy() for h in gold["superb_hops"]]
    assert len(hops) == world
    gathered = hops[0].size // 2
    d_hop = g.to_device(hops[rank])
    sh = gpu.SuperbShard(g, world, rank, gathered, sif)

    def down(ptr, n):
        a = np.empty(n, np.float32)
        g._ck(g.lib.tsdrgpu_download(g.h, a.ctypes.data, ptr, a.nbytes))
        g.sync()
        return a

    def up(ptr, a):
        g._ck(g.lib.tsdrgpu_upload(g.h, ptr, a.ctypes.data, a.nbytes))
        g.sync()

    p, n = sh.reference(d_hop)
    ref = torch.from_numpy(down(p, n))
    dist.broadcast(ref, src=0)
    up(p, ref.numpy())
    p, n, off = sh.spectrum(d_hop)
    mine = torch.from_numpy(down(p + 4 * rank * n, n))
    parts = [torch.empty(n, dtype=torch.float32) for _ in range(world)]
    dist.all_gather(parts, mine)
    up(p, torch.cat(parts).numpy())
    d_out = g.empty(world * n)
    total = sh.finish(d_out)
    got = d_out.download()
    offs_all = [None] * world
    dist.all_gather_object(offs_all, off)
    ok = True
    if rank == 0:
        want, offs = orc.superb_stitch(hops, sif)
        ok = bool(2 * total == want.size and offs_all == [int(o) for o in offs] and
                  np.max(np.abs(got - want)) <= 1e-4 * np.max(np.abs(want)))
        print("sharded stitch equals the oracle:", ok, flush=True)
    # every rank ends with the same stitched signal
    heads = [None] * world
    dist.all_gather_object(heads, got[:4096].tobytes())
    ok = ok and all(h == heads[0] for h in heads)
    g.close()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
