"""the frame path sharded by row bands (tsdrgpu_postproc_band_begin / _finish).

Two "ranks" (two post-processing objects; in production one process per GPU) each hold a band of rows of every frame.
Their exchange — a sum all-reduce of the strip partials and a max all-reduce of {-min, max, pixel 0} — is done here by
the test on the host (adding / maximising the two device buffers), the production collective being
tsdrgpu_comm_allreduce_f64 / _f32max over RCCL, which is exercised with a one-rank communicator.  The concatenated
band outputs, the per-frame sync / autogain records and the state carried over several batches must equal the
single-GPU run bit for bit (its fast mode: a band cannot walk a column through the other bands for the literal
re-collapse of toss-up strips).  Reference arithmetic: dsp.c:41-110, syncdetector.c:171-225."""
import numpy as np
import pytest

from tempestsdr_amd import gpu
from gpu_util import ctx

pytestmark = pytest.mark.gpu


def _frames(rng, F, W, H, k0):
    y, x = np.mgrid[0:H, 0:W]
    out = np.empty((F, H, W), np.float32)
    for f in range(F):
        img = 0.3 + 0.5 * (((x + 3 * (k0 + f)) // 37) % 2) + 0.1 * (((y // 16) + (x // 16)) % 2)
        img[:, : W // 9] = 0.05  # horizontal blanking
        img[: H // 20, :] = 0.05  # vertical blanking
        out[f] = (img + rng.standard_normal((H, W)) * 0.02).astype(np.float32)
    return out


def _exchange(g, bufs):
    """all-reduce of the ranks' exchange buffers on the host: sum for the doubles, max for the floats"""
    (ps0, ns, pm0, nm), (ps1, _, pm1, _) = bufs
    s = [np.empty(ns, np.float64) for _ in range(2)]
    m = [np.empty(nm, np.float32) for _ in range(2)]
    for k, (ps, pm) in enumerate(((ps0, pm0), (ps1, pm1))):
        g._ck(g.lib.tsdrgpu_download(g.h, s[k].ctypes.data, ps, s[k].nbytes))
        g._ck(g.lib.tsdrgpu_download(g.h, m[k].ctypes.data, pm, m[k].nbytes))
    g.sync()
    ssum, mmax = s[0] + s[1], np.maximum(m[0], m[1])
    for ps, pm in ((ps0, pm0), (ps1, pm1)):
        g._ck(g.lib.tsdrgpu_upload(g.h, ps, ssum.ctypes.data, ssum.nbytes))
        g._ck(g.lib.tsdrgpu_upload(g.h, pm, mmax.ctypes.data, mmax.nbytes))
    g.sync()


@pytest.mark.parametrize("W,H,split,blur,sentinels", [(507, 525, 256, 0.0, False), (1033, 806, 416, 0.5, False), (2962, 2250, 1152, 0.9375, False),
                                                        (300, 200, 96, 0.0, True)])
def test_two_row_bands_equal_the_single_gpu_run(W, H, split, blur, sentinels):
    g = ctx()
    rng = np.random.default_rng(W + H)
    ref = gpu.PostProcess(g)
    ref.set_exact_ties(False)
    bands = [gpu.PostProcess(g), gpu.PostProcess(g)]
    rows = [(0, split), (split, H - split)]
    for batch, F in enumerate((3, 5, 1)):
        fr = _frames(rng, F, W, H, 10 * batch)
        if sentinels:
            fr[0, 0, 0] = 512.0      # pixel 0 a sentinel: the v[0] quirk of dsp.c:50-51 crosses the exchange
            fr[F - 1, 150, 7] = 1024.0  # a sentinel in band 1
        d_full, d_out = g.to_device(fr.reshape(-1)), g.empty(F * W * H)
        infos = ref.run(d_full, F, W, H, d_out, motionblur=blur)
        want = d_out.download().reshape(F, H, W)
        outs, bufs, d_bands = [], [], []
        for pp, (y0, n) in zip(bands, rows):
            d_b = g.to_device(np.ascontiguousarray(fr[:, y0:y0 + n, :]).reshape(-1))
            d_bands.append(d_b)
            bufs.append(pp.band_begin(d_b, F, W, H, y0, n, motionblur=blur))
        _exchange(g, bufs)
        for pp, (y0, n) in zip(bands, rows):
            d_ob = g.empty(F * W * n)
            binfo = pp.band_finish(d_ob)
            outs.append(d_ob.download().reshape(F, n, W))
            for a, b in zip(infos, binfo):  # every rank ends up with the same record
                ra = (a.lastmin, a.lastmax, a.dx, a.vx, a.stripx, a.dy, a.vy, a.stripy, a.locked, a.avg_speed)
                rb = (b.lastmin, b.lastmax, b.dx, b.vx, b.stripx, b.dy, b.vy, b.stripy, b.locked, b.avg_speed)
                assert ra == rb
        got = np.concatenate(outs, axis=1)
        assert np.array_equal(got, want), (batch, int(np.sum(got != want)))
        if blur == 0.0:
            assert (want == 512.0).any()  # the green sync lines are painted in the right rows / columns of each band


def test_band_exchange_over_rccl_one_rank():
    """A single band covering the whole frame with the exchange done by the production collectives
    (tsdrgpu_comm_allreduce_f64 / _f32max on a one-rank RCCL communicator) equals tsdrgpu_postproc_run."""
    g = ctx()
    W, H, F = 507, 525, 4
    rng = np.random.default_rng(9)
    fr = _frames(rng, F, W, H, 0)
    d_in, d_a, d_b = g.to_device(fr.reshape(-1)), g.empty(F * W * H), g.empty(F * W * H)
    ref = gpu.PostProcess(g)
    ref.set_exact_ties(False)
    ref.run(d_in, F, W, H, d_a)
    pp = gpu.PostProcess(g)
    comm = gpu.Comm(g, 1, 0, gpu.Comm.unique_id(g))
    ps, ns, pm, nm = pp.band_begin(d_in, F, W, H, 0, H)
    comm.allreduce_f64(ps, ns)
    comm.allreduce_f32max(pm, nm)
    pp.band_finish(d_b)
    assert np.array_equal(d_a.download(), d_b.download())
    comm.destroy()
    with pytest.raises(gpu.TsdrGpuError):
        pp.band_begin(d_in, F, W, H, 16, 100)  # bands start on multiples of 32 rows
