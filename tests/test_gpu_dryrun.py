"""Dry run of `bench.py --gpus 8` on the one GPU of the test box: eight processes under torch.distributed.run exactly
as the driver launches them on an 8-GPU node, but all on device 0 and with every exchange routed device -> host -> gloo ->
device (`--dist-backend gloo --one-device`; RCCL refuses two ranks on one device).  What is exercised is everything
but the wire: argument handling, the band edges, each rank's share of the capture windows, `total_windows`, the order
of begin / exchange / relay / finish calls on every rank, the certificate taken on the merged plots, the barrier + max-over-ranks
timing and the ONE JSON line of rank 0.  The throughput of such a run means nothing and is not looked at.
BASELINE configs[3] (100 MS/s sweep over 8 GPUs) = --scaling strong / weak; configs[4] (200 MS/s, 2160p, 8 GPUs) =
--config 4 --bands."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(extra, world=8, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
           "--passes", "2", "--dist-backend", "gloo", "--one-device"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    return json.loads(lines[0])


def _common(d, world, scaling):
    assert d["n_gpus"] == world and d["steps"] == 1 and d["warmup"] == 1 and d["scaling"] == scaling
    assert d["unit"] == "Msamples/s" and d["higher_is_better"] is True and d["value"] > 0 and d["dtype"] == "f32"
    assert [r["rank"] for r in d["ranks"]] == list(range(world))
    assert len({tuple(r["argmax"]) for r in d["ranks"]}) == 1          # every rank holds the same merged plots
    assert all(r["epochs_replayed_exact"] == 0 for r in d["ranks"])     # the raster stream is certified everywhere
    assert "gloo" in d["collective"]


def test_bench_gpus8_strong_scaling_dry_run():
    d = _run(["--scaling", "strong", "--seconds", "1"])
    _common(d, 8, "strong")
    assert [r["windows_per_pass"] for r in d["ranks"]] == [3, 2, 2, 2, 2, 2, 2, 2] and all(r["of"] == 17 for r in d["ranks"])
    # the merged plots' detection is the single-GPU run's on this stream (BENCH_r03.json `detected`: the reference's own
    # answer — a pair of line periods beats one)
    assert d["detected"]["frame_lag"] == 1666667 and d["detected"]["line_lag"] == 2963 and d["detected"]["height"] == 562


def test_bench_gpus8_weak_scaling_dry_run():
    d = _run(["--scaling", "weak", "--seconds", "1"])
    _common(d, 8, "weak")
    assert all(r["windows_per_pass"] == 17 and r["of"] == 8 * 17 for r in d["ranks"])
    assert d["config"]["samples_per_step_per_gpu"] == 2 * 99_999_600


def test_bench_gpus8_config4_row_bands_dry_run():
    d = _run(["--config", "4", "--bands", "--seconds", "0.5"])
    _common(d, 8, "strong")
    H = 2250
    edges = [0] + [32 * ((H * k // 8) // 32) for k in range(1, 8)] + [H]
    assert [r["rows"] for r in d["ranks"]] == [[a, b] for a, b in zip(edges[:-1], edges[1:])]
    rb = d["config"]["row_bands"]
    assert rb["bands"] == 8 and rb["of"] == H
    assert sum(r["windows_per_pass"] for r in d["ranks"]) == d["ranks"][0]["of"]
    assert d["frames_per_s"] > 0
