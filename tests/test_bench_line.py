"""bench.py's ONE line: strict JSON, at most 8 KB, the contract's keys — whatever the run behind it.

Round 5's line had grown to 26 KB and the driver could not parse it; these tests hold the formatter (bench.compact_line)
to its promises on canned records: that very 26 KB record (profiles/round5_bench_final.json), a record padded far beyond
it, a 64-rank record, and a record with non-finite numbers."""
import copy
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "alg_bytes_per_launch")
CPU = ("value", "unit", "cores", "kind", "sample")


@pytest.fixture(scope="module")
def bench():
    import bench as b  # (imports torch: the CPU build is enough)
    return b


@pytest.fixture()
def canned():
    return json.load(open(os.path.join(ROOT, "profiles", "round5_bench_final.json")))


def strict(line):
    def no_constants(x):
        raise ValueError("non-standard JSON constant " + x)
    return json.loads(line, parse_constant=no_constants)


def check(line):
    assert "\n" not in line
    assert len(line.encode()) <= 8192
    d = strict(line)
    for k in REQUIRED:
        assert k in d, k
    assert isinstance(d["value"], (int, float)) and d["value"] > 0
    assert d["unit"] == "Msamples/s" and d["dtype"] == "f32"
    assert len(d["config"]["workload"]) <= 200 and "model" not in d["config"]
    for k in ROOFLINE:
        assert k in d["roofline"], k
    assert d["roofline"]["bound"] in ("hbm", "mfma")
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 2e-3
    for k in CPU:
        assert k in d["cpu_baseline"], k
    return d


def test_round5_record_fits(bench, canned):
    assert len(json.dumps(canned)) > 20000  # the record that could not be parsed
    d = check(bench.compact_line(canned, "gpurun_out/bench_detail.json"))
    assert d["value"] == canned["value"] and d["ms_per_step"] == canned["ms_per_step"]
    assert d["roofline"]["frac_rocprof"] == canned["roofline"]["frac_rocprof"]
    assert d["cpu_baseline"]["pipeline"]["value"] == canned["cpu_baseline"]["pipeline"]["value"]
    assert "configs" not in d and "kernels" not in d and "stage_ms_per_pass" not in d
    assert d["detail"] == "gpurun_out/bench_detail.json"


def test_prose_cannot_grow_the_line(bench, canned):
    r = copy.deepcopy(canned)
    r["config"]["workload"] = "w" * 5000
    r["config"]["essay"] = "x" * 100000
    r["roofline"]["kernel"] = "k" * 5000
    r["roofline"]["launch"] = "y" * 50000
    r["cpu_baseline"]["sample"] = "s" * 9000
    r["collective"] = "c" * 9000
    r["device"] = "d" * 9000
    r["note"] = "z" * 100000
    check(bench.compact_line(r, "gpurun_out/bench_detail.json"))


def test_many_ranks_still_fit(bench, canned):
    r = copy.deepcopy(canned)
    r["n_gpus"] = 64
    r["ranks"] = [{"rank": k, "windows_per_pass": 3, "of": 17, "rows": [k * 32, k * 32 + 32], "rccl_ranks": 64, "device": k % 8,
                   "argmax": [516855, 715], "epochs_replayed_exact": 0, "essay": "e" * 1000} for k in range(64)]
    d = check(bench.compact_line(r, None))
    assert len(d["ranks"]) == 64 and all(x["rccl_ranks"] == 64 for x in d["ranks"])
    r["ranks"] = r["ranks"][:8]
    d = check(bench.compact_line(r, None))
    assert d["ranks"][3] == {"rank": 3, "windows_per_pass": 3, "of": 17, "rows": [96, 128], "rccl_ranks": 64, "device": 3, "argmax": [516855, 715],
                             "epochs_replayed_exact": 0}


def test_non_finite_numbers_become_null(bench, canned):
    r = copy.deepcopy(canned)
    r["roofline"]["traffic"] = float("nan")
    r["frame_path"]["frac_moved"] = float("inf")
    r = bench._json_safe(r)
    d = check(bench.compact_line(r, None))
    assert d["roofline"]["traffic"] is None


def test_missing_objects_do_not_break_the_line(bench, canned):
    """a run without the instrumented passes / without the CPU baseline still prints a parseable line"""
    r = copy.deepcopy(canned)
    r["roofline"] = None
    del r["cpu_baseline"]
    r["e2e"] = None
    line = bench.compact_line(r, None)
    d = strict(line)
    assert d["roofline"] is None and "cpu_baseline" not in d and len(line) < 2048
