"""bench.py's contract with the driver (no GPU): the last stdout line is a compact JSON object that carries the roofline and cpu_baseline
objects (round 3's 26.8 KB line was not parsed), and `--gpus N` without a launcher starts N ranks itself.
Reference shape of the report: benchmarks/src/tpch/run.rs:120-156 (one query: per-iteration times and their average)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports neither torch nor the library at module level)


def canned(n_workloads=40, note_len=4000):
    kernels = {"k_kernel_%d" % i: 0.123 for i in range(40)}
    w = {"workload_%d" % i: {"ms_per_step": 1.0 + i, "rows_per_s": 1e9, "note": "x" * note_len, "kernel_ms_per_step": kernels, "step_ms": [1.0, 2.0, 3.0]} for i in range(n_workloads)}
    return {"metric": bench.METRIC, "value": 2.17e11, "unit": "rows/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 3.52, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": "TPC-H SF100 Q3", "input_rows": 765043119, "result_rows": 1130448, "parallelism": "1 GPU", "paths": "p" * 200},
            "roofline": {"bound": "hbm", "kernel": "k_probe_match_bitmap", "achieved": 4047.2, "peak": 8000.0, "unit": "GB/s", "frac": 0.5059, "traffic": 3234802972, "launches_per_step": 2.0,
                         "avg_launch_ms": 0.7645, "algorithmic_bytes_per_launch": 3093927865, "measured_copy_GBps": 4483.8, "kernel_ms_per_step": kernels, "host_syncs_per_step": {"a": 1, "b": 1},
                         "traffic_source": "t" * 500},
            "cpu_baseline": {"value": 3.06e8, "unit": "rows/s", "cores": 16, "cpu_model": "AMD EPYC", "kind": "port", "sample": "s" * 300, "acero": {"value": 1.4e8, "note": "n" * 500}},
            "result_check": {"ok": True, "what": "w" * 500, "device": [1, 2, 3]}, "ranks_seen": 1,
            "q3_general_paths": {"ms_per_step": 6.6, "kernel_ms_per_step": kernels}, "q3_shuffled_inputs": {"ms_per_step": 11.3, "kernel_ms_per_step": kernels}, "workloads": w}


def test_line_is_compact_and_carries_the_objects():
    d = canned()
    assert len(json.dumps(d)) > 100_000                      # the full object is far beyond what a line may hold
    text = bench.compact_line(d)
    assert "\n" not in text and len(text) < 8192
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["roofline"]["frac"] == 0.5059 and line["roofline"]["bound"] == "hbm" and line["roofline"]["traffic"] == 3234802972
    assert set(line["roofline"]) == set(bench.ROOFLINE_KEYS)
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 16 and "acero" not in line["cpu_baseline"]
    assert line["result_check"] == {"ok": True} and line["ranks_seen"] == 1 and line["host_syncs_per_step"] == 2
    assert line["ms_per_step_other"]["workload_3"] == 4.0 and line["ms_per_step_other"]["q3_shuffled_inputs"] == 11.3
    assert all(isinstance(v, (int, float)) for v in line["ms_per_step_other"].values())


def test_line_of_the_default_workload_set_is_under_4k():
    d = canned(n_workloads=24)
    assert len(bench.compact_line(d)) < 4096


def test_line_never_exceeds_the_cap():
    d = canned(n_workloads=2000)                              # an absurd number of nested workloads: the optional part is dropped, the headline stays
    line = json.loads(bench.compact_line(d))
    assert "ms_per_step_other" not in line and line["roofline"]["frac"] == 0.5059 and line["cpu_baseline"]["value"] == 3.06e8


def test_line_of_the_recorded_round3_result():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_e_bench_sf100_default_run.json")))
    text = bench.compact_line(d)
    assert len(text) < 4096
    assert json.loads(text)["roofline"]["kernel"] == "k_probe_match_bitmap"


def run_bench(argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` alone (no WORLD_SIZE): two ranks rendezvous (the launcher branch; gloo on the CPU, no GPU touched)."""
    r = run_bench(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert last["n_gpus"] == 2 and last["ranks_seen"] == 2


def test_world_size_must_equal_gpus():
    r = run_bench(["--gpus", "4", "--launch-check"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    r = run_bench(["--gpus", "1", "--launch-check"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0
