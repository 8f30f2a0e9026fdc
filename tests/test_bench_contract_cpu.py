"""CPU-only checks of bench.py's contract (the driver parses ONE JSON line): the committed line of the newest round (profiles/rNN_bench.json, written by
`python bench.py` on an MI355X) carries every key the contract names, `roofline` and `cpu_baseline` included, and its numbers are consistent
with each other (value = units / time, frac = achieved / peak, the committed rocprofv3 average of the dominant kernel agrees with the
in-library timing)."""
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=lambda p: int(re.search(r"r(\d+)_", os.path.basename(p)).group(1)))
    assert files, pattern
    return files[-1]


def bench_line():
    text = open(newest("r*_bench.json")).read().strip().splitlines()
    return json.loads(text[-1])


def test_the_line_has_the_contracts_keys():
    d = bench_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "probe_rows_per_sec" and d["unit"] == "rows/s"      # BASELINE.json: "probe rows/sec ... TPCH-SF100 ... Q3 join"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1


def test_the_lines_numbers_are_consistent():
    d = bench_line()
    probe_rows = d["config"]["lineitem_probe_rows"]
    assert abs(d["value"] - probe_rows / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_step"] / (r["kernel_ms_per_step"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["kernel_ms_per_step"] < d["ms_per_step"]                    # the dominant kernel fits inside the step
    assert r["traffic"] is None or 0.5 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 2.0
    assert all(v if isinstance(v, bool) else v.get("ok", True) for v in d["checks"].values() if not isinstance(v, dict) or "ok" in v)
    for name, sub in d["sub_benchmarks"].items():
        assert sub["ok"], name
        rr = sub["request_roofline"]
        assert rr["bound"] == "random_access" and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-9, name


def test_rocprof_average_of_the_dominant_kernel_is_committed_next_to_the_line():
    d = bench_line()
    stats = newest("r*_kernel_stats.csv")
    rows = {r["Name"]: r for r in csv.DictReader(open(stats))}
    assert "fj_probe_direct" in rows, "the dominant kernel of the headline (fused_filter_probe = fj_probe_direct) is missing from the rocprofv3 summary"
    avg_ms = float(rows["fj_probe_direct"]["AverageNs"]) / 1e6
    # different runs (and boxes) of the same command: within 25 % of the line's HIP-event average
    assert 0.75 < avg_ms / d["roofline"]["avg_launch_ms"] < 1.25, (avg_ms, d["roofline"]["avg_launch_ms"])
    pmc = json.load(open(newest("r*_pmc_traffic.json")))["kernels"]
    assert "fused_filter_probe" in pmc and pmc["fused_filter_probe"]["traffic_bytes_per_launch_avg"] > 0
