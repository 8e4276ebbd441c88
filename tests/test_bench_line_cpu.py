"""bench.py prints ONE line the driver can parse: the compact view (< 4 KB) of the full record, which goes to
bench_detail.json.  Round 4's 25 KB line came back as `parsed: null`; this pins the assembler on that very record
(profiles/r04_final_bench_line.json, the full line of the round-4 run) and on degenerate records."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def canned():
    return json.loads(open(os.path.join(ROOT, "profiles", "r04_final_bench_line.json")).read())


def test_compact_line_is_small_and_round_trips():
    full = canned()
    assert len(json.dumps(full)) > 20000  # the record that broke the driver's parser
    s = bench.compact_line(full)
    assert "\n" not in s and len(s) < bench.MAX_LINE_BYTES - 512  # (head room for longer CPU model names / more digits)
    line = json.loads(s)
    for k in REQUIRED:
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert line["config"]["workload"].startswith("GpuIndexFlatL2")
    r = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_kernel_ms", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert isinstance(r["traffic"], (int, float))  # HBM bytes per launch, a number (or None), never a nested block
    for k in ("value", "unit", "cores", "kind"):
        assert k in line["cpu_baseline"], k
    for name in ("ivfpq", "ivfflat", "ivfflat_10m", "ivfpq_100m", "ivfpq_shards"):
        leg = line["legs"][name]
        assert leg["qps"] == full[name]["qps"] and leg["ms_per_step"] == full[name]["ms_per_step"]
        assert 0 < leg["frac"] < 1.5
    for name in ("ivfpq", "ivfflat", "ivfflat_10m", "ivfpq_100m"):
        assert line["legs"][name]["real_mismatches"] == 0
        assert line["legs"][name]["cpu_qps"] == full[name]["cpu_baseline"]["value"]
    assert line["detail"] == bench.DETAIL_NAME


def test_compact_line_survives_errors_and_oversized_values():
    full = canned()
    full["ivfpq"] = {"error": "x" * 5000}
    full["ivfflat_10m"] = {"skipped": "budget"}
    full["ivfpq_100m"]["workload"] = "w" * 3000
    full["cpu_baseline"] = {"error": "e" * 3000}
    full["config"]["workload"] = "GpuIndexFlatL2 " + "y" * 3000
    s = bench.compact_line(full)
    assert len(s) < bench.MAX_LINE_BYTES
    line = json.loads(s)
    assert len(line["legs"]["ivfpq"]["error"]) <= 120 and line["legs"]["ivfflat_10m"] == {"skipped": "budget"}
    # the minimal record of an N > 1 run (no legs, no cpu baseline)
    minimal = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data", "config", "roofline")}
    minimal.update({"ranks_seen": 8, "devices": list(range(8))})
    line = json.loads(bench.compact_line(minimal))
    assert line["ranks_seen"] == 8 and line["devices"] == list(range(8)) and "legs" not in line


def test_cpu_model_is_reported():
    m = bench.cpu_model()
    assert isinstance(m, str) and 0 < len(m) <= 48
