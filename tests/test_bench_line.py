"""bench.py's stdout line must stay small enough for the driver to parse (BENCH_r04.json.parsed was null: the line was > 20 KB)."""
import json
import os

import bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(text):
    assert "\n" not in text
    assert len(text.encode()) < 4096
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert "workload" in line["config"] and "model" not in line["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    assert len(line["roofline"]["kernel"]) <= 80
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key

    def scalars_only(obj, depth=0):
        for v in obj.values():
            if isinstance(v, dict):
                assert depth < 3
                scalars_only(v, depth + 1)
            else:
                assert v is None or isinstance(v, (bool, int, float, str)), v
    scalars_only(line)
    return line


def test_line_from_the_round4_record():
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    line = _check(bench_line.compact_line(full, "gpurun_out/bench_full.json"))
    assert line["value"] == float("%.6g" % full["value"])
    assert abs(line["ms_per_step"] - full["ms_per_step"]) < 1e-3
    assert line["roofline"]["bound"] == "mfma" and abs(line["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-4
    assert line["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
    assert line["config"]["checks"]["topk_ids_bit_exact"] is True
    assert line["config"]["checks"]["cfg"] == {"cfg0": True, "cfg1": True, "cfg3_shard": True, "cfg3_full": True, "cfg4": True}
    assert abs(line["fit_epochs_per_s"] - full["fit"]["fit_epochs_per_sec"]) < 1e-2


def test_line_stays_small_on_a_bloated_record():
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    full["step_ms_by_hip_events"]["all"] = [90.123456789] * 5000
    full["config"]["workload"] = "w" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["configs"].update({"cfg%d" % i: {"green": True} for i in range(5, 40)})
    full["parity"] = {"error": "e" * 5000}
    full["roofline"]["traffic"], full["roofline"]["traffic_stale"] = None, "r02_fit_pmc_summary.txt"
    line = _check(bench_line.compact_line(full, "gpurun_out/bench_full.json"))
    assert line["roofline"]["traffic"] is None and line["roofline"]["traffic_stale"]


def test_line_survives_missing_records():
    full = {"metric": "user-item predictions/sec", "value": 1.0e13, "unit": "predictions/s", "n_gpus": 8, "steps": 20, "warmup": 5,
            "ms_per_step": 12.5, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i8/bf16/fp32",
            "data": "synthetic", "config": {"workload": "x", "users": 10, "items": 10, "n_components": 128, "top_k": 10,
                                            "parallelism": "items sharded x8, users replicated"},
            "roofline": {"kernel": "k", "bound": "mfma", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5, "traffic": None},
            "cpu_baseline": None, "fit": {"error": "boom"}}
    text = bench_line.compact_line(full)
    assert len(text) < 4096
    line = json.loads(text)
    assert line["cpu_baseline"] is None and line["config"]["checks"]["errors"]["fit"] == "boom"
