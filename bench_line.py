"""
bench_line.py -- the ONE JSON line bench.py prints, kept small enough for the driver to parse (round 4's line had grown past
20 KB and BENCH_r04.json.parsed came back null).

bench.py assembles the FULL record (every roofline object, config record, per-step list, note) exactly as before; that record goes
to gpurun_out/bench_full.json.  compact_line() projects it onto scalars: the contract keys (metric, value, unit, n_gpus, steps,
warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config), `roofline` of the dominant kernel, `cpu_baseline`,
and the verdicts of every other record as scalars under config.checks.  tests/test_bench_line.py builds the line from the committed
round-4 record and from a worst-case record and asserts len(line) < 4096 and that json.loads round-trips.
"""
import json
import math

MAX_LINE_BYTES = 4096


def _get(obj, *path):
    for key in path:
        if not isinstance(obj, dict) or key not in obj:
            return None
        obj = obj[key]
    return obj


def _num(x, digits=5):
    """Scalars only, floats at `digits` significant digits (1.1098e13 reads the same as 11098412345123.77 and is 10 bytes shorter)."""
    if isinstance(x, bool) or x is None or isinstance(x, int):
        return x
    if isinstance(x, float):
        if math.isnan(x) or math.isinf(x):
            return None
        return float("%.*g" % (digits, x))
    return x


def _short(s, n):
    if s is None:
        return None
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "~"


def _roofline(r, kernel_chars=80):
    if not isinstance(r, dict) or "error" in r:
        return None
    out = {"kernel": _short(r.get("kernel"), kernel_chars), "bound": r.get("bound"), "achieved": _num(r.get("achieved")),
           "peak": _num(r.get("peak")), "unit": r.get("unit"), "frac": _num(r.get("frac")), "traffic": _num(r.get("traffic")),
           "avg_launch_ms": _num(r.get("avg_launch_ms"))}
    if r.get("traffic_stale"):
        out["traffic_stale"] = _short(r["traffic_stale"], 48)
    if r.get("traffic") is not None:
        out["traffic_from_profile"] = True            # PMC counters cannot be read inside the run: committed profiles/ file of the same sources
    return out


def _cfg_verdicts(configs):
    """One boolean per config record: every verdict the record carries is true."""
    if not isinstance(configs, dict):
        return None
    out = {}
    for name, rec in configs.items():
        if not isinstance(rec, dict) or "error" in rec:
            out[name] = False
            continue
        vs = [v for v in (_get(rec, "green"), _get(rec, "parity_one_step_vs_oracle", "green"),
                          _get(rec, "parity", "topk_ids_bit_exact_vs_oracle")) if v is not None]
        out[name] = bool(vs) and all(bool(v) for v in vs)
    return out


def compact_line(full, full_path=None):
    """Project bench.py's full record onto the driver's line.  Every value is a scalar or a flat object of scalars."""
    cfg = full.get("config") or {}
    parity, fit = full.get("parity"), full.get("fit")
    step_ms = _get(full, "step_ms_by_hip_events") or {}
    checks = {
        "parity_users": _get(parity, "sample_users"),
        "topk_ids_bit_exact": _get(parity, "topk_ids_bit_exact_vs_oracle"),
        "topk_values_bit_exact": _get(parity, "topk_values_bit_exact_vs_oracle"),
        "fp32_mfma_equals_all_users": _get(full, "fp32_mfma_mode", "equals_all_users"),
        "fp32_mfma_frac": _num(_get(full, "fp32_mfma_mode", "frac_of_fp32_mfma_peak"), 4),
        "bf16_filter_equals_cascade": _get(full, "bf16_filter_mode", "equals_timed_cascade_output"),
        "bf16_dense_frac": _num(_get(full, "roofline_bf16_dense", "frac"), 4),
        "k1_frac_of_hbm": _num(_get(full, "roofline_k1", "frac"), 4),
        "k1_multi_nnz_frac": _num(_get(full, "roofline_k1_multi_nnz", "frac"), 4),
        "refine_ms": _num(_get(full, "roofline_bf16_stage", "avg_launch_ms"), 4),
        "refined_fraction_of_pairs": _num(_get(full, "roofline_bf16_stage", "refined_fraction_of_pairs"), 3),
        "public_api_ms": _num(_get(full, "public_api_mode", "ms_per_call_min"), 4),
        "public_api_equals_step": _get(full, "public_api_mode", "equals_timed_step_output"),
        "trained_ms": _num(_get(full, "trained_weights_mode", "ms_per_step"), 4),
        "trained_int8_ms": _num(_get(full, "trained_weights_mode", "kernels_avg_ms", "score_gemm_blockmax_i8"), 4),
        "trained_bit_exact": _get(full, "trained_weights_mode", "parity", "topk_ids_bit_exact_vs_oracle"),
        "parity_fit_green": _get(full, "parity_fit", "green"),
        "parity_fit_binned": _get(full, "parity_fit_binned", "green"),
        "parity_multi_nnz_ids": _get(full, "parity_multi_nnz", "topk_ids_bit_exact_vs_oracle"),
        "step_ms_first": _num(step_ms.get("first"), 4), "step_ms_last": _num(step_ms.get("last"), 4),
        "step_ms_min": _num(step_ms.get("min"), 4), "step_ms_max": _num(step_ms.get("max"), 4),
        "prewarm_steps": full.get("prewarm_steps"),
        "dispatches_per_step": _get(full, "dispatches_per_step"),
        "emulated_rank_ms_predict_n8": _num(_get(full, "scale_emulation", "predict", "per_rank_step_ms"), 4),
        "emulated_rank_ms_fit_n8": _num(_get(full, "scale_emulation", "fit", "per_rank_compute_ms_per_step"), 4),
        "cfg": _cfg_verdicts(full.get("configs")),
    }
    if checks.get("emulated_rank_ms_predict_n8") is not None or checks.get("emulated_rank_ms_fit_n8") is not None:
        checks["emulated_from_profile"] = True        # one-GPU emulations of one rank of 8, read from profiles/ -- not measured in this run
    if _get(full, "fp32_mfma_mode") and checks["fp32_mfma_equals_all_users"] is None:
        # (records of round 4 carry the user count inside the key)
        for key, val in full["fp32_mfma_mode"].items():
            if key.startswith("equals_timed_exact_mode_output_all_"):
                checks["fp32_mfma_equals_all_users"] = val
    for name in ("parity", "fit", "fp32_mfma_mode", "bf16_filter_mode", "public_api_mode", "trained_weights_mode", "parity_fit",
                 "parity_fit_binned", "parity_multi_nnz", "cpu_baseline_fit", "roofline_k1_multi_nnz"):
        err = _get(full, name, "error")
        if err is not None:
            checks.setdefault("errors", {})[name] = _short(err, 60)
    checks = {k_: v for k_, v in checks.items() if v is not None}

    config = {"workload": _short(cfg.get("workload"), 200)}
    for key in ("users", "items", "n_components", "top_k", "parallelism", "exchange", "collective_selfcheck", "topk_method"):
        if cfg.get(key) is not None:
            config[key] = cfg[key] if not isinstance(cfg[key], str) else _short(cfg[key], 80)
    config["checks"] = checks

    cpu = full.get("cpu_baseline")
    if isinstance(cpu, dict) and "error" not in cpu:
        cpu = {"value": _num(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
               "sample": _short(cpu.get("sample"), 160)}
    elif cpu is not None:
        cpu = {"error": _short(_get(cpu, "error"), 80)}

    line = {
        "metric": full.get("metric"), "value": _num(full.get("value"), 6), "unit": full.get("unit"), "n_gpus": full.get("n_gpus"),
        "steps": full.get("steps"), "warmup": full.get("warmup"), "ms_per_step": _num(full.get("ms_per_step"), 6),
        "higher_is_better": full.get("higher_is_better"), "scaling": full.get("scaling"), "vs_baseline": full.get("vs_baseline"),
        "dtype": full.get("dtype"), "result_precision": _short(full.get("result_precision"), 48), "data": full.get("data"),
        "config": config,
        "roofline": _roofline(full.get("roofline")),
        "cpu_baseline": cpu,
        "fit_epochs_per_s": _num(_get(fit, "fit_epochs_per_sec")),
        "fit_ms_per_epoch": _num(1e3 * _get(fit, "sec_per_epoch"), 5) if _get(fit, "sec_per_epoch") else None,
        "roofline_fit_frac": _num(_get(full, "roofline_fit", "frac"), 4),
        "roofline_fit_frac_of_gather_ceiling": _num(_get(full, "roofline_fit", "frac_of_gather_ceiling"), 4),
        "roofline_fit": _roofline(full.get("roofline_fit"), 48),
        "cpu_baseline_fit_value": _num(_get(full, "cpu_baseline_fit", "value")),
        "full_record": full_path,
    }
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= MAX_LINE_BYTES:
        # never let the line outgrow the driver again: shed the optional parts in order of (un)importance
        for drop in ("roofline_fit", "result_precision", "full_record"):
            line.pop(drop, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < MAX_LINE_BYTES:
                break
        if len(text) >= MAX_LINE_BYTES:
            line["config"]["checks"] = {k_: v for k_, v in checks.items() if isinstance(v, bool)}
            line["config"]["workload"] = _short(line["config"]["workload"], 100)
            text = json.dumps(line, separators=(",", ":"))
    assert len(text) < MAX_LINE_BYTES, len(text)
    return text
