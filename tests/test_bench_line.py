"""
The driver parses ONE JSON line from bench.py's stdout.  Round 4's line had grown to 20 406 bytes and BENCH_r04.json recorded
"parsed": null -- the round's headline was unmeasured.  These tests bound the line: bench.slim_line() is a pure function of the full
result dict, fed here with round 4's full record (profiles/r4_lda_k50_bench.json, 20 KB: four configurations with per-iteration
parity dumps) and with a synthetic multi-GPU record; the GPU test tests/test_bench_torchrun_gpu.py asserts the same on a real line.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports no torch at module level)


def _full():
    return json.load(open(os.path.join(ROOT, "profiles", "r4_lda_k50_bench.json")))


def test_slim_line_is_small_and_round_trips():
    full = _full()
    assert len(json.dumps(full)) > 15000                      # the canned record really is the oversized one
    line = bench.slim_line(full)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 6000
    r = json.loads(line)
    # the contract fields
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in r, k
    assert r["config"]["workload"].startswith("LDA K=50") and "model" not in r["config"]
    assert abs(r["value"] - full["value"]) / full["value"] < 1e-6
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    assert rf["algorithmic_bytes"] == full["roofline"]["algorithmic_bytes_per_estep"] and rf["traffic"] and rf["estep_ms"] > 0
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and len(cb["sample"]) <= 90
    assert r["parity"]["pass"] is True and r["parity"]["documents"] == 128804 and "per_iteration" not in r["parity"]
    # the side configurations keep their own roofline / cpu_baseline / parity verdicts
    assert set(r["other_configs"]) == {"lda100", "ctm", "ctpf"}
    for name, o in r["other_configs"].items():
        assert o["roofline"]["frac"] > 0 and o["cpu_baseline"]["value"] > 0 and o["parity"]["pass"] is True, name
        assert "per_iteration" not in o["parity"] and "tolerances" not in o["parity"]
    assert r["other_configs"]["ctm"]["roofline"]["bound"] == "valu"
    assert r["elbo_plateau"]["reached"] is True and "elbo_vs_wallclock" not in r["elbo_plateau"]


def test_slim_line_with_multi_gpu_arrays_and_errors():
    full = _full()
    full["n_gpus"] = 8
    full["multi_gpu_check"] = {"iterations": 5, "globals_hash_equal": True, "globals_hash_per_rank": ["%016x" % i for i in range(8)],
                               "elbo_rel_vs_n1": 3e-8, "elbo_rel_tolerance": 2e-6, "elbo_n": [-1.0e8] * 5, "elbo_n1": [-1.0e8] * 5, "pass": True,
                               "form": "fused", "shard_nnz_max_over_min": 1.001}
    full["other_configs"]["ctm"] = {"error": "RuntimeError: " + "x" * 5000}
    full["cpu_baseline"] = None
    full["parity"] = None
    line = bench.slim_line(full)
    assert len(line) < 6000
    r = json.loads(line)
    assert r["multi_gpu_check"]["pass"] is True and "elbo_n" not in r["multi_gpu_check"] and "globals_hash_per_rank" not in r["multi_gpu_check"]
    assert len(r["other_configs"]["ctm"]["error"]) <= 200 and r["cpu_baseline"] is None


def test_slim_line_sheds_before_it_overflows():
    """whatever the side configurations carry, the printed line stays under the limit (they are shed first, the headline never)"""
    full = _full()
    for o in full["other_configs"].values():
        o["parity"]["worst"] = {f"metric_{i}": 1e-7 * i for i in range(120)}
    line = bench.slim_line(full)
    assert len(line) < 6000
    r = json.loads(line)
    assert r["roofline"]["frac"] > 0 and r["cpu_baseline"]["value"] > 0 and r["parity"]["pass"] is True


def test_non_finite_numbers_do_not_break_json():
    full = _full()
    full["roofline"]["estep_ms"] = float("nan")
    r = json.loads(bench.slim_line(full))                      # strict JSON: NaN would not parse everywhere
    assert r["roofline"]["estep_ms"] is None
