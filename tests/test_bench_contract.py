"""The bench line the driver parses: every key of the contract, on the line recorded with the final build of the round
(profiles/r03_bench_final.json = stdout of `python bench.py` on one MI355X)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_bench_line_has_the_contract_keys():
    line = open(os.path.join(ROOT, "profiles", "r03_bench_final.json")).read().strip()
    assert "\n" not in line                                   # ONE JSON line
    d = json.loads(line)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert base["metric"].startswith(d["metric"]) and d["metric"].startswith("MDoF/s per V-cycle")   # BASELINE's headline clause
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4097 * 4097 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # a roofline fraction: bytes the launch must move / time / peak, never above 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] <= 1.0
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["unfused_equivalent_gbs"] > r["achieved"]                      # the per-operator accounting lives under its own key
    assert r["traffic"] is None or (0.9 * r["bytes_per_launch"] < r["traffic"] < 1.5 * r["bytes_per_launch"] and r["traffic_source"])
    for p in ("f32", "f64"):
        for name, k in r["kernels"][p].items():
            assert 0.0 < k["frac"] <= 1.0, (p, name)
    s = r["smoother_hbm"]                                                   # the north-star kernel, HBM proper, beside its ceiling
    assert 0.0 < s["frac"] <= 1.0 and 0.5 < s["frac_of_stream_ceiling"] <= 1.05 and s["target_frac"] == 0.70
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"]
    ref = d["reference_cpu_captured"]                                       # the reference's own CPU V-cycle at the bench size
    assert ref["kind"] == "reference" and ref["cores"] == 1 and "4097^2" in ref["sample"] and 20.0 < ref["seconds_per_cycle"] < 40.0
    assert d["iterations"] == d["steps"] and d["residual_floor"] > 0 and 1 <= d["iterations_to_floor"] <= 40
    assert d["iterations_to_1e-10_absolute"] is None                        # unreachable at 4097^2 in fp64: the floor says why
    # round 3: time to solution per policy, the adaptive policy never behind double, the size sweep
    t = d["time_to_solution"]
    for pol in ("adaptive", "double", "defect"):
        assert t[pol]["iterations_to_floor"] >= 1 and t[pol]["time_to_floor_ms"] > 0
    assert t["adaptive"]["time_to_floor_ms"] <= 1.05 * t["double"]["time_to_floor_ms"]
    assert d["switch_reason"] in ("fp32_skipped", "fp32_floor", "threshold", "stagnation")
    assert [row["n"] for row in d["smoother_size_sweep"]["rows"]] == [4097, 8193, 16385]
    assert "span" in r["kernel"] and "span_leg_nomid" in r["kernels"]["f64"]      # the spanning leg is the dominant kernel
    assert r["build"] and ("traffic_is_from_this_build" in r)
