"""The bench line the driver parses: the committed profile of the last GPU run (profiles/r02_bench.json, written by
tools/profile_round.sh from `python bench.py`) must carry every field of the contract with consistent values. Runs on CPU."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "k-mers/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert "k-mers" in d["metric"] and "k-mers" in base["metric"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / measured kernel duration
    alg = d["config"]["kmers_per_step_per_gpu"] * r["algorithmic_bytes_per_kmer"]
    assert abs(r["achieved"] - alg / (r["kernel_avg_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    # whole-job throughput from the step time
    assert abs(d["value"] - d["config"]["kmers_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "by_threads"):
        assert key in c, key
    assert c["kind"] == "port" and c["unit"] == "k-mers/s" and c["cores"] >= 1
    assert "1" in c["by_threads"] and all(v["seconds"] > 2.0 for v in c["by_threads"].values())     # >= a few seconds of wall each
    assert c["value"] == max(v["value"] for v in c["by_threads"].values())
    # round-2 fields: the end-to-end leg, the JVM probe, where the traffic figure comes from
    assert "jvm" in d and "end_to_end" in d and d["end_to_end"]["seconds_wall"] > 0 and d["end_to_end"]["windows"] == d["config"]["windows"]
    assert r["kernel"] == "screen_k31c_kernel" and (r["traffic"] is None or "not measured in this run" in r["traffic_source"])
    assert d["parity"]["bit_exact"] is True
    # the rocprofv3 summary committed next to it agrees with the HIP-event figure within 10 %
    s = json.load(open(os.path.join(ROOT, "profiles", "r02_summary.json")))
    assert abs(s["rocprof_kernel_avg_ms"] - r["kernel_avg_ms"]) < 0.1 * r["kernel_avg_ms"]
    assert r["traffic"] is None or r["traffic"] > alg      # HBM traffic by PMC is above the algorithmic bytes
