"""The bench line the driver parses: the committed profile of the last GPU run (profiles/r06_bench.json, written by
tools/profile_round.sh from `python bench.py`) must carry every field of the contract with consistent values. Runs on CPU."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
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
    s = json.load(open(os.path.join(ROOT, "profiles", "r06_summary.json")))
    assert abs(s["rocprof_kernel_avg_ms"] - r["kernel_avg_ms"]) < 0.1 * r["kernel_avg_ms"]
    assert r["traffic"] is None or r["traffic"] > alg      # HBM traffic by PMC is above the algorithmic bytes
    # ... and what the memory system really moves while the kernel runs stands beside the algorithmic rate
    assert r["traffic"] is None or abs(r["traffic_GBps"] - r["traffic"] / (r["kernel_avg_ms"] * 1e-3) / 1e9) < 1e-6 * r["traffic_GBps"]
    assert r["traffic"] is None or (r["frac"] < r["traffic_frac_of_peak"] < 1.0)
    # since round 5: the step is the pass the product runs for one sample -- ONE launch on a text-only reference (K1 - K3) + reduce -- at the
    # REQUESTED warm-up; plan + screen as two launches and the amortised form are measured beside it and the fractions follow
    assert d["step_form"] == "one-pass" and d["step"].startswith("one pass") and d["warmup"] == 5
    assert r["kernel_instance"] == "screen_k31c_kernel<31, true, false>" and r["kernel_instance"] in s["rocprof_kernel"]
    o = d["other_forms"]
    assert d["forms_identical"] is True and o["amortised"]["value"] > d["value"] > o["two-launch"]["value"]
    assert abs(r["frac_one_pass"] - r["frac"]) < 1e-12
    assert abs(r["frac_two_launch"] - alg / ((o["two-launch"]["kernel_avg_ms"] + o["two-launch"]["ref_plan_ms"]) * 1e-3) / 1e9 / r["peak"]) < 1e-9
    assert abs(r["frac_amortised"] - alg / (o["amortised"]["kernel_avg_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-9
    assert abs(s["rocprof_ref_plan_avg_ms"] - r["ref_plan_ms"]) < 0.1 * r["ref_plan_ms"] and r["ref_plan_ms"] <= 0.030
    assert d["ramped"]["untimed_steps_before"] > d["warmup"] + d["steps"]             # the clock-ramped repeat stands BESIDE the line's figures
    # what a launch costs by what came before it: a read-only sweep of the caches costs little (nothing of the hot loop's speed is
    # owed to the previous launch's data beyond what the Infinity Cache serves), a dirty one and the first launch on a fresh handle do
    cs = d["cache_state"]
    assert cs["hot_loop_ms"] < 1.02 * cs["read_swept_ms"] < 1.2 * cs["hot_loop_ms"] and cs["read_swept_ms"] < cs["write_swept_ms"]
    assert len(cs["first_pass_ms"]) == 5 and cs["first_pass_median_ms"] > 1.2 * cs["second_pass_median_ms"]      # (on the one-pass table; 2.2 x on round 5's 17 GB table)
    # inputs that look like data: every window of each against the oracle, in both product configurations (round 6): the one-pass job
    # (text-only reference, the table sized for one pass) and the resident engine (planned reference, the table of a kept handle)
    for name in ("read_error_db", "read_error_depth30", "k25", "repeats"):
        x = d["realistic"][name]
        assert x["parity"]["bit_exact"] is True and x["forms_identical"] is True and x["parity"]["windows_checked"] >= 3000, name
        assert 0.1 < x["one_pass"]["frac"] < x["resident"]["frac"] < 1.0
        assert x["one_pass"]["table"]["lines"] < x["resident"]["table"]["lines"] and 0.15 < x["one_pass"]["table"]["kmers_per_cell"] <= 0.30
    # round 6: the step's table IS the one kcfx_db_open_opts builds for one pass (~0.3 k-mers per cell), the PMC traffic figure is of that
    # table, and the same kernels against a resident handle's table stand beside it
    assert r["table"]["policy"] == "product" and r["table"]["lines"] == json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["table_lines"]
    assert 0.15 < r["table"]["kmers_per_cell"] <= 0.30 and r["table"]["expected_queries"] > 0
    rt = d["resident_table"]
    assert rt["identical_to_the_step"] is True and rt["table"]["lines"] > r["table"]["lines"] and rt["table"]["kmers_per_cell"] <= 0.075
    assert abs(r["frac_resident_one_pass"] - rt["one-pass"]["frac"]) < 1e-12 and abs(r["frac_resident_amortised"] - rt["amortised"]["frac"]) < 1e-12
    assert abs(rt["one-pass"]["frac"] - alg / (rt["one-pass"]["kernel_avg_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-9
    # ... the end-to-end leg carries every run with its phases, and two more runs right after the process released its device memory
    e = d["end_to_end"]
    assert len(e["runs"]) == 4 and e["seconds_wall_all_runs"] == [x["seconds_wall"] for x in e["runs"]]
    for run in e["runs"] + e["after_churn"]["runs"]:
        ph = run["phases_ms"]
        assert any(k.startswith("cli: kcfx_screen_sharded") for k in ph) and any(k.startswith("load: ") for k in ph)
        rest = "process start + exit (wall - the driver's phases)"
        assert abs(sum(v for k, v in ph.items() if k.startswith("cli: ") or k == rest) - 1e3 * run["seconds_wall"]) < 1.0
        # ... and the rest split by the driver's wall-clock stamps: spawn -> main(), the return from main() -> reaped
        split = [ph[k] for k in ("process start (spawn -> main)", "process exit (leaving -> reaped)")]
        assert all(v >= 0 for v in split) and abs(sum(split) - ph[rest]) < 5.0
    assert len(e["after_churn"]["runs"]) == 2 and max(e["seconds_wall_all_runs"] + e["after_churn"]["seconds_wall_all_runs"]) < d["cpu_end_to_end"]["seconds_wall"]
    assert d["setup_seconds"]["db_open"] < 0.3
    # the CPU leg says how many cores it really had
    assert c["host_cores"] <= c["host_cores_logical"] and str(c["host_cores"]) in c["by_threads"]


def test_gpus_n_run_plainly_starts_the_ranks_as_a_child(tmp_path, monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE must not die on argument handling: it starts
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process (before anything imports torch or
    touches a GPU), relays rank 0's JSON line on stdout and returns the child's exit code."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "20"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511" and cmd[-4:] == ["--gpus", "4", "--steps", "20"]
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], os.path.join(ROOT, "bench.py"))

    # the relay, with a stand-in for the ranks: a fake `torch.distributed.run` module earlier on the path
    pkg = tmp_path / "torch" / "distributed"
    pkg.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "run.py").write_text(
        "import os, sys\n"
        "assert os.environ.get('KCFX_BENCH_CHILD') == '1' and os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'\n"
        "print('rank chatter')\n"
        "print('{\"metric\": \"k-mers screened/sec, getVariations k=31\", \"value\": 1.0, \"n_gpus\": 2}')\n"
        "sys.exit(int(os.environ.get('FAKE_RC', '0')))\n")
    env = dict(os.environ, PYTHONPATH=str(tmp_path))
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3"], capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stderr
    out = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 2            # ONE JSON line on stdout ...
    assert "rank chatter" in p.stderr                                     # ... the rest on stderr
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(env, FAKE_RC="7"))
    assert p.returncode == 7                                              # a failing job fails the command


def test_usable_cores_is_the_affinity_mask_not_the_logical_count():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))
