#!/usr/bin/env python3
"""Summarise a tools/profile_round.sh run (gpurun_out/<tag>/) into profiles/<tag>_*: kernel stats, per-launch HBM
traffic of the screening kernel (gfx950 correction from MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte requests at 64 B, so
read bytes = 2 x FETCH_SIZE; checked on the membench calibration kernel), counters, and profiles/traffic.json that
bench.py reads for roofline.traffic."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
COHORT = "--cohort" in sys.argv          # a tools/profile_cohort.sh run: summary only, profiles/traffic.json stays the single-sample one
src = f"gpurun_out/{tag}"
os.makedirs("profiles", exist_ok=True)
# the dominant kernel of the bench workload, as the bench line names it (screen_k31c_kernel on a cell table)
KERNEL = json.load(open(f"{src}/bench.json"))["roofline"]["kernel"]
# A bench process runs three kernels of interest -- the screening kernel on a text-only reference (round 5: the step's own,
# K1 - K3 in one launch), the same template on a planned reference, and ref_plan_kernel; rocprofv3 prints template arguments, which
# tell them apart. HEAD = the instantiation the bench line's roofline is about (roofline.kernel_instance).
_BENCH = json.load(open(f"{src}/bench.json"))
HEAD = _BENCH["roofline"].get("kernel_instance") or (KERNEL + "<31, true, true>")
PLANNED = KERNEL + "<31, true, true>" if KERNEL in ("screen_k31c_kernel", "screen_cohort_kernel") else KERNEL
TEXT_ONLY = KERNEL + "<31, true, false>"
PLAN = "ref_plan_kernel<31>"
OTHER = PLANNED if HEAD == TEXT_ONLY else TEXT_ONLY

def counters(pattern, kernel):
    agg = collections.defaultdict(list)
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


out = {"tag": tag}
bench = json.load(open(f"{src}/bench.json"))
out["kernel"] = KERNEL
out["bench"] = {k: bench.get(k) for k in ("value", "ms_per_step", "step", "step_form", "other_forms", "forms_identical", "ramped", "cache_state",
                                          "realistic", "roofline", "cpu_baseline", "parity", "config")}
for f in glob.glob(f"{src}/stats/*kernel_stats.csv"):
    shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
    for r in csv.DictReader(open(f)):
        if HEAD in r["Name"]:
            out["rocprof_kernel"] = r["Name"].split("(")[0]
            out["rocprof_kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
            out["rocprof_kernel_calls"] = int(r["Calls"])
        elif OTHER in r["Name"]:
            out["rocprof_other_form_kernel"] = r["Name"].split("(")[0]
            out["rocprof_other_form_kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
            out["rocprof_other_form_kernel_calls"] = int(r["Calls"])
        elif PLAN in r["Name"]:
            out["rocprof_ref_plan_avg_ms"] = float(r["AverageNs"]) / 1e6
            out["rocprof_ref_plan_calls"] = int(r["Calls"])
fetch, n1 = counters(f"{src}/pmc_fetch/*counter_collection.csv", HEAD)
write, _ = counters(f"{src}/pmc_write/*counter_collection.csv", HEAD)
tcc, _ = counters(f"{src}/pmc_tcc/*counter_collection.csv", HEAD)
sq, _ = counters(f"{src}/pmc_sq/*counter_collection.csv", HEAD)
sq2, _ = counters(f"{src}/pmc_sq2/*counter_collection.csv", HEAD)
sq.update(sq2)
# the other two kernels: requests, VALU instructions, bytes
others = {}
for name, pat in (("planned_reference" if OTHER == PLANNED else "text_only", OTHER), ("ref_plan", PLAN)):
    f2, _ = counters(f"{src}/pmc_fetch/*counter_collection.csv", pat)
    w2, _ = counters(f"{src}/pmc_write/*counter_collection.csv", pat)
    t2, _ = counters(f"{src}/pmc_tcc/*counter_collection.csv", pat)
    q2, _ = counters(f"{src}/pmc_sq/*counter_collection.csv", pat)
    if f2 or t2 or q2:
        others[name] = {"hbm_read_bytes_per_launch": 2.0 * f2["FETCH_SIZE"] * 1024 if f2.get("FETCH_SIZE") is not None else None,
                        "hbm_write_bytes_per_launch": w2["WRITE_SIZE"] * 1024 if w2.get("WRITE_SIZE") is not None else None, **t2, **q2}
out["other_kernels"] = others
cal, _ = counters(f"{src}/pmc_cal/*counter_collection.csv", "gather_coop<8, 1>")
out["counters"] = {"FETCH_SIZE_KB": fetch.get("FETCH_SIZE"), "WRITE_SIZE_KB": write.get("WRITE_SIZE"), **tcc, **sq}
if cal.get("FETCH_SIZE"):
    known = 64e6 * 128          # kcfx_membench 4 64: 64 M accesses of one whole 128-byte line each
    out["calibration"] = {"kernel": "gather_coop<8,1> (8 lanes x 16 B per random 128-byte line)", "known_bytes": known,
                          "FETCH_SIZE_KB": cal["FETCH_SIZE"], "bytes_over_fetch_size": known / (cal["FETCH_SIZE"] * 1024)}
if fetch.get("FETCH_SIZE") is not None:
    rd = 2.0 * fetch["FETCH_SIZE"] * 1024
    wr = (write.get("WRITE_SIZE") or 0.0) * 1024
    out["hbm_bytes_per_launch"] = rd + wr
    out["hbm_read_bytes_per_launch"] = rd
    out["hbm_write_bytes_per_launch"] = wr
    cfg = bench["config"]
    if not COHORT:
        json.dump({"tag": tag, "kernel": KERNEL, "kernel_instance": HEAD, "step_form": bench.get("step_form", "two-launch"), "hbm_bytes_per_launch": rd + wr,
                   "rocprof_kernel_avg_ms": out.get("rocprof_kernel_avg_ms"),
                   "hbm_read_requests_per_launch": tcc.get("TCC_EA0_RDREQ_sum"),
                   "ref_len": 30427671, "db_kmers": 150000000, "window": 10000, "table_lines": bench["roofline"].get("table", {}).get("lines"),
                   "note": "2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes), separate --pmc passes, averaged over launches"},
                  open("profiles/traffic.json", "w"), indent=1)
    kms = bench["roofline"]["kernel_avg_ms"]
    out["hbm_GBps_during_kernel"] = (rd + wr) / (kms * 1e-3) / 1e9
    out["algorithmic_bytes_per_launch"] = cfg["kmers_per_step_per_gpu"] * bench["roofline"]["algorithmic_bytes_per_kmer"]
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
shutil.copy(f"{src}/bench.json", f"profiles/{tag}_bench.json")
if os.path.exists(f"{src}/membench.log"):
    shutil.copy(f"{src}/membench.log", f"profiles/{tag}_membench_4GiB.log")
print(json.dumps(out, indent=1)[:3000])
