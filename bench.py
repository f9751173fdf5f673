#!/usr/bin/env python3
"""bench.py -- k-mers screened/s of the getVariations hot path on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: every window of the reference screened against one sample's KMC
database, inputs resident in HBM -- the reference as TEXT, the database as its table: K1 to K3 of SURVEY.md section 8(d). The
default form of the step is the one the product runs for one sample per reference (kcftools-amd getVariations, kcfx_screen_sharded,
kcfx_ref_opts.expected_screens = 1): ONE launch of screen_k31c_kernel<k, both, false>, which plans its tiles itself (2-bit pack,
validity, minimizer choice), + reduce_windows. `--step two-launch` is the same work as ref_plan_kernel + screening kernel on a
planned reference, `--step amortised` leaves the plan out of the step (a job that screens MANY samples against one reference plans
it once); at N = 1 the line carries all three (`other_forms`, `roofline.frac_one_pass` / `frac_two_launch` / `frac_amortised`).
The line's figures are those of the W requested warm-up steps + K timed steps (`--spin-ms` > 0 adds a clock-ramped repeat as
`ramped`, beside them); `cache_state` says what one launch costs after a read-only sweep of the caches and right after kcfx_db_open.

Workload at N=1 (BASELINE.json configs[1]): synthetic A. thaliana chr1-sized reference (30,427,671 bp, seeded),
one synthetic KMC sample (k=31, counter_size 2, ~150 M k-mers = sample-genome k-mers + random noise k-mers),
10 kb tiling windows (3,052 windows).
At N>1 the default is BASELINE.json configs[3]: the whole-genome-sized reference (119,667,750 bp, 10 kb windows) and one
sample per GPU, sample-sharded (`--shard samples`): rank r screens the whole window list against its own sample
database; per-GPU work is fixed ("weak").  `--shard windows` is the S < G case (configs[1]/[2] on several GPUs): ONE
sample, replicated, every rank screens a contiguous window range of equal k-mer count ("strong").  Either way the units are
independent, so there is no collective in the data path; the job ends with ONE RCCL gather of the 40-byte per-window
records to rank 0, inside the timed region.  One process per GPU; launched for N>1 as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`, `cpu_baseline` and `end_to_end` objects.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
K = 31
COUNTER_SIZE = 2
B_ALG = 1 + 8 + COUNTER_SIZE   # algorithmic bytes per k-mer screened (SURVEY.md section 8 d)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def ensure_built():
    """The native pieces are normally there (built in-tree, shipped with the snapshot); a fresh checkout builds them once,
    under a file lock so that the ranks of one torchrun job do not run `make` on top of each other."""
    from kcftools_amd import engine
    synth_bin = os.path.join(ROOT, "kcftools_amd", "bin", "kcfx_synth")
    cli_bin = os.path.join(ROOT, "kcftools_amd", "bin", "kcftools-amd")
    need = (engine.LIB_PATH, synth_bin, cli_bin)
    if not all(os.path.exists(p) for p in need):
        import fcntl
        with open(os.path.join(ROOT, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if not all(os.path.exists(p) for p in need):
                import __graft_entry__ as g
                g.build()
    return synth_bin


def jvm_probe():
    """the Java reference is timed only where a JVM and a kcftools jar exist; say which."""
    import shutil
    java = shutil.which("java")
    if not java:
        return None
    try:
        out = subprocess.run([java, "-version"], capture_output=True, text=True, timeout=20)
        return (out.stderr or out.stdout).strip().splitlines()[0]
    except Exception:
        return "present (version unreadable)"


def pin_probe(jvm):
    """The pin fires the moment it can: with a JVM on the box and KCFTOOLS_JAR=<kcftools jar> in the environment the nine
    golden cases are run through the reference (tools/pin_against_jar.sh) and the outcome lands in the bench line."""
    jar = os.environ.get("KCFTOOLS_JAR")
    if not jvm or not jar:
        return {"status": "unpinned", "why": ("no JVM on this box" if not jvm else "KCFTOOLS_JAR is not set")
                + "; tools/pin_against_jar.sh is the one-command pin"}
    try:
        p = subprocess.run([os.path.join(ROOT, "tools", "pin_against_jar.sh")], capture_output=True, text=True, timeout=900)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("pin:")]
        return {"status": "pinned" if p.returncode == 0 else "FAILED", "jar": jar, "cases_identical_to_oracle":
                sum("reference == oracle" in ln for ln in lines), "cases_byte_identical_kcf": sum("byte-identical" in ln for ln in lines),
                "tail": lines[-3:]}
    except Exception as exc:
        return {"status": "error", "why": str(exc)[:300]}


def java_reference_leg(jvm, fa, prefix, window, min_count, threads, kmers_per_step):
    """the Java reference itself, timed on this box's host cores, where a JVM and a kcftools jar exist."""
    jar = os.environ.get("KCFTOOLS_JAR")
    if not jvm or not jar or not os.path.exists(fa):
        return None
    try:
        out = prefix + ".java.kcf"
        t0 = time.perf_counter()
        p = subprocess.run(["java", "-jar", jar, "getVariations", "-r", fa, "-k", prefix, "-o", out, "-s", "bench", "-f", "window",
                            "-w", str(window), "-c", str(min_count), "-t", str(threads), "-m"], capture_output=True, text=True, timeout=3600)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": (p.stdout + p.stderr)[-300:]}
        return {"value": kmers_per_step / wall, "unit": "k-mers/s", "cores": threads, "kind": "reference", "seconds_wall": wall,
                "sample": "the whole step (every window), java -jar kcftools getVariations -m, database open and KCF output included"}
    except Exception as exc:
        return {"error": str(exc)[:300]}


def make_dataset(synth_bin: str, workdir: str, ref_len: int, db_kmers: int, sample_seed: int, threads: int) -> tuple[str, dict]:
    """Reference text + one sample's KMC database, written by the native generator (cached per parameter set)."""
    os.makedirs(workdir, exist_ok=True)
    prefix = os.path.join(workdir, f"r{ref_len}_d{db_kmers}_s{sample_seed}")
    meta_path = prefix + ".meta.json"
    if os.path.exists(meta_path):
        return prefix, json.load(open(meta_path))
    # the sample genome contributes ~0.88 * ref_len distinct k-mers; the rest is random noise k-mers
    noise = max(0, db_kmers - int(0.885 * ref_len))
    cmd = [synth_bin, "--out", prefix, "--ref-len", str(ref_len), "--seed", "1", "--sample-seed", str(sample_seed),
           "--noise", str(noise), "--k", str(K), "--p", "7", "--cs", str(COUNTER_SIZE), "--bins", "512", "--threads", str(threads)]
    t0 = time.time()
    out = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    meta = json.loads(out.strip().splitlines()[-1])
    meta["fasta"] = False
    meta["generate_seconds"] = time.time() - t0
    meta["command"] = " ".join(cmd[1:])
    json.dump(meta, open(meta_path, "w"))
    return prefix, meta


def launcher_command(gpus: int, argv: list[str], port: int) -> list[str]:
    """`python bench.py --gpus N ...` run plainly: the command of the one-process-per-GPU job it stands for."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(gpus: int, argv: list[str]) -> int:
    """Start the N ranks as a CHILD process (this process has not touched a GPU and never will), relay rank 0's JSON line on
    stdout and everything else on stderr, return the child's exit code."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["KCFX_BENCH_CHILD"] = "1"
    cmd = launcher_command(gpus, argv, port)
    log("bench.py: --gpus", gpus, "without WORLD_SIZE: starting", " ".join(cmd))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    line = None
    for ln in p.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            log(ln)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        log("bench.py: the ranks exited without a bench line")
        rc = 4
    return rc


def usable_cores() -> int:
    """Host cores this process may actually run on: the affinity mask, capped by the cgroup CPU quota (a box can show 256
    logical CPUs and grant 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:                                                    # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:                                                # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(math.ceil(quota))))
    return n


def cache_state_leg(engine, torch, prefix, ref, db, out_ptr, stream, min_count, expected, device, launches=40, opens=5):
    """What ONE screening launch costs depending on what the launches before it left on the chip (kernel time by HIP events on the
    launch itself, so the sweeps between launches are not in it):
      hot_loop     the launch repeated back to back (what every other figure of the line is);
      read_swept   a READ-ONLY pass over a 1 GiB buffer (torch sum) between launches: L2s and Infinity Cache hold other data, nothing dirty;
      write_swept  a 1 GiB fill between launches: the caches are left DIRTY, and the launch also pays their write-back (round 4's
                   probe; kept for comparison, it is not what a clean cold start costs);
      first_pass   a fresh kcfx_db_open followed at once by its first screening launch (the build kernels have just written the
                   table: what a one-pass job sees), `opens` open / close cycles."""
    def timed(pre, n):
        ref.prof_enable(True)
        ref.prof_read(reset=True)
        for _ in range(n):
            if pre is not None:
                pre()
            ref.screen_async(db, out_ptr, stream, min_count)
        torch.cuda.synchronize()
        ms, c = ref.prof_read(reset=True)
        ref.prof_enable(False)
        return ms / max(1, c)
    res = {}
    for _ in range(5):
        ref.screen_async(db, out_ptr, stream, min_count)
    torch.cuda.synchronize()
    res["hot_loop_ms"] = timed(None, launches)
    sweep = torch.ones(1 << 28, dtype=torch.int32, device=f"cuda:{device}")        # 1 GiB
    sink = []
    res["read_swept_ms"] = timed(lambda: sink.append(sweep.sum()), launches)
    res["write_swept_ms"] = timed(lambda: sweep.fill_(1), launches)
    del sweep, sink
    # a fresh handle's first launches: right after kcfx_db_open (its build kernels have just written the table: L2s and Infinity
    # Cache full of dirty lines of it), the launch after that one, and -- each on a handle of its own -- the first launch after a
    # read-only sweep / after 50 ms of idle time between the open and the launch
    def first_of(pre, n_launch=1):
        d2 = engine.KmcDb.open(prefix, device=device, expected_queries=expected)
        if pre is not None:
            pre()
        out_ms = []
        for _ in range(n_launch):
            ref.prof_enable(True)
            ref.prof_read(reset=True)
            ref.screen_async(d2, out_ptr, stream, min_count)
            torch.cuda.synchronize()
            ms, c = ref.prof_read(reset=True)
            ref.prof_enable(False)
            out_ms.append(ms / max(1, c))
        d2.close()
        return out_ms
    sweep = torch.ones(1 << 28, dtype=torch.int32, device=f"cuda:{device}")
    firsts, seconds, after_sweep, after_idle = [], [], [], []
    for _ in range(opens):
        a, b = first_of(None, 2)
        firsts.append(a)
        seconds.append(b)
        after_sweep.append(first_of(lambda: (sweep.sum(), torch.cuda.synchronize()))[0])
        after_idle.append(first_of(lambda: time.sleep(0.05))[0])
    del sweep
    med = lambda v: sorted(v)[len(v) // 2]
    res["first_pass_ms"] = [round(x, 5) for x in firsts]
    res["first_pass_median_ms"] = med(firsts)
    res["second_pass_median_ms"] = med(seconds)
    res["first_pass_after_read_sweep_median_ms"] = med(after_sweep)
    res["first_pass_after_50ms_idle_median_ms"] = med(after_idle)
    res["launches_per_state"] = launches
    return res


def realistic_leg(engine, torch, synth_bin, workdir, ref_len, window, min_count, device, threads, launches=100):
    """Inputs that look like data, beside the uniform-noise database of the headline (every one on the chr1-sized reference, every
    window of every one against the oracle):
      read_error_db       a database as `kmc -ci1` writes it from reads (README.md:171-177): the sample genome's k-mers + ~5 x as many
                          count-1 k-mers in runs around substitution errors, which share minimizers and cells with their originals
                          (genomic counters uniform in 1 .. 30: one in thirty a 1 that an error k-mer of the same cell beats on a coin toss);
      read_error_depth30  the same reads with the counters the error load implies (~30 x: no genomic k-mer seen once);
      k25                 KMC's default k (W = 10 of a line's 16 cells: shorter minimizer runs, more lines per k-mer);
      repeats             reference and sample with genome structure (TE families, satellite arrays, segmental duplications,
                          microsatellites: kcfx_synth --repeats 0.3): crowded minimizers.
    Per input the product's two configurations (round 6): `one_pass` = the one-sample job -- a text-only reference (K1 - K3 in one launch)
    against the table kcfx_db_open_opts builds for ONE pass (expected_queries = the pass: ~0.3 k-mers per cell); `resident` = a handle
    somebody keeps -- a planned reference against the table of expected_queries = 0 (~0.07 per cell). The screening kernel's average
    (HIP events on the launch) as a fraction of the HBM roofline at 1 + 8 + counter_size bytes per k-mer."""
    from oracle import binding as oracle
    out = {}
    noise = str(max(0, 150_000_000 - int(0.885 * ref_len)))
    cases = (("read_error_db", 31, 7, ["--error-events", "4500000", "--noise", "0"]),
             ("read_error_depth30", 31, 7, ["--error-events", "4500000", "--noise", "0", "--depth", "30"]),
             ("k25", 25, 5, ["--noise", noise]),
             ("repeats", 31, 7, ["--repeats", "0.3", "--noise", noise]))
    for name, k, p, extra in cases:
        try:
            prefix = os.path.join(workdir, f"real_{name}_r{ref_len}")
            if not os.path.exists(prefix + ".meta.json"):
                cmd = [synth_bin, "--out", prefix, "--ref-len", str(ref_len), "--seed", "1", "--sample-seed", "2", "--k", str(k), "--p", str(p),
                       "--cs", str(COUNTER_SIZE), "--bins", "512", "--threads", str(threads)] + extra
                meta = json.loads(subprocess.run(cmd, check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1])
                json.dump(meta, open(prefix + ".meta.json", "w"))
            meta = json.load(open(prefix + ".meta.json"))
            text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
            starts, ends = engine.windows(len(text), window, 0, k)
            wb, wl = starts.astype(np.int64), (ends - starts).astype(np.int32)
            stream = torch.cuda.current_stream().cuda_stream
            res = {}
            rec = None
            for form, screens in (("one_pass", 1), ("resident", 0)):
                ref = engine.Reference(text, wb, wl, k, device=device, expected_screens=screens)
                t_open = time.perf_counter()
                db = engine.KmcDb.open(prefix, device=device, expected_queries=int(ref.n_kmer_slots) if form == "one_pass" else 0)
                t_open = time.perf_counter() - t_open
                o = torch.zeros(len(wb) * 10, dtype=torch.int32, device=f"cuda:{device}")
                for _ in range(5):
                    ref.screen_async(db, o.data_ptr(), stream, min_count)
                torch.cuda.synchronize()
                ref.prof_enable(True)
                ref.prof_read(reset=True)
                for _ in range(launches):
                    ref.screen_async(db, o.data_ptr(), stream, min_count)
                torch.cuda.synchronize()
                ms, n = ref.prof_read(reset=True)
                r = o.cpu().numpy().view(engine.RESULT_DTYPE)[:len(wb)].copy()
                nk = int(r["total_kmers"].sum())
                info = db.info
                res[form] = {"kernel_avg_ms": ms / max(1, n), "frac": nk * B_ALG / (ms / max(1, n) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "db_open_seconds": t_open,
                             "table": {"lines": int(info.table_lines), "kmers_per_cell": float(info.stored_kmers) / max(1, 16 * int(info.table_lines)),
                                       "stash_share": float(info.stash_kmers) / max(1, int(info.stored_kmers)),
                                       "outside_first_choice_share": float(info.second_choice_kmers + info.stash_kmers) / max(1, int(info.stored_kmers)),
                                       "extended": int(info.extended)}}
                if rec is None:
                    rec = r
                else:
                    res["forms_identical"] = bool(all(np.array_equal(r[f], rec[f]) for f in engine.INT_FIELDS))
                ref.close()
                db.close()
                del o
            want = oracle.OracleDb(prefix).screen(text, wb, wl, min_count, threads=threads)
            res["parity"] = {"windows_checked": int(len(wb)), "bit_exact": bool(all(np.array_equal(rec[f], want[f]) for f in engine.INT_FIELDS))}
            res["kmers_per_pass"] = int(rec["total_kmers"].sum())
            res["hit_rate"] = float(rec["obs"].sum() / max(1, rec["total_kmers"].sum()))
            res["db_kmers"] = int(meta["db_kmers"])
            res["k"] = k
            out[name] = res
        except Exception as exc:                     # the bench line must not die on an extra leg
            out[name] = {"error": str(exc)[:300]}
    engine.release_cached()
    return out


_EPOCH = re.compile(r"^\[kcftools-amd\] epoch (.*?)\s+([0-9]+\.[0-9]+)\s*$", re.M)
_PHASE = re.compile(r"^\[(kcftools-amd|kcfx \w+)\]\s+(.*?)\s+([0-9.]+) ms\s*$")


def end_to_end_runs(cli, fa, prefix, window, min_count, device, n_runs):
    """`kcftools-amd getVariations` n_runs times, each in a fresh process with its phase trace on (KCFX_CLI_TRACE / KCFX_LOAD_TRACE):
    per run the wall time and where it went -- the driver's own phases (options + .kmc_pre header, FASTA index, windows + text, the
    rest of the GPU runtime's start, the sharded call, scores + KCF) and inside the sharded call the database load's (device + stream,
    buffers, the pass over the file, bids + settle, ...)."""
    env = dict(os.environ)
    env["KCFX_CLI_TRACE"] = "1"
    env["KCFX_LOAD_TRACE"] = "1"
    runs = []
    kcf = prefix + ".bench.kcf"
    for _ in range(n_runs):
        t0 = time.perf_counter()
        e0 = time.time()
        pr = subprocess.run([cli, "getVariations", "-r", fa, "-k", prefix, "-o", kcf, "-s", "bench", "-f", "window",
                             "-w", str(window), "-c", str(min_count), "--device", str(device)], capture_output=True, text=True, env=env)
        wall = time.perf_counter() - t0
        e1 = time.time()
        if pr.returncode != 0:
            raise RuntimeError(pr.stdout[-500:] + pr.stderr[-500:])
        phases = {}
        for ln in pr.stderr.splitlines():
            m = _PHASE.match(ln)
            if m:
                key = ("cli: " if m.group(1) == "kcftools-amd" else "load: ") + m.group(2).strip()
                phases[key] = round(phases.get(key, 0.0) + float(m.group(3)), 1)
        accounted = sum(v for k_, v in phases.items() if k_.startswith("cli: "))
        phases["process start + exit (wall - the driver's phases)"] = round(wall * 1e3 - accounted, 1)
        # ... split by the driver's own wall-clock stamps: spawn -> main() (exec, the dynamic linker, the ROCm libraries' initialisers)
        # and "leaving" -> reaped (the kernel taking the process's device memory and queues down)
        stamps = dict(_EPOCH.findall(pr.stderr))
        if "main entered" in stamps and "leaving" in stamps:
            phases["process start (spawn -> main)"] = round((float(stamps["main entered"]) - e0) * 1e3, 1)
            phases["process exit (leaving -> reaped)"] = round((e1 - float(stamps["leaving"])) * 1e3, 1)
        split = [ln for ln in pr.stdout.splitlines() if "Screened" in ln]
        runs.append({"seconds_wall": round(wall, 4), "phases_ms": phases, "breakdown": split[-1].split(" - ")[-1] if split else ""})
    rows = sum(1 for ln in open(kcf) if not ln.startswith("#"))
    return runs, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500, help="default sized for a timed region of ~0.1 s")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ref-len", type=int, default=0, help="0 = 30,427,671 (configs[1]) on one GPU, 119,667,750 (configs[3]) on several")
    ap.add_argument("--shard", choices=("samples", "windows", "cohort", "table"), default="samples",
                    help="N > 1: one sample per GPU against the whole window list (weak scaling, configs[3]) or one sample "
                         "replicated and the windows split (strong scaling, the S < G case); cohort = --samples S (the JOB's samples) "
                         "in cohort tables of eight, tables x window ranges over the GPUs (kcfx_shard_plan, KCFX_SHARD_COHORT: "
                         "8 samples on 8 GPUs = every GPU the same table and an eighth of the windows; 32 = four tables, two GPUs each); "
                         "table = ONE sample whose TABLE is cut into one slice per GPU by minimizer (kcfx_db_open_slice: the database a single "
                         "GPU could only hold densely), every GPU screens the whole window list against its slice and the tiles' hit bitmaps "
                         "meet on rank 0 in EVERY step -- the one data-path exchange of the engine")
    ap.add_argument("--db-kmers", type=int, default=150_000_000)
    ap.add_argument("--window", type=int, default=10_000)
    ap.add_argument("--workdir", default=os.environ.get("KCFX_BENCH_DIR", "/tmp/kcfx_bench"))
    ap.add_argument("--cpu-baseline-seconds", type=float, default=5.0,
                    help="wall seconds per thread count of the CPU baseline leg (T = 1, 16, 64, all; 0 = skip)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the FASTA -> KCF command-line run")
    ap.add_argument("--no-realistic", action="store_true",
                    help="skip the inputs that look like data (read-error database, k = 25, repeat-structured genome: ~1 min of generation, "
                         "opens and oracle checks)")
    ap.add_argument("--no-cache-state", action="store_true", help="skip the hot / swept / first-pass launches of the cache_state leg")
    ap.add_argument("--no-resident", action="store_true", help="skip the step's kernels against the table of a resident handle (expected_queries = 0)")
    ap.add_argument("--min-count", type=int, default=1)
    ap.add_argument("--spin-ms", type=float, default=60.0,
                    help="the line's own figures (value, ms_per_step, roofline) are ALWAYS those of the W requested warm-up steps + K timed "
                         "steps. After them, when this is > 0: a clock ramp -- untimed repetitions of the step for this long -- and the K steps "
                         "once more, reported beside as \"ramped\" (its warm-up count stated there), never as the headline; 0 = skip")
    ap.add_argument("--step", choices=("one-pass", "two-launch", "amortised"), default="one-pass",
                    help="one-pass (default): K1 - K3 of ONE sample against a reference the way the product runs it -- the reference "
                         "resident as TEXT (kcfx_ref_opts.expected_screens = 1: what the command-line driver and kcfx_screen_sharded "
                         "create), the screening launch plans its tiles itself: screen_k31c_kernel<K, both, false> + reduce_windows. "
                         "two-launch: the same work as ref_plan_kernel + screening kernel + reduce on a planned reference. amortised: "
                         "the plan stays out of the step (many samples per reference). With --samples > 1 the reference is planned "
                         "and one-pass means two-launch")
    ap.add_argument("--table", choices=("product", "sparse"), default="product",
                    help="product: the table kcfx_db_open_opts builds for ONE pass (expected_queries = the step's k-mers: what the "
                         "command-line driver and kcfx_screen_sharded pass; ~0.3 k-mers per cell since round 6); sparse: kcfx_db_open's "
                         "table for an unknown / large number of look-ups (~0.07 per cell; measured beside the step either way: resident_table)")
    ap.add_argument("--samples", type=int, default=1,
                    help="samples per GPU; > 1 screens them in one pass over the reference (kcfx_screen_ref_multi, SURVEY f-4). "
                         "The default 1 is BASELINE.json's configuration")
    ap.add_argument("--cohort", action="store_true",
                    help="with --samples S (2 .. 8): the S samples in ONE cohort table (kcfx_cohort_open), one look-up per reference "
                         "k-mer for all of them; algorithmic bytes per sample-k-mer = (1 + 8) / S + counter_size")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 control flow on a one-GPU box (not a measurement): every rank on device 0, gloo instead of RCCL
    rehearsal = os.environ.get("KCFX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not os.environ.get("KCFX_BENCH_CHILD"):
            sys.exit(launch_ranks(args.gpus, sys.argv[1:]))  # (before torch is imported: nothing here has touched a GPU)
        args.gpus = world

    if args.ref_len <= 0:
        args.ref_len = 30_427_671 if world == 1 else 119_667_750
    cohort_sharded = world > 1 and args.shard == "cohort"
    if cohort_sharded:
        args.cohort = True
    window_sharded = world > 1 and args.shard in ("windows", "cohort")      # (the samples are the same on every rank, the windows are cut)
    table_sharded = world > 1 and args.shard == "table"                      # (ONE sample, the same on every rank; its table is cut)
    same_data = window_sharded or table_sharded
    synth_bin = ensure_built()
    import torch
    import torch.distributed as dist
    from kcftools_amd import engine

    if not torch.cuda.is_available():
        log("bench.py: no GPU visible; the engine has no CPU fallback")
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- data: same reference on every rank, one sample database per rank -------------------------------
    ncpu = usable_cores()
    ncpu_logical = os.cpu_count() or ncpu
    t0 = time.time()
    S = max(1, args.samples)
    prefixes = []
    for i in range(S):
        # window-sharded: every rank holds the SAME sample (replicated database); sample-sharded: its own
        seed = 2 + (0 if same_data else rank * S) + i
        if same_data and rank != 0:
            if world > 1 and not os.path.exists(os.path.join(args.workdir, f"r{args.ref_len}_d{args.db_kmers}_s{seed}.meta.json")):
                time.sleep(1.0)
            while not os.path.exists(os.path.join(args.workdir, f"r{args.ref_len}_d{args.db_kmers}_s{seed}.meta.json")):
                time.sleep(0.5)                # rank 0 writes it
        pfx, meta = make_dataset(synth_bin, args.workdir, args.ref_len, args.db_kmers, seed,
                                 ncpu if same_data else max(1, ncpu // world))
        prefixes.append(pfx)
    prefix = prefixes[0]
    text = np.fromfile(prefix + ".ref.txt", dtype=np.uint8).tobytes()
    gen_s = time.time() - t0
    starts, ends = engine.windows(len(text), args.window, 0, K)
    win_begin = starts.astype(np.int64)
    win_len = (ends - starts).astype(np.int32)
    n_win_total = len(win_begin)
    if cohort_sharded:
        # this rank's unit of the cohort plan: a group of up to eight of the job's samples and a window range
        from kcftools_amd import shard
        segs = shard.cohort_segments_of_rank(win_len, K, S, world, rank)
        if len(segs) != 1:
            log(f"bench.py: --shard cohort with {S} samples on {world} GPUs gives rank {rank} {len(segs)} units; use a GPU count that is "
                f"a multiple of the {-(-S // 8)} table(s)")
            sys.exit(2)
        first, n_grp, lo, hi = segs[0]
        prefixes = prefixes[first:first + n_grp]
        prefix = prefixes[0]
        S = n_grp
        win_begin, win_len = win_begin[lo:hi], win_len[lo:hi]
    elif window_sharded:
        from kcftools_amd import shard
        lo, hi = shard.window_range_of_rank(win_len, K, world, rank)
        win_begin, win_len = win_begin[lo:hi], win_len[lo:hi]

    # the form of the step (see --step). One sample: the product's one-launch form on a text-only reference unless asked otherwise.
    single = (max(1, args.samples) == 1 and not args.cohort) or (cohort_sharded and S == 1)
    form = args.step
    if form == "one-pass" and not single:
        form = "two-launch"
    t0 = time.time()
    # upload + tile list (+ for a planned reference, k = 19 .. 31: codes and anchor bytes by ref_plan_kernel)
    ref = engine.Reference(text, win_begin, win_len, K, device=local_rank, expected_screens=1 if form == "one-pass" else 0)
    ref_s = time.time() - t0
    n_win = len(win_begin)
    # The table a handle gets depends on the look-ups it is opened for (kcfx_db_open_opts.expected_queries): the command-line
    # driver and kcfx_screen_sharded pass one pass over their windows, and that is what is measured here by default.
    expected = int(ref.n_kmer_slots) if args.table == "product" else 0
    t0 = time.time()
    cohort = None
    if args.cohort and S > 1:                      # (a group of ONE sample -- e.g. 9 samples on 2 ranks -- takes the single-database path, as kcfx_screen_sharded does)
        cohort = engine.Cohort(prefixes, device=local_rank, expected_queries=expected)
        dbs = [engine.KmcDb.open(prefixes[0], device=local_rank, expected_queries=expected)]   # (the first sample on its own: the parity leg's single-sample twin)
    elif table_sharded:
        dbs = [engine.KmcDb.open(prefixes[0], device=local_rank, expected_queries=expected, n_slices=world, slice=rank)]
    else:
        dbs = [engine.KmcDb.open(pfx, device=local_rank, expected_queries=expected) for pfx in prefixes]
    db = dbs[0]
    load_s = (time.time() - t0) / S
    if form != "one-pass" and not ref.planned:          # (k outside 19 .. 31, KCFX_REF_PLAN=0: there is no plan to make)
        form = "amortised"

    # every rank gathers the same number of records (a window-sharded rank's range is padded to the longest)
    n_rec = S * n_win
    if world > 1:
        m = torch.tensor([n_rec], dtype=torch.int64, device="cpu" if rehearsal else f"cuda:{local_rank}")
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        n_rec = int(m.item())
    out = torch.zeros(n_rec * 10, dtype=torch.int32, device=f"cuda:{local_rank}")
    cdev = "cpu" if rehearsal else f"cuda:{local_rank}"     # where the tensors of the collectives live
    gathered = [torch.zeros_like(out, device=cdev) for _ in range(world)] if (world > 1 and rank == 0) else None
    stream = torch.cuda.current_stream()

    if table_sharded:
        from kcftools_amd import shard
        nb, ns = ref.partial_bytes()
        part_bits = torch.zeros(nb // 8, dtype=torch.int64, device=f"cuda:{local_rank}")
        part_sums = torch.zeros(ns // 8, dtype=torch.int64, device=f"cuda:{local_rank}")

    def screen(r, o):
        if table_sharded:
            # this rank's slice against the whole window list; the tiles' bitmaps and count sums to rank 0 (RCCL over xGMI: 520 B
            # per tile and rank), OR-ed / added there, reduced once
            r.screen_partial_async(db, stream.cuda_stream, args.min_count)
            r.export_partial_async(part_bits.data_ptr(), part_sums.data_ptr(), stream.cuda_stream)
            gb, gs = shard.gather_table_partials(part_bits.to(cdev), part_sums.to(cdev), dst=0)
            if rank == 0:
                for b, c in zip(gb[1:], gs[1:]):
                    b, c = b.to(f"cuda:{local_rank}"), c.to(f"cuda:{local_rank}")
                    r.merge_partial_async(b.data_ptr(), c.data_ptr(), stream.cuda_stream)
                r.finish_async(o.data_ptr(), stream.cuda_stream)
            return
        if cohort is not None:
            cohort.screen_async(r, o.data_ptr(), stream.cuda_stream, args.min_count)
        elif S == 1:
            r.screen_async(db, o.data_ptr(), stream.cuda_stream, args.min_count)
        else:
            r.screen_multi_async(dbs, o.data_ptr(), stream.cuda_stream, args.min_count)

    def make_step(r, o, frm):
        """one step of form `frm` on reference r: one-pass = the one launch that plans its tiles itself (K1 - K3) + reduce;
        two-launch = ref_plan_kernel, then screening kernel + reduce; amortised = screening kernel + reduce"""
        def run():
            if frm == "two-launch":
                r.plan_async(stream.cuda_stream)     # K1 + the minimizer choice: ref_plan_kernel over the resident text
            screen(r, o)                             # (K1 +) K2 + K3: screen_k31c / screen_cohort + reduce_windows
        return run

    step = make_step(ref, out, form)

    def final_gather():
        if world > 1 and not table_sharded:                 # (table slices: rank 0 already holds the job's records)
            dist.gather(out.to(cdev), gathered, dst=0)      # RCCL over xGMI: 40 B per (window, sample), once per job

    def timed(run, r, n_steps, with_gather):
        """the contract's timed region: barrier + synchronize on both sides of exactly n_steps steps; kernel times by HIP events"""
        r.prof_enable(True)
        r.prof_read(reset=True)
        if r.planned:
            r.prof_read_plan(reset=True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        for _ in range(n_steps):
            run()
        if with_gather:
            final_gather()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t_e = time.perf_counter() - t_a
        k_ms, k_n = r.prof_read(reset=True)
        p_ms, p_n = r.prof_read_plan(reset=True) if r.planned else (0.0, 0)
        r.prof_enable(False)
        return t_e, k_ms, k_n, p_ms, p_n

    for _ in range(args.warmup):
        step()
    final_gather()                                 # (also warms the communicator up)
    torch.cuda.synchronize()
    res_all = out.cpu().numpy().view(engine.RESULT_DTYPE)[:S * n_win].reshape(S, n_win).copy()
    res = res_all[0]                               # the sample checked against the oracle below
    kmers_per_step = int(res_all["total_kmers"].sum())

    # ---- timed region: the W requested warm-up steps above, then exactly K steps ----------------------------------
    elapsed, kernel_ms, launches, plan_ms_sum, plan_launches = timed(step, ref, args.steps, True)

    # the other forms of the step, beside it (same process, no gather), each on its own reference handle and output buffer
    others = {}
    forms_same = None
    if world == 1 and S == 1 and cohort is None and K >= 19:
        ref_o = engine.Reference(text, win_begin, win_len, K, device=local_rank, expected_screens=0 if form == "one-pass" else 1)
        out_o = torch.zeros_like(out)
        for frm in ("one-pass", "two-launch", "amortised"):
            if frm == form:
                continue
            r, o = (ref_o, out_o) if (frm == "one-pass") != (form == "one-pass") else (ref, out)
            if (frm == "one-pass") == r.planned:           # (KCFX_REF_PLAN in the environment overrides the handle's form: skip)
                continue
            run = make_step(r, o, frm)
            for _ in range(max(3, args.warmup)):
                run()
            o_el, o_kms, o_kn, o_pms, o_pn = timed(run, r, args.steps, False)
            others[frm] = {"ms_per_step": o_el / args.steps * 1e3, "kernel_avg_ms": o_kms / max(1, o_kn),
                           "ref_plan_ms": (o_pms / o_pn) if o_pn else None, "value": kmers_per_step * args.steps / o_el}
        if any(((f == "one-pass") != (form == "one-pass")) for f in others):     # (the other handle ran)
            forms_same = bool(torch.equal(out_o, out))
        if forms_same is False:
            log("bench.py: the text-only reference gives other results than the planned one")

    # The same two kernels against the table a RESIDENT handle gets (expected_queries = 0: ~0.07 k-mers per cell, 17 GB for this
    # database -- kcfx_db_open's table, what an engine that keeps its handles and serves thousands of passes builds), beside the
    # step's own table (sized for ONE pass since round 6: ~0.3 per cell, 4.4 GB -- what the one-sample job can afford to allocate)
    resident = None
    if world == 1 and S == 1 and cohort is None and K >= 19 and args.table == "product" and not args.no_resident:
        try:
            t_r = time.time()
            db_res = engine.KmcDb.open(prefix, device=local_rank, expected_queries=0)
            t_r = time.time() - t_r
            resident = {"table": {"lines": int(db_res.info.table_lines), "bytes": int(db_res.info.table_bytes),
                                  "kmers_per_cell": float(db_res.info.stored_kmers) / max(1, 16 * int(db_res.info.table_lines)),
                                  "stash_kmers": int(db_res.info.stash_kmers)}, "db_open_seconds": t_r}
            out_r = torch.zeros_like(out)
            for frm, r in (("one-pass", ref if form == "one-pass" else ref_o), ("amortised", ref_o if form == "one-pass" else ref)):
                if (frm == "one-pass") == r.planned:
                    continue
                def run_r(r=r):
                    r.screen_async(db_res, out_r.data_ptr(), stream.cuda_stream, args.min_count)
                for _ in range(max(3, args.warmup)):
                    run_r()
                r_el, r_kms, r_kn, _, _ = timed(run_r, r, args.steps, False)
                resident[frm] = {"ms_per_step": r_el / args.steps * 1e3, "kernel_avg_ms": r_kms / max(1, r_kn),
                                 "value": kmers_per_step * args.steps / r_el,
                                 "frac": kmers_per_step * B_ALG / (r_kms / max(1, r_kn) * 1e-3) / 1e9 / HBM_PEAK_GBS}
            resident["identical_to_the_step"] = bool(torch.equal(out_r, out))
            if not resident["identical_to_the_step"]:
                forms_same = False
            del out_r
            db_res.close()                           # (its 17 GB stay with the library's pool: nothing goes back to the driver here)
        except Exception as exc:                     # the bench line must not die on the extra leg
            resident = {"error": str(exc)[:300]}

    # Optional clock ramp (--spin-ms > 0): the step repeated untimed for that long, then the K steps once more -- reported
    # beside the line's own figures, never instead of them.
    ramped = None
    if args.spin_ms > 0:
        spin_steps = 0
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < args.spin_ms:
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            spin_steps += 50
        r_el, r_kms, r_kn, r_pms, r_pn = timed(step, ref, args.steps, False)
        ramped = {"ms_per_step": r_el / args.steps * 1e3, "kernel_avg_ms": r_kms / max(1, r_kn),
                  "ref_plan_ms": (r_pms / r_pn) if r_pn else None, "value": kmers_per_step * args.steps / r_el,
                  "untimed_steps_before": args.warmup + args.steps * (1 + len(others)) + spin_steps, "spin_ms": args.spin_ms}

    # what ONE launch costs by what the launches before it left on the chip (hot loop / caches swept / first launch after an open)
    cache_state = None
    if world == 1 and S == 1 and cohort is None and not args.no_cache_state:
        try:
            cache_state = cache_state_leg(engine, torch, prefix, ref, db, out.data_ptr(), stream.cuda_stream, args.min_count, expected,
                                          local_rank)
        except Exception as exc:                     # the bench line must not die on the extra leg
            cache_state = {"error": str(exc)[:300]}

    el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    tot = torch.tensor([kmers_per_step], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    kmers_all = int(tot.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = kmers_all * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    kern_avg_ms = kernel_ms / max(1, launches)
    launches_per_step = max(1.0, launches / args.steps)                # multi-sample: one launch per <= 8 samples
    # one ASCII base per reference position is read once for all S samples (SURVEY f-4: B_alg = 1/S + 8 + counter_size); in a
    # cohort table the key is compared once for all of them too: (1 + 8) / S + counter_size
    b_alg = ((1.0 + 8.0) / S + COUNTER_SIZE) if cohort is not None else (1.0 / S + 8 + COUNTER_SIZE)
    achieved_gbs = (kmers_per_step * b_alg / launches_per_step) / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0
    # every form of the step that was measured: its screening kernel's average, its plan kernel's (two-launch), and the same
    # algorithmic bytes over what the form launches for K1 - K3
    plan_avg_ms = (plan_ms_sum / plan_launches) if plan_launches else None
    forms = {form: {"ms_per_step": ms_per_step, "kernel_avg_ms": kern_avg_ms, "ref_plan_ms": plan_avg_ms, "value": value}}
    forms.update(others)
    if plan_avg_ms is None:
        plan_avg_ms = (forms.get("two-launch") or {}).get("ref_plan_ms")
    step_bytes = kmers_per_step * b_alg / launches_per_step

    def frac_of(frm):
        f = forms.get(frm)
        if not f or not f["kernel_avg_ms"]:
            return None
        ms = f["kernel_avg_ms"] + ((f["ref_plan_ms"] or 0.0) if frm == "two-launch" else 0.0)
        return step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS

    kernel_name = ("screen_k31c_kernel" if db.info.table_format == 1 else "screen_k31_kernel") if (S == 1 or db.info.table_format == 1) \
        else "screen_k31_multi_kernel"
    # (the instantiation, as rocprofv3 --kernel-trace names it: <k, both strands, planned reference>)
    kernel_inst = (f"screen_k31c_kernel<{K}, {'true' if db.info.both_strands else 'false'}, {'false' if form == 'one-pass' else 'true'}>"
                   if kernel_name == "screen_k31c_kernel" else kernel_name)
    if cohort is not None:
        kernel_name = "screen_cohort_kernel"
    if os.environ.get("KCFX_GENERIC_KERNEL"):           # A/B hook: the general kernel on a line table
        kernel_name = "screen_tiles_kernel"
    # HBM traffic comes from PMC counters, which need their own rocprofv3 passes (tools/profile_round.sh): the figure below
    # is the committed one for the same kernel and workload, NOT measured in this run -- traffic_source says so, and it is
    # dropped when the committed profile is of another kernel or workload.
    traffic = None
    traffic_source = None
    rocprof_ms = None
    requests = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if (S == 1 and world == 1 and tj.get("kernel") == kernel_name and tj.get("ref_len") == args.ref_len
                    and tj.get("db_kmers") == args.db_kmers and tj.get("window") == args.window
                    and tj.get("table_lines") in (None, int(db.info.table_lines)) and tj.get("step_form", "two-launch") == form):
                traffic = tj.get("hbm_bytes_per_launch")
                requests = tj.get("hbm_read_requests_per_launch")
                rocprof_ms = tj.get("rocprof_kernel_avg_ms")    # the committed rocprofv3 --stats average of the same kernel
                traffic_source = (f"profiles/traffic.json (committed rocprofv3 --pmc passes, tag {tj.get('tag')}, kernel "
                                  f"{tj.get('kernel')}; not measured in this run)")
        except Exception:
            traffic = None

    # ---- CPU baseline: the oracle (C restatement of the reference path, probing the on-disk KMC layout) -----------
    cpu = None
    parity = None
    if table_sharded:
        # the merged records of the slices against ONE table of the whole database on this rank (it fits here; the point of the
        # mode is the database that does not)
        whole = engine.KmcDb.open(prefix, device=local_rank, expected_queries=expected)
        ref_w = engine.Reference(text, win_begin, win_len, K, device=local_rank, expected_screens=1)
        want = ref_w.screen(whole, args.min_count)
        parity = {"windows_checked": int(n_win), "against": "the unsharded table on rank 0 (kcfx_screen_ref)",
                  "bit_exact": bool(all(np.array_equal(res[f], want[f]) for f in engine.INT_FIELDS))}
        ref_w.close()
        whole.close()
    if world == 1 and args.cpu_baseline_seconds > 0:
        from oracle import binding as oracle
        odb = oracle.OracleDb(prefix)
        kmers_per_window = max(1.0, kmers_per_step / S / n_win)
        # calibrate on a few windows, single thread
        cal = min(n_win, 8)
        t0 = time.perf_counter()
        w1 = odb.screen(text, win_begin[:cal], win_len[:cal], args.min_count, threads=1)
        rate1 = int(w1["total_kmers"].sum()) / (time.perf_counter() - t0)     # (warms the page cache of the database files too)
        by_threads = {}
        best = None
        ok = True
        checked = 0
        for T in sorted({1, min(16, ncpu), min(64, ncpu), ncpu}):
            # calibrate this thread count on a short sample, then run ~cpu_baseline_seconds of wall
            n_c = int(min(n_win, max(4 * T, 64)))
            idx = np.linspace(0, n_win - 1, n_c).astype(np.int64)
            t0 = time.perf_counter()
            wc = odb.screen(text, win_begin[idx], win_len[idx], args.min_count, threads=T)
            rate_c = int(wc["total_kmers"].sum()) / (time.perf_counter() - t0)
            n_s = int(max(T, rate_c * args.cpu_baseline_seconds / kmers_per_window))
            # evenly spaced windows; more than one pass over the window list when the list is shorter than the budget
            idx = (np.linspace(0, n_win - 1, n_s).astype(np.int64) if n_s <= n_win
                   else np.resize(np.arange(n_win, dtype=np.int64), n_s))
            t0 = time.perf_counter()
            wN = odb.screen(text, win_begin[idx], win_len[idx], args.min_count, threads=T)
            tN = time.perf_counter() - t0
            cpu_kmers = int(wN["total_kmers"].sum())
            rate = cpu_kmers / tN
            by_threads[str(T)] = {"value": rate, "seconds": tN, "windows": n_s, "kmers": cpu_kmers}
            # the same windows from the GPU run must be bit-identical
            ok = ok and all(np.array_equal(res[f][idx], wN[f]) for f in engine.INT_FIELDS)
            checked = max(checked, min(n_s, n_win))
            if best is None or rate > best[1]:
                best = (T, rate, n_s, cpu_kmers, tN)
        cpu = {"value": best[1], "unit": "k-mers/s", "cores": best[0], "kind": "port",
               "sample": f"{best[2]} windows of the step's {n_win} ({best[3]} k-mers, {best[4]:.1f} s wall; evenly spaced, or the whole "
                         f"list repeated when it is shorter than the time budget), same database files; "
                         f"oracle/kcf_oracle.c (C restatement of the Java path, KMC on-disk layout, thread pool over windows "
                         f"like GetVariants.java:129-151); the best of the thread counts in by_threads",
               "host_cores": ncpu, "host_cores_logical": ncpu_logical,
               "host_cores_source": "sched_getaffinity capped by the cgroup CPU quota", "by_threads": by_threads}
        parity = {"windows_checked": int(checked), "bit_exact": bool(ok)}
        if not ok:
            log("bench.py: GPU results differ from the oracle on the baseline sample")

    # ---- end to end: the command-line driver, FASTA read -> KCF written, same reference / sample / windows ----------------
    # (four runs here -- the first writes the .fai and warms the page cache -- with this process's handles alive, and two more at the
    # very end, right after this process has given ~20 GB back to the driver: `after_churn`, what a job sees on a node where another
    # job has just finished. Every run with its phases.)
    e2e = None
    cli = os.path.join(ROOT, "kcftools_amd", "bin", "kcftools-amd")
    fa = prefix + ".fa"
    if world == 1 and S == 1 and not args.no_end_to_end:
        try:
            from kcftools_amd import synth
            if not os.path.exists(fa):
                synth.write_fasta(fa, [("chr1", text)])
            runs, rows = end_to_end_runs(cli, fa, prefix, args.window, args.min_count, local_rank, 4)
            warm = sorted(runs[1:], key=lambda r_: r_["seconds_wall"])
            med = warm[len(warm) // 2]
            e2e = {"seconds_wall": med["seconds_wall"], "seconds_wall_first_run": runs[0]["seconds_wall"],
                   "seconds_wall_all_runs": [r_["seconds_wall"] for r_ in runs],
                   "kmers_per_s": kmers_per_step / med["seconds_wall"],
                   "windows": rows, "what": "kcftools-amd getVariations: process start, FASTA index + sequence read, database load, "
                   "reference upload, screening, scores, KCF written (median of three runs after a first one that writes the "
                   ".fai and warms the page cache); runs[i].phases_ms: where each run's wall time went",
                   "breakdown": med["breakdown"], "runs": runs}
        except Exception as exc:                     # the bench line must not die on the extra leg
            e2e = {"error": str(exc)[:300]}

    # ---- the same end to end on the host cores: the C port opens the database files, screens every window, writes the KCF ----
    cpu_e2e = None
    if world == 1 and S == 1 and not args.no_end_to_end and cpu is not None:
        try:
            from oracle import binding as oracle
            from oracle import pyhost
            T = int(cpu["cores"])
            t0 = time.perf_counter()
            odb2 = oracle.OracleDb(prefix)
            t_open = time.perf_counter() - t0
            wN = odb2.screen(text, win_begin, win_len, args.min_count, threads=T)
            t_screen = time.perf_counter() - t0 - t_open
            sc, md = oracle.score_array(wN)
            with open(prefix + ".cpu.kcf", "w") as f:
                for i in range(n_win):
                    f.write(pyhost.kcf_row("chr1", int(win_begin[i]), int(win_begin[i] + win_len[i]), f"chr1_{int(win_begin[i])}",
                                           {fld: int(wN[fld][i]) for fld in engine.INT_FIELDS}, float(sc[i]), float(md[i])) + "\n")
            wall = time.perf_counter() - t0
            cpu_e2e = {"seconds_wall": wall, "kmers_per_s": kmers_per_step / wall, "cores": T, "kind": "port",
                       "breakdown": f"database open {t_open:.3f} s, screening {t_screen:.3f} s ({T} threads), scores + KCF rows {wall - t_open - t_screen:.3f} s",
                       "what": "oracle/kcf_oracle.c (C restatement of the Java path) + the KCF rows of oracle/pyhost.py, same files and windows as "
                               "end_to_end; the FASTA is already in memory (not read again)"}
        except Exception as exc:
            cpu_e2e = {"error": str(exc)[:300]}

    # ---- this box's own random-request rate (the wall the kernel runs against differs by +- 8 % between boxes) ----------
    box = None
    if world == 1 and S == 1 and not args.no_end_to_end:
        try:
            mb = os.path.join(ROOT, "kcftools_amd", "bin", "kcfx_membench")
            pr = subprocess.run([mb, "16", "128", "box"], capture_output=True, text=True, timeout=120)
            box = json.loads(pr.stdout.strip().splitlines()[-1])
            if requests and kern_avg_ms > 0:
                box["kernel_G_requests_per_s"] = requests / (kern_avg_ms * 1e-3) / 1e9
                # (against the better of the two gather forms: a pure gather gains nothing from the non-temporal hint -- it has no
                # other data in the caches to protect -- which is why the kernel, which has, can come out above it)
                box["kernel_fraction_of_the_request_rate"] = box["kernel_G_requests_per_s"] / max(
                    box["G_lines_per_s"], box.get("G_lines_per_s_one_nontemporal_load", 0.0))
                box["requests_per_launch_source"] = "profiles/traffic.json (TCC_EA0_RDREQ_sum of the committed --pmc pass)"
        except Exception as exc:
            box = {"error": str(exc)[:200]}

    # ---- inputs that look like data (after the handles of the headline are gone: each case opens a 17 GB table of its own) ----
    realistic = None
    if world == 1 and S == 1 and cohort is None and not args.no_realistic:
        for d_ in dbs:
            d_.close()
        ref.close()
        engine.release_cached()
        realistic = realistic_leg(engine, torch, synth_bin, args.workdir, args.ref_len, args.window, args.min_count, local_rank, ncpu)
    if e2e is not None and "error" not in e2e:
        try:
            for d_ in dbs:
                d_.close()
            ref.close()
            if "ref_o" in locals():
                ref_o.close()
            if cohort is not None:
                cohort.close()
            engine.release_cached()                  # everything this process held goes back to the driver NOW (tables, pooled blocks)
            runs2, _ = end_to_end_runs(cli, fa, prefix, args.window, args.min_count, local_rank, 2)
            e2e["after_churn"] = {"seconds_wall_all_runs": [r_["seconds_wall"] for r_ in runs2], "runs": runs2,
                                  "what": "the same command right after this process released its device memory to the driver"}
        except Exception as exc:
            e2e["after_churn"] = {"error": str(exc)[:300]}

    jvm = jvm_probe()
    java_leg = None
    if world == 1 and S == 1:
        java_leg = java_reference_leg(jvm, prefix + ".fa", prefix, args.window, args.min_count, ncpu, kmers_per_step)

    line = {
        "metric": "k-mers screened/sec, getVariations k=31",
        "value": value,
        "unit": "k-mers/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "step": {"one-pass": "one pass of one sample as the product runs it: the reference resident as text, ONE launch that plans its "
                             "tiles itself (K1 - K3: screen_k31c_kernel<k, both, false>) + reduce_windows",
                 "two-launch": "one pass as two launches on a planned reference: ref_plan_kernel + screening kernel + reduce_windows (K1 - K3)",
                 "amortised": "screening kernel + reduce_windows (the reference's plan made once, outside the step: many samples per reference)"}[form],
        "step_form": form,
        "other_forms": others or None,
        "forms_identical": forms_same,
        "ramped": ramped,
        "cache_state": cache_state,
        "resident_table": resident,
        "realistic": realistic,
        "higher_is_better": True,
        "scaling": "strong" if (window_sharded or table_sharded) else "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": (f"getVariations: synthetic {'chr1' if args.ref_len < 100_000_000 else 'whole-genome'}-sized reference "
                                f"{args.ref_len} bp (BASELINE configs[{1 if args.ref_len < 100_000_000 else 3}] shape), "
                                + (f"ONE KMC sample, its table in {world} slices on {world} GPUs" if table_sharded else
                                   (f"ONE KMC sample replicated on {world} GPUs" if S == 1 else
                                    f"{S} KMC samples ({'one cohort table' if cohort is not None else 'one table each'}) replicated on {world} GPUs")
                                   if window_sharded else
                                   f"{S} KMC sample{'s' if S > 1 else ''} per GPU ({S * world} in all)")
                                + f" (k=31, counter_size {COUNTER_SIZE}, {meta['db_kmers']} k-mers each), {args.window // 1000} kb tiling "
                                f"windows ({n_win_total} windows), -c {args.min_count}"),
                   "windows": n_win_total, "samples_per_gpu": S, "kmers_per_step_per_gpu": kmers_per_step, "db_kmers": meta["db_kmers"],
                   "parallelism": ((f"table-sharded x{world}: the sample's table in {world} slices by minimizer, every GPU screens every window against "
                                    f"its slice, the tiles' hit bitmaps gathered to rank 0 and OR-ed in every step" if table_sharded else
                                    f"window-sharded x{world} (contiguous ranges of equal k-mer count, database replicated)" if window_sharded
                                    else f"sample-sharded x{world} (reference replicated)")
                                   + (", one RCCL gather of 520 B per tile and rank per step" if table_sharded else
                                      ", no data-path collective, one final RCCL gather of 40 B records")
                                   + (" -- REHEARSAL: all ranks on one GPU over gloo, not a measurement" if rehearsal else ""))
                   if world > 1 else "single GPU",
                   "table_bytes": int(cohort.info.table_bytes if cohort is not None else db.info.table_bytes),
                   "cohort_table": (None if cohort is None else {"samples": S, "records": int(cohort.info.total_records),
                                                                   "lines": int(cohort.info.table_lines),
                                                                   "stash_records": int(cohort.info.stash_records),
                                                                   "open_seconds": float(cohort.info.load_seconds)}),
                   "hit_rate": float(res["obs"].sum() / max(1, res["total_kmers"].sum()))},
        "windows_per_s": (n_win_total * max(1, args.samples) if (window_sharded or table_sharded) else n_win * S * world) * args.steps / elapsed,
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": kernel_name, "kernel_instance": kernel_inst, "kernel_avg_ms": kern_avg_ms,
                     "launches_timed": int(launches), "algorithmic_bytes_per_kmer": b_alg if S > 1 else B_ALG,
                     "rocprof_kernel_avg_ms": rocprof_ms,
                     # what the memory system really moves while the kernel runs (the committed PMC traffic per launch over this
                     # run's launch duration), beside the algorithmic rate above: how close the ACCESS PATTERN is to the peak
                     "traffic_GBps": (None if not traffic or not kern_avg_ms else traffic / (kern_avg_ms * 1e-3) / 1e9),
                     "traffic_frac_of_peak": (None if not traffic or not kern_avg_ms else traffic / (kern_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
                     # the same bytes over each form of the step that was measured: the one launch on a text-only reference (K1 - K3),
                     # plan + screen on a planned one (K1 - K3 in two launches), the screening kernel alone (plan amortised)
                     "ref_plan_ms": plan_avg_ms, "frac_one_pass": frac_of("one-pass"), "frac_two_launch": frac_of("two-launch"),
                     "frac_amortised": frac_of("amortised"),
                     # the same kernels against a resident handle's table (expected_queries = 0: the sparse one), see "resident_table"
                     "frac_resident_one_pass": ((resident or {}).get("one-pass") or {}).get("frac"),
                     "frac_resident_amortised": ((resident or {}).get("amortised") or {}).get("frac"),
                     "table": {"policy": args.table, "expected_queries": expected, "lines": int(db.info.table_lines),
                               "bytes": int(db.info.table_bytes), "kmers_per_cell": float(db.info.stored_kmers) / max(1, 16 * int(db.info.table_lines)),
                               "stash_kmers": int(db.info.stash_kmers)}},
        "cpu_baseline": cpu,
        "end_to_end": e2e,
        "cpu_end_to_end": cpu_e2e,
        "box": box,
        "jvm": jvm,
        "java_reference": java_leg,
        "pin": pin_probe(jvm) if world == 1 else None,
        "parity": parity,
        # (ref_create: once per reference -- text upload, tile plan and the plan of the reference for cell tables: codes + one
        # anchor byte per k-mer, what every screening launch against this reference then starts from; outside the timed region
        # like the database's table, inside end_to_end)
        "setup_seconds": {"generate": gen_s, "db_open": load_s, "db_open_inner": float(db.info.load_seconds),
                          "ref_create": ref_s},
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    # a number measured on results that differ from the oracle's (or on two forms of the kernel that disagree) is not a
    # measurement: the line above says so ("parity", "identical_to_planned"), and so does the exit code
    wrong = (parity is not None and not parity["bit_exact"]) or (forms_same is False)
    if wrong:
        log("bench.py: PARITY FAILURE -- the line above is not a valid measurement")
        sys.exit(3)


if __name__ == "__main__":
    main()
