#!/bin/bash
# The "two speeds" of a 206 GB table (profiles/r05_notes.md section 5: a pass over the 9.1 Gb genome takes 43 ms in most processes and 52 ms
# in some, a property of the process): the same open + passes in several fresh processes, every one under rocprofv3 --pmc with the
# memory-latency and translation counters, the table allocated after the loader's buffers (default) or before everything else
# (KCFX_LOAD_TABLE_EARLY=1: never slow in round 5). ON THE GPU BOX:   bash tools/two_speeds_pmc.sh <tag> [ref_len]
TAG=${1:-r06_two_speeds}; LEN=${2:-9100000000}
export KCFX_TEST_HOOKS=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT
D=/dev/shm/kcfx_large; mkdir -p $D
[ -f $D/w.kmc_suf ] || ./kcftools_amd/bin/kcfx_synth --out $D/w --ref-len $LEN --seed 41 --sample-seed 42 --noise 0 > $OUT/synth.log 2>&1
cat $D/w.kmc_suf > /dev/null
i=0
for mode in default default early default default early; do
  i=$((i + 1))
  if [ $mode = early ]; then export KCFX_LOAD_TABLE_EARLY=1; else unset KCFX_LOAD_TABLE_EARLY; fi
  timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 tools/large_table_probe.py --prefix $D/w --passes 4 > $OUT/p$i.log 2> $OUT/p$i.err
  python3 - $OUT/p$i $mode $i <<'PY'
import csv, glob, sys, collections
d, mode, i = sys.argv[1:4]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "screen_k31c_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "screen_k31c_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
c = {k: sum(v) / len(v) for k, v in agg.items()}
lat = c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(1.0, c.get("TCP_TCC_READ_REQ_sum", 1))
print(f"process {i} ({mode}): pass under the counters {sorted(dur)[len(dur) // 2] if dur else 0:.1f} ms (median of {len(dur)}), L2 read latency {lat:.0f} cycles, "
      f"HBM requests {c.get('TCC_EA0_RDREQ_sum', 0) / 1e6:.0f} M, UTCL1 misses {c.get('TCP_UTCL1_TRANSLATION_MISS_sum', 0) / 1e6:.0f} M, "
      f"UTCL2 out-of-credit stall cycles {c.get('TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum', 0) / 1e6:.0f} M", flush=True)
PY
  grep -h "pass_ms\|open_s" $OUT/p$i.log | cut -c1-300
done
rm -f $D/w.*
