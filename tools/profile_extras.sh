#!/bin/bash
# Other inputs under rocprofv3 --kernel-trace --stats, ON THE GPU BOX (tools/realistic_db_probe.py checks every window against
# the oracle): k = 25 (KMC's default; uniform-noise database), a read-error-like database (k = 31), a repeat-structured genome.
#   bash tools/profile_extras.sh <tag>   ->  gpurun_out/<tag>_{k25,readerror_db,repeats}/ + .log
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() {  # name, args...
  local name=$1; shift
  local OUT=gpurun_out/${TAG}_$name
  mkdir -p $OUT
  python3 tools/realistic_db_probe.py "$@" --steps 5 > /dev/null 2>&1            # (generates the database outside the profiler)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 tools/realistic_db_probe.py "$@" --steps 300 > $OUT.log 2> $OUT.err
  tail -2 $OUT.log
}
run k25 --k 25 --error-events 0 --noise 120000000
run readerror_db --error-events 4500000
run repeats --repeats 0.45 --error-events 0 --noise 120000000
