#!/bin/bash
# rocprofv3 kernel stats for one bench.py workload: bash tools/prof_wl.sh <tag> <workload> [extra bench args]
tag=$1; wl=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 3 --no-cpu-baseline --configs none "$@" > $out/prof_$wl.log 2>&1
echo "rocprof $wl rc=$?"
f=$(find $out/prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cat "$f" | cut -c1-400
