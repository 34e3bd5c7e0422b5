#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench lines, rocprofv3 kernel stats.
# usage (via gpurun): bash tools/gpu_round.sh <tag>
set -o pipefail
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout -k 10 300 python bench.py --gpus 1 > $out/bench_add.json 2> $out/bench_add.err; echo "bench rc=$?"; cat $out/bench_add.json
for wl in bcast_mul pow add_sum transpose_add; do
  timeout -k 10 120 python bench.py --workload $wl --steps 100 --warmup 10 > $out/bench_$wl.json 2> $out/bench_$wl.err; cat $out/bench_$wl.json
done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_add -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_add.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find $out/prof_add -name "*stats*" | head; 
f=$(find $out/prof_add -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
for wl in bcast_mul pow add_sum transpose_add; do bash tools/prof_wl.sh $tag $wl > $out/prof_$wl.txt 2>&1; done
