#!/bin/bash
# round 2, first GPU visit: the new multi-GPU / pool / jit code, then the whole GPU suite and the bench modes
set -o pipefail
out=gpurun_out/r02a
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_multi_gpu.py tests/test_gpu_cpp.py -x -q -m gpu > $out/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -15 $out/pytest_new.log
timeout -k 10 700 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest_gpu.log
timeout -k 10 200 python bench.py > $out/bench_add.json 2> $out/bench_add.err; echo "bench rc=$?"; cat $out/bench_add.json; tail -3 $out/bench_add.err
timeout -k 10 200 python bench.py --gpus 1 --mode single --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_single.json 2> $out/bench_single.err; echo "single rc=$?"; cat $out/bench_single.json; tail -3 $out/bench_single.err
for wl in bcast_mul pow add_sum; do
  timeout -k 10 200 python bench.py --workload $wl --steps 100 --warmup 10 > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "$wl rc=$?"; cat $out/bench_$wl.json; tail -3 $out/bench_$wl.err
done
