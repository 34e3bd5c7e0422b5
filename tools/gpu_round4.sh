#!/bin/bash
# Round 4's evidence in three GPU-box visits (each under gpurun's 20-minute cap):
#   SMHIP_COMMIT=<sha> bash tools/gpu_round4.sh <tag> a    parity tests, smoke, the bench line (N = 1 with configs incl. "chain"; --mode single with the
#                                                           sharded config-3 leg; per workload, replayed and cold)
#   SMHIP_COMMIT=<sha> bash tools/gpu_round4.sh <tag> c    rocprofv3 kernel stats of the same bench commands, PMC traffic, the C++ benchmarks and fusion
#                                                           test, the small-array breakdown, f64 pow rates
#   SMHIP_COMMIT=<sha> bash tools/gpu_round4.sh <tag> b    rate tables: chains, cold operands on one / two queues, mid-size reductions, the r03 matrices,
#                                                           fuzzers (chains, views, policy)
# tools/collect_round4.sh copies the judged summaries into profiles/r04_*.
set -o pipefail
tag=${1:-r04}; part=${2:-a}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
prof() {  # prof <name> <bench args...>: rocprofv3 kernel stats of one bench.py command
  local name=$1; shift
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_$name -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/$out/prof_$name.log 2>&1 )
  echo "rocprof $name rc=$?"
}
if [ "$part" = a ]; then
  timeout -k 10 900 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
  tail -3 $out/pytest_gpu.log
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
  timeout -k 10 400 python bench.py > $out/bench_add.json 2> $out/bench_add.err; echo "bench rc=$?"; cut -c1-300 $out/bench_add.json
  timeout -k 10 300 python bench.py --gpus 1 --mode single --no-cpu-baseline > $out/bench_add_single.json 2> $out/bench_add_single.err; echo "bench single rc=$?"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --configs none > $out/bench_torchrun1.json 2> $out/bench_torchrun1.err; echo "bench torchrun rc=$?"
  for wl in bcast_mul pow add_sum transpose_add chain; do
    timeout -k 10 200 python bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "bench $wl rc=$?"
  done
  for wl in bcast_mul pow add_sum chain; do
    timeout -k 10 200 python bench.py --workload $wl --setting cold --no-cpu-baseline > $out/bench_${wl}_cold.json 2> $out/bench_${wl}_cold.err; echo "bench $wl cold rc=$?"
  done
  echo "part a done"
elif [ "$part" = c ]; then
  prof add --steps 200 --warmup 20 --no-cpu-baseline --configs none
  for wl in bcast_mul pow add_sum transpose_add chain; do prof $wl --workload $wl --steps 200 --warmup 3 --no-cpu-baseline --configs none; done
  for wl in bcast_mul pow add_sum chain; do prof ${wl}_cold --workload $wl --setting cold --steps 200 --warmup 20 --no-cpu-baseline --configs none; done
  # the cold legs on ONE queue: what a kernel's own duration is when nothing overlaps it
  for wl in bcast_mul chain; do SMHIP_QUEUES=1 prof ${wl}_cold_one_queue --workload $wl --setting cold --steps 200 --warmup 20 --no-cpu-baseline --configs none; done
  bash tools/pmc_traffic.sh $tag add bcast_mul pow add_sum transpose_add chain > $out/pmc_traffic.txt 2>&1; grep ratio $out/pmc_traffic.txt
  timeout -k 10 100 simplemath_amd/bin/benchmark_add > $out/cpp_benchmarks.txt 2>&1; timeout -k 10 100 simplemath_amd/bin/benchmark_pow >> $out/cpp_benchmarks.txt 2>&1
  tail -16 $out/cpp_benchmarks.txt
  timeout -k 10 100 simplemath_amd/bin/test_chain_fusion > $out/test_chain_fusion.txt 2>&1; tail -1 $out/test_chain_fusion.txt
  timeout -k 10 100 tools/bin/small_breakdown > $out/small_breakdown.txt 2>&1
  python -c "
from oracle import oracle as orc
r = orc.Reference()
print('the reference on this host (oracle/ref_shim.cpp: ref_bench_tiny), ns per iteration: simple_check %.0f  BM_SMArrayPow_1D %.0f  BM_SMArrayPow_2D %.0f' % tuple(r.bench_tiny(k, 300000) for k in range(3)))" >> $out/small_breakdown.txt 2>&1
  tail -4 $out/small_breakdown.txt
  timeout -k 10 200 python tools/pow64_rate.py > $out/pow64_rate.txt 2>&1; tail -14 $out/pow64_rate.txt
  (echo "# SMHIP_TINY_BATCH=0 (one launch per operator)"; SMHIP_TINY_BATCH=0 timeout 200 python tools/tiny_latency.py; echo "# recorded (default)"; timeout 200 python tools/tiny_latency.py; \
   echo "# SMHIP_TINY_BATCH=0 again"; SMHIP_TINY_BATCH=0 timeout 200 python tools/tiny_latency.py) > $out/tiny_latency.txt 2>&1
  timeout -k 10 200 simplemath_amd/bin/pool_streams > $out/pool_streams.txt 2>&1; tail -1 $out/pool_streams.txt
  echo "part c done"
else
  timeout -k 10 300 python tools/chain_fused_rates.py > $out/chain_fused_rates.txt 2>&1; cat $out/chain_fused_rates.txt
  for q in 2 1; do echo "SMHIP_QUEUES=$q"; SMHIP_QUEUES=$q timeout -k 10 300 python tools/cold_rates.py --sizes 8,16,32,64,128,256; done > $out/cold_rates_queues.txt 2>&1; echo "cold rates rc=$?"
  timeout -k 10 300 python tools/reduce_mid_rates.py > $out/reduce_mid_rates.txt 2>&1; echo "reduce mid rc=$?"
  timeout -k 10 200 python tools/op_matrix.py > $out/op_matrix.txt 2>&1
  timeout -k 10 200 python tools/bcast_matrix.py > $out/bcast_matrix.txt 2>&1
  timeout -k 10 200 python tools/reduce_rates.py > $out/reduce_rates.txt 2>&1
  timeout -k 10 200 python tools/misc_rates.py > $out/misc_rates.txt 2>&1
  timeout -k 10 200 python tools/chain_rates.py > $out/chain_rates.txt 2>&1
  timeout -k 10 110 python tools/tile_shapes.py - odd > $out/tile_odd.txt 2>&1
  echo "matrices done"
  for s in 11 12; do timeout -k 10 400 python tests/fuzz_chain.py 2000 $s; done > $out/fuzz_chain.txt 2>&1; tail -2 $out/fuzz_chain.txt
  timeout -k 10 400 python tests/fuzz_views.py 6000 21 > $out/fuzz_views.txt 2>&1; tail -1 $out/fuzz_views.txt
  timeout -k 10 400 python tests/fuzz_policy.py 300 21 > $out/fuzz_policy.txt 2>&1; tail -1 $out/fuzz_policy.txt
  timeout -k 10 300 python tests/fuzz_flat.py 1500 21 > $out/fuzz_flat.txt 2>&1; tail -1 $out/fuzz_flat.txt
  for y in 2.5 1.5; do timeout -k 10 120 simplemath_amd/bin/pow_exhaustive $y; done > $out/pow_exhaustive.txt 2>&1
  echo "part b done"
fi
