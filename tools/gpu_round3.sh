#!/bin/bash
# Round 3's evidence in two GPU-box visits (each under gpurun's 20-minute cap):
#   SMHIP_COMMIT=<sha> bash tools/gpu_round3.sh <tag> a    parity tests, smoke, the bench line (N = 1 with configs; --mode single; per
#                                                           workload, replayed and cold) with rocprofv3 kernel stats of the same
#                                                           commands, PMC traffic, SQ cycles, the C++ benchmarks
#   SMHIP_COMMIT=<sha> bash tools/gpu_round3.sh <tag> b    rate matrices and the round's sweeps (cold operands, launch order, large
#                                                           arrays, fused add+sum ingredients, double pow)
# tools/collect_round3.sh copies the judged summaries into profiles/r03_*.
set -o pipefail
tag=${1:-r03}; part=${2:-a}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ "$part" = a ]; then
  timeout -k 10 900 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
  tail -3 $out/pytest_gpu.log
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
  timeout -k 10 300 python bench.py > $out/bench_add.json 2> $out/bench_add.err; echo "bench rc=$?"; cut -c1-400 $out/bench_add.json
  timeout -k 10 300 python bench.py --gpus 1 --mode single --no-cpu-baseline > $out/bench_add_single.json 2> $out/bench_add_single.err; echo "bench single rc=$?"
  # the driver's N > 1 launch line with one rank: torch.distributed over RCCL + libsmhip's communicator
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --configs none > $out/bench_torchrun1.json 2> $out/bench_torchrun1.err; echo "bench torchrun rc=$?"
  for wl in bcast_mul pow add_sum transpose_add; do
    timeout -k 10 200 python bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "bench $wl rc=$?"
  done
  for wl in bcast_mul pow add_sum; do
    timeout -k 10 200 python bench.py --workload $wl --setting cold --no-cpu-baseline > $out/bench_${wl}_cold.json 2> $out/bench_${wl}_cold.err; echo "bench $wl cold rc=$?"
  done
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_add -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --configs none > $GRAFT_REPO_ROOT/$out/prof_add.log 2>&1; echo "rocprof add rc=$?"
  cd $GRAFT_REPO_ROOT
  for wl in bcast_mul pow add_sum transpose_add; do bash tools/prof_wl.sh $tag $wl --steps 200 > $out/prof_$wl.txt 2>&1; tail -3 $out/prof_$wl.txt | cut -c1-200; done
  for wl in bcast_mul pow add_sum; do  # the same commands in the cold setting: the kernel's average over rotating operand sets
    cd /tmp
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_${wl}_cold -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --setting cold --steps 200 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_${wl}_cold.log 2>&1; echo "rocprof $wl cold rc=$?"
    cd $GRAFT_REPO_ROOT
  done
  bash tools/pmc_traffic.sh $tag add bcast_mul pow add_sum transpose_add > $out/pmc_traffic.txt 2>&1; grep ratio $out/pmc_traffic.txt
  bash tools/pmc_sq.sh $tag add pow bcast_mul add_sum > $out/pmc_sq.txt 2>&1; grep "workload\|of the wave" $out/pmc_sq.txt
  timeout -k 10 100 simplemath_amd/bin/benchmark_add > $out/cpp_benchmarks.txt 2>&1; timeout -k 10 100 simplemath_amd/bin/benchmark_pow >> $out/cpp_benchmarks.txt 2>&1
  tail -12 $out/cpp_benchmarks.txt
  echo "part a done"
else
  timeout -k 10 200 python tools/op_matrix.py > $out/op_matrix.txt 2>&1
  timeout -k 10 200 python tools/bcast_matrix.py > $out/bcast_matrix.txt 2>&1
  timeout -k 10 200 python tools/reduce_rates.py > $out/reduce_rates.txt 2>&1
  timeout -k 10 200 python tools/misc_rates.py > $out/misc_rates.txt 2>&1
  timeout -k 10 100 python tools/pow_shapes.py > $out/pow_shapes.txt 2>&1
  timeout -k 10 200 python tools/chain_rates.py > $out/chain_rates.txt 2>&1
  timeout -k 10 200 python tools/pow64_rate.py > $out/pow64_rate.txt 2>&1
  timeout -k 10 110 python tools/tile_shapes.py - > $out/tile_shapes_auto.txt 2>&1
  timeout -k 10 110 python tools/tile_shapes.py - f64 > $out/tile_shapes_f64.txt 2>&1
  echo "matrices done"
  timeout -k 10 200 python tools/cold_rates.py > $out/cold_rates.txt 2>&1; cat $out/cold_rates.txt
  SMHIP_RESIDENCY=off timeout -k 10 200 python tools/cold_rates.py > $out/cold_rates_size_rule.txt 2>&1
  bash tools/pmc_cold.sh $tag > $out/pmc_cold.txt 2>&1; echo "pmc cold rc=$?"
  timeout -k 10 300 tools/bin/sweep_cold > $out/sweep_cold.txt 2>&1; echo "sweep_cold rc=$?"
  timeout -k 10 100 tools/bin/sweep_anyorder > $out/sweep_anyorder.txt 2>&1; cat $out/sweep_anyorder.txt
  bash tools/big_add.sh $tag > $out/big_add.txt 2>&1; cat $out/big_add.txt
  timeout -k 10 300 tools/bin/sweep_vmm big > $out/sweep_vmm.txt 2>&1
  bash tools/pmc_vmm.sh $tag > $out/pmc_vmm.txt 2>&1
  timeout -k 10 300 tools/bin/sweep_distance > $out/sweep_distance.txt 2>&1
  timeout -k 10 200 tools/bin/sweep_fused2 > $out/sweep_fused2.txt 2>&1
  for y in 2.5 1.5; do timeout -k 10 120 simplemath_amd/bin/pow_exhaustive $y; done > $out/pow_exhaustive.txt 2>&1
  echo "part b done"
fi
