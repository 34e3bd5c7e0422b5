#!/bin/bash
# One GPU-box visit that regenerates the round's evidence: parity tests, smoke, the bench line per workload, rocprofv3
# kernel stats of the same commands, PMC traffic, SQ cycle breakdown, the C++ benchmarks and the rate matrices.
# usage (via gpurun): SMHIP_COMMIT=<sha> bash tools/gpu_round.sh <tag>
set -o pipefail
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
tail -3 $out/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout -k 10 300 python bench.py > $out/bench_add.json 2> $out/bench_add.err; echo "bench rc=$?"; cut -c1-600 $out/bench_add.json
timeout -k 10 300 python bench.py --gpus 1 --mode single --no-cpu-baseline > $out/bench_add_single.json 2> $out/bench_add_single.err; echo "bench single rc=$?"
for wl in bcast_mul pow add_sum transpose_add; do
  timeout -k 10 200 python bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "bench $wl rc=$?"
done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_add -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_add.log 2>&1; echo "rocprof add rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $out/prof_add -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -4 "$f" | cut -c1-300
for wl in bcast_mul pow add_sum transpose_add; do bash tools/prof_wl.sh $tag $wl --steps 200 > $out/prof_$wl.txt 2>&1; tail -3 $out/prof_$wl.txt | cut -c1-300; done
bash tools/pmc_traffic.sh $tag add bcast_mul pow add_sum transpose_add > $out/pmc_traffic.txt 2>&1; grep ratio $out/pmc_traffic.txt
bash tools/pmc_sq.sh $tag add pow bcast_mul add_sum > $out/pmc_sq.txt 2>&1; grep "workload\|of the wave\|INSTS_VALU" $out/pmc_sq.txt
timeout -k 10 100 simplemath_amd/bin/benchmark_add > $out/cpp_benchmarks.txt 2>&1; timeout -k 10 100 simplemath_amd/bin/benchmark_pow >> $out/cpp_benchmarks.txt 2>&1
echo "with SMHIP_STORE_POLICY=nt (results always stored non-temporally):" >> $out/cpp_benchmarks.txt
SMHIP_STORE_POLICY=nt timeout -k 10 100 simplemath_amd/bin/benchmark_add 2>&1 | grep chain_check >> $out/cpp_benchmarks.txt; cat $out/cpp_benchmarks.txt
timeout -k 10 200 python tools/op_matrix.py > $out/op_matrix.txt 2>&1
timeout -k 10 200 python tools/bcast_matrix.py > $out/bcast_matrix.txt 2>&1
timeout -k 10 200 python tools/reduce_rates.py > $out/reduce_rates.txt 2>&1
timeout -k 10 200 python tools/misc_rates.py > $out/misc_rates.txt 2>&1
timeout -k 10 100 python tools/pow_shapes.py > $out/pow_shapes.txt 2>&1
timeout -k 10 200 python tools/chain_rates.py > $out/chain_rates.txt 2>&1
timeout -k 10 200 python tools/pow64_rate.py > $out/pow64_rate.txt 2>&1
timeout -k 10 100 python tools/tile_modes.py > $out/tile_modes.txt 2>&1
timeout -k 10 100 python tools/pitch_views.py > $out/pitch_views.txt 2>&1
bash tools/pmc_sq_pow_shapes.sh $tag > $out/pmc_sq_pow_shapes.txt 2>&1
mkdir -p tools/bin && g++ -std=c++20 -O2 -Iinclude tools/small_breakdown.cpp -Lsimplemath_amd/lib -lsmhip -Wl,-rpath,$PWD/simplemath_amd/lib -o tools/bin/small_breakdown 2> $out/small_breakdown_build.log \
  && timeout -k 10 100 tools/bin/small_breakdown > $out/small_breakdown.txt 2>&1
for y in 2.5 1.5 3.25; do timeout -k 10 120 simplemath_amd/bin/pow_exhaustive $y; done > $out/pow_exhaustive.txt 2>&1
echo done
