#!/bin/bash
# Copies the judged summaries of one tools/gpu_round.sh visit from gpurun_out/<tag>/ into profiles/<prefix>_*.
# usage: bash tools/collect_round.sh <tag> <prefix>      e.g.  bash tools/collect_round.sh r02p r02
set -e
src=gpurun_out/$1; pre=profiles/$2
for wl in add bcast_mul pow add_sum transpose_add; do
  cp $src/bench_$wl.json ${pre}_bench_$wl.json
  f=$(find $src/prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" ${pre}_${wl}_kernel_stats.csv
done
cp $src/bench_add_single.json ${pre}_bench_add_single.json
cp $src/traffic.json ${pre}_pmc_traffic.json
cp $src/traffic.json profiles/traffic_latest.json
cp $src/pmc_sq.txt ${pre}_pmc_sq_cycles.txt
cp $src/pmc_sq_pow_shapes.txt ${pre}_pmc_sq_pow_shapes.txt
cp $src/cpp_benchmarks.txt ${pre}_cpp_benchmarks.txt
[ -s $src/small_breakdown.txt ] && ! grep -q "failed to run" $src/small_breakdown.txt && cp $src/small_breakdown.txt ${pre}_small_array_breakdown.txt
for t in op_matrix bcast_matrix reduce_rates misc_rates pow_shapes pow_exhaustive; do cp $src/$t.txt ${pre}_$t.txt; done
for t in chain_rates pow64_rate tile_modes; do [ -s $src/$t.txt ] && cp $src/$t.txt ${pre}_${t}_final.txt; done
[ -s $src/pitch_views.txt ] && cp $src/pitch_views.txt ${pre}_pitch_views.txt
echo collected
