#!/bin/bash
# Copies the judged summaries of tools/gpu_round3.sh (parts a and b, same tag) from gpurun_out/<tag>/ into profiles/<prefix>_*.
# usage: bash tools/collect_round3.sh <tag> <prefix>      e.g.  bash tools/collect_round3.sh r03z r03
set -e
src=gpurun_out/$1; pre=profiles/$2
for wl in add bcast_mul pow add_sum transpose_add; do
  cp $src/bench_$wl.json ${pre}_bench_$wl.json
  f=$(find $src/prof_$wl -name "*kernel_stats.csv" -printf "%T@ %p\n" | sort -rn | head -1 | cut -d" " -f2-); [ -n "$f" ] && cp "$f" ${pre}_${wl}_kernel_stats.csv
done
for wl in bcast_mul pow add_sum; do
  cp $src/bench_${wl}_cold.json ${pre}_bench_${wl}_cold.json
  f=$(find $src/prof_${wl}_cold -name "*kernel_stats.csv" -printf "%T@ %p\n" | sort -rn | head -1 | cut -d" " -f2-); [ -n "$f" ] && cp "$f" ${pre}_${wl}_cold_kernel_stats.csv
done
cp $src/bench_add_single.json ${pre}_bench_add_single.json
[ -s $src/bench_torchrun1.json ] && cp $src/bench_torchrun1.json ${pre}_bench_torchrun1.json
cp $src/traffic.json ${pre}_pmc_traffic.json
cp $src/traffic.json profiles/traffic_latest.json
cp $src/pmc_sq.txt ${pre}_pmc_sq_cycles.txt
cp $src/cpp_benchmarks.txt ${pre}_cpp_benchmarks.txt
for t in op_matrix bcast_matrix reduce_rates misc_rates pow_shapes pow_exhaustive chain_rates pow64_rate cold_rates cold_rates_size_rule sweep_cold \
         sweep_anyorder big_add sweep_vmm pmc_vmm sweep_distance sweep_fused2 tile_shapes_auto tile_shapes_f64; do
  [ -s $src/$t.txt ] && cp $src/$t.txt ${pre}_$t.txt
done
# the cold-operand counter table: the summary lines only (the pass log above them is scratch)
[ -s $src/pmc_cold.txt ] && sed -n '/^per launch of the case/,$p' $src/pmc_cold.txt > ${pre}_pmc_cold.txt
echo collected
