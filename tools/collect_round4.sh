#!/bin/bash
# Copies the judged summaries of tools/gpu_round4.sh (parts a and b, same tag) from gpurun_out/<tag>/ into profiles/<prefix>_*.
# usage: bash tools/collect_round4.sh <tag> <prefix>      e.g.  bash tools/collect_round4.sh r04z r04
set -e
src=gpurun_out/$1; pre=profiles/$2
stats() { find $src/prof_$1 -name "*kernel_stats.csv" -printf "%T@ %p\n" 2>/dev/null | sort -rn | head -1 | cut -d" " -f2-; }
for wl in add bcast_mul pow add_sum transpose_add chain; do
  [ -s $src/bench_$wl.json ] && cp $src/bench_$wl.json ${pre}_bench_$wl.json
  f=$(stats $wl); [ -n "$f" ] && cp "$f" ${pre}_${wl}_kernel_stats.csv
done
for wl in bcast_mul pow add_sum chain; do
  [ -s $src/bench_${wl}_cold.json ] && cp $src/bench_${wl}_cold.json ${pre}_bench_${wl}_cold.json
  f=$(stats ${wl}_cold); [ -n "$f" ] && cp "$f" ${pre}_${wl}_cold_kernel_stats.csv
done
for wl in bcast_mul chain; do f=$(stats ${wl}_cold_one_queue); [ -n "$f" ] && cp "$f" ${pre}_${wl}_cold_one_queue_kernel_stats.csv; done
[ -s $src/bench_add_single.json ] && cp $src/bench_add_single.json ${pre}_bench_add_single.json
[ -s $src/bench_torchrun1.json ] && cp $src/bench_torchrun1.json ${pre}_bench_torchrun1.json
[ -s $src/traffic.json ] && cp $src/traffic.json ${pre}_pmc_traffic.json && cp $src/traffic.json profiles/traffic_latest.json
for t in cpp_benchmarks small_breakdown:small_array_breakdown test_chain_fusion chain_fused_rates cold_rates_queues reduce_mid_rates:reduce_mid_rates_final op_matrix bcast_matrix reduce_rates misc_rates \
         chain_rates tile_odd fuzz_chain pow_exhaustive; do
  from=${t%%:*}; to=${t##*:}
  [ -s $src/$from.txt ] && cp $src/$from.txt ${pre}_$to.txt
done
echo collected
