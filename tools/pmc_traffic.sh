#!/bin/bash
# HBM traffic of bench.py workloads from rocprofv3 PMC counters, per the MI355X guide:
# FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots), no trace domains beside
# --kernel-trace.   bash tools/pmc_traffic.sh <tag> <workload>...
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for wl in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${wl}_$ctr -- \
      python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --configs none > $out/pmc_${wl}_$ctr.log 2>&1
    echo "pmc $wl $ctr rc=$?"
  done
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summarise.py $out "$@"
