#!/bin/bash
# A.T + B at 16384 x 16384 (1 GiB per operand) runs at 68 % where 8192 x 8192 runs at 88 %: translation? memory-side latency?
# bash tools/pmc_tile_big.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for pass in "utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "utcl2 GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "ea TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" "tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_32B_sum"; do
  set -- $pass; name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_tilebig_$name -- python3 $GRAFT_REPO_ROOT/tools/tile_sizes.py > $out/pmc_tilebig_$name.log 2>&1
  echo "pmc $name rc=$?"
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc_tilebig_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tile_kernel" not in r["Kernel_Name"]: continue
        acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for grid in sorted(acc, key=int):
    print("tile_kernel grid_size", grid)
    for c, v in sorted(acc[grid].items()): print("   %-34s %14.5g per launch (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
