#!/bin/bash
# Ragged-pitch transposes: what the memory side sees for out = A.T + B at 12288^2 (rows on 128-byte lines)
# and 12287^2 (rows off them): bytes fetched / written, read and write requests and their sizes.   bash tools/pmc_tile_odd.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "tcc TCC_REQ_sum TCC_MISS_sum"; do
  set -- $pass; name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_tileodd_$name -- python3 $GRAFT_REPO_ROOT/tools/tile_pair.py > $out/pmc_tileodd_$name.log 2>&1
  echo "pmc $name rc=$?"
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc_tileodd_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tile_kernel" not in r["Kernel_Name"]: continue
        acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for grid in sorted(acc, key=int):
    print("tile_kernel grid_size", grid)
    for c, v in sorted(acc[grid].items()): print("   %-30s %16.6g per launch (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
