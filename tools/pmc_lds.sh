#!/bin/bash
# LDS bank-conflict share of the tile kernel (bench.py --workload transpose_add) from rocprofv3 PMC counters.
# bash tools/pmc_lds.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/pmc_lds -- \
  python3 $GRAFT_REPO_ROOT/bench.py --workload transpose_add --steps 10 --warmup 2 --no-cpu-baseline > $out/pmc_lds.log 2>&1
echo "pmc lds rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $out/pmc_lds -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].replace("smhip::(anonymous namespace)::", "").split("(")[0][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, c in acc.items():
    if "tile_kernel" not in k: continue
    n = max(cnt[k], 1)
    print(k, "launches", n)
    for name, v in c.items(): print("  %-24s %.4g per launch" % (name, v / n))
    if c.get("SQ_LDS_IDX_ACTIVE"): print("  bank-conflict cycles / LDS active cycles = %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
PY
