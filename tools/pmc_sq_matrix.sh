#!/bin/bash
# SQ cycle breakdown per kernel over tools/bcast_matrix.py (all broadcast kernels).  bash tools/pmc_sq_matrix.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/pmc_sq_matrix -- \
  python3 $GRAFT_REPO_ROOT/${SM_SCRIPT:-tools/bcast_matrix.py} > $out/pmc_sq_matrix.log 2>&1
echo "rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $out/pmc_sq_matrix -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("smhip::(anonymous namespace)::", "").replace("smhip::dev::", "").split("(")[0]
    k = k + " grid=" + r["Grid_Size"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, c in sorted(acc.items()):
    if cnt[k] < 20 or "uniform" in k: continue
    w = c["SQ_WAVE_CYCLES"]
    print("%-100s wait %.0f%% stall %.0f%% issue %.0f%% (VALU %.1f%% LDS %.1f%%)  VALU insts/launch %.3g" % (k[:100], 100 * c["SQ_WAIT_ANY"] / w, 100 * c["SQ_WAIT_INST_ANY"] / w,
          100 * c["SQ_ACTIVE_INST_ANY"] / w, 100 * c["SQ_ACTIVE_INST_VALU"] / w, 100 * c["SQ_ACTIVE_INST_LDS"] / w, c["SQ_INSTS_VALU"] / cnt[k]))
PY
