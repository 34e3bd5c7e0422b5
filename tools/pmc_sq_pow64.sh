#!/bin/bash
# SQ cycle breakdown and LDS bank conflicts of the f64 pow kernels (tools/pow64_rate.py).   bash tools/pmc_sq_pow64.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_sq_pow64 -- \
    python3 $GRAFT_REPO_ROOT/tools/pow64_rate.py > $out/pmc_sq_pow64.log 2>&1
echo "rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/pmc_lds_pow64 -- \
    python3 $GRAFT_REPO_ROOT/tools/pow64_rate.py > $out/pmc_lds_pow64.log 2>&1
echo "rc=$?"
for d in pmc_sq_pow64 pmc_lds_pow64; do
f=$(find $out/$d -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
first = None
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("smhip::(anonymous namespace)::", "").replace("smhip::dev::", "").split("(")[0][:90]
    if "double" not in k: continue
    # launches are told apart by their dispatch order: keep (kernel, which-source) = every run of 45 launches
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if first is None: first = r["Counter_Name"]
    if r["Counter_Name"] == first:
        cnt[k] += 1; dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(dur, key=dur.get, reverse=True):
    c, n = acc[k], cnt[k]
    print("%-92s x%d  %.1f us/launch" % (k, n, dur[k] / n / 1e3))
    print("    " + "  ".join("%s %.4g" % (name, v / n) for name, v in sorted(c.items())))
PY
done
