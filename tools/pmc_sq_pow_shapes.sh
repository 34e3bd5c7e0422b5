#!/bin/bash
# SQ cycle breakdown of every pow kernel tools/pow_shapes.py launches (scalar / array / row / column exponents at 2^24).
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_sq_powshapes -- \
    python3 $GRAFT_REPO_ROOT/tools/pow_shapes.py > $out/pmc_sq_powshapes.log 2>&1
echo "rc=$?"
f=$(find $out/pmc_sq_powshapes -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("smhip::(anonymous namespace)::", "").replace("smhip::dev::", "").replace("smhip::bk::", "").split("(")[0][:90]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[k] += 1; dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(dur, key=dur.get, reverse=True):
    if cnt[k] < 50: continue
    c, n = acc[k], cnt[k]
    w = c["SQ_WAVE_CYCLES"]
    print("%-92s x%d  %.1f us/launch" % (k, n, dur[k] / n / 1e3))
    print("    waves %.3g  VALU insts %.3g  LDS insts %.3g | of wave cycles: waiting %.1f %%, issue-stalled %.1f %%, issuing %.1f %% (VALU %.1f %%)" % (
        c["SQ_WAVES"] / n, c["SQ_INSTS_VALU"] / n, c["SQ_INSTS_LDS"] / n, 100 * c["SQ_WAIT_ANY"] / w, 100 * c["SQ_WAIT_INST_ANY"] / w, 100 * c["SQ_ACTIVE_INST_ANY"] / w, 100 * c["SQ_ACTIVE_INST_VALU"] / w))
PY
