#!/bin/bash
# Where do the waves of a workload's dominant kernel spend their cycles?  rocprofv3 SQ counters (one pass, 8 SQ slots).
# bash tools/pmc_sq.sh <tag> <workload>...
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for wl in "$@"; do
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $out/pmc_sq_$wl -- \
    python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --configs none > $out/pmc_sq_$wl.log 2>&1
  echo "pmc sq $wl rc=$?"
  f=$(find $out/pmc_sq_$wl -name "*counter_collection.csv" | head -1)
  python3 - "$f" $wl <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
for r in rows:
    k = r["Kernel_Name"].replace("smhip::(anonymous namespace)::", "").replace("smhip::dev::", "").split("(")[0][:80]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[k] += 1; dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
k = max(dur, key=dur.get)  # the kernel the workload spends its time in
c, n = acc[k], max(cnt[k], 1)
print("workload %s: %s, %d launches" % (sys.argv[2], k, n))
for name, v in sorted(c.items()): print("  %-22s %.4g per launch" % (name, v / n))
w = c["SQ_WAVE_CYCLES"]
print("  of the wave cycles: waiting (s_waitcnt / barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %% (VALU %.1f %%)" % (
    100 * c["SQ_WAIT_ANY"] / w, 100 * c["SQ_WAIT_INST_ANY"] / w, 100 * c["SQ_ACTIVE_INST_ANY"] / w, 100 * c["SQ_ACTIVE_INST_VALU"] / w))
PY
done
