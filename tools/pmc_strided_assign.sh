#!/bin/bash
# What does the memory system move for A[:, ::2] = B[:, ::2]?  FETCH_SIZE / WRITE_SIZE (separate passes) and the L2's
# request counters for the strided copy kernel next to a dense copy.   bash tools/pmc_strided_assign.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
pass() {
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_sa_$name -- \
      python3 $GRAFT_REPO_ROOT/tools/strided_assign.py > $out/pmc_sa_$name.log 2>&1
  echo "pmc $name rc=$?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum TCC_MISS_sum
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_WRITEBACK_sum
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
order = []
for f in sorted(glob.glob(f"{out}/pmc_sa_*/**/*counter_collection.csv", recursive=True)):
    # the three timed loops launch the same kernel template with different arguments: tell them apart by dispatch order
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "strided_copy" not in r["Kernel_Name"] and "copy" not in r["Kernel_Name"].lower(): continue
        per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), r["Kernel_Name"].replace("smhip::(anonymous namespace)::", "").split("(")[0][:60], float(r["Counter_Value"])))
    for c, rows in per.items():
        rows.sort()
        third = len(rows) // 3
        for i, label in enumerate(("dense copy", "A[:, ::2] = B[:, ::2]", "A[:, :4096] = B[:, :4096]")):
            part = rows[i * third:(i + 1) * third]
            if not part: continue
            print("%-28s %-26s %-50s %.5g per launch" % (c, label, part[0][1], sum(v for _, _, v in part) / len(part)))
PY
