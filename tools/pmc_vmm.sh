#!/bin/bash
# UTCL1 translation misses of the f32 add at N = 2^28 / 2^30 by how its memory was obtained (tools/sweep_vmm.hip pmc):
# does a VMM mapping with large handles / 1 GiB-aligned addresses buy translation reach?   bash tools/pmc_vmm.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for pass in "utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "utcl2 GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_vmm_$name -- $GRAFT_REPO_ROOT/tools/bin/sweep_vmm pmc > $out/pmc_vmm_$name.log 2>&1
  echo "pmc vmm $name rc=$?"
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
names = [ln.strip().split("4 launches", 1)[0].strip() for ln in open(f"{out}/pmc_vmm_utcl1.log") if "4 launches" in ln]
table = collections.defaultdict(dict)
for name in ("utcl1", "utcl2"):
    for f in glob.glob(f"{out}/pmc_vmm_{name}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "add_k" in r["Kernel_Name"]: per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, vals in per.items():
            vals.sort()
            for i, nm in enumerate(names):
                grp = [v for _, v in vals[4 * i + 1: 4 * i + 4]]  # the last three of each variant's four launches
                if grp: table[nm][c] = sum(grp) / len(grp)
cols = sorted({c for v in table.values() for c in v})
print("%-36s " % "allocation" + " ".join("%30s" % c for c in cols))
for nm in names: print("%-36s " % nm + " ".join("%30.5g" % table[nm].get(c, float("nan")) for c in cols))
PY
