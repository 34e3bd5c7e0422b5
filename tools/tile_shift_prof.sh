#!/bin/bash
# Kernel trace (registers, LDS, scratch, durations) of the tile kernels on 12288^2 / 12287^2.   bash tools/tile_shift_prof.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for mode in 0 1 2; do
  SMHIP_TILE_SHIFT=$mode timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$mode -- python3 $GRAFT_REPO_ROOT/tools/tile_pair.py 12288 12287 > $out/trace_$mode.log 2>&1
  echo "mode $mode rc=$?"; cat $out/trace_$mode.log | grep "A.T"
  f=$(find $out/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list); meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "tile" not in k: continue
    k = k.replace("smhip::(anonymous namespace)::", "").replace("smhip::dev::", "").split("(")[0][:70]
    key = (k, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
    acc[key].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    meta[key] = {c: r[c] for c in r if any(w in c for w in ("VGPR", "SGPR", "LDS", "Scratch", "Workgroup_Size_X"))}
for key, v in acc.items():
    print("  %-72s grid %-9s x%d  %.1f us  %s" % (key[0], key[1], len(v), sum(v) / len(v), meta[key]))
PY
done
