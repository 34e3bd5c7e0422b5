#!/bin/bash
# Memory-side and translation counters of the LDS tile kernel with one and two turned operands (tools/tile_modes.py).
# bash tools/pmc_tile.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
pass() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_tile_$name -- \
      python3 $GRAFT_REPO_ROOT/tools/tile_modes.py > $out/pmc_tile_$name.log 2>&1
  echo "pmc $name rc=$?"
}
pass utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
pass utcl2 GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass ea TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
pass lvl TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict); names = set()
for f in glob.glob(f"{out}/pmc_tile_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "tile_kernel" not in k and "contiguous_vec" not in k: continue
        k = k.replace("smhip::(anonymous namespace)::", "").replace("smhip::dev::", "").split("(")[0][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in acc:
        names.add(k)
        for c, v in acc[k].items(): rows[c][k] = v / cnt[k][c]
names = sorted(names)
print("per launch".ljust(48) + "".join(n[-34:].rjust(36) for n in names))
for c in sorted(rows): print(c.ljust(48) + "".join(("%.5g" % rows[c].get(n, float("nan"))).rjust(36) for n in names))
PY
