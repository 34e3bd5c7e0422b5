#!/bin/bash
# Why does the f32 add fall from 82 % of HBM peak at N = 2^28 to 78 % / 77 % at 2^30 / 2^31?  Address-translation and
# memory-side counters for the same kernel at the three sizes (separate passes; none combined with trace domains).
# bash tools/pmc_tlb.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
pass() {  # name, counters...
  name=$1; shift
  for lg in 28 30 31; do
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_${name}_$lg -- \
      python3 $GRAFT_REPO_ROOT/bench.py --log2n $lg --steps 5 --warmup 2 --no-cpu-baseline --configs none --prewarm 0.05 > $out/pmc_${name}_$lg.log 2>&1
    echo "pmc $name 2^$lg rc=$?"
  done
}
pass utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
pass utcl2 GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
pass ea TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
pass lvl TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
print("counter                                        per launch, contiguous_vec_kernel<float, AddOp>:   N=2^28          N=2^30          N=2^31     (and per MiB of traffic)")
rows = collections.defaultdict(dict)
dur = {}
for lg in (28, 30, 31):
    for f in glob.glob(f"{out}/pmc_*_{lg}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); cnt = collections.Counter(); d = 0.0
        for r in csv.DictReader(open(f)):
            if "contiguous_vec_kernel" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        for k, v in acc.items(): rows[k][lg] = v / max(cnt[k], 1)
for k in sorted(rows):
    vals = [rows[k].get(lg) for lg in (28, 30, 31)]
    mib = [3 * 4 * (1 << lg) / 2**20 for lg in (28, 30, 31)]
    print("%-46s %s" % (k, "   ".join("%14.4g (%8.3g)" % (v, v / m) if v is not None else "%25s" % "-" for v, m in zip(vals, mib))))
PY
