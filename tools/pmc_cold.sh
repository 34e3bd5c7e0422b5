#!/bin/bash
# Where do mid-size launches on COLD operands lose their 8-25 points?  The same library kernel, 30 launches re-reading the
# same operands and then 30 launches rotating through >= 2.5 GiB of disjoint operand sets (tools/cold_rates.py --pmc),
# under separate rocprofv3 --pmc passes (none combined with trace domains other than --kernel-trace):
#   translation (UTCL1 / UTCL2), dispatch (waves, busy cycles, GRBM_GUI_ACTIVE against the kernel's duration),
#   memory side (EA request levels = average requests in flight, DRAM vs total requests).
# bash tools/pmc_cold.sh <tag> [case:MiB ...]
tag=$1; shift
cases=${@:-"scalar:16 scalar:64 scalar:128 row:64 add:16"}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
pass() {  # name, counters...
  name=$1; shift
  for c in $cases; do
    d=$out/pmc_${name}_${c/:/_}
    timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- \
      python3 $GRAFT_REPO_ROOT/tools/cold_rates.py --pmc $c > $d.log 2>&1
    echo "pmc $name $c rc=$?"
  done
}
pass utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass utcl2 GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
pass ea TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
pass dram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum
python3 - $out $cases <<'PY'
import csv, glob, sys, collections
out, cases = sys.argv[1], sys.argv[2:]
print("per launch of the case's kernel: first 30 launches = SAME operands, last 30 = ROTATING cold operands")
print("%-12s %-34s %16s %16s %8s" % ("case", "counter", "same", "rotate", "ratio"))
for c in cases:
    tagc = c.replace(":", "_")
    seen_dur = False
    for d in sorted(glob.glob(f"{out}/pmc_*_{tagc}")):
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            rows = list(csv.DictReader(open(f)))
            if not rows: continue
            # the case's kernel is the last one launched; its cold and its replayed launches may be different instantiations
            # (config 3's shape takes one vector per lane cold, two replayed), so launches are matched by the function's name
            base = lambda name: name.split("<")[0]
            kern = base(max(rows, key=lambda r: int(r["Dispatch_Id"]))["Kernel_Name"])
            per = collections.defaultdict(list); dur = {}
            for r in rows:
                if base(r["Kernel_Name"]) != kern: continue
                per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                dur[int(r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            if not seen_dur:
                ids = sorted(dur)
                # the first 2000+ launches of this kernel name may be the clock-ramp loop of another size: take the LAST 30+K+30
                s = [dur[i] for i in ids[-30:]]; first = None
                print("%-12s %-34s %16s %16.0f" % (c, "kernel ns under the profiler (rotate)", "", sum(s) / len(s)))
                seen_dur = True
            for name, vals in sorted(per.items()):
                vals.sort()
                rot = [v for _, v in vals[-30:]]
                # the 30 `same` launches sit right before the K + 30 rotate launches
                n_tail = None
                logf = d + ".log"
                try:
                    for line in open(logf):
                        if line.startswith("pmc case"):
                            n_tail = int(line.split("then")[1].split("+")[0])
                except OSError:
                    pass
                if n_tail is None: continue
                lo = len(vals) - 30 - n_tail - 30
                same = [v for _, v in vals[lo:lo + 30]]
                a, b = sum(same) / len(same), sum(rot) / len(rot)
                print("%-12s %-34s %16.5g %16.5g %8.2f" % (c, name, a, b, b / a if a else float("nan")))
PY
