#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output to HBM bytes per launch of each
workload's dominant kernel, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports exactly half of a wide (16 B/lane)
coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
Writes <dir>/traffic.json (copy to profiles/traffic_latest.json)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

DOMINANT = {"add": "contiguous_vec_kernel", "bcast_mul": "flat_tile_kernel", "pow": "flat_tile_kernel", "add_sum": "reduce_kernel",
            "transpose_add": "tile_kernel", "chain": "chain_kernel"}
# per LAUNCH of the dominant kernel: since round 3 the library issues operands above 512 MiB in pieces -- the 2^28-element add
# and fused add+sum are two launches of 2^27 elements each
ALGORITHMIC = {"add": 12 * 2 ** 27, "bcast_mul": 4 * (2 * 4096 * 4096 + 4096), "pow": 8 * 2 ** 26, "add_sum": 12 * 2 ** 27,
               "transpose_add": 12 * 8192 * 8192, "chain": 4 * (3 * 4096 * 4096 + 4096)}


def per_launch(dirname, kernel_substr, counter):
    vals = []
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel_substr in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    vals.append(float(row["Counter_Value"]))
    return vals


def main():
    out, wls = sys.argv[1], sys.argv[2:]
    res = {}
    for wl in wls:
        k = DOMINANT[wl]
        f = per_launch(os.path.join(out, f"pmc_{wl}_FETCH_SIZE"), k, "FETCH_SIZE")
        w = per_launch(os.path.join(out, f"pmc_{wl}_WRITE_SIZE"), k, "WRITE_SIZE")
        if not f or not w:
            res[wl] = {"error": f"no counter rows (fetch {len(f)}, write {len(w)})"}
            continue
        fetch_kib = sum(f) / len(f)
        write_kib = sum(w) / len(w)
        read_b = 2.0 * fetch_kib * 1024.0   # gfx950: FETCH_SIZE = half the bytes of a 16 B/lane stream
        write_b = write_kib * 1024.0
        res[wl] = {"kernel": k, "launches": len(f), "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
                   "read_bytes": read_b, "write_bytes": write_b, "hbm_bytes_per_launch": read_b + write_b,
                   "algorithmic_bytes": ALGORITHMIC[wl], "ratio_to_algorithmic": (read_b + write_b) / ALGORITHMIC[wl]}
    import datetime
    res["_meta"] = {"date": datetime.date.today().isoformat(), "commit": os.environ.get("SMHIP_COMMIT", "unknown"),
                    "how": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, of `bench.py --workload W --steps 10`"}
    with open(os.path.join(out, "traffic.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
