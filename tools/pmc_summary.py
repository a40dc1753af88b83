#!/usr/bin/env python3
"""MFMA utilisation per kernel and HBM traffic per step from rocprofv3 PMC passes (MI355X_MICROARCH.md conventions).

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d A -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline   (the default schedule is single-stream)
  rocprofv3 --pmc FETCH_SIZE ... -d B -- (same command)        rocprofv3 --pmc WRITE_SIZE ... -d C -- (same command)
  python tools/pmc_summary.py A B C profiles/r01_step_pmc_summary.json

MfmaUtil of a dispatch = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (cycles * 1024), cycles = GRBM_GUI_ACTIVE / 8
(rocprofv3 reports the sum over the 8 XCDs).  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction).
Per-step figures divide the totals of the vg_* kernels by the number of vg_adamw launches / 2.
"""
import collections
import csv
import glob
import json
import sys


def load(d):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    return per, names


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:48]


def main():
    a, b, c, out = sys.argv[1:5]
    mf, names = load(a)
    util = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for d, v in mf.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or "GRBM_GUI_ACTIVE" not in v:
            continue
        k = short(names[d])
        util[k][0] += v["SQ_VALU_MFMA_BUSY_CYCLES"]
        util[k][1] += v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        util[k][2] += 1
    steps = max(1, sum(1 for n in names.values() if "vg_adamw" in n) // 2)
    rec = {"note": __doc__.strip().split("\n\n")[1], "steps_in_profile": steps, "mfma_util_percent": {}, "hbm": {}}
    tot_busy = tot_cyc = 0.0
    for k, (busy, cyc, n) in sorted(util.items(), key=lambda kv: -kv[1][1]):
        if not k.startswith(("vg_", "_Z")):
            continue
        tot_busy += busy
        tot_cyc += cyc
        if busy > 0:
            rec["mfma_util_percent"][k] = {"launches_per_step": round(n / steps, 1), "util": round(100.0 * busy / cyc, 1)}
    rec["mfma_util_percent"]["<whole step, all vg_ kernels>"] = {"util": round(100.0 * tot_busy / max(tot_cyc, 1.0), 1)}
    fetch, fn = load(b)
    write, wn = load(c)
    fb = sum(v.get("FETCH_SIZE", 0.0) for d, v in fetch.items() if short(fn[d]).startswith(("vg_", "_Z")))
    wb = sum(v.get("WRITE_SIZE", 0.0) for d, v in write.items() if short(wn[d]).startswith(("vg_", "_Z")))
    rec["hbm"] = {"read_GB_per_step": round(2 * fb * 1024 / steps / 1e9, 2), "write_GB_per_step": round(wb * 1024 / steps / 1e9, 2)}
    rec["hbm"]["total_GB_per_step"] = round(rec["hbm"]["read_GB_per_step"] + rec["hbm"]["write_GB_per_step"], 2)
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tree_hash import tree_hash
    rec["tree_hash"] = tree_hash()                      # the sources this was measured on (tools/tree_hash.py); bench.py checks it
    rec["commit"] = os.environ.get("VG_COMMIT") or None  # the commit of that tree, handed in by tools/profile_round.sh's caller
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
