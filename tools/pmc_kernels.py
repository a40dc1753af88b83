#!/usr/bin/env python3
"""MfmaUtil and VALU share per kernel from one rocprofv3 PMC pass (any workload):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d OUT -- python3 tools/chain_bench.py
    python tools/pmc_kernels.py OUT
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs), cycles = GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs), as tools/pmc_summary.py."""
import collections
import csv
import glob
import sys


def main(d):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for disp, v in per.items():
        k = names[disp]
        if not k.startswith(("vg_", "_Z")):
            continue
        cnt[k] += 1
        for c, x in v.items():
            agg[k][c] += x
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc <= 0:
            continue
        line = f"{k:62s} n={cnt[k]:3d} cycles/launch {cyc / cnt[k]:9.0f}"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            line += f"  MfmaUtil {100.0 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):5.1f} %"
        for c in sorted(v):
            if c not in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"):
                line += f"  {c} {v[c] / cnt[k]:.3e}"
        print(line)


if __name__ == "__main__":
    main(sys.argv[1])
