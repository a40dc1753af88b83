#!/usr/bin/env python3
"""Derive HBM bytes per launch of every GEMM shape in tools/gemm_bench.py from two rocprofv3 PMC passes.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -- python3 tools/gemm_bench.py   (REPS=3)
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -- python3 tools/gemm_bench.py
  python tools/pmc_traffic.py A B profiles/r01_gemm_pmc_traffic.json

Counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read stream
(MI355X_MICROARCH.md, HBM / rocprofv3 section), hence bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
The dispatch order of gemm_bench.py is fixed (3 warm-ups + REPS launches per shape, SHAPES in order), so shapes are recovered by
position; only the GEMM kernels (vg_gemm_*) are counted - the slab folds and the weight packing are not.
"""
import csv
import glob
import json
import shutil
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench  # noqa: E402  (the shape list; importing it runs nothing on the GPU)

NAMES = [n for n, _, _ in gemm_bench.SHAPES]
ALGO = {n: (f, b) for n, f, b in gemm_bench.SHAPES}


def per_shape(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "vg_gemm_" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    per = len(rows) // len(NAMES)  # 3 warm-ups + REPS launches per shape, in gemm_bench.py's fixed order
    if per * len(NAMES) != len(rows):
        raise SystemExit(f"{len(rows)} GEMM dispatches do not divide into {len(NAMES)} shapes: did gemm_bench.py change?")
    groups = [[((r["Kernel_Name"], r["Grid_Size"]), float(r["Counter_Value"])) for r in rows[i * per:(i + 1) * per]] for i in range(len(NAMES))]
    return f, [(g[0][0][0].split("(")[0].replace("void ", ""), sum(v for _, v in g[-3:]) / len(g[-3:])) for g in groups]


def main():
    a, b, out = sys.argv[1:4]
    fa, fetch = per_shape(a, "FETCH_SIZE")
    fb, write = per_shape(b, "WRITE_SIZE")
    if len(fetch) != len(NAMES) or len(write) != len(NAMES):
        raise SystemExit(f"expected {len(NAMES)} shapes, found {len(fetch)} / {len(write)}: did gemm_bench.py change?")
    rec = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/gemm_bench.py; HBM bytes per launch = "
                   "(2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads half the bytes of a wide coalesced stream, "
                   "MI355X_MICROARCH.md section HBM); produced by tools/pmc_traffic.py", "kernels": {}}
    for i, name in enumerate(NAMES):
        rec["kernels"][name] = {"kernel": fetch[i][0], "rows_M": gemm_bench.M, "FETCH_SIZE_KB": fetch[i][1],
                                "WRITE_SIZE_KB": write[i][1], "hbm_bytes_per_launch": int((2 * fetch[i][1] + write[i][1]) * 1024),
                                "algorithmic_flops": ALGO[name][0], "algorithmic_bytes": ALGO[name][1]}
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tree_hash import tree_hash
    rec["tree_hash"] = tree_hash()                      # the sources this was measured on (tools/tree_hash.py); bench.py checks it
    rec["commit"] = os.environ.get("VG_COMMIT") or None  # the commit of that tree, handed in by tools/profile_round.sh's caller
    json.dump(rec, open(out, "w"), indent=1)
    base = out.rsplit("_traffic.json", 1)[0]
    shutil.copy(fa, base + "_FETCH_SIZE.csv")
    shutil.copy(fb, base + "_WRITE_SIZE.csv")
    for k, v in rec["kernels"].items():
        print(f"{k:22s} {v['kernel']:34s} {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB")


if __name__ == "__main__":
    main()
