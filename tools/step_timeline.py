#!/usr/bin/env python3
"""One step of bench.py as a launch-by-launch timeline, from a rocprofv3 --kernel-trace csv:
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
    python tools/step_timeline.py OUT > profiles/rNN_step_timeline.txt
Prints every dispatch of the last complete step (between two vg_step_inputs launches): start offset, duration, gap to the next
dispatch, kernel, grid; then the totals (sum of durations, sum of gaps, launches) and the per-kernel sums of that step."""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    n = re.sub(r"at::native::(\(anonymous namespace\)::)?", "at::", n)
    return n[:72]


def main(d):
    files = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=lambda f: -len(open(f).readlines()))
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "vg_step_inputs" in r["Kernel_Name"]]
    if len(idx) < 3:
        raise SystemExit("fewer than three steps in the trace")
    step = rows[idx[-2]:idx[-1]]
    t0 = int(step[0]["Start_Timestamp"])
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in step]
    gap = [(int(step[i + 1]["Start_Timestamp"]) - int(step[i]["End_Timestamp"])) / 1e3 for i in range(len(step) - 1)] + [0.0]
    print(f"# {files[0]}: last complete step, {len(step)} dispatches")
    print("#   i   start_us  dur_us  gap_us  kernel  [workgroups]")
    per = defaultdict(lambda: [0, 0.0])
    for i, r in enumerate(step):
        wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
        print(f"{i:5d} {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {dur[i]:7.1f} {gap[i]:7.1f}  {short(r['Kernel_Name'])}  [{wg}]")
        k = per[short(r["Kernel_Name"])]
        k[0] += 1
        k[1] += dur[i]
    span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
    print(f"# span {span:.1f} us; sum of durations {sum(dur):.1f} us; sum of gaps {sum(gap):.1f} us (negative = overlap); "
          f"dispatches under 6 us: {sum(1 for x in dur if x < 6.0)} ({sum(x for x in dur if x < 6.0):.1f} us)")
    print("# per kernel (this step): launches, us")
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"#  {n:4d} {t:8.1f}  {k}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
