#!/usr/bin/env python3
"""Per-(kernel, grid size) summary of a rocprofv3 --kernel-trace CSV: the same kernel template serves several GEMM shapes of
the model, which --stats folds into one line.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py ... ; python tools/kernel_trace_by_shape.py DIR [out.csv]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = re.sub(r"^void ", "", row["Kernel_Name"]).split("(")[0].replace("dfot::", "").replace("(anonymous namespace)::", "")
            grid = int(row["Grid_Size_X"]) // max(int(row["Workgroup_Size_X"]), 1)
            a = acc[(name[:100], grid, int(row["Workgroup_Size_X"]))]
            a[0] += 1
            a[1] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    tot = sum(v[1] for v in acc.values())
    lines = ["name,workgroups,threads,calls,total_us,avg_us,percent"]
    for (name, grid, wg), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        lines.append('"%s",%d,%d,%d,%.1f,%.2f,%.2f' % (name, grid, wg, n, t / 1e3, t / n / 1e3, 100.0 * t / tot))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print("\n".join(l[:220] for l in lines[:45]))
    print("total ms", tot / 1e6)


if __name__ == "__main__":
    main()
