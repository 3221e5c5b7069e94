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
    spans = []
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            spans.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
            name = re.sub(r"^void ", "", row["Kernel_Name"]).replace("dfot::", "").replace("(anonymous namespace)::", "").split("(")[0]
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
    # timeline of the busiest second half of the run (the timed steps): busy time, wall span and the gaps between kernels
    spans.sort()
    half = spans[len(spans) // 2:]
    busy, gaps, end = 0, [], half[0][0]
    for a, b in half:
        if a > end:
            gaps.append(a - end)
        busy += max(0, b - max(a, end))
        end = max(end, b)
    span = end - half[0][0]
    big = [x for x in gaps if x > 20000]
    summary = ("timeline (second half of the dispatches): %d kernels, span %.2f ms, busy %.2f ms (%.1f %%), %d gaps avg %.2f us, "
               "%d gaps > 20 us totalling %.2f ms" % (len(half), span / 1e6, busy / 1e6, 100.0 * busy / span, len(gaps),
                                                     (sum(gaps) / max(len(gaps), 1)) / 1e3, len(big), sum(big) / 1e6))
    print(summary)
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write("# " + summary + "\n")


if __name__ == "__main__":
    main()
