#!/usr/bin/env python3
"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately, no trace domains) into per-kernel HBM
bytes per launch:  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950 correction, MI355X_MICROARCH.md HBM section).

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write "note text" > profiles/rNN_pmc_....json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(directory: str, counter: str):
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"^void ", "", row["Kernel_Name"])
            name = name.split("(")[0].replace("dfot::", "").replace("(anonymous namespace)::", "")
            a = acc[name]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"note": sys.argv[3] if len(sys.argv) > 3 else "", "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, [0.0, 0]), write.get(name, [0.0, 0])
        n = max(f[1], w[1])
        if n == 0 or "at::native" in name or name.startswith("__amd"):
            continue
        fa, wa = f[0] / max(f[1], 1), w[0] / max(w[1], 1)
        out["kernels"][name] = {"launches": n, "FETCH_SIZE_KiB_avg": round(fa, 2), "WRITE_SIZE_KiB_avg": round(wa, 2),
                                "hbm_bytes_per_launch_corrected": round((2 * fa + wa) * 1024, 1)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
