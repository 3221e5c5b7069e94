#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 PMC counters (one or more passes) + the MFMA utilisation derived from them.

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES \
            SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_sq -- python bench.py ...
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_grbm -- python bench.py ...
  python tools/pmc_counters.py "note" gpurun_out/pmc_sq gpurun_out/pmc_grbm > profiles/rNN_pmc_mfma_util.json

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8
XCDs, MI355X_MICROARCH.md "DVFS give-back"); SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 bf16 MFMA).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main():
    note, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                name = re.sub(r"^void ", "", row["Kernel_Name"]).replace("dfot::", "").replace("(anonymous namespace)::", "").split("(")[0]
                if "at::native" in name or name.startswith("__amd"):
                    continue
                a = acc[name][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    out = {"note": note, "kernels": {}}
    for name, counters in sorted(acc.items()):
        k = {c: round(v[0] / v[1], 1) for c, v in counters.items()}
        k["launches"] = max(v[1] for v in counters.values())
        if k.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in k:
            k["mfma_util"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * k["GRBM_GUI_ACTIVE"]), 4)
        if k.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            k["mfma_valu_coexec_frac_of_mfma_busy"] = round(k.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / k["SQ_VALU_MFMA_BUSY_CYCLES"], 4)
        out["kernels"][name] = k
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
