#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel-name substring, per dispatch, the counter sums."""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict


def main():
    dirs, pat = sys.argv[1:-1], sys.argv[-1]
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            rows = defaultdict(OrderedDict)          # dispatch id -> counter -> value
            meta = {}
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if pat not in r["Kernel_Name"]:
                        continue
                    k = r["Dispatch_Id"]
                    rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    meta[k] = (r["Kernel_Name"][:70], r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"))
            print("==", f)
            for k in sorted(rows, key=lambda v: int(v)):
                print(k, meta[k], " ".join("%s=%.4g" % kv for kv in rows[k].items()))


if __name__ == "__main__":
    main()
