#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs of bench.py: per counter, the mean over the last launches of
the search kernel (steady state).  Usage: python tests/pmc_summary.py OUT.json [--launches-per-search N] DIR [DIR ...]
where every DIR holds a *_counter_collection.csv of one pass.  The figures are per KERNEL LAUNCH; a search of
a move may be several launches (N, recorded as "_kernel_launches_per_search")."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    per_search = 1
    if dirs and dirs[0] == "--launches-per-search":
        per_search, dirs = int(dirs[1]), dirs[2:]
    res = {"_kernel_launches_per_search": per_search}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            per = defaultdict(lambda: defaultdict(float))          # counter -> dispatch -> value
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    if "search_" not in row["Kernel_Name"]:            # search_kernel*, search_dual_kernel
                        continue
                    per[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
            for name, by_dispatch in per.items():
                ids = sorted(by_dispatch)
                last = ids[-4:]
                res[name] = {"launches": len(ids), "mean_of_last_4": sum(by_dispatch[i] for i in last) / len(last)}
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
