#!/usr/bin/env python3
"""Mean counter values per kernel and launch from rocprofv3 --pmc runs.

    python tools/pmc_summary.py <dir with *_counter_collection.csv> [more dirs ...]

Prints kernel,counter,mean_per_launch,launches (the layout of profiles/*_pmc_summary.csv).
"""
import collections
import csv
import glob
import os
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, counter) -> dispatch -> value
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    key = (row["Kernel_Name"], row["Counter_Name"])
                    acc[key][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    print("kernel,counter,mean_per_launch,launches")
    for (kern, ctr), per in sorted(acc.items()):
        vals = list(per.values())
        print('"%s",%s,%.1f,%d' % (kern, ctr, sum(vals) / len(vals), len(vals)))


if __name__ == "__main__":
    main()
