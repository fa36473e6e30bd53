#!/usr/bin/env python
"""Median per (kernel, counter) of rocprofv3 `--pmc ... --kernel-trace --output-format csv` output directories.
Usage: python tools/pmc_summary.py <kernel substring> <dir> [<dir> ...]   (prints a markdown table)"""
import csv
import glob
import os
import statistics
import sys


def main():
    sub, dirs = sys.argv[1], sys.argv[2:]
    for d in dirs:
        vals, dur = {}, []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if sub in r["Kernel_Name"]:
                    vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if sub in r["Kernel_Name"]:
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        print(f"## {d}  (kernel ~ '{sub}', {len(dur)} launches, median duration {statistics.median(dur) if dur else float('nan'):.3f} ms)")
        print("| counter | median per launch | launches |\n|---|---|---|")
        for k in sorted(vals):
            print(f"| {k} | {statistics.median(vals[k]):.4e} | {len(vals[k])} |")
        print()


if __name__ == "__main__":
    main()
