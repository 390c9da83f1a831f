#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: per-kernel launch count / mean duration (kernel trace) and per-kernel mean of every
PMC counter (counter collection). Usage: prof_summarize.py <dir> [<dir> ...]  -> prints a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0].split("<")[0][-60:]


def main():
    for d in sys.argv[1:]:
        for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
            agg = defaultdict(list)
            with open(path) as fh:
                for r in csv.DictReader(fh):
                    agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            print("## kernel trace:", os.path.relpath(path, d))
            print("%-62s %8s %14s %14s" % ("kernel", "calls", "mean_us", "total_ms"))
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                print("%-62s %8d %14.2f %14.3f" % (k, len(v), sum(v) / len(v), sum(v) / 1e3))
        for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            agg = defaultdict(lambda: defaultdict(list))
            with open(path) as fh:
                for r in csv.DictReader(fh):
                    agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            print("## counters:", os.path.relpath(path, d))
            for k, cs in sorted(agg.items()):
                for c, v in sorted(cs.items()):
                    print("%-62s %-28s n=%-6d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
