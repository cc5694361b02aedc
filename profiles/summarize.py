"""Summarise a profiles/collect.sh output directory: per-kernel time from the kernel trace and
per-kernel counter sums from every PMC pass (averaged per launch)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def main(d):
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel trace (--stats):", os.path.relpath(f, d))
        rows = list(csv.DictReader(open(f)))
        for r in rows[:12]:
            print("  %-60s calls %4s  avg %10.1f us  total %9.3f ms  %5s%%" % (
                short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"][:5]))
    for p in sorted(glob.glob(os.path.join(d, "pmc*"))):
        if not os.path.isdir(p):
            continue
        files = glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(set)
        for f in files:
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k].add(r["Dispatch_Id"])
        print("== counters per launch:", os.path.basename(p))
        for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
            if not any(t in k for t in ("k_bounce", "k_resolve", "k_trace", "k_shade", "k_shadow_gen", "k_stream")):
                continue
            n = max(len(calls[k]), 1)
            print("  %-46s launches %3d  " % (k, n) + "  ".join("%s=%.4g" % (c, v / n) for c, v in sorted(acc[k].items())))


if __name__ == "__main__":
    main(sys.argv[1])
