"""Sums rocprofv3 --pmc CSV output per kernel family and counter.  usage: python tools/pmc_summary.py <pmc_dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def family(name):
    for k in ("k_trace_shadow", "k_trace_batch", "k_trace", "k_shade_post", "k_shade_pre", "k_backward", "k_raygen", "k_resolve"):
        if k in name:
            return k
    return None


def load(pmc_dir):
    out = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for f in glob.glob(os.path.join(pmc_dir, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            fam = family(row["Kernel_Name"])
            if fam is None:
                continue
            out[fam][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[(fam, row["Counter_Name"])].add(row["Dispatch_Id"])
    return out, calls


if __name__ == "__main__":
    out, calls = load(sys.argv[1])
    for fam in sorted(out):
        print("==", fam)
        for c in sorted(out[fam]):
            print("  %-32s %18.0f   (%d dispatches)" % (c, out[fam][c], len(calls[(fam, c)])))
