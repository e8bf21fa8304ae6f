"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (dev tool).  usage: pmc_summary.py dir..."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "lgar_forward" not in k and "lgar_tangent" not in k:
                continue
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(d, k)
            for c, v in sorted(cs.items()):
                v = sorted(v)
                print("   %-28s n=%d median=%.6g" % (c, len(v), v[len(v) // 2]))
