#!/usr/bin/env python3
"""Mean per-dispatch value of every counter in rocprofv3 --pmc CSV outputs under the given directories."""
import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/*/*_counter_collection.csv")):
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            by[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in by.items():
            print("%-62s %-22s n=%-5d mean=%.1f" % (k[0], k[1], len(v), sum(v) / len(v)))
