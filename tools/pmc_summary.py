#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 counter_collection.csv (one counter per pass):
kernel, counter, launches, mean value per launch (library kernels only)."""
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "uzk::" not in name: continue
    key = (name.split("(")[0].replace("void ", ""), r["Counter_Name"])
    a = acc.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["Kernel", "Counter", "Launches", "Mean_Value_Per_Launch"])
for (k, c), (n, s) in acc.items(): w.writerow([k, c, n, f"{s / n:.3f}"])
