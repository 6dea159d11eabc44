"""Summarise tools/pmc.sh output: per kernel name, mean counter value per dispatch (last N dispatches)."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "packed_kernel"
vals = defaultdict(list)
for f in sorted(glob.glob(out + "/pass*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in vals.items():
    v = v[-5:]
    print("%-32s %16.1f  (n=%d)" % (k, sum(v) / len(v), len(v)))
