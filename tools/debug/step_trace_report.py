import csv, glob, sys
from collections import OrderedDict
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last 20 replays: find the period by the score_kernel occurrences
idx = [i for i, r in enumerate(rows) if "score_kernel" in r["Kernel_Name"]]
last = idx[-6:-1]
a, b = last[0], last[1]
step = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
print("kernels per step:", len(step), " wall %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))
busy = 0
prev_end = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("%7.1f us  gap %5.1f  %s" % ((e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]))
    prev_end = e
print("busy %.1f us" % (busy / 1e3))
