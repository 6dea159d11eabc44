import csv, glob, sys
from collections import Counter
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "edge_removal_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
step = rows[a:b]
print("kernels per step:", len(step), " wall %.1f us" % ((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3))
small = [r for r in step if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 9000]
print("kernels < 9 us:", len(small), " total %.1f us" % (sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in small) / 1e3))
prev = None
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = 0.0 if prev is None else (int(r["Start_Timestamp"]) - prev) / 1e3
    prev = int(r["End_Timestamp"])
    print("%7.1f  gap %5.1f  %s" % (d, gap, r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("at::native::", "")[:110]))
