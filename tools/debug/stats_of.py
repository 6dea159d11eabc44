import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in sys.argv[2:]):
            print("   %-60s calls %4s avg %8.1f us" % (r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
