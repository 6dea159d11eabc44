#!/bin/bash
# Regenerates the judged profile artifacts on the GPU box (writes under gpurun_out/, copy into profiles/ afterwards):
#   bench.py under rocprofv3 --kernel-trace --stats (csv), its JSON line, and the per-kernel summary.
# usage (gpurun): bash tools/refresh_profiles.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/refresh
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench" -o bench -- python3 bench.py --steps 20 --warmup 5 > "$out/bench_under_rocprof.log" 2>&1
grep "^{\"metric\"" "$out/bench_under_rocprof.log" | tail -1 > "$out/bench_under_rocprof.json"
find "$out/bench" -name "*kernel_stats.csv" -exec cp {} "$out/bench_kernel_stats.csv" \;
python3 - "$out" <<'PY'
import csv, json, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/bench_kernel_stats.csv")))
ours = {r["Name"]: {"launches": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "percent": float(r["Percentage"])}
        for r in rows if "anonymous namespace" in r["Name"] and "at::native" not in r["Name"]}
json.dump({"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5",
           "kernels": ours}, open(out + "/bench_dominant_kernel.json", "w"), indent=1)
PY
ls "$out"
