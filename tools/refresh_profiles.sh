#!/bin/bash
# Regenerates the judged profile artifacts on the GPU box (writes under gpurun_out/, copy into profiles/ afterwards):
#   bench.py under rocprofv3 --kernel-trace --stats (csv), its JSON line, and the per-kernel summary.
# usage (gpurun): bash tools/refresh_profiles.sh [round-tag]
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(command -v python3)")     # the ELF interpreter itself: no shim may exec after the profiler's preload
if ! head -c 4 "$PY" | grep -q ELF; then echo "python3 resolves to $PY, which is not an ELF binary" >&2; exit 1; fi
out=gpurun_out/refresh
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench" -o bench -- "$PY" bench.py --steps 200 --warmup 20 > "$out/bench_under_rocprof.log" 2>&1
grep "^{\"metric\"" "$out/bench_under_rocprof.log" | tail -1 > "$out/${tag}_bench_under_rocprof.json"
find "$out/bench" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_bench_kernel_stats.csv" \;
"$PY" - "$out" "$tag" <<'PY'
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open("%s/%s_bench_kernel_stats.csv" % (out, tag))))
ours = {r["Name"]: {"launches": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "percent": float(r["Percentage"])}
        for r in rows if "anonymous namespace" in r["Name"] and "at::native" not in r["Name"]}
json.dump({"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 200 --warmup 20",
           "kernels": ours}, open("%s/%s_bench_dominant_kernel.json" % (out, tag), "w"), indent=1)
PY
ls "$out"
