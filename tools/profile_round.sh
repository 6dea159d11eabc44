#!/bin/bash
# Regenerates the judged profile artefacts of a round on the GPU box, under gpurun_out/<tag>/ (copy into profiles/):
#   <tag>_bench_under_rocprof.json      bench.py's JSON line, run under rocprofv3 --kernel-trace --stats
#   <tag>_bench_kernel_stats.csv        rocprofv3's per-kernel summary of that command
#   <tag>_bench_dominant_kernel.json    the library's kernels out of it (launches, average us, share)
#   <tag>_traffic_stress.json           S-stress rowgroup kernel: HBM bytes per launch from separate --pmc passes
#   <tag>_traffic_fwd_fb15k237.json     headline graph's forward kernel (F = 2048): the same
#   <tag>_stress_rowgroup_pmc.txt / <tag>_kbench_fwd_pmc.txt   the counter means those come from
#   <tag>_train_{wn18rr,fb15k237}_kernel_stats.csv + _step.txt   fine-tuning steps (hipGraph replays): kernel summary, ms/step
#   <tag>_step_trace_{fb15k237,codexs}.txt   one evaluation batch, kernel by kernel (rocprofv3 --kernel-trace)
#   <tag>_stress_predict_kernel_stats.csv + _b1.json / _b4.json   config 5 as inference: whole predict on S-stress (round 6)
#   <tag>_pmc_layer_fused_raw.txt        PMC A/B of one S-stress layer, fused launch vs two launches (tools/pmc_layer_fused.sh)
# usage (gpurun): bash tools/profile_round.sh r03
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(command -v python3)")     # the ELF interpreter itself: no shim may exec after the profiler's preload
if ! head -c 4 "$PY" | grep -q ELF; then echo "python3 resolves to $PY, which is not an ELF binary" >&2; exit 1; fi
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench" -o bench -- "$PY" bench.py --steps 200 --warmup 20 > "$out/bench_under_rocprof.log" 2>&1
grep "^{\"metric\"" "$out/bench_under_rocprof.log" | tail -1 > "$out/${tag}_bench_under_rocprof.json"
find "$out/bench" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_bench_kernel_stats.csv" \;
rm -rf "$out/bench"
# fine-tuning steps (hipGraph replays) of the two fine-tuning shapes: per-kernel summary
for wl in wn18rr fb15k237; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train_$wl" -o train -- "$PY" tools/train_bench.py --workload S-$wl --graphed --steps 20 > "$out/train_$wl.log" 2>&1
  find "$out/train_$wl" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_train_${wl}_kernel_stats.csv" \;
  grep -h "ms/step" "$out/train_$wl.log" > "$out/${tag}_train_${wl}_step.txt"
  rm -rf "$out/train_$wl"
done
# one evaluation batch replayed as a hipGraph, kernel by kernel (headline graph and the small config-2 graph)
for wl in fb15k237 codexs; do
  WORKLOAD=S-$wl timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/step_$wl" -o st -- "$PY" tools/debug/step_trace.py > "$out/step_$wl.log" 2>&1
  d=$(dirname "$(find "$out/step_$wl" -name "*kernel_trace.csv" | tail -1)")
  "$PY" tools/debug/step_trace_report.py "$d" > "$out/${tag}_step_trace_${wl}.txt" 2>&1
  rm -rf "$out/step_$wl"
done
# BASELINE config 5 as inference (round 6): the whole predict on S-stress, kernel by kernel; and the PMC A/B of one layer, fused vs two launches
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stress_predict" -o sp -- "$PY" tools/stress_predict.py --batch 1 --reps 3 --json "$out/${tag}_stress_predict_b1.json" > "$out/stress_predict.log" 2>&1
find "$out/stress_predict" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_stress_predict_kernel_stats.csv" \;
rm -rf "$out/stress_predict"
timeout 300 "$PY" tools/stress_predict.py --batch 4 --reps 3 --json "$out/${tag}_stress_predict_b4.json" > "$out/stress_predict_b4.log" 2>&1
bash tools/pmc_layer_fused.sh "$out/pmc_layer" 2 > "$out/${tag}_pmc_layer_fused_raw.txt" 2>&1
rm -rf "$out/pmc_layer"
"$PY" - "$out/${tag}_pmc_layer_fused_raw.txt" "$out/${tag}_traffic_layer_fused.json" <<'PY'
import json, re, sys
t = open(sys.argv[1]).read()
def get(kernel, counter):
    m = re.search(r"%s\s+%s\s+([0-9.]+)" % (kernel, counter), t)
    return float(m.group(1)) if m else float("nan")
fused = 2 * get("rowgroup_layer_kernel", "FETCH_SIZE") * 1024 + get("rowgroup_layer_kernel", "WRITE_SIZE") * 1024
split = 2 * (get("rowgroup_kernel", "FETCH_SIZE") + get("combine_kernel", "FETCH_SIZE")) * 1024 \
    + (get("rowgroup_kernel", "WRITE_SIZE") + get("combine_kernel", "WRITE_SIZE")) * 1024
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_layer_fused.sh, 2 queries) -- python3 "
                     "tools/layer_bench.py --queries 2; rowgroup_layer_kernel vs rowgroup_kernel + combine_kernel",
           "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
           "hbm_bytes_per_launch": int(fused), "two_launch_hbm_bytes_per_layer": int(split),
           "algorithmic_bytes_per_launch": 62680512004, "two_launch_algorithmic_bytes_per_layer": 72920512004, "F": 128},
          open(sys.argv[2], "w"), indent=1)
PY
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d "$out/stress_$name" -- "$PY" tools/stress_bench.py --reps 2 --knob 0 > "$out/stress_$name.log" 2>&1
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d "$out/fwd_$name" -- "$PY" tools/kbench.py --workload S-fb15k237 --batch 32 --reps 4 > "$out/fwd_$name.log" 2>&1
done
"$PY" - "$out" "$tag" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open("%s/%s_bench_kernel_stats.csv" % (out, tag))))
ours = {r["Name"]: {"launches": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "percent": float(r["Percentage"])}
        for r in rows if "anonymous namespace" in r["Name"] and "at::native" not in r["Name"]}
json.dump({"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 200 --warmup 20",
           "kernels": ours}, open("%s/%s_bench_dominant_kernel.json" % (out, tag), "w"), indent=1)


def means(prefix, pattern):
    vals = defaultdict(list)
    for f in sorted(glob.glob("%s/%s_*/**/*_counter_collection.csv" % (out, prefix), recursive=True)):
        for r in csv.DictReader(open(f)):
            if pattern in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v[-2:]) / len(v[-2:]) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


for prefix, pattern, name, algo, cmd, txt in (
        ("stress", "rowgroup_kernel", "traffic_stress", 29400256004, "python3 tools/stress_bench.py --reps 2 --knob 0", "stress_rowgroup_pmc"),
        ("fwd", "quad_kernel", "traffic_fwd_fb15k237", 4587923968, "python3 tools/kbench.py --workload S-fb15k237 --batch 32 --reps 4", "kbench_fwd_pmc")):
    m, n = means(prefix, pattern)
    with open("%s/%s_%s.txt" % (out, tag, txt), "w") as f:
        f.write("# rocprofv3 --pmc <one counter set per pass> -- %s ; mean of the last 2 dispatches of *%s*\n" % (cmd, pattern))
        for k in sorted(m):
            f.write("%-28s %18.1f  (dispatches seen: %d)\n" % (k, m[k], n[k]))
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        hbm = int(2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024)
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) -- %s; kernel *%s*" % (cmd, pattern),
                   "FETCH_SIZE_KB_per_launch": m["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": m["WRITE_SIZE"],
                   "TCC_HIT_sum": m.get("TCC_HIT_sum"), "TCC_MISS_sum": m.get("TCC_MISS_sum"),
                   "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
                   "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": algo, "F": 64 if prefix == "stress" else 2048},
                  open("%s/%s_%s.json" % (out, tag, name), "w"), indent=1)
PY
rm -rf "$out"/stress_* "$out"/fwd_*
ls -la "$out"
"$PY" tools/configs_table.py "$out/${tag}_bench_under_rocprof.json" "$out/${tag}_configs.md"
