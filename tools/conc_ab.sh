#!/bin/bash
# A/B of the headline kernel's column-tile concurrency (quad.inc teams: 1 / 2 / 4 tiles of an XCD label side by side) on
# S-fb15k237 at the fused launch width F = 2 048: kernel duration (HIP events, no profiler), the whole evaluation step
# (bench.py, hipGraph replay), and the L2 / fabric counters of the kernel from separate rocprofv3 --pmc passes.
# ULTRA_CONC forces the team count; ULTRA_CONC_MIN_ROWS=4096 leaves the relation graphs (474 rows, x in LDS) on their default.
# usage (gpurun): bash tools/conc_ab.sh r04   ->  gpurun_out/r04_conc/r04_l2_conc_ab.json (copy into profiles/)
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(command -v python3)")     # the ELF interpreter itself: no shim may exec after the profiler's preload
if ! head -c 4 "$PY" | grep -q ELF; then echo "python3 resolves to $PY, which is not an ELF binary" >&2; exit 1; fi
out=gpurun_out/${tag}_conc
rm -rf "$out"; mkdir -p "$out"
export ULTRA_CONC_MIN_ROWS=4096
for c in 1 2 4; do
  export ULTRA_CONC=$c
  "$PY" tools/kbench.py --workload S-fb15k237 --batch 32 --reps 60 --boundary > "$out/kbench_c$c.txt" 2>&1
  "$PY" bench.py --steps 200 --warmup 20 --no-stress --no-configs --no-cpu-baseline --mrr-queries 0 > "$out/bench_c$c.json" 2> "$out/bench_c$c.err"
  for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_c${c}_$name" -- "$PY" tools/kbench.py --workload S-fb15k237 --batch 32 --reps 4 --boundary > "$out/pmc_c${c}_$name.log" 2>&1
  done
done
unset ULTRA_CONC ULTRA_CONC_MIN_ROWS
"$PY" - "$out" "$tag" <<'PY'
import csv, glob, json, re, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
rows = []
for c in (1, 2, 4):
    line = [l for l in open("%s/kbench_c%d.txt" % (out, c)) if "median" in l][-1]
    us = float(re.search(r"median ([0-9.]+) us", line).group(1))
    bench = json.loads([l for l in open("%s/bench_c%d.json" % (out, c)) if l.startswith("{")][-1])
    vals = defaultdict(list)
    for f in sorted(glob.glob("%s/pmc_c%d_*/**/*_counter_collection.csv" % (out, c), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "quad_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v[-2:]) / len(v[-2:]) for k, v in vals.items()}
    fetch = 2 * m.get("FETCH_SIZE", float("nan")) * 1024            # gfx950: 128-B requests counted at 64 B (MI355X_MICROARCH.md)
    hit, miss = m.get("TCC_HIT_sum", float("nan")), m.get("TCC_MISS_sum", float("nan"))
    rows.append({"tiles_side_by_side_per_label": c, "kernel_us": us, "step_ms": bench["ms_per_step"],
                 "step_block_median_ms": bench["config"]["ms_per_step_block_median"],
                 "fetch_bytes_beyond_l2_per_launch": fetch, "x_compulsory_249MB": fetch / 249e6,
                 "fabric_read_TBps": fetch / (us * 1e-6) / 1e12, "tcc_hit_rate": hit / (hit + miss),
                 "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "TCC_REQ_sum": m.get("TCC_REQ_sum"),
                 "TCC_EA0_RDREQ_sum": m.get("TCC_EA0_RDREQ_sum")})
json.dump({"what": "quad_kernel<FWD,add,mul> on S-fb15k237, F = 2 048 (32 column tiles, 4 per XCD label), sparse boundary "
                   "epilogue; ULTRA_CONC = tiles of a label worked on at the same time (default heuristic: 4)",
           "source": "tools/conc_ab.sh: kernel_us = median of 60 HIP-event timed launches (no profiler); step_ms = bench.py "
                     "--steps 200 (hipGraph replay); counters = mean of the last 2 dispatches under rocprofv3 --pmc, one "
                     "counter set per pass",
           "settings": rows}, open("%s/%s_l2_conc_ab.json" % (out, tag), "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
