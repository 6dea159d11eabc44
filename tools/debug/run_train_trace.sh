#!/bin/bash
# one fine-tuning step kernel by kernel (tools/debug/train_trace_report.py): usage tools/debug/run_train_trace.sh S-wn18rr
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(command -v python3)")
wl=${1:-S-wn18rr}
out=gpurun_out/train_trace; rm -rf "$out"; mkdir -p "$out"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$out/t" -o tr -- "$PY" tools/train_bench.py --workload $wl --graphed --steps 6 ${BATCH:+--batch $BATCH} > "$out/log.txt" 2>&1
d=$(dirname "$(find "$out/t" -name "*kernel_trace.csv" | tail -1)")
"$PY" tools/debug/train_trace_report.py "$d" > "$out/train_trace_$wl.txt" 2>&1
rm -rf "$out/t"
