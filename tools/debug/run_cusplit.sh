#!/bin/bash
# how the two halves of an entity layer scale with the compute units they get: the rspmm kernel and the fused epilogue at 256 / 224 / 192 / 128 / 64 / 32 CUs
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/cusplit; rm -rf "$out"; mkdir -p "$out"
for r in 0 32 64 128 192 224; do
  timeout 200 python tools/kbench.py --workload S-fb15k237 --batch 32 --reps 30 --boundary --reserve $r 2>&1 | tail -1 | sed "s/^/[rspmm, $r CUs reserved] /" >> "$out/t.txt"
  timeout 200 python tools/kbench.py --workload S-fb15k237 --batch 32 --reps 30 --combine --reserve $r 2>&1 | tail -1 | sed "s/^/[epilogue, $r CUs reserved] /" >> "$out/t.txt"
done
cut -c1-160 "$out/t.txt"
