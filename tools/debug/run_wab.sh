#!/bin/bash
# what per-edge weights (the training step's edge removal) cost the rspmm kernels: unit vs weighted, forward and backward
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/wab; rm -rf "$out"; mkdir -p "$out"
for wl in S-fb15k237 S-wn18rr; do
  for b in 16 64; do
    for w in "" "--weights"; do
      timeout 200 python tools/kbench.py --workload $wl --batch $b --reps 30 $w 2>&1 | tail -1 | sed "s/^/[$w] /" >> "$out/wab.txt"
      timeout 200 python tools/kbench.py --workload $wl --batch $b --reps 30 --backward $w 2>&1 | tail -1 | sed "s/^/[$w] /" >> "$out/wab.txt"
    done
  done
done
cat "$out/wab.txt"
