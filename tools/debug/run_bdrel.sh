#!/bin/bash
# first layer's d_relation from the boundary nodes' out-edges: parity tests, then the training steps with / without it
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/bdrel; rm -rf "$out"; mkdir -p "$out"
timeout 900 python -m pytest tests/test_rspmm_gpu.py -x -q -k "activity or active" > "$out/tests.txt" 2>&1; echo "tests rc $?" >> "$out/tests.txt"
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_reference_definition_gpu.py -x -q >> "$out/tests.txt" 2>&1; echo "tests rc $?" >> "$out/tests.txt"
for wl in S-wn18rr S-fb15k237; do
  for v in 1 0; do
    ULTRA_BOUNDARY_DRELATION=$v timeout 300 python tools/train_bench.py --workload $wl --graphed --steps 40 2>&1 | tail -2 | sed "s/^/$wl boundary_drel=$v: /" >> "$out/times.txt"
  done
done
tail -5 "$out/tests.txt"; cat "$out/times.txt"
