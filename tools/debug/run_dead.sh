#!/bin/bash
# removed edges as marked words (quad.inc DEAD): parity tests, then the training steps with / without
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/dead; rm -rf "$out"; mkdir -p "$out"
timeout 900 python -m pytest tests/test_rspmm_gpu.py tests/test_frontier_sampler_gpu.py -x -q -k "marked or removal or activity or graphed_train" > "$out/tests.txt" 2>&1; echo "tests rc $?" >> "$out/tests.txt"
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_reference_definition_gpu.py tests/test_configs_gpu.py -x -q >> "$out/tests.txt" 2>&1; echo "tests rc $?" >> "$out/tests.txt"
for wl in S-wn18rr S-fb15k237; do
  for v in 1 0; do
    ULTRA_DEAD_EDGE_WORDS=$v timeout 300 python tools/train_bench.py --workload $wl --graphed --steps 40 2>&1 | tail -1 | sed "s/^/$wl dead_words=$v: /" >> "$out/times.txt"
  done
done
for wl in S-fb15k237 S-codexm; do
  for v in 1 0; do
    ULTRA_DEAD_EDGE_WORDS=$v timeout 300 python tools/train_bench.py --workload $wl --graphed --steps 20 --batch 64 2>&1 | tail -1 | sed "s/^/$wl B=64 dead_words=$v: /" >> "$out/times.txt"
  done
done
grep -h "passed\|failed\|rc" "$out/tests.txt" | tail -6; cat "$out/times.txt"
