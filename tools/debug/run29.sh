cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_rspmm_gpu.py -m gpu -q -p no:cacheprovider -k "accumulates or backward" > gpurun_out/t29.log 2>&1; grep -E "passed|failed" gpurun_out/t29.log; grep -E "^FAILED|Error" gpurun_out/t29.log | head
python tools/train_bench.py --workload S-wn18rr --steps 30 --graphed 2>&1 | tail -1
