cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_rspmm_gpu.py tests/test_model_gpu.py tests/test_reference_definition_gpu.py -m gpu -q -p no:cacheprovider -x > gpurun_out/t27.log 2>&1; grep -E "passed|failed" gpurun_out/t27.log; grep -E "^FAILED" gpurun_out/t27.log
python tools/train_bench.py --workload S-fb15k237 --steps 30 --graphed 2>&1 | tail -1
python tools/train_bench.py --workload S-wn18rr --steps 30 --graphed 2>&1 | tail -1
python tools/pretrain_bench.py --steps 12 2>&1 | tail -1
