cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/t28.log 2>&1; grep -E "passed|failed" gpurun_out/t28.log; grep -E "^FAILED" gpurun_out/t28.log
python tools/train_bench.py --workload S-fb15k237 --steps 30 --graphed 2>&1 | tail -1
python tools/train_bench.py --workload S-wn18rr --steps 30 --graphed 2>&1 | tail -1
python tools/pretrain_bench.py --steps 12 2>&1 | tail -1
