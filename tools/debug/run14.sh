cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/t14.log 2>&1; grep -E "passed|failed" gpurun_out/t14.log
for b in 1 2 4; do python tools/stress_bench.py --batch $b 2>&1 | grep S-stress; python tools/stress_bench.py --batch $b --knob 8 2>&1 | grep S-stress; done | tee gpurun_out/stress3.log
