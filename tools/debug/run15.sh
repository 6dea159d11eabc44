cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/t15.log 2>&1; grep -E "passed|failed" gpurun_out/t15.log; grep -E "^FAILED" gpurun_out/t15.log
python bench.py > gpurun_out/bench_r2d.json 2> gpurun_out/bench_r2d.err; tail -c 300 gpurun_out/bench_r2d.err
