cd "$GRAFT_REPO_ROOT"
python tools/debug/fb_time.py 2>&1 | grep rows
python tools/debug/fb_time.py 232656 2>&1 | grep rows
python -m pytest tests/test_frontier_sampler_gpu.py tests/test_rspmm_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
for w in S-wn18rr S-fb15k237; do python tools/train_bench.py --workload $w --steps 20 --graphed 2>&1 | grep "ms/step"; done
