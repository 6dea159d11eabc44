cd "$GRAFT_REPO_ROOT"
echo "== L2-resident table"; timeout 120 ./tools/variants/gather_l2 2>&1 | grep -v "half rows"
echo "== DRAM-resident table"; timeout 120 ./tools/variants/gather_big 2>&1 | grep "policy"
python -m pytest tests/test_frontier_sampler_gpu.py tests/test_torch_ext_gpu.py tests/test_configs_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/t21.log 2>&1; grep -E "passed|failed" gpurun_out/t21.log; grep -E "^FAILED" gpurun_out/t21.log
python tools/stress_bench.py --knob 0
python tools/stress_bench.py --knob 8
python tools/stress_bench.py --knob 0 --batch 2
python tools/stress_bench.py --knob 0 --batch 4
