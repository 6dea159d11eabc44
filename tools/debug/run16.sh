cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_frontier_sampler_gpu.py tests/test_torch_ext_gpu.py tests/test_configs_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/t16.log 2>&1; grep -E "passed|failed" gpurun_out/t16.log; grep -E "^FAILED" gpurun_out/t16.log
python tools/stress_bench.py --knob 8
python tools/stress_bench.py --knob 0
for B in 768 1024; do echo "block $B"; ULTRA_BINDING=ctypes ULTRA_RSPMM_LIB=$GRAFT_REPO_ROOT/gpurun_variants/libultra_rspmm_rg$B.so python tools/stress_bench.py --knob 0; done
python tools/stress_bench.py --knob 0 --batch 4
python tools/stress_bench.py --knob 0 --batch 2
