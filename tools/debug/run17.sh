cd "$GRAFT_REPO_ROOT"
timeout 120 ./tools/variants/gather_big
python tools/stress_bench.py --knob 0
for V in B768 B1024 E1 E2 E3 E4; do echo "variant $V"; ULTRA_BINDING=ctypes ULTRA_RSPMM_LIB=$GRAFT_REPO_ROOT/tools/variants/libultra_rspmm_$V.so python tools/stress_bench.py --knob 0 2>&1 | tail -1; done
python tools/stress_bench.py --knob 8
