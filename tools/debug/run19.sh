cd "$GRAFT_REPO_ROOT"
python tools/stress_bench.py --knob 0
for V in P; do echo "variant $V"; ULTRA_BINDING=ctypes ULTRA_RSPMM_LIB=$GRAFT_REPO_ROOT/tools/variants/libultra_rspmm_$V.so python tools/stress_bench.py --knob 0 2>&1 | tail -1; done
python tools/stress_bench.py --knob 8
python tools/stress_bench.py --knob 0
