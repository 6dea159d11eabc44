cd "$GRAFT_REPO_ROOT"
for v in 0 1 2 4 7 16 96 103 119; do ULTRA_RSPMM_LIB=$PWD/gpurun_variants/libfb_skip$v.so python tools/debug/fb_time.py 2>&1 | grep rows; done
