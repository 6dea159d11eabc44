#!/bin/bash
# A/B: quad_kernel with and without gathering each run of parallel edges once (tools/variants/libultra_rspmm_dedup.so: -DULTRA_QUAD_DEDUP=1)
export ULTRA_BINDING=ctypes
for w in S-fb15k237 S-wn18rr S-codexm; do
  for lib in ultra_torchdrug_amd/libultra_rspmm.so tools/variants/libultra_rspmm_dedup.so; do
    ULTRA_RSPMM_LIB=$PWD/$lib python tools/kbench.py --workload $w --batch 32 --boundary --reps 60 2>&1 | tail -1
    ULTRA_RSPMM_LIB=$PWD/$lib python tools/kbench.py --workload $w --batch 16 --backward --reps 40 2>&1 | tail -1
  done
done
