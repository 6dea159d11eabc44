#!/bin/bash
# PMC A/B of quad_kernel<FWD> with and without the gather dedup: separate passes per counter set and library.
# (a pass with TA_TA_BUSY_sum / TA_BUFFER_* / TCP_TOTAL_* counters never finished on this pool: only the sets of tools/pmc.sh)
# usage (on the GPU box): tools/pmc_dedup.sh gpurun_out/pmc_dedup
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
PY=$(readlink -f "$(command -v python3)")
export ULTRA_BINDING=ctypes
for lib in ${LIBS:-base dedup}; do
  if [ $lib = base ]; then export ULTRA_RSPMM_LIB=$PWD/ultra_torchdrug_amd/libultra_rspmm.so; else export ULTRA_RSPMM_LIB=$PWD/tools/variants/libultra_rspmm_dedup.so; fi
  i=0; mkdir -p "$out/$lib"
  for set in \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
    "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_WAVES" \
    "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" ; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $set --output-format csv -d "$out/$lib/pass$i" -- "$PY" tools/kbench.py --workload S-fb15k237 --batch 32 --boundary --reps 4 > "$out/$lib/pass$i.log" 2>&1
  done
  echo "== $lib"
  python3 tools/pmc_summary.py "$out/$lib" "quad_kernel<0"
done
