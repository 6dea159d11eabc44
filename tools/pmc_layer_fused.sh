#!/bin/bash
# PMC A/B of one entity layer on S-stress: fused (rowgroup_layer_kernel) vs the two launches (rowgroup_kernel + combine_kernel).
# FETCH_SIZE / WRITE_SIZE in their own passes (MI355X_MICROARCH.md's HBM recipe); the summary lists bytes per launch per kernel.
# usage (on the GPU box): tools/pmc_layer_fused.sh gpurun_out/pmc_layer [queries]
out=$1; q=${2:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
PY=$(readlink -f "$(command -v python3)")
for form in fused split; do
  i=0; mkdir -p "$out/$form"
  for set in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
    "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES" ; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d "$out/$form/pass$i" -- "$PY" tools/layer_bench.py --queries $q --reps 2 --form $form > "$out/$form/pass$i.log" 2>&1
  done
  echo "== $form (Q = $q)"
  for k in rowgroup_layer_kernel rowgroup_kernel combine_kernel; do
    python3 tools/pmc_summary.py "$out/$form" "$k" | sed "s/^/$k  /"
  done
done
timeout 300 "$PY" tools/layer_bench.py --queries $q --reps 6 2>&1 | tail -3
