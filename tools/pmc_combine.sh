#!/bin/bash
# PMC passes for the fused layer epilogue (combine_kernel): MFMA busy, LDS conflicts, waits.
# usage: tools/pmc_combine.sh <outdir> <python-script-and-args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
PY=$(readlink -f "$(command -v python3)")     # the ELF interpreter itself: no shim may exec after the profiler's preload
if ! head -c 4 "$PY" | grep -q ELF; then echo "python3 resolves to $PY, which is not an ELF binary" >&2; exit 1; fi
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- "$PY" "$@" > "$out/pass$i.log" 2>&1
done
ls "$out"
