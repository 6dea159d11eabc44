#!/bin/bash
# PMC passes for one command (each pass its own run; gfx950 slot limits: 8 SQ, 4 TCC per pass).
# usage: tools/pmc.sh <outdir> <python-script-and-args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
PY=$(readlink -f "$(command -v python3)")     # the ELF interpreter itself: no shim may exec after the profiler's preload
if ! head -c 4 "$PY" | grep -q ELF; then echo "python3 resolves to $PY, which is not an ELF binary" >&2; exit 1; fi
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_WAVES" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- "$PY" "$@" > "$out/pass$i.log" 2>&1
done
ls "$out"
