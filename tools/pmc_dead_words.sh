#!/bin/bash
# PMC A/B of the forward kernel of a training step: removed edges as marked words (quad_kernel<.., DEAD>, default) vs as zero weights
# (knob bit 7: the weighted kernel on the same plans).  Separate passes per counter set; only the sets of tools/pmc.sh.
# usage (on the GPU box): tools/pmc_dead_words.sh gpurun_out/pmc_dead
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
PY=$(readlink -f "$(command -v python3)")
for form in marked weighted; do
  knob=0; [ $form = weighted ] && knob=128
  i=0; mkdir -p "$out/$form"
  for set in \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
    "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_WAVES" \
    "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" ; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $set --output-format csv -d "$out/$form/pass$i" -- "$PY" tools/kbench.py --workload S-fb15k237 --batch 16 --boundary --removed --knob $knob --reps 4 > "$out/$form/pass$i.log" 2>&1
  done
  echo "== $form"
  python3 tools/pmc_summary.py "$out/$form" "quad_kernel<0"
  timeout 200 "$PY" tools/kbench.py --workload S-fb15k237 --batch 16 --boundary --removed --knob $knob --reps 40 2>&1 | tail -1
  timeout 200 "$PY" tools/kbench.py --workload S-fb15k237 --batch 16 --backward --removed --knob $knob --reps 40 2>&1 | tail -1
done
