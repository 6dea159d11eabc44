cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(command -v python3)")
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/t10.log 2>&1; tail -4 gpurun_out/t10.log
for w in S-fb15k237 S-wn18rr; do
  python tools/kbench.py --workload $w --batch 32 --boundary --reps 40 2>&1 | grep median
  for v in q_blk512 q_blk512_u12 q_blk512_u16 q_wordnt; do
    ULTRA_BINDING=ctypes ULTRA_RSPMM_LIB=$PWD/gpurun_variants/lib$v.so python tools/kbench.py --workload $w --batch 32 --boundary --reps 40 2>&1 | grep median
  done
done | tee gpurun_out/quad_variants.log
for w in S-wn18rr S-fb15k237; do python tools/train_bench.py --workload $w --steps 20 --graphed 2>&1 | grep "ms/step"; done
