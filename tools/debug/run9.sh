cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(command -v python3)")
for lib in "" gpurun_variants/libbig_aux2.so gpurun_variants/libbig_un16.so; do
  for k in 8 0; do
    if [ -n "$lib" ]; then ULTRA_BINDING=ctypes ULTRA_RSPMM_LIB=$PWD/$lib python tools/stress_bench.py --knob $k 2>&1 | grep S-stress | sed "s|^|$lib |"; else python tools/stress_bench.py --knob $k 2>&1 | grep S-stress; fi
  done
done | tee gpurun_out/stress2.log
for w in S-wn18rr S-fb15k237; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train3_$w -o train -- "$PY" tools/train_bench.py --workload $w --steps 20 --graphed > gpurun_out/prof_train3_$w.log 2>&1
done
python tools/train_bench.py --workload S-codexm --batch 64 --steps 10 --graphed 2>&1 | grep "ms/step"
