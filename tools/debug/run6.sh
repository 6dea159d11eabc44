cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(command -v python3)")
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/t6.log 2>&1; tail -5 gpurun_out/t6.log
for k in 0 8 16 24; do python tools/stress_bench.py --knob $k 2>&1 | grep S-stress; done | tee gpurun_out/stress1.log
for w in S-wn18rr S-fb15k237; do
  python tools/train_bench.py --workload $w --steps 20 --graphed 2>&1 | grep "ms/step"
  ULTRA_COMBINE_BWD=split python tools/train_bench.py --workload $w --steps 20 --graphed 2>&1 | grep "ms/step"
done | tee gpurun_out/train1.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train2_wn -o train -- "$PY" tools/train_bench.py --workload S-wn18rr --steps 20 --graphed > gpurun_out/prof_train2_wn.log 2>&1
python bench.py --no-stress --finetune-steps 0 --no-cpu-baseline > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err; python -c "
import json; d=json.load(open('gpurun_out/bench_r2c.json')); print(d['value'], d['ms_per_step'], d['composition']['ms_per_step_without_first_layer_frontier'], d['composition']['first_layer_frontier_kernel_ms'])"
