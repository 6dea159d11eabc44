cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(command -v python3)")
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/t5.log 2>&1; tail -5 gpurun_out/t5.log
python bench.py --no-stress --finetune-steps 0 > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err
python tools/stress_bench.py > gpurun_out/stress0.log 2>&1; python tools/stress_bench.py --knob 8 >> gpurun_out/stress0.log 2>&1; python tools/stress_bench.py --batch 4 >> gpurun_out/stress0.log 2>&1; cat gpurun_out/stress0.log | grep S-stress
bash tools/pmc.sh gpurun_out/pmc_stress tools/stress_bench.py --reps 3 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_stress rowgroup_kernel > gpurun_out/pmc_stress_summary.txt 2>&1; cat gpurun_out/pmc_stress_summary.txt
for w in S-wn18rr S-fb15k237; do
  python tools/train_bench.py --workload $w --steps 20 > gpurun_out/train_$w.log 2>&1
  python tools/train_bench.py --workload $w --steps 20 --graphed >> gpurun_out/train_$w.log 2>&1
  grep "ms/step" gpurun_out/train_$w.log
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_$w -o train -- "$PY" tools/train_bench.py --workload $w --steps 20 --graphed > gpurun_out/prof_train_$w.log 2>&1
done
