cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(command -v python3)")
bash tools/refresh_profiles.sh r02 > gpurun_out/refresh_r02.log 2>&1; tail -2 gpurun_out/refresh_r02.log
bash tools/pmc.sh gpurun_out/pmc_quad_r02 tools/kbench.py --batch 32 --reps 5 --boundary > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_quad_r02 quad_kernel > gpurun_out/r02_kbench_fwd_pmc.txt 2>&1; cat gpurun_out/r02_kbench_fwd_pmc.txt | grep -E "FETCH|WRITE|TCC"
bash tools/pmc.sh gpurun_out/pmc_stress_r02 tools/stress_bench.py --reps 3 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_stress_r02 rowgroup_kernel > gpurun_out/r02_stress_rowgroup_pmc.txt 2>&1; cat gpurun_out/r02_stress_rowgroup_pmc.txt | grep -E "FETCH|WRITE|TCC"
for w in S-wn18rr S-fb15k237; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train4_$w -o train -- "$PY" tools/train_bench.py --workload $w --steps 20 --graphed > gpurun_out/prof_train4_$w.log 2>&1
grep "ms/step" gpurun_out/prof_train4_$w.log
done
python tools/run_configs.py > gpurun_out/r02_configs.md 2> gpurun_out/r02_configs.err; tail -12 gpurun_out/r02_configs.md
