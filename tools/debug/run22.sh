cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/t22.log 2>&1; grep -E "passed|failed" gpurun_out/t22.log; grep -E "^FAILED" gpurun_out/t22.log
python tools/stress_bench.py --knob 0
python tools/stress_bench.py --knob 8
python tools/stress_bench.py --knob 0 --batch 2
python tools/stress_bench.py --knob 0 --batch 4
bash tools/pmc.sh gpurun_out/pmc_stress_r2b tools/stress_bench.py --reps 2 --knob 0 > /dev/null 2>&1
bash tools/pmc_combine.sh gpurun_out/pmc_stress_r2b rowgroup_kernel > gpurun_out/r02_stress_rowgroup_pmc.txt 2>&1; tail -30 gpurun_out/r02_stress_rowgroup_pmc.txt
