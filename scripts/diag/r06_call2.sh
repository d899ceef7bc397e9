set -x
python -m pytest tests/test_gpu_config4.py -x -q -k rccl > gpurun_out/r06_rccl2_pytest.log 2>&1; tail -3 gpurun_out/r06_rccl2_pytest.log
uname -r; cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag
for mode in 64 0; do
  MOMLEVEL_AMD_HUGE_RESULT_MIB=$mode python scripts/example_call.py --reps 4 --source numpy > gpurun_out/r06_example_numpy_huge$mode.json 2> gpurun_out/r06_example_numpy_huge$mode.err; cat gpurun_out/r06_example_numpy_huge$mode.json
done
python scripts/example_call.py --reps 3 --source masked > gpurun_out/r06_example_masked.json 2> gpurun_out/r06_example_masked.err; cat gpurun_out/r06_example_masked.json
python scripts/example_call.py --reps 3 --source masked_lazy > gpurun_out/r06_example_masked_lazy.json 2> gpurun_out/r06_example_masked_lazy.err; cat gpurun_out/r06_example_masked_lazy.json
for t in 2 8; do
  MOMLEVEL_AMD_PREFAULT_THREADS=$t python scripts/example_call.py --reps 3 --source numpy > gpurun_out/r06_example_numpy_pf$t.json 2>/dev/null; cat gpurun_out/r06_example_numpy_pf$t.json
done
