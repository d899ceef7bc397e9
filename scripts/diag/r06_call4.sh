echo "== pool on (default), prefault off"
python scripts/example_call.py --reps 6 --source numpy 2>/dev/null | tee gpurun_out/r06_example_numpy_pool.json
echo "== pool off"
MOMLEVEL_AMD_RESULT_POOL_GIB=0 python scripts/example_call.py --reps 6 --source numpy 2>/dev/null | tee gpurun_out/r06_example_numpy_nopool.json
echo "== huge off (np.empty)"
MOMLEVEL_AMD_HUGE_RESULT_MIB=0 python scripts/example_call.py --reps 6 --source numpy 2>/dev/null | tee gpurun_out/r06_example_numpy_huge0_c.json
echo "== pool on, masked"
python scripts/example_call.py --reps 6 --source masked 2>/dev/null | tee gpurun_out/r06_example_masked_pool.json
echo "== pool off, prefault by atomic touch, 2 threads"
MOMLEVEL_AMD_RESULT_POOL_GIB=0 MOMLEVEL_AMD_PREFAULT=touch MOMLEVEL_AMD_PREFAULT_THREADS=2 python scripts/example_call.py --reps 4 --source numpy 2>/dev/null | tee gpurun_out/r06_example_numpy_touch2.json
