echo "== duplex probe"
python scripts/link_duplex_probe.py 2>/dev/null | tee gpurun_out/r06_link_duplex_probe.json
echo "== huge off (np.empty)"
MOMLEVEL_AMD_HUGE_RESULT_MIB=0 python scripts/example_call.py --reps 6 --source numpy 2>/dev/null | tee gpurun_out/r06_example_numpy_huge0_d.json
echo "== pool on"
python scripts/example_call.py --reps 6 --source numpy 2>/dev/null | tee gpurun_out/r06_example_numpy_pool_b.json
echo "== pool off"
MOMLEVEL_AMD_RESULT_POOL_GIB=0 python scripts/example_call.py --reps 6 --source numpy 2>/dev/null | tee gpurun_out/r06_example_numpy_nopool_b.json
echo "== duplex probe again"
python scripts/link_duplex_probe.py 2>/dev/null | tee gpurun_out/r06_link_duplex_probe_b.json
