nproc; cat /sys/fs/cgroup/cpu.max
for pf in 0 1; do
  echo "== huge 64, prefault threads $pf"
  MOMLEVEL_AMD_PREFAULT_THREADS=$pf python scripts/example_call.py --reps 5 --source numpy 2>/dev/null | tee gpurun_out/r06_example_numpy_huge64_pf$pf.json
done
echo "== huge 0"
MOMLEVEL_AMD_HUGE_RESULT_MIB=0 python scripts/example_call.py --reps 5 --source numpy 2>/dev/null | tee gpurun_out/r06_example_numpy_huge0_b.json
echo "== masked, huge 0"
MOMLEVEL_AMD_HUGE_RESULT_MIB=0 python scripts/example_call.py --reps 5 --source masked 2>/dev/null | tee gpurun_out/r06_example_masked_huge0.json
echo "== masked_lazy, huge 0"
MOMLEVEL_AMD_HUGE_RESULT_MIB=0 python scripts/example_call.py --reps 5 --source masked_lazy 2>/dev/null | tee gpurun_out/r06_example_masked_lazy_huge0.json
echo "== masked, huge 64 pf 0"
MOMLEVEL_AMD_PREFAULT_THREADS=0 python scripts/example_call.py --reps 5 --source masked 2>/dev/null | tee gpurun_out/r06_example_masked_huge64_pf0.json
