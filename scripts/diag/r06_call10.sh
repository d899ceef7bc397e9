python -m pytest tests/test_gpu_steric.py -x -q -m gpu > gpurun_out/r06_steric_tests.log 2>&1; tail -3 gpurun_out/r06_steric_tests.log
python scripts/example_call.py --reps 8 --source numpy 2>/dev/null | tee gpurun_out/r06_example_ramp.json | cut -c100-600
python scripts/link_duplex_probe.py --reps 3 2>/dev/null | cut -c1-400
