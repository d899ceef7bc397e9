export TMPDIR=/tmp
bash scripts/gpu_pytest.sh r06_suite6 || exit 1
python bench.py --force-collective --steps 20 --warmup 5 2> gpurun_out/r06_forced_b.err | tail -n 1 > gpurun_out/r06_bench_forced_collective_b.json
python -c "
import json; l=json.load(open('gpurun_out/r06_bench_forced_collective_b.json')); print(l['value'], l['ms_per_step'], l['forced_collective'])"
rocprofv3 --output-format csv --kernel-trace -d gpurun_out/prof_r06forced_b -o run -- python3 bench.py --force-collective --steps 3 --warmup 1 > gpurun_out/r06_forced_trace_b.log 2>&1
