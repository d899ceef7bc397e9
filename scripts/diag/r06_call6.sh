export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 --detail-file gpurun_out/r06_bench_detail_a.json > gpurun_out/r06_bench_stdout_a.log 2> gpurun_out/r06_bench_a.err
tail -n 1 gpurun_out/r06_bench_stdout_a.log > gpurun_out/r06_bench_line_a.json
python -c "
import json; l=json.load(open('gpurun_out/r06_bench_line_a.json'))
print({k:l[k] for k in ('value','ms_per_step','checks_all_true','checks_failed','reference_example_call')})
print(l['roofline']['frac'], l['roofline']['traffic'], l['cpu_baseline'])"
python bench.py --force-collective --steps 20 --warmup 5 > gpurun_out/r06_bench_forced_collective.json 2> gpurun_out/r06_forced.err
python -c "
import json; l=json.load(open('gpurun_out/r06_bench_forced_collective.json')); print(l['value'], l['ms_per_step'], l['forced_collective'])"
rocprofv3 --output-format csv --kernel-trace -d gpurun_out/prof_r06forced -o run -- python3 bench.py --force-collective --steps 3 --warmup 1 > gpurun_out/r06_forced_trace.log 2>&1
