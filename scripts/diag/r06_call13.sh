MOMLEVEL_TEST_STOCK_TRANSFERS=1 bash scripts/gpu_pytest.sh r06_suite_stock || exit 1
python bench.py --steps 20 --warmup 5 --detail-file gpurun_out/r06_bench_detail_e.json 2> gpurun_out/r06_bench_e.err | tail -n 1 > gpurun_out/r06_bench_line_e.json
python -c "
import json; l=json.load(open('gpurun_out/r06_bench_line_e.json'))
print({k:l[k] for k in ('value','ms_per_step','checks_all_true','checks_failed','reference_example_call')})
print(l['roofline']['frac'], l['roofline']['traffic'], l['roofline'].get('frac_of_matching_probe'))"
