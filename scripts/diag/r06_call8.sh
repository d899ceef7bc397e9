for tag in b c; do
  python bench.py --steps 20 --warmup 5 --detail-file gpurun_out/r06_bench_detail_$tag.json 2> gpurun_out/r06_bench_$tag.err | tail -n 1 > gpurun_out/r06_bench_line_$tag.json
  python -c "
import json; l=json.load(open('gpurun_out/r06_bench_line_$tag.json'))
print({k:l[k] for k in ('value','ms_per_step','checks_all_true','checks_failed','reference_example_call')})
print(l['roofline']['frac'], l['roofline']['traffic'], l['roofline'].get('frac_of_matching_probe'))"
done
python bench.py 2> gpurun_out/r06_bench_d.err | tail -n 1 > gpurun_out/r06_bench_line_d.json
