bash scripts/gpu_pytest.sh r06_suite7 || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
python scripts/host_path_soak.py --calls 12 --nt 24 2>/dev/null | tee gpurun_out/r06_host_path_soak.log | cut -c1-400
python scripts/host_path_soak.py --calls 8 --nt 60 2>/dev/null | tee gpurun_out/r06_host_path_soak_nt60.log | cut -c1-400
MOMLEVEL_AMD_RESULT_POOL_GIB=0 python scripts/host_path_soak.py --calls 6 --nt 60 2>/dev/null | tee gpurun_out/r06_host_path_soak_nt60_nopool.log | cut -c1-400
