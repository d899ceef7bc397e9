echo "== pool on"; python scripts/host_path_soak.py --calls 24 --nt 60 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['wall_s'])"
echo "== pool off"; MOMLEVEL_AMD_RESULT_POOL_GIB=0 python scripts/host_path_soak.py --calls 24 --nt 60 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['wall_s'])"
echo "== np.empty"; MOMLEVEL_AMD_HUGE_RESULT_MIB=0 python scripts/host_path_soak.py --calls 24 --nt 60 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['wall_s'])"
echo "== pool on again"; python scripts/host_path_soak.py --calls 24 --nt 60 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['wall_s'])"
