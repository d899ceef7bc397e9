run() { for k in 1 2 3 4 5; do REPS=10 python scripts/diag/r06_timeline.py 2>/dev/null | python -c "
import sys, json; d=json.load(sys.stdin); print(d['wall_s'])"; done; }
echo "== default (MADV_FREE, async rho0)"; run
echo "== no MADV_FREE"; MOMLEVEL_AMD_POOL_MADV_FREE=0 run
echo "== sync rho0"; MOMLEVEL_AMD_SYNC_RHO0=1 run
echo "== np.empty results"; MOMLEVEL_AMD_HUGE_RESULT_MIB=0 run
