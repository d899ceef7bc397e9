#!/bin/bash
# rocprofv3 evidence for the stratification kernels (csrc/momlevel_strat.hip), on the GPU box:
#   bash scripts/run_profiles_strat.sh r04
# Each counter set is its own pass with --kernel-trace only; the profiled program is python3 itself.
set -e -o pipefail
TAG=${1:-r04}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
V=gpurun_out/prof_${TAG}strat
mkdir -p $V
VAR="python3 scripts/profile_strat.py --nt 16 --reps 2"
echo "== strat: plain";        $VAR --plan-out $V/plan.json > $V/plain.log 2>&1
echo "== strat: kernel trace"; rocprofv3 --output-format csv --kernel-trace --stats -d $V/trace -o run -- $VAR > $V/trace.log 2>&1
echo "== strat: FETCH_SIZE";   rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $V/pmc_fetch -o run -- $VAR > $V/pmc_fetch.log 2>&1
echo "== strat: WRITE_SIZE";   rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $V/pmc_write -o run -- $VAR > $V/pmc_write.log 2>&1
echo "== strat: SQ";           rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $V/pmc_sq -o run -- $VAR > $V/pmc_sq.log 2>&1
find $V -name "*_agent_info.csv" -delete
MLX_SUMMARY_MAIN=k_stratification,k_adjust_n2 python3 scripts/summarize_variants.py $V profiles/${TAG}_strat
echo "strat profiles done"
