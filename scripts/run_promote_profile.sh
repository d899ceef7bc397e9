#!/bin/bash
# rocprofv3 evidence for the any-dtype EOS map (k_eos_promote) and the mixed-dtype K1 / K2 twins:
#   bash scripts/run_promote_profile.sh r03      (on the GPU box, through gpurun)
# kernel trace + stats in one pass, FETCH_SIZE and WRITE_SIZE each in their own pass (never combined
# with other trace domains); the profiled program is python3 itself.
set -e -o pipefail
TAG=${1:-r03}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
P=gpurun_out/prof_${TAG}p
mkdir -p $P
CMD="python3 scripts/promote_probe.py --nt 16"
rocprofv3 --output-format csv --kernel-trace --stats -d $P/trace -o run -- $CMD > $P/trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $P/pmc_fetch -o run -- $CMD > $P/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $P/pmc_write -o run -- $CMD > $P/pmc_write.log 2>&1
find $P -name "*_agent_info.csv" -delete
echo "promote profile done"
