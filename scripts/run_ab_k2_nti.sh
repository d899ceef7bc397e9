#!/bin/bash
# K2 time steps per thread, A/B on ONE box: every tuning library in its own process, two rounds
# interleaved (box drift shows up as round-to-round scatter).  bash scripts/run_ab_k2_nti.sh <tag>
set -e -o pipefail
cd "$(dirname "$0")/.."
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_tune_k2_nti.log
: > $OUT
for round in 1 2; do
  for v in nti64_16 nti64_12 nti64_8 nti32_6 nti32_4 nti32_8; do
    MOMLEVEL_AMD_LIB=scripts/variants/lib_$v.so python3 scripts/ab_k2.py --nt 48 >> $OUT 2>> gpurun_out/${TAG}_tune_k2_nti.err
    echo "done $v round $round"
  done
done
