#!/bin/bash
# Next-level register prefetch in the held-field K2 kernels (MLX_TUNE_K2_PF64 / _PF32), A/B on ONE
# box: every library in its own process, two rounds interleaved.  bash scripts/run_ab_k2_pf.sh <tag>
set -e -o pipefail
cd "$(dirname "$0")/.."
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_tune_k2_prefetch.log
: > $OUT
for round in 1 2; do
  for v in pf0 pf64 pf64_32 pf64_n16; do
    MOMLEVEL_AMD_LIB=scripts/variants/lib_$v.so python3 scripts/ab_k2.py --nt 48 >> $OUT 2>> gpurun_out/${TAG}_tune_k2_prefetch.err
    echo "done $v round $round"
  done
done
