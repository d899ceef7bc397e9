#!/bin/bash
# The fused policies' tuning switches (eos_device.hpp MLX_TUNE_FMA_ACC), A/B on
# ONE box: every library in its own process, two rounds interleaved.
set -e -o pipefail
cd "$(dirname "$0")/.."
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_tune_k1_fused.log
: > $OUT
for round in 1 2; do
  for v in base fmaacc; do
    MOMLEVEL_AMD_LIB=scripts/variants/lib_$v.so python3 scripts/ab_k1.py --nt 40 >> $OUT 2>> gpurun_out/${TAG}_tune_k1_fused.err
    echo "done $v round $round"
  done
done
