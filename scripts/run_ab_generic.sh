#!/bin/bash
# A/B of tuning libraries on ONE box, one process per library, two rounds interleaved:
#   bash scripts/run_ab_generic.sh <out tag> <ab script> <nt> <lib name> <lib name> ...
set -e -o pipefail
cd "$(dirname "$0")/.."
TAG=$1; SCRIPT=$2; NT=$3; shift 3
OUT=gpurun_out/${TAG}.log
: > $OUT
for round in 1 2; do
  for v in "$@"; do
    MOMLEVEL_AMD_LIB=scripts/variants/lib_$v.so python3 scripts/$SCRIPT --nt $NT >> $OUT 2>> gpurun_out/${TAG}.err
    echo "done $v round $round"
  done
done
