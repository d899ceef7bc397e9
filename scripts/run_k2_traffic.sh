#!/bin/bash
# FETCH_SIZE of the eta-only held-field pass for each tuning library:
#   bash scripts/run_k2_traffic.sh <tag> <lib name> ...
set -e -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/${TAG}.log
: > $OUT
for dtype in f64 f32; do
  for v in "$@"; do
    export MOMLEVEL_AMD_LIB=scripts/variants/lib_$v.so
    D=gpurun_out/${TAG}_${v}_${dtype}
    rm -rf $D
    python3 scripts/k2_traffic.py --dtype $dtype >> $OUT 2>> gpurun_out/${TAG}.err
    rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $D -o run -- python3 scripts/k2_traffic.py --dtype $dtype > $D.stdout 2>&1
    echo "{\"lib\": \"$v\", \"dtype\": \"$dtype\", \"fetch\":" >> $OUT
    python3 scripts/k2_traffic.py --dtype $dtype --summarize $D >> $OUT
    echo "}" >> $OUT
    find $D -name "*.csv" -size +2M -delete
    echo "done $v $dtype"
  done
done
