#!/bin/bash
# FETCH_SIZE (and WRITE_SIZE) of the eta-only held-field pass at nt = 120 for each tuning library
# ("default" = the in-tree library):   bash scripts/run_k2_traffic.sh <tag> <lib name> ...
set -e -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/${TAG}.log
: > $OUT
for dtype in f64 f32; do
  for v in "$@"; do
    if [ "$v" = "default" ]; then unset MOMLEVEL_AMD_LIB; else export MOMLEVEL_AMD_LIB=scripts/variants/lib_$v.so; fi
    python3 scripts/k2_traffic.py --dtype $dtype >> $OUT 2>> gpurun_out/${TAG}.err
    for counter in FETCH_SIZE WRITE_SIZE; do
      D=gpurun_out/${TAG}_${v}_${dtype}_${counter}
      rm -rf $D
      rocprofv3 --output-format csv --kernel-trace --pmc $counter -d $D -o run -- python3 scripts/k2_traffic.py --dtype $dtype > $D.stdout 2>&1
      echo "{\"lib\": \"$v\", \"dtype\": \"$dtype\", \"counter\": \"$counter\", \"per_kernel\":" >> $OUT
      python3 scripts/k2_traffic.py --dtype $dtype --summarize $D --counter $counter >> $OUT
      echo "}" >> $OUT
      find $D -name "*.csv" -size +2M -delete
    done
    echo "done $v $dtype"
  done
done
