#!/bin/bash
# The GPU suite with the evidence kept if it dies:  bash scripts/gpu_pytest.sh <log name> [pytest args]
# On a non-zero exit the transfer-range log (tests/conftest.py _transfer_log: every staging / result
# buffer and every uploaded host array, by address) stays under gpurun_out/, and the head of any GPU
# core dump the runtime wrote (gpucore.*) is copied there too -- what round 3's abort lacked
# (DESIGN.md section 7).  On success the range logs are removed.
set -o pipefail
cd "$(dirname "$0")/.."
NAME=${1:-pytest_gpu}
shift || true
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q "$@" > gpurun_out/${NAME}.log 2>&1
rc=$?
if [ $rc -ne 0 ]; then
  for core in gpucore.* /tmp/gpucore.*; do
    [ -f "$core" ] || continue
    ls -l "$core" >> gpurun_out/${NAME}.log
    head -c 16777216 "$core" | gzip > gpurun_out/$(basename "$core").head16M.gz
  done
else
  rm -f gpurun_out/transfer_ranges_*.log
fi
tail -5 gpurun_out/${NAME}.log
exit $rc
