#!/bin/bash
# Every GPU call of a round goes through this wrapper, so that a call the box ends by force is judged
# from its OWN record and not from a sentence about it:
#
#     bash scripts/gpurun_keep.sh <tag> [--timeout SECONDS] -- '<command>'
#
# What gpurun printed when the call ended (exit code, the tail of the command's stdout / stderr, and
# -- when the box killed the command -- the guard's own message: "process guard", "memory cap", the
# time limit, a hang) is kept as gpurun_out/<tag>_call.log.  When the call did not end with exit
# code 0 that record, gpurun's verdict file (gpurun_out/.last_call.json) and the command line are
# ALSO copied to gpurun_out/<tag>_killed.log and to profiles/<tag>_killed.log (tracked) BEFORE
# anything else runs: nothing is retried here, the next call is the caller's decision.
set -o pipefail
cd "$(dirname "$0")/.."
TAG=${1:?tag}
shift
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
GPURUN=${GPURUN:-/usr/local/graft/bin/gpurun}
{
  echo "# $(date -u +%FT%TZ) gpurun ${ARGS[*]} -- $*"
  "$GPURUN" "${ARGS[@]}" -- "$@" 2>&1
  echo "# gpurun exit code: $?"
} | tee gpurun_out/${TAG}_call.log | tail -40
rc=$(sed -n 's/^# gpurun exit code: //p' gpurun_out/${TAG}_call.log | tail -1)
# gpurun's own verdict line: "status=ok" / "status=fail" = the command ran to its end (a failing test
# is an ordinary failure, its log is under gpurun_out/); anything else -- a time limit, the process
# guard, the memory cap, a lost box -- ended it by force
status=$(sed -n 's/^\[gpurun\] status=\([a-z_]*\).*/\1/p' gpurun_out/${TAG}_call.log | tail -1)
if [ "${rc:-1}" != "0" ] && [ "$status" != "fail" ]; then
  {
    echo "# call '${TAG}' did not end cleanly (gpurun exit code ${rc:-?}); its record, unedited:"
    cat gpurun_out/${TAG}_call.log
    echo "# gpurun_out/.last_call.json:"
    cat gpurun_out/.last_call.json 2>/dev/null
  } > gpurun_out/${TAG}_killed.log
  # exit codes 2 / 3 (refused, no box free) charged nothing and ran nothing: not a kill
  if [ "${rc:-1}" != "2" ] && [ "${rc:-1}" != "3" ]; then
    cp gpurun_out/${TAG}_killed.log profiles/${TAG}_killed.log
  fi
fi
exit ${rc:-1}
