#!/bin/bash
# One GPU-box session: runs the given steps in order, each under its own timeout, logs under gpurun_out/, and stops at the first
# step that had to be killed (a hung kernel must be understood before anything else touches the card).
# usage: scripts/gpu_session.sh "<name>|<timeout s>|<command>" ...
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (timeout ${tmo}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc $(( $(date +%s) - start ))s"; tail -n 12 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name was killed at its limit: stopping the session"; exit $rc; fi
done
exit 0
