#!/bin/bash
# Runs the given commands one after the other on the GPU box, each under its own `timeout -k 10`, output per step under gpurun_out/<tag>/;
# a step that is killed at its limit (rc 124 / 137) ends the call -- no further GPU step is started after a hang.
# usage: scripts/gpu_steps.sh <tag> "<seconds> <name> <command>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
for spec in "$@"; do
  secs=${spec%% *}; rest=${spec#* }; name=${rest%% *}; cmd=${rest#* }
  echo "=== step $name (limit $secs s): $cmd"
  t0=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "$OUT/$name.out" 2> "$OUT/$name.err"
  rc=$?
  echo "=== step $name rc $rc in $(( $(date +%s) - t0 )) s"
  tail -n 6 "$OUT/$name.out"
  echo "$name rc=$rc seconds=$(( $(date +%s) - t0 ))" >> "$OUT/steps.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its limit: stopping"; exit $rc; fi
done
exit 0
