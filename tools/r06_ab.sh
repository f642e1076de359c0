#!/bin/bash
# round 6: A/B of library variants in one session: tools/r06_ab.sh <reps> name1 name2 ...  (base = the shipped library)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
reps=$1; shift
{
for r in $(seq 1 $reps); do bash tools/var_try.sh "$@"; done
} > gpurun_out/r06_ab.txt 2>&1
tail -60 gpurun_out/r06_ab.txt
