#!/bin/bash
# round 6: kernel times of library variants (tools/buildvar.sh; base = the shipped library) for one bench config, repeated:
#   tools/r06_var.sh <config> <reps> name1 name2 ...   -> gpurun_out/r06_var_<config>.txt
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cfg=$1; reps=$2; shift 2
{
for r in $(seq 1 $reps); do
for v in "$@"; do
    lib=$PWD/pyiga_amd/libigx_$v.so
    [ "$v" = base ] && lib=$PWD/pyiga_amd/libigx.so
    echo "== $v"
    IGX_LIB=$lib timeout 600 python bench.py --config $cfg --no-cpu-baseline --steps 6 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"
done
done
} > gpurun_out/r06_var_$cfg.txt 2>&1
tail -40 gpurun_out/r06_var_$cfg.txt
