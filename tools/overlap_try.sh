#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3b
run() {
    echo "== $*"
    env "$@" timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 8 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['step_ms'], d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"
}
{
run IGX_OVERLAP=1
run IGX_OVERLAP=4 IGX_OVERLAP_BFS=2
run IGX_OVERLAP=8 IGX_OVERLAP_BFS=2
run IGX_OVERLAP=16 IGX_OVERLAP_BFS=2
run IGX_OVERLAP=8 IGX_OVERLAP_BFS=2 IGX_OVERLAP_LEAN=0
run IGX_OVERLAP=8 IGX_OVERLAP_BFS=3
run IGX_OVERLAP=1
} > gpurun_out/r3b/overlap4.txt 2>&1
cat gpurun_out/r3b/overlap4.txt
