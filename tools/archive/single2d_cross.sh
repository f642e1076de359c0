#!/bin/bash
# 2D stiffness p=3: the single-launch kernel against the stage-kernel chain over patch sizes (device ms of an assembly)
cd "$GRAFT_REPO_ROOT"
for n in 8 15 24 32 48 64 96 128 256; do
    for path in single unfused; do
        echo -n "n=$n $path: "
        env IGX_PATH=$path timeout 120 python bench.py --config c2 --n $n --no-cpu-baseline --no-api-call --steps 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['step_ms']['median'], 4), d['roofline']['kernel_ms'])
"
    done
done
