#!/bin/bash
cd "$GRAFT_REPO_ROOT"
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['step_ms'], d['roofline']['kernel_ms'], d['config']['path'])
    else: print(l.rstrip()[-300:])
"; }
echo "== C5 k_geoA (two-source arrays first)"; timeout 300 python bench.py --config c5 --no-cpu-baseline --no-api-call --steps 6 2>&1 | line
echo "== mirror block order (ablation build, one process)"
IGX_LIB=$PWD/pyiga_amd/libigx_ablate.so timeout 300 python tools/mirror_order_ab.py IGX_MIRROR_ORDER 0 1 2>&1 | tail -4
echo "== rhs kernels"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/rhs_prof -o rhs -- python3 $GRAFT_REPO_ROOT/bench.py --op rhs --steps 10 > /tmp/rhs.log 2>&1
tail -2 /tmp/rhs.log | cut -c1-600
find $GRAFT_REPO_ROOT/gpurun_out/rhs_prof -name "*kernel_stats.csv" | head -1 | xargs head -12
