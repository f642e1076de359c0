#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for c in c4k c4m; do timeout 600 python bench.py --config $c --no-cpu-baseline --no-api-call --steps 5 > gpurun_out/r04_${c}_bench.json 2> gpurun_out/r04_${c}.err; tail -1 gpurun_out/r04_${c}_bench.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['workload'], '|', round(d['ms_per_step'], 3), 'ms', d['roofline']['kernel_ms'], 'frac', round(d['roofline']['frac'], 4), 'nnz', d['config']['nnz'])
"; tail -2 gpurun_out/r04_${c}.err; done
