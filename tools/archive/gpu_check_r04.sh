#!/bin/bash
# final state of the round: full gpu suite, smoke(), default bench line
cd "$GRAFT_REPO_ROOT"
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/final_bench_c4.json; python -c "
import json; d = json.load(open('gpurun_out/final_bench_c4.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['placement'], d['cpu_baseline']['value'], d['roofline']['traffic']['bytes'] if d['roofline']['traffic'] else None)"
