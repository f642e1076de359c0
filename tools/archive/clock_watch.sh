#!/bin/bash
# samples clocks / power while the C4 bench runs (DVFS diagnosis: how far does the chain pull the shader clock down?)
cd "$GRAFT_REPO_ROOT"
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power|fclk" | head -8
timeout 120 python bench.py --config c4 --no-cpu-baseline --no-api-call --steps 300 > /tmp/bw.json 2>&1 &
BENCH_PID=$!
sleep 9
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 1; done
wait $BENCH_PID
sleep 3
tail -1 /tmp/bw.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
