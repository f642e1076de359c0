#!/bin/bash
# full GPU suite + the three bench ops
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
timeout 400 python bench.py > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "bench rc=$?"; cat gpurun_out/bench_c4.json; tail -3 gpurun_out/bench_c4.err
timeout 300 python bench.py --op rhs --no-cpu-baseline > gpurun_out/bench_rhs.json 2> gpurun_out/bench_rhs.err; echo "rhs rc=$?"; cat gpurun_out/bench_rhs.json; tail -3 gpurun_out/bench_rhs.err
timeout 300 python bench.py --op entries --no-cpu-baseline > gpurun_out/bench_entries.json 2> gpurun_out/bench_entries.err; echo "entries rc=$?"; cat gpurun_out/bench_entries.json; tail -3 gpurun_out/bench_entries.err
IGX_ENTRIES=thread timeout 300 python bench.py --op entries --no-cpu-baseline > gpurun_out/bench_entries_thread.json 2>&1; cat gpurun_out/bench_entries_thread.json
timeout 300 python bench.py --emulate 3/8 --no-cpu-baseline > gpurun_out/bench_c4_emul.json 2>gpurun_out/bench_c4_emul.err; cat gpurun_out/bench_c4_emul.json; tail -3 gpurun_out/bench_c4_emul.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
