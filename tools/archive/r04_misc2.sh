#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== new parity tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_core or convdiff_geometry_in_sweep" 2>&1 | tail -6
echo "== rhs kernels"
mkdir -p gpurun_out/rhs_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rhs_prof -- python3 bench.py --op rhs --steps 10 > gpurun_out/rhs_prof/rhs.log 2>&1
tail -1 gpurun_out/rhs_prof/rhs.log | cut -c1-400
find gpurun_out/rhs_prof -name "*kernel_stats.csv" | head -1 | xargs head -12
