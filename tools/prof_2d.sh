#!/bin/bash
# rocprofv3 kernel stats of the 2D configs (kernels without event markers): tools/prof_2d.sh [config...]  -> gpurun_out/r06_2d_<cfg>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in ${@:-c2}; do
  rm -rf gpurun_out/p2d
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2d -- python3 bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline --no-api-call > gpurun_out/r06_2d_${c}_bench.json 2> gpurun_out/p2d.log
  find gpurun_out/p2d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06_2d_${c}_kernel_stats.csv
  rm -rf gpurun_out/p2d
  head -6 gpurun_out/r06_2d_${c}_kernel_stats.csv | cut -c1-200
done
