cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in c1 c2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2d_$c -- python3 bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline --no-api-call > gpurun_out/p2d_${c}_bench.json 2> gpurun_out/p2d_$c.log
  find gpurun_out/p2d_$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/p2d_${c}_kernel_stats.csv
  rm -rf gpurun_out/p2d_$c
  head -8 gpurun_out/p2d_${c}_kernel_stats.csv
done
