#!/bin/bash
# Round-2 evidence.  usage (GPU box, repo root): bash tools/prof_r02.sh <stage> [outdir]
#   trace : rocprofv3 --kernel-trace --stats of the headline bench command + bench lines of every config + slab emulation
#   fetch | write | tcc | sq : one PMC pass each (separate runs, kernel-trace only besides the counters)
set -u
STAGE=${1:-trace}
OUT=${2:-gpurun_out/r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
pmc() { local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc/$name" -- python3 bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?"; python3 tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt" 2>&1; cat "$OUT/pmc_summary.txt"
  rm -f "$OUT"/pmc/*/*/*kernel_trace.csv; }
case $STAGE in
trace)
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/c4_bench_under_rocprof.json" 2> "$OUT/trace.log"
  find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/c4_kernel_stats.csv"
  rm -rf "$OUT/trace"
  timeout 600 python3 bench.py > "$OUT/c4_bench.json" 2>> "$OUT/bench.log"
  for c in c1 c2 c3 c5; do timeout 300 python3 bench.py --config $c --no-cpu-baseline > "$OUT/${c}_bench.json" 2>> "$OUT/bench.log"; done
  for r in 0 1 2 3 4 5 6 7; do timeout 300 python3 bench.py --emulate $r/8 --no-cpu-baseline > "$OUT/c4_slab${r}of8_bench.json" 2>> "$OUT/bench.log"; done
  for r in 0 3 7; do timeout 300 python3 bench.py --config c5 --emulate $r/8 --no-cpu-baseline > "$OUT/c5_slab${r}of8_bench.json" 2>> "$OUT/bench.log"; done
  for r in 0 1; do timeout 300 python3 bench.py --emulate $r/2 --no-cpu-baseline > "$OUT/c4_slab${r}of2_bench.json" 2>> "$OUT/bench.log"; done
  timeout 300 python3 bench.py --op rhs --no-cpu-baseline > "$OUT/c4_rhs_bench.json" 2>> "$OUT/bench.log"
  timeout 300 python3 bench.py --op entries --no-cpu-baseline > "$OUT/c4_entries_bench.json" 2>> "$OUT/bench.log"
  IGX_PATH=unfused IGX_GEOA=0 timeout 300 python3 bench.py --no-cpu-baseline > "$OUT/c4_bench_r01_kernels.json" 2>> "$OUT/bench.log"
  head -12 "$OUT/c4_kernel_stats.csv"; cut -c1-300 "$OUT/c4_bench.json"; tail -5 "$OUT/bench.log"
  ;;
fetch) pmc fetch FETCH_SIZE ;;
write) pmc write WRITE_SIZE ;;
tcc)   pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum ;;
sq)    pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS ;;
esac
ls "$OUT"
