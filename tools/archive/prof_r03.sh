#!/bin/bash
# Round-3 evidence.  usage (GPU box, repo root): bash tools/prof_r03.sh <stage> [outdir]
#   trace : rocprofv3 --kernel-trace --stats of the headline bench command + the default bench line (cpu_baseline, cold path,
#           api call) + bench lines of the other configs and ops
#   slabs : every slab of the strong-scaling split of C4 for W = 2, 4, 8 (+ three C5 slabs of 8), emulated on this one GPU
#   fetch | write | tcc | sq : one PMC pass each (separate runs, kernel-trace only besides the counters)
#   calib : FETCH_SIZE calibration (tools/ubench/fetch_calib) for the read patterns of the chain
set -u
STAGE=${1:-trace}
OUT=${2:-gpurun_out/r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
pmc() { local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc/$name" -- python3 bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline --no-api-call > "$OUT/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?"; python3 tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt" 2>&1; cat "$OUT/pmc_summary.txt"
  rm -f "$OUT"/pmc/*/*/*kernel_trace.csv; }
case $STAGE in
trace)
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-api-call > "$OUT/c4_bench_under_rocprof.json" 2> "$OUT/trace.log"
  find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/c4_kernel_stats.csv"
  rm -rf "$OUT/trace"
  timeout 600 python3 bench.py > "$OUT/c4_bench.json" 2>> "$OUT/bench.log"
  for c in c1 c2 c3 c5; do timeout 300 python3 bench.py --config $c --no-cpu-baseline > "$OUT/${c}_bench.json" 2>> "$OUT/bench.log"; done
  timeout 300 python3 bench.py --op rhs --no-cpu-baseline > "$OUT/c4_rhs_bench.json" 2>> "$OUT/bench.log"
  timeout 300 python3 bench.py --op entries --no-cpu-baseline > "$OUT/c4_entries_bench.json" 2>> "$OUT/bench.log"
  for c in c2 c3; do timeout 300 python3 bench.py --op fast --config $c > "$OUT/fast_${c}_bench.json" 2>> "$OUT/bench.log"; done
  IGX_PATH=unfused IGX_GEOA=0 timeout 300 python3 bench.py --no-cpu-baseline --no-api-call > "$OUT/c4_bench_r01_kernels.json" 2>> "$OUT/bench.log"
  head -12 "$OUT/c4_kernel_stats.csv"; cut -c1-400 "$OUT/c4_bench.json"; tail -5 "$OUT/bench.log"
  ;;
slabs)
  for W in 2 4 8; do for ((r=0; r<W; r++)); do timeout 300 python3 bench.py --emulate $r/$W --no-cpu-baseline > "$OUT/c4_slab${r}of${W}_bench.json" 2>> "$OUT/bench.log"; done; done
  for r in 0 3 7; do timeout 300 python3 bench.py --config c5 --emulate $r/8 --no-cpu-baseline > "$OUT/c5_slab${r}of8_bench.json" 2>> "$OUT/bench.log"; done
  python3 tools/slab_table.py "$OUT" | tee "$OUT/c4_slab_table.txt"
  ;;
fetch) pmc fetch FETCH_SIZE ;;
write) pmc write WRITE_SIZE ;;
tcc)   pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum ;;
sq)    pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS ;;
calib)
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_calib" -- ./tools/ubench/fetch_calib > "$OUT/fetch_calib.txt" 2> "$OUT/fetch_calib.err"
  python3 tools/fetch_calib_summary.py "$OUT/fetch_calib" "$OUT/fetch_calib.txt" | tee "$OUT/fetch_calib_summary.txt"
  rm -rf "$OUT/fetch_calib"
  ;;
esac
ls "$OUT"
