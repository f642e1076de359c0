#!/bin/bash
# Round-6 evidence.  usage (GPU box, repo root): bash tools/prof_r06.sh <stage> [outdir]
#   trace : rocprofv3 --kernel-trace --stats of the headline bench command (C4) and of C5 / rhs + the default bench line
#           (cpu_baseline, cold path, api call) + bench lines of the other configs and ops
#   pmc   : FETCH_SIZE and WRITE_SIZE passes (separate runs, kernel-trace only besides the counter) for EVERY config and op
#           that carries a roofline: c2 c3 c4 c5, c4 rhs, c4 entries   -> tools/make_traffic.py -> profiles/r06_traffic.json
#   sq    : SQ / TCC passes of C4 (VALU busy, waits, L2 hit rate)
#   slabs : every slab of the strong-scaling split of C4 for W = 2, 4, 8, emulated on this one GPU
set -u
STAGE=${1:-trace}
OUT=${2:-gpurun_out/r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
pmc() { local key=$1 name=$2; shift 2; local ctr=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc_$key/$name" -- python3 bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-api-call > "$OUT/pmc_${key}_$name.log" 2>&1
  echo "pmc $key $name rc=$?"; rm -f "$OUT"/pmc_$key/*/*/*kernel_trace.csv "$OUT"/pmc_$key/*/*/*agent_info.csv; }
case $STAGE in
trace)
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-api-call > "$OUT/c4_bench_under_rocprof.json" 2> "$OUT/trace.log"
  find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/c4_kernel_stats.csv"; rm -rf "$OUT/trace"
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline --no-api-call > "$OUT/c5_bench_under_rocprof.json" 2>> "$OUT/trace.log"
  find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/c5_kernel_stats.csv"; rm -rf "$OUT/trace"
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --op rhs --steps 10 --warmup 2 > "$OUT/c4_rhs_bench_under_rocprof.json" 2>> "$OUT/trace.log"
  find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/c4_rhs_kernel_stats.csv"; rm -rf "$OUT/trace"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --config c2 --steps 50 --warmup 5 --no-cpu-baseline --no-api-call > "$OUT/c2_bench_under_rocprof.json" 2>> "$OUT/trace.log"
  find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/c2_kernel_stats.csv"; rm -rf "$OUT/trace"
  timeout 600 python3 bench.py > "$OUT/c4_bench.json" 2>> "$OUT/bench.log"
  for c in c1 c2 c3 c5 c4m c4k c4g c4l c5s c3mass c4mass; do timeout 300 python3 bench.py --config $c > "$OUT/${c}_bench.json" 2>> "$OUT/bench.log"; done
  timeout 300 python3 bench.py --op rhs > "$OUT/c4_rhs_bench.json" 2>> "$OUT/bench.log"
  timeout 300 python3 bench.py --op entries > "$OUT/c4_entries_bench.json" 2>> "$OUT/bench.log"
  timeout 300 python3 bench.py --op form --no-cpu-baseline > "$OUT/c4_form_bench.json" 2>> "$OUT/bench.log"
  IGX_PATH=unfused IGX_GEOA=0 timeout 300 python3 bench.py --no-cpu-baseline --no-api-call > "$OUT/c4_bench_r01_kernels.json" 2>> "$OUT/bench.log"
  IGX_BF=2 timeout 300 python3 bench.py --no-cpu-baseline --no-api-call > "$OUT/c4_bench_bf2_mirror.json" 2>> "$OUT/bench.log"
  IGX_GEOA=0 timeout 300 python3 bench.py --config c5 --no-cpu-baseline --no-api-call > "$OUT/c5_bench_r03_kernels.json" 2>> "$OUT/bench.log"
  head -8 "$OUT/c4_kernel_stats.csv"; cut -c1-400 "$OUT/c4_bench.json"; tail -5 "$OUT/bench.log"
  ;;
pmc4)
  pmc c4 fetch FETCH_SIZE --config c4
  pmc c4 write WRITE_SIZE --config c4
  python3 tools/make_traffic.py "$OUT" | tail -40
  ;;
pmc)
  for c in c4 c5 c3 c2; do
    pmc $c fetch FETCH_SIZE --config $c
    pmc $c write WRITE_SIZE --config $c
  done
  pmc c4_rhs fetch FETCH_SIZE --op rhs; pmc c4_rhs write WRITE_SIZE --op rhs
  pmc c4_entries fetch FETCH_SIZE --op entries; pmc c4_entries write WRITE_SIZE --op entries
  pmc c4_form fetch FETCH_SIZE --op form; pmc c4_form write WRITE_SIZE --op form
  python3 tools/make_traffic.py "$OUT" | tail -40
  ;;
sq)
  pmc c4sq tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" --config c4
  pmc c4sq sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS" --config c4
  pmc c4sq sq2 "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" --config c4
  pmc c4sq sq3 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --config c4
  python3 tools/pmc_summary.py "$OUT/pmc_c4sq" | tee "$OUT/c4_pmc_summary.txt"
  ;;
sq5)
  pmc c5sq tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" --config c5
  pmc c5sq sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS" --config c5
  pmc c5sq sq2 "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" --config c5
  pmc c5sq sq3 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --config c5
  python3 tools/pmc_summary.py "$OUT/pmc_c5sq" | tee "$OUT/c5_pmc_summary.txt"
  ;;
all)
  bash "$0" trace "$OUT"; bash "$0" sq "$OUT"; bash "$0" sq5 "$OUT"; bash "$0" pmc "$OUT"; bash "$0" slabs "$OUT"
  ;;
slabs)
  for W in 2 4 8; do for ((r=0; r<W; r++)); do timeout 300 python3 bench.py --emulate $r/$W --no-cpu-baseline > "$OUT/c4_slab${r}of${W}_bench.json" 2>> "$OUT/bench.log"; done; done
  python3 tools/slab_table.py "$OUT" | tee "$OUT/c4_slab_table.txt"
  ;;
sq3)
  pmc c4sq sq3 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --config c4
  pmc c4sq sq4 "SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" --config c4
  python3 tools/pmc_summary.py "$OUT/pmc_c4sq" | tee "$OUT/c4_lds_summary.txt"
  ;;
esac
ls "$OUT"
