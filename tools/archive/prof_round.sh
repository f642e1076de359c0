#!/bin/bash
# End-of-round evidence: kernel-trace stats, PMC passes (separate runs), bench lines of every config.
# usage (on the GPU box, repo root): bash tools/prof_round.sh <outdir>
set -u
OUT=${1:-gpurun_out/round}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
# 1. kernel trace + stats of the headline bench (C4)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/c4_bench_under_rocprof.json" 2> "$OUT/trace.log"
# 2. PMC passes, one counter group per run (no trace domains besides kernel-trace)
pmc() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc/$name" -- python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
pmc fetch FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pmc write WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pmc tcc TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt" 2>&1
# 3. bench lines
for c in c1 c2 c3 c5; do timeout 300 python3 bench.py --config $c --no-cpu-baseline > "$OUT/${c}_bench.json" 2>> "$OUT/bench.log"; done
timeout 600 python3 bench.py > "$OUT/c4_bench.json" 2>> "$OUT/bench.log"
for r in 0 3 7; do timeout 300 python3 bench.py --emulate $r/8 --no-cpu-baseline > "$OUT/c4_slab${r}of8_bench.json" 2>> "$OUT/bench.log"; done
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/c4_kernel_stats.csv"
rm -rf "$OUT/trace" "$OUT"/pmc/*/*/*kernel_trace.csv
ls -la "$OUT"; head -12 "$OUT/c4_kernel_stats.csv"; cat "$OUT/c4_bench.json" | cut -c1-400
