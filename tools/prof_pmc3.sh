#!/bin/bash
# PMC passes focused on the memory side of the final stage.  usage: bash tools/prof_pmc3.sh <outdir> [config]
set -u
OUT=${1:-gpurun_out/pmc3}
CFG=${2:-c4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --config "$CFG" --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/$name.log" 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run tcc2 TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum
run tcc3 TCC_EA0_WRREQ_STALL_sum TCC_WRITEBACK_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_TAG_STALL_sum
run ta TA_BUSY_avr TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
tail -5 "$OUT"/*.log | head -60
