#!/bin/bash
# PMC counters of the fused stage (C4), one pass per counter group
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_bf; mkdir -p $OUT
run() { local name=$1; shift; IGX_PATH=fused rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/$name.log" 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
python3 tools/pmc_summary.py $OUT | grep -A40 "k_bf\|k_mirror" | head -80
