#!/bin/bash
# LDS bank-conflict counters of the C4 chain (separate PMC pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04lds; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS --output-format csv -d $OUT/p -- python3 bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline --no-api-call --placement-tries 1 > $OUT/log.txt 2>&1
python3 tools/pmc_summary.py $OUT/p | grep -A8 -E "k_bf2|k_geoA|k_mirror2" | head -40
