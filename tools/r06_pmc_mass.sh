#!/bin/bash
# round 6: SQ counters of the mass chain at C4's size (k_geoA one-array form + k_bf3<SYM=3>)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_mass; rm -rf $OUT; mkdir -p $OUT
i=0
for ctr in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/p$i -- python3 bench.py --config c4mass --steps 1 --warmup 0 --no-cpu-baseline --no-api-call > $OUT/p$i.log 2>&1
  rm -f $OUT/p$i/*/*kernel_trace.csv $OUT/p$i/*/*agent_info.csv
done
python3 tools/pmc_summary.py $OUT | tee $OUT/summary.txt | grep -A22 "k_bf3\|k_geoA<" | head -70
