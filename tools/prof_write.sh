#!/bin/bash
# HBM write-side counters of one bench configuration.  usage: bash tools/prof_write.sh <outdir> [bench args...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { local name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $BARGS > "$OUT/$name.log" 2>&1; }
BARGS="$*"
run write WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run fetch FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -A8 "k_final" "$OUT/summary.txt"
