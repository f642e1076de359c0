#!/bin/bash
# C5 (convection-diffusion, 3D p=5 n=96) against the launch/LDS knobs of the stage kernels
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/c5_sweep.txt
: > $OUT
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 5 --warmup 1 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-44s %7.2f ms   %s' % ('$label', d['step_ms']['median'], d['roofline']['kernel_ms']))
" >> $OUT
}
for w in 2 4 8; do for d in 2 3; do run "k_final_q IGX_Q_WAVES=$w IGX_Q_DEPTH=$d" IGX_LIB=$PWD/pyiga_amd/libigx_w${w}d${d}.so; done; done
run "IGX_FINAL=valu (k_final, LDS basis table)" IGX_FINAL=valu
for t in 1 2 4 8 16; do run "k_final_q IGX_FINALQ_TILE=$t" IGX_FINALQ_TILE=$t; done
for w in 1024 2048 4096; do run "k_final_q IGX_FINALQ_WAVES=$w" IGX_FINALQ_WAVES=$w; done
cat $OUT
