#!/bin/bash
# k_geoA of a general form (table) with parts switched off (ablation build): IGX_GEOA_DBG bits 1 no geometry, 2 no K1 stores
cd "$GRAFT_REPO_ROOT"
for d in 0 1 2 3; do
  echo "== IGX_GEOA_DBG=$d"
  IGX_LIB=$PWD/pyiga_amd/libigx_ablate.so IGX_GEOA_DBG=$d IGX_STAGE_EVENTS=1 python tools/r06_form_paths.py 128 4 2>&1 | grep "^(inner(grad(u), grad(v)) + u \* v\|convection\|x\[1\]" | cut -c1-60,100-400
done
