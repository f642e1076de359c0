#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/var_try.sh base nstw3 base nstw3
IGX_LIB=$PWD/pyiga_amd/libigx_nstw3.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_equals or fullsize_vs_reference or row_slabs or tiny or fixtures or full_size_c4 or repeat or golden_matrices" 2>&1 | tail -5
