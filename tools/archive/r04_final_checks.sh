#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_chunks or load_vector or inner_products" 2>&1 | tail -4
timeout 900 python tools/fuzz_convdiff.py 30 3 2>&1 | tail -34
IGX_GEOA=mfma timeout 900 python tools/fuzz_paths.py 30 5 - 4 2>&1 | tail -6
bash tools/prof_r04.sh slabs gpurun_out/r04 2>&1 | tail -12
