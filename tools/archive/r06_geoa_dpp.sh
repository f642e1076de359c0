#!/bin/bash
# round 6: the pair-product (DPP row_newbcast) axis-0 sweep of k_geoA against the multiply + FMA sweep, same box and session
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
./tools/ubench/valu_f64 2>&1 | grep -i "dpp\|fmac_f64 v, s\|pattern\|device"
bash tools/var_try.sh base nodpp base nodpp base nodpp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_equals or fullsize_vs_reference or row_slabs or tiny or fixtures or full_size_c4 or repeat or golden" 2>&1 | tail -5
} > gpurun_out/r06_geoa_dpp.txt 2>&1
tail -40 gpurun_out/r06_geoa_dpp.txt
