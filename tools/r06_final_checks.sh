#!/bin/bash
# final checks of round 6 on the shipped library: whole gpu suite + the fuzzers (paths, paths with the matrix-core sweep, the twin patch,
# convection-diffusion, load vectors, forms, parametric forms)
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 1500 python tools/fuzz_paths.py 60 51 - 5 2>&1 | grep -E "FAIL|all .* cases|worst" | tail -3
IGX_GEOA=mfma timeout 900 python tools/fuzz_paths.py 30 53 - 4 2>&1 | grep -E "FAIL|all .* cases|worst" | tail -3
timeout 1500 python tools/fuzz_convdiff.py 40 55 2>&1 | grep -E "FAIL|worst" | tail -3
timeout 900 python tools/fuzz_rhs.py 50 57 2>&1 | grep -E "FAIL|worst" | tail -3
timeout 900 python tools/fuzz_forms.py 30 59 2>&1 | tail -3
timeout 900 python tools/fuzz_pforms.py 30 61 2>&1 | tail -3
timeout 900 python tools/fuzz_form_tables.py 40 63 2>&1 | tail -2
timeout 900 python tools/fuzz_twin.py 60 65 2>&1 | tail -1
