#!/bin/bash
# round 6, extra: longer randomised runs on the final library (other seeds than tools/r06_final_checks.sh)
cd "$GRAFT_REPO_ROOT"
timeout 2400 python tools/fuzz_paths.py 200 601 - 5 2>&1 | grep -E "FAIL|all .* cases|worst" | tail -3
timeout 1800 python tools/fuzz_form_tables.py 150 603 2>&1 | tail -2
timeout 1800 python tools/fuzz_convdiff.py 100 605 2>&1 | grep -E "FAIL|worst" | tail -3
timeout 900 python tools/fuzz_forms.py 80 607 2>&1 | tail -2
timeout 900 python tools/fuzz_rhs.py 100 609 2>&1 | grep -E "FAIL|worst" | tail -3
timeout 1800 python tools/fuzz_twin.py 200 611 2>&1 | tail -1
