#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --op rhs --steps 10 2>&1 | tail -1 | cut -c1-700
timeout 300 python bench.py --op rhs --config c5 --steps 10 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py --op rhs --config c3 --steps 10 2>&1 | tail -1 | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inner_products or load_vector or vector or poisson or functional or jet" 2>&1 | tail -6
