#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grouped_lists or query_list_in_flight or concurrent_host" 2>&1 | tail -3
