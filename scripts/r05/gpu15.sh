#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05o; mkdir -p $O
timeout 2400 python -X faulthandler -m pytest tests -m gpu -v -x 2>&1 | tail -150 > $O/pytest_full.txt; tail -60 $O/pytest_full.txt
