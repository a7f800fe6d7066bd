#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05r; mkdir -p $O
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -s > $O/pytest_full_s.txt 2>&1; echo "rc=$?"; tail -c 2500 $O/pytest_full_s.txt
