#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -k "ragged" 2>&1 | grep -v "^  \|^$" | cut -c1-220 | tail -80
