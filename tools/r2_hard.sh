#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_hardening.py -x -q 2>&1 | tail -15
