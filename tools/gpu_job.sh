#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02_d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log
timeout 600 python tools/strip_overhead.py --out $O/strip_overhead.json > $O/strip_overhead.log 2>&1; cat $O/strip_overhead.log | cut -c1-330 | grep -v amdgpu
