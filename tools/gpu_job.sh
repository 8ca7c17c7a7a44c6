#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02_j; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|Error|assert" $O/pytest_gpu.txt | head
python tools/experiments/tuning_ab.py 14 0 1 2>/dev/null | tee $O/next_raycast_ab.txt
for c in 1920x1080:8:sparse 1920x1080:4:sparse 1920x1080:2:sparse 3840x2160:8:sparse; do timeout 100 python tools/strip_overhead.py --only $c 2>/dev/null | cut -c1-330; done
