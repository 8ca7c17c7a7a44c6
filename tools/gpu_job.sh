#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02_j; mkdir -p $O
python tools/experiments/ws_variants.py 2>/dev/null | tee $O/ws_variants3.txt
for l in gpurun_variants/*.so; do RT_LIB_PATH=$PWD/$l timeout 200 python tools/experiments/ws_variants.py 2>/dev/null | tee -a $O/ws_variants3.txt; done
