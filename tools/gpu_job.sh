#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/experiments/ws_variants.py 2>/dev/null
for l in gpurun_variants/*.so; do RT_LIB_PATH=$PWD/$l timeout 200 python tools/experiments/ws_variants.py 2>/dev/null; done
python tools/experiments/tuning_ab.py 14 0 2>/dev/null | grep 1920 | head -1
RT_LIB_PATH=$PWD/gpurun_variants/lib_noscratch.so python tools/experiments/tuning_ab.py 14 0 2>/dev/null | grep 1920 | head -1
