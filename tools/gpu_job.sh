#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02_f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest.log | head -20
timeout 300 python tools/bvh_builders.py > $O/bvh_builders.json 2> $O/bvh_builders.err; cat $O/bvh_builders.json; tail -3 $O/bvh_builders.err
timeout 300 python tools/spatial_variants.py > $O/spatial_variants.json 2> $O/spatial_variants.err; cat $O/spatial_variants.json
