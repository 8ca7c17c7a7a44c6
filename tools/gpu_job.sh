#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do for t in "8=1" "8=0" "8=1,14=0" "8=0,14=0"; do for c in 1920x1080:8:sparse 1920x1080:4:sparse 3840x2160:8:sparse; do echo -n "$t  "; RT_TUNING=$t timeout 100 python tools/strip_overhead.py --only $c 2>/dev/null | cut -c1-110; done; done; done
