#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r02_g > /tmp/prof.log 2>&1; tail -3 /tmp/prof.log | cut -c1-300
O=gpurun_out/r02_g
timeout 600 python tools/config_table.py > $O/config_table.json 2> $O/config_table.err
timeout 300 python tools/experiments/shadowed_mode.py > $O/shadowed_mode.txt 2>&1; tail -4 $O/shadowed_mode.txt
du -sh gpurun_out
