#!/bin/bash
# scratch driver for gpurun jobs of this round (edited per job)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_b; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 3 > $O/bench1.json 2> $O/bench1.err; tail -2 $O/bench1.err
BENCH_DEV_MIRROR=1 BENCH_NO_4K=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench2_mirror.json 2> $O/bench2_mirror.err; tail -5 $O/bench2_mirror.err
BENCH_DEV_MIRROR=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 8 --steps 10 --warmup 3 > $O/bench8_mirror.json 2> $O/bench8_mirror.err; tail -5 $O/bench8_mirror.err
timeout 600 python tools/strip_overhead.py --out $O/strip_overhead.json > $O/strip_overhead.log 2>&1; tail -20 $O/strip_overhead.log
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cat $O/bench1.json $O/bench2_mirror.json $O/bench8_mirror.json
