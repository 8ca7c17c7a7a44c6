#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02_l; mkdir -p $O
for mode in "BENCH_FORCE_FALLBACK=1" "BENCH_DEV_SHM=1"; do
  env $mode BENCH_DEV_MIRROR=1 BENCH_VERIFY=1 BENCH_NO_4K=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 > $O/b2.json 2> $O/b2.err
  echo "$mode rc=$?"; python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r02_l/b2.json").read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","verified_vs_single_context")}, d["config"]["parallelism"][:200])
except Exception as e:
    print("no json", e); print(open("gpurun_out/r02_l/b2.err").read()[-1500:])
PY
done
