#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02_l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream" 2>&1 | grep -E "passed|failed|Error|assert" | head -5
{
echo "resolve as a stream (rt_tuning key 15 = 1: k_resolve_stream, persistent wavefronts, lane refill) against the default (k_resolve<work-sharing walk>), config #4 1080p"
echo "== HIP-event time per kernel (tools/experiments/tuning_ab.py 15 0 1)"
python tools/experiments/tuning_ab.py 15 0 1 2>/dev/null | grep 1920
for pm in "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
echo "== rocprofv3 --pmc $pm  (mean per launch)"
for t in "15=1" "15=0"; do
rm -rf /tmp/pm; RT_TUNING=$t rocprofv3 --pmc $pm --output-format csv -d /tmp/pm -- python3 tools/experiments/run_frames.py 6 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:50]
        if 'resolve' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[k]+=1
for k,v in acc.items():
    n=cnt[k] or 1
    print(k, {c:round(x/n) for c,x in v.items()})
PY
done
done
} > $O/stream_resolve_ab.txt 2>&1
cat $O/stream_resolve_ab.txt | cut -c1-400
