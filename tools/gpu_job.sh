#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_k.txt 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/pytest_gpu_k.txt | head
bash tools/profile_round.sh r02_k > /dev/null 2>&1
cat gpurun_out/r02_k/bench.json | cut -c1-400
python tools/config_table.py 2>/dev/null > gpurun_out/r02_k/summary/r02_k_config_table.json
python tools/wave_tail.py 2>/dev/null > gpurun_out/r02_k/summary/r02_k_wave_tail.txt
python tools/strip_overhead.py 2>/dev/null > gpurun_out/r02_k/summary/r02_k_strip_overhead.json
rm -rf /tmp/tl; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -- python3 tools/strip_overhead.py --only 1920x1080:8:sparse > /dev/null 2>&1
python tools/strip_timeline.py /tmp/tl > gpurun_out/r02_k/summary/r02_k_strip_timeline_1080p_n8.txt 2>&1
rm -rf /tmp/tl; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -- python3 tools/strip_overhead.py --only 3840x2160:8:sparse > /dev/null 2>&1
python tools/strip_timeline.py /tmp/tl > gpurun_out/r02_k/summary/r02_k_strip_timeline_4k_n8.txt 2>&1
