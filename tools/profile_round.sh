#!/bin/bash
# Produces the raw material of one profiles/<tag>_* set on the GPU box (run from the repo root):
#   bash tools/profile_round.sh r01_f
# bench line, rocprofv3 kernel-trace stats, and separate PMC passes (never combined with traces).
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
# BENCH_VERIFY=0 in the profiled runs below: the kernel-sequence verification context would add its launches to the traces
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
export BENCH_VERIFY=0
# per-kernel profiles are of frames whose kernels run back to back on one stream (RT_TUNING=14=0: without the pipelined stage 0
# the kernels of consecutive frames do not overlap (17=0: resolve on the main stream too), so a kernel's duration and counters are its own; the bench line records
# rt_tuning_env). The headline bench.json above is the default (pipelined) run.
export RT_TUNING=14=0,17=0
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o st -- python3 bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d $OUT/pmc_$c -o pmc -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 2 > $OUT/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_valu -o pmc -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 2 > $OUT/pmc_valu.log 2>&1
# dynamic instruction classes (priced with the measured per-class issue costs: valu_issue_frac_weighted)
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $OUT/pmc_class -o pmc -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 2 > $OUT/pmc_class.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $OUT/pmc_mix -o pmc -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 2 > $OUT/pmc_mix.log 2>&1
# where a wavefront's cycles go (quad-cycles; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, MI355X_MICROARCH.md PMC slots)
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d $OUT/pmc_stall -o pmc -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 2 > $OUT/pmc_stall.log 2>&1
# vector-L1 / L2 request counters of the frame kernels (tools/pmc_probe.py; what the four-lanes-per-record fetch of the spatial pass changes)
: > $OUT/cache_counters.txt
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum SQ_WAVES"; do
  rm -rf $OUT/pmc_cache
  timeout 300 rocprofv3 --pmc $set -d $OUT/pmc_cache -o pmc -- python3 bench.py --no-cpu-baseline --steps 12 --warmup 2 > $OUT/pmc_cache.log 2>&1
  echo "== $set" >> $OUT/cache_counters.txt
  python3 tools/pmc_probe.py $OUT/pmc_cache >> $OUT/cache_counters.txt 2>&1
done
# the shadowed-target mode (README key 3): kernel trace + the same counter passes, 20 frames
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/sh_stats -o st -- python3 tools/shadowed_frames.py 20 > $OUT/sh_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d $OUT/sh_pmc_$c -o pmc -- python3 tools/shadowed_frames.py 12 > $OUT/sh_pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/sh_pmc_valu -o pmc -- python3 tools/shadowed_frames.py 12 > $OUT/sh_pmc_valu.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $OUT/sh_pmc_class -o pmc -- python3 tools/shadowed_frames.py 12 > $OUT/sh_pmc_class.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d $OUT/sh_pmc_stall -o pmc -- python3 tools/shadowed_frames.py 12 > $OUT/sh_pmc_stall.log 2>&1
unset RT_TUNING BENCH_VERIFY
sha256sum cedec_2024_rt_amd/librestir_rt.so > $OUT/lib.sha256
python3 -c "from cedec_2024_rt_amd import api; print(api.build_id())" > $OUT/lib.build_id
# summarise on the box (the raw rocprofv3 databases are too large to travel back) and keep the summaries only
mkdir -p $OUT/summary
cp -r profiles /tmp/profiles_before_$TAG
python3 tools/profile_collect.py $TAG > $OUT/summary/collect.log 2>&1
cp $OUT/cache_counters.txt profiles/${TAG}_cache_counters.txt
for f in profiles/${TAG}_* profiles/spatial_pmc_latest.json; do cp $f $OUT/summary/; done
rm -rf $OUT/stats $OUT/pmc_* $OUT/sh_stats $OUT/sh_pmc_*
cat $OUT/bench.json
