#!/bin/bash
# A/B builds of librestir_rt.so with different -D flags, for RT_LIB_PATH (the tools).
#   tools/build_variants.sh name1 "-DX=1" name2 "-DX=2 -DY=3" ...
cd "$(dirname "$0")/../cedec_2024_rt_amd/csrc"
mkdir -p ../../gpurun_variants
pids=()
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  ( make OBJDIR=/tmp/rt_variant_$n OUT=../../gpurun_variants/lib_$n.so EXTRA="$f" ../../gpurun_variants/lib_$n.so > /tmp/rt_variant_$n.log 2>&1 || echo "variant $n FAILED" ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
ls -la ../../gpurun_variants/
