#!/bin/bash
# A/B builds of the same ABI: tools/build_variant.sh NAME "-DFLAG=..."  ->  chatttsplus_amd/_lib/libctts_hip_NAME.so  (select with CTTS_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
OUT=chatttsplus_amd/_lib/ab_$NAME
mkdir -p $OUT
for f in gpt_engine skinny_gemm prefill_gemm prefill_split lora attention sampler vocoder encoder persist_layer; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=8 "$@" -c chatttsplus_amd/csrc/$f.hip -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o chatttsplus_amd/_lib/libctts_hip_$NAME.so $OUT/*.o -ldl
echo built chatttsplus_amd/_lib/libctts_hip_$NAME.so
