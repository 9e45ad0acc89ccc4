#!/bin/bash
# experiment builds of the product library with cache-policy bits on the pixel loads / stores:
#   tools/exp/build_aux.sh "<ld> <st>" ...   ->  build/exp/libhevcdbk_ld<ld>_st<st>.so   (build/ is git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/../../gpu_video_codec_amd/csrc"
make -s all
mkdir -p ../../build/exp
for v in "$@"; do
  set -- $v; ld=$1; st=$2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DDBK_AUX_LD=$ld -DDBK_AUX_ST=$st -c -o ../../build/exp/k_${ld}_${st}.o deblock_kernels.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build/exp/libhevcdbk_ld${ld}_st${st}.so ../../build/exp/k_${ld}_${st}.o deblock_h265.o sao.o deblock_host.o deblock_host_h265.o execute_gpu_shim.o -Wl,-rpath,/opt/rocm/lib
  echo built ld=$ld st=$st
done
