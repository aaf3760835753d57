#!/bin/bash
# timing / traffic-attribution builds of the library with gemm3.hip compiled under -DARIA_ABL=<bits> (bits documented at their uses in
# gemm3.hip; results are wrong by construction) -> build/abl/libaria_gemm3_<bits>.so.  On the GPU box a session step `lib=<path>` copies one
# over aria_amd/libaria_hip.so (the box's tree is a scratch copy), `lib=restore` puts the product library back (tools/gpu_session.sh).
#   tools/probes/build_gemm_abl.sh 8192 16384
cd "$(dirname "$0")/../.."; mkdir -p build/abl
OTHERS=$(ls build/*.o | grep -v gemm3.o)
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iaria_amd/csrc -DARIA_ABL=$v -c aria_amd/csrc/gemm3.hip -o build/abl/gemm3_$v.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/abl/gemm3_$v.o $OTHERS -o build/abl/libaria_gemm3_$v.so && rm build/abl/gemm3_$v.o ) &
  while [ $(jobs -r | wc -l) -ge 3 ]; do sleep 1; done
done
wait; ls -la build/abl/libaria_gemm3_*.so
