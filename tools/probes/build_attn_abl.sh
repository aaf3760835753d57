#!/bin/bash
# timing-only builds of the library with attn.hip compiled under -D<MACRO>=<bits> (see attn.hip) -> build/abl/libaria_<tag>_<bits>.so
#   tools/probes/build_attn_abl.sh ARIA_ATTN_ABL attn 0 1 2 4 ...      (attn_fwd2_kernel)
#   tools/probes/build_attn_abl.sh ARIA_DKDV_ABL dkdv 0 1 2 4 ...      (attn_bwd3_dkdv_kernel)
cd "$(dirname "$0")/../.."; mkdir -p build/abl
MACRO=$1; TAG=$2; shift 2
OTHERS=$(ls build/*.o | grep -v attn.o)
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iaria_amd/csrc -D$MACRO=$v -c aria_amd/csrc/attn.hip -o build/abl/${TAG}_$v.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/abl/${TAG}_$v.o $OTHERS -o build/abl/libaria_${TAG}_$v.so && rm build/abl/${TAG}_$v.o ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait; ls build/abl/libaria_${TAG}_*.so | wc -l
