#!/bin/bash
# timing-only builds of the library with attn.hip compiled under -DARIA_ATTN_ABL=<bits> (see attn.hip) -> build/abl/libaria_attn_<bits>.so
cd "$(dirname "$0")/../.."; mkdir -p build/abl
OTHERS=$(ls build/*.o | grep -v attn.o)
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iaria_amd/csrc -DARIA_ATTN_ABL=$v -c aria_amd/csrc/attn.hip -o build/abl/attn_$v.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/abl/attn_$v.o $OTHERS -o build/abl/libaria_attn_$v.so && rm build/abl/attn_$v.o ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait; ls -la build/abl/libaria_attn_*.so
