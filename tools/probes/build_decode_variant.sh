#!/bin/bash
# variant builds of the library with decode.hip compiled under extra -D flags -> build/abl/libaria_decode_<tag>.so
#   tools/probes/build_decode_variant.sh fence -DARIA_STREAM_ACQ_FENCE=1
cd "$(dirname "$0")/../.."; mkdir -p build/abl
TAG=$1; shift
OTHERS=$(ls build/*.o | grep -v decode.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iaria_amd/csrc "$@" -c aria_amd/csrc/decode.hip -o build/abl/decode_$TAG.o \
  && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/abl/decode_$TAG.o $OTHERS -o build/abl/libaria_decode_$TAG.so && rm build/abl/decode_$TAG.o
ls -la build/abl/libaria_decode_$TAG.so
