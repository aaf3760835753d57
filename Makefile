# Builds the product library (gfx950) and the test-only emulated library (host).
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
HOSTCXX ?= $(ROCM)/lib/llvm/bin/clang++
SRC := $(wildcard aria_amd/csrc/*.hip)
HDR := $(wildcard aria_amd/csrc/*.h) include/aria_hip.h
OBJ := $(patsubst aria_amd/csrc/%.hip,build/%.o,$(SRC))

all: aria_amd/libaria_hip.so

build/%.o: aria_amd/csrc/%.hip $(HDR)
	@mkdir -p build
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iaria_amd/csrc -c $< -o $@

aria_amd/libaria_hip.so: $(OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(OBJ) -o $@

# test infrastructure: same sources, SIMT emulator instead of a GPU
emu: tests/emu/libaria_emu.so
tests/emu/libaria_emu.so: $(SRC) $(HDR) tests/emu/hip_emu.cpp tests/emu/hip_emu.h
	$(HOSTCXX) -x c++ -std=c++17 -O3 -g -fPIC -shared -DARIA_EMU -Wno-unknown-attributes -Wno-unused-value -Wno-psabi \
	    -Iinclude -Iaria_amd/csrc -Itests/emu $(SRC) tests/emu/hip_emu.cpp -o $@

# test infrastructure: hardware-semantics probes (lane layout of ds_read_b64_tr_b16, fp32 atomic rates) -- their own shared object
probes: tests/probes/libaria_probe.so
tests/probes/libaria_probe.so: tests/probes/probe.hip
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $< -o $@

clean:
	rm -rf build aria_amd/libaria_hip.so tests/emu/libaria_emu.so
.PHONY: all emu probes clean
