# C++-only build of the product (no Python needed): the HIP library with its C ABI and the main() drop-in.
#   make            -> our_first_climate_model_amd/librcm_hip.so + our_first_climate_model_amd/rcm_main
#   make examples   -> examples/abi_minimal (the C ABI from a plain C++ program)
#   make oracle     -> the CPU checker (test infrastructure, never linked into the product)
# `python -m our_first_climate_model_amd.build` does the same from Python (and is what __graft_entry__.build() calls).
HIPCC     ?= /opt/rocm/bin/hipcc
ROCM_PATH ?= /opt/rocm
PKG  := our_first_climate_model_amd
CSRC := $(PKG)/csrc
LIB  := $(PKG)/librcm_hip.so
MAIN := $(PKG)/rcm_main
HDRS := include/rcm.h $(wildcard $(CSRC)/*.hpp)

comma := ,
RCCL_FLAGS := $(if $(wildcard $(ROCM_PATH)/include/rccl/rccl.h),-DRCM_WITH_RCCL -D__HIP_PLATFORM_AMD__ \
  -I$(ROCM_PATH)/include -L$(ROCM_PATH)/lib -lrccl -lamdhip64 -Wl$(comma)-rpath$(comma)$(ROCM_PATH)/lib,)

all: $(LIB) $(MAIN)

examples: examples/abi_minimal

examples/abi_minimal: examples/abi_minimal.cpp $(LIB) include/rcm.h
	g++ -O2 -std=c++17 -Iinclude -o $@ $< -L$(PKG) -lrcm_hip -Wl,-rpath,'$$ORIGIN/../$(PKG)'

$(LIB): $(CSRC)/rcm_capi.hip $(CSRC)/rcm_host.cpp $(CSRC)/rcm_nc.cpp $(HDRS)
	$(HIPCC) -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$(CSRC) -Wall -Wno-unused-function -shared -o $@ \
	  $(CSRC)/rcm_capi.hip $(CSRC)/rcm_host.cpp $(CSRC)/rcm_nc.cpp

$(MAIN): $(CSRC)/rcm_main.cpp $(LIB) include/rcm.h
	g++ -O2 -std=c++17 -Iinclude -o $@ $(CSRC)/rcm_main.cpp -L$(PKG) -lrcm_hip -Wl,-rpath,'$$ORIGIN' $(RCCL_FLAGS)

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB) $(MAIN) examples/abi_minimal

.PHONY: all examples oracle clean
