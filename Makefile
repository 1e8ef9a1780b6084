# C++-only build of the product (no Python needed): the HIP library with its C ABI and the main() drop-in.
#   make            -> our_first_climate_model_amd/librcm_hip.so + our_first_climate_model_amd/rcm_main
#   make examples   -> examples/abi_minimal (the C ABI from a plain C++ program)
#   make oracle     -> the CPU checker (test infrastructure, never linked into the product)
#   make asan-nc    -> mutation fuzz of the .nc reader under ASan/UBSan (100 000 cases, host code)
#   make asan-host  -> the same for the .rtab / .atm / ASCII-table / lbl-cache readers (20 000 cases each)
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

$(LIB): $(CSRC)/rcm_capi.hip $(CSRC)/rcm_host.cpp $(CSRC)/rcm_nc.cpp $(HDRS) $(CSRC)/rcm_exports.map
	$(HIPCC) -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$(CSRC) -Wall -Wno-unused-function -shared -o $@ \
	  -Wl,--version-script=$(CSRC)/rcm_exports.map $(CSRC)/rcm_capi.hip $(CSRC)/rcm_host.cpp $(CSRC)/rcm_nc.cpp

$(MAIN): $(CSRC)/rcm_main.cpp $(LIB) include/rcm.h
	g++ -O2 -std=c++17 -Iinclude -o $@ $(CSRC)/rcm_main.cpp -L$(PKG) -lrcm_hip -Wl,-rpath,'$$ORIGIN' $(RCCL_FLAGS)

oracle:
	$(MAKE) -C oracle

# The HDF5-subset reader under ASan + UBSan on mutated tables (host code; tests/test_sanitize_nc.py runs 4 000 cases)
asan-nc:
	g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -Iinclude \
	  -I$(CSRC) -o /tmp/rcm_nc_fuzz tests/sanitize/nc_fuzz.cpp $(CSRC)/rcm_nc.cpp $(CSRC)/rcm_host.cpp
	/tmp/rcm_nc_fuzz tests/golden/Reduced10Forcing.nc 100000 /tmp/rcm_nc_fuzz_scratch.nc \
	  $(wildcard /root/reference/repwvl_V2.01_cpp/Reduced20Forcing.nc /root/reference/repwvl_V2.01_cpp/Reduced100Forcing.nc)

asan-host:
	g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -Iinclude \
	  -I$(CSRC) -o /tmp/rcm_host_fuzz tests/sanitize/host_fuzz.cpp $(CSRC)/rcm_nc.cpp $(CSRC)/rcm_host.cpp
	/tmp/rcm_host_fuzz tests/golden/Reduced10Forcing.rtab tests/golden/test.atm 20000 /tmp/rcm_host_fuzz_scratch

clean:
	rm -f $(LIB) $(MAIN) examples/abi_minimal

.PHONY: all examples oracle asan-nc asan-host clean
