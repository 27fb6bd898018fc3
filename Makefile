# Builds everything in-tree (built artefacts are git-ignored but travel to the GPU box):
#   md_neighbor_list_amd/lib/libnl_hip.so     HIP kernels + C ABI (gfx950 only)
#   md_neighbor_list_amd/lib/libnl_inputs.so  synthetic particle boxes (host C++)
#   tools/make_list                           driver with the flow of the reference's make_list.cu
#   oracle/liboracle.so, oracle/_ref/*        the CPU oracle (test infrastructure; see oracle/Makefile)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
LIBDIR := md_neighbor_list_amd/lib
CSRC := md_neighbor_list_amd/csrc

# -ffp-contract=off: r2 = (dx*dx + dy*dy) + dz*dz must not be contracted into FMAs (bit-exact pair set).
# -fno-slp-vectorize: v_pk_add/mul_f32 issue slower than two plain ops on gfx950 (profiles/r01_valu_microbench.txt)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-result

all: lib inputs tools oracle

lib: $(LIBDIR)/libnl_hip.so
inputs: $(LIBDIR)/libnl_inputs.so

$(LIBDIR)/libnl_hip.so: $(CSRC)/nl_api.hip $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.inc) include/nl_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/nl_api.hip

$(LIBDIR)/libnl_inputs.so: $(CSRC)/nl_inputs.cpp
	@mkdir -p $(LIBDIR)
	$(CXX) -O2 -std=c++17 -fPIC -shared -o $@ $<

tools: tools/make_list
tools/make_list: tools/make_list.cpp include/neighlist_gpu.hpp include/nl_hip.h $(LIBDIR)/libnl_hip.so $(LIBDIR)/libnl_inputs.so
	$(CXX) -O2 -std=c++17 -Iinclude -o $@ tools/make_list.cpp -L$(LIBDIR) -lnl_hip -lnl_inputs -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

oracle:
	$(MAKE) -C oracle

# CPU sanitizer build: the host shims over a host-memory stand-in of the C ABI, the input generator
# and the oracle's restatement under AddressSanitizer + UndefinedBehaviorSanitizer.  CPU only -- never run on the GPU box.
SANFLAGS := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1
asan: build/sanitize_test
build/sanitize_test: tests/sanitize/main.cpp tests/sanitize/abi_stub.cpp oracle/nl_oracle.c oracle/nl_oracle_impl.h $(CSRC)/nl_inputs.cpp include/neighlist_cpu.hpp include/neighlist_gpu.hpp include/nl_hip.h
	@mkdir -p build
	gcc $(SANFLAGS) -std=c11 -ffp-contract=off -fopenmp -Wall -Wextra -c -o build/san_oracle.o oracle/nl_oracle.c
	$(CXX) $(SANFLAGS) -std=c++17 -Wall -Iinclude -o $@ tests/sanitize/main.cpp tests/sanitize/abi_stub.cpp $(CSRC)/nl_inputs.cpp build/san_oracle.o -fopenmp

# ISA + resource usage of the kernels, for DESIGN.md / tuning
asm:
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -o build/nl_api.s $(CSRC)/nl_api.hip 2> build/resource_usage.txt

clean:
	rm -f $(LIBDIR)/*.so tools/make_list build/*
	$(MAKE) -C oracle clean

.PHONY: all lib inputs tools oracle asm asan clean
