# Builds the gfx950 library (in-tree, so the .so travels with the source tree)
# and the CPU oracle used by the tests.
HIPCC     ?= /opt/rocm/bin/hipcc
ARCH      ?= gfx950
# -ffp-contract=off: the update is specified without fused multiply-add.
HIPFLAGS  ?= -O3 --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -Wall -Wextra -Iinclude

LIB       := particles_amd/libparticles_hip.so
SRC       := particles_amd/csrc/particles_hip.hip
SORT_SRC  := particles_amd/csrc/pair_sort.hip
SORT_OBJ  := build/pair_sort.o
HDRS      := particles_amd/csrc/step_kernels.hpp particles_amd/csrc/reorder_kernels.hpp particles_amd/csrc/brush_kernels.hpp particles_amd/csrc/handle_access.hpp include/particles_hip.h
COMM_SRC  := particles_amd/csrc/particles_comm.hip
COMM_OBJ  := build/particles_comm.o

CPPTEST   := tests/cpp/compute_wrapper_test
APIWALK   := tests/cpp/api_walk

FAKE_RCCL := tests/fake_rccl/libfake_rccl.so

all: $(LIB) oracle $(CPPTEST) $(APIWALK) $(FAKE_RCCL)

# the radix sort of the binned pair term's cell order, in its own object
$(SORT_OBJ): $(SORT_SRC)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -Wno-unused-parameter -c -o $@ $(SORT_SRC)

MAIN_OBJ  := build/particles_hip.o
$(MAIN_OBJ): $(SRC) $(HDRS)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $(SRC)

# the multi-GPU layer; RCCL itself is loaded at run time (dlopen), only its header is needed here
$(COMM_OBJ): $(COMM_SRC) $(HDRS)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $(COMM_SRC)

# host-only authoring helpers (force curves through sample points, initial fields)
AUTH_OBJ  := build/authoring.o
$(AUTH_OBJ): particles_amd/csrc/authoring.cpp include/particles_hip.h
	mkdir -p build
	g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wextra -Iinclude -c -o $@ $<

$(LIB): $(MAIN_OBJ) $(SORT_OBJ) $(COMM_OBJ) $(AUTH_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -o $@ $(MAIN_OBJ) $(SORT_OBJ) $(COMM_OBJ) $(AUTH_OBJ) -ldl

oracle:
	$(MAKE) -C oracle

# Test infrastructure: a host-staged stand-in for librccl.so that lets several ranks share one GPU
$(FAKE_RCCL): tests/fake_rccl/fake_rccl.cpp
	g++ -O1 -std=c++17 -fPIC -shared -Wall -Wextra -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o $@ $< -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lrt -lpthread

# Host-only program over the C++ mirror of the reference's Compute; plain g++.
$(CPPTEST): tests/cpp/compute_wrapper_test.cpp include/particles_compute.hpp include/particles_hip.h $(LIB)
	g++ -O2 -std=c++17 -Wall -Wextra -Iinclude -o $@ $< -Lparticles_amd -lparticles_hip -Wl,-rpath,'$$ORIGIN/../../particles_amd' -Wl,-rpath-link,/opt/rocm/lib

# The whole C ABI walked once by a native program (status codes only; results are the Python tests' business).
$(APIWALK): tests/cpp/api_walk.cpp include/particles_hip.h $(LIB)
	g++ -O1 -std=c++17 -Wall -Wextra -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include -o $@ $< -Lparticles_amd -lparticles_hip -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,'$$ORIGIN/../../particles_amd' -Wl,-rpath,/opt/rocm/lib

# The same program and a copy of the library with the HOST code under sanitizers (device code is not
# instrumented): by default AddressSanitizer + UndefinedBehaviorSanitizer into build/sanitize/; run on a GPU box as
#   ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 build/sanitize/api_walk
# `make sanitize SANKIND=thread SANDIR=build/tsan` builds the ThreadSanitizer copies the same way.
SANKIND   ?= address,undefined
SANDIR    ?= build/sanitize
SANHOST   := -fsanitize=$(SANKIND) -fno-sanitize-recover=all -fno-omit-frame-pointer -g
SAN       := $(foreach f,$(SANHOST),-Xarch_host $(f))
CLANGXX   := /opt/rocm/lib/llvm/bin/clang++
sanitize: $(SRC) $(SORT_SRC) $(COMM_SRC) $(HDRS) particles_amd/csrc/authoring.cpp tests/cpp/api_walk.cpp
	mkdir -p $(SANDIR)
	$(HIPCC) $(HIPFLAGS) $(SAN) -c -o $(SANDIR)/particles_hip.o $(SRC)
	$(HIPCC) $(HIPFLAGS) $(SAN) -Wno-unused-parameter -c -o $(SANDIR)/pair_sort.o $(SORT_SRC)
	$(HIPCC) $(HIPFLAGS) $(SAN) -c -o $(SANDIR)/particles_comm.o $(COMM_SRC)
	$(CLANGXX) -O1 -std=c++17 -ffp-contract=off -fPIC $(SANHOST) -Iinclude -c -o $(SANDIR)/authoring.o particles_amd/csrc/authoring.cpp
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -fsanitize=$(SANKIND) -shared-libsan -o $(SANDIR)/libparticles_hip.so $(SANDIR)/particles_hip.o $(SANDIR)/pair_sort.o $(SANDIR)/particles_comm.o $(SANDIR)/authoring.o -ldl
	$(CLANGXX) -O1 -std=c++17 $(SANHOST) -shared-libsan -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include -o $(SANDIR)/api_walk tests/cpp/api_walk.cpp -L$(SANDIR) -lparticles_hip -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$$(dirname $$($(CLANGXX) -print-file-name=libclang_rt.asan-x86_64.so))

asm: $(SRC) $(HDRS)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o build/particles_hip.s $(SRC) -Rpass-analysis=kernel-resource-usage 2> build/resource_usage.txt

clean:
	rm -f $(LIB) $(CPPTEST) $(APIWALK) $(FAKE_RCCL)
	rm -rf build/sanitize build/tsan
	$(MAKE) -C oracle clean

.PHONY: all oracle asm clean sanitize
