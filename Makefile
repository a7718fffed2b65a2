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
APIWALK   := tests/cpp/api_walk
$(APIWALK): tests/cpp/api_walk.cpp include/particles_hip.h $(LIB)
	g++ -O1 -std=c++17 -Wall -Wextra -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include -o $@ $< -Lparticles_amd -lparticles_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../../particles_amd' -Wl,-rpath,/opt/rocm/lib

# The same program and a copy of the library with the HOST code under AddressSanitizer and
# UndefinedBehaviorSanitizer (device code is not instrumented): build/sanitize/, run on a GPU box as
#   ASAN_OPTIONS=detect_leaks=0 build/sanitize/api_walk
SAN       := -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -fno-omit-frame-pointer -g
sanitize: $(SRC) $(SORT_SRC) $(COMM_SRC) $(HDRS) particles_amd/csrc/authoring.cpp tests/cpp/api_walk.cpp
	mkdir -p build/sanitize
	$(HIPCC) $(HIPFLAGS) $(SAN) -c -o build/sanitize/particles_hip.o $(SRC)
	$(HIPCC) $(HIPFLAGS) $(SAN) -Wno-unused-parameter -c -o build/sanitize/pair_sort.o $(SORT_SRC)
	$(HIPCC) $(HIPFLAGS) $(SAN) -c -o build/sanitize/particles_comm.o $(COMM_SRC)
	/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -ffp-contract=off -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Iinclude -c -o build/sanitize/authoring.o particles_amd/csrc/authoring.cpp
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -fsanitize=address,undefined -shared-libsan -o build/sanitize/libparticles_hip.so build/sanitize/particles_hip.o build/sanitize/pair_sort.o build/sanitize/particles_comm.o build/sanitize/authoring.o -ldl
	/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libsan -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include -o build/sanitize/api_walk tests/cpp/api_walk.cpp -Lbuild/sanitize -lparticles_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$$(dirname $$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so))

asm: $(SRC) $(HDRS)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o build/particles_hip.s $(SRC) -Rpass-analysis=kernel-resource-usage 2> build/resource_usage.txt

clean:
	rm -f $(LIB) $(CPPTEST) $(APIWALK) $(FAKE_RCCL)
	rm -rf build/sanitize
	$(MAKE) -C oracle clean

.PHONY: all oracle asm clean sanitize
