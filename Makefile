# Builds the gfx950 library (in-tree, so the .so travels with the source tree)
# and the CPU oracle used by the tests.
HIPCC     ?= /opt/rocm/bin/hipcc
ARCH      ?= gfx950
# -ffp-contract=off: the update is specified without fused multiply-add.
HIPFLAGS  ?= -O3 --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -Wall -Wextra -Iinclude

LIB       := particles_amd/libparticles_hip.so
SRC       := particles_amd/csrc/particles_hip.hip
HDRS      := particles_amd/csrc/step_kernels.hpp include/particles_hip.h

all: $(LIB) oracle

$(LIB): $(SRC) $(HDRS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(SRC)

oracle:
	$(MAKE) -C oracle

asm: $(SRC) $(HDRS)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o build/particles_hip.s $(SRC) -Rpass-analysis=kernel-resource-usage 2> build/resource_usage.txt

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle asm clean
